/* include/hp_hip.h — C ABI of libhp_hip.so, the MI355X (gfx950) implementation of HyperPose's hot path.
 *
 * HyperPose has no FFI of its own: the boundary of its hot path is a set of C++17 classes
 * (SURVEY.md section 8b).  This header is the thin C layer the north star asks for: the C++ mirror
 * classes in include/hyperpose/ (same names, signatures and error behaviour as the reference headers)
 * are implemented purely on top of these entry points, and any other host language can bind them
 * directly (INTEGRATION.md shows the ctypes / C++ stubs).
 *
 * Conventions: plain pointers and sizes only; opaque handles; every function returns HP_OK (0) or a
 * negative HP_ERR_* code and never throws or exits across the ABI; hp_last_error() returns a
 * thread-local description of the last failure.  One handle is used by one thread at a time (same rule
 * as the reference: include/hyperpose/stream/stream.hpp:139-144, openpifpaf_postprocessor.hpp:23-26).
 * "dev" pointers are HIP device pointers on the device given to hp_init(); "host" pointers are
 * ordinary (ideally pinned) host memory.
 */
#ifndef HP_HIP_H
#define HP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HP_OK 0
#define HP_ERR_INVALID (-1)   /* bad argument / shape */
#define HP_ERR_HIP (-2)       /* a HIP runtime call failed */
#define HP_ERR_CAPACITY (-3)  /* a fixed-capacity device list overflowed (peaks, candidates, humans, batch) */
#define HP_ERR_STATE (-4)     /* call order violated (e.g. collect without enqueue; shape changed after first call) */
#define HP_ERR_NO_DEVICE (-5) /* no gfx950 device visible */

#define HP_COCO_N_PARTS 18 /* include/hyperpose/utility/human.hpp:10 */
#define HP_COCO_N_PAIRS 19 /* include/hyperpose/utility/human.hpp:11 */

/* hyperpose::body_part_t / human_t (include/hyperpose/utility/human.hpp:14-31), same 292-byte layout
 * (has_value is the reference's `bool` widened to its 4-byte slot: 0 or 1). */
typedef struct hp_body_part {
    int32_t has_value;
    float x, y, score;
} hp_body_part;

typedef struct hp_human {
    hp_body_part parts[HP_COCO_N_PARTS];
    float score;
} hp_human;

/* peak_info (src/post_process.hpp:126-131) and connection (src/paf.cpp:7-13) — exposed only for the
 * stage-wise parity taps below. */
typedef struct hp_peak {
    int32_t part_id;
    int32_t x, y;
    float score;
    int32_t id;
} hp_peak;

typedef struct hp_conn {
    int32_t pair_id;
    int32_t cid1, cid2;
    float score;
} hp_conn;

/* ---- runtime ------------------------------------------------------------------------------------ */
int hp_init(int device);            /* hipSetDevice(device) for the calling thread; checks the arch is gfx950 */
int hp_device_count(void);          /* number of visible HIP devices, or a negative HP_ERR_* */
const char* hp_last_error(void);    /* thread-local, never NULL */
const char* hp_version(void);

/* Device memory helpers so that a host language needs no HIP binding of its own. */
int hp_malloc(void** dev, size_t nbytes);
int hp_free(void* dev);
int hp_malloc_host(void** host, size_t nbytes); /* pinned */
int hp_free_host(void* host);
int hp_memcpy_h2d(void* dev, const void* host, size_t nbytes);
int hp_memcpy_d2h(void* host, const void* dev, size_t nbytes);
int hp_device_synchronize(void);

/* ---- pre-processing: replaces hyperpose::nhwc_images_append_nchw_batch (src/data.cpp:21-51) -------
 * u8 HWC (BGR) frames [n,h,w,3] -> f32 CHW [n,3,h,w], value = (float)((double)u8 * factor), channel
 * order {2,1,0} when flip_rb.  Both pointers are device pointers; `stream` is a hipStream_t (NULL = default). */
int hp_preproc_u8hwc_to_f32nchw(const uint8_t* dev_hwc, int n, int h, int w, double factor, int flip_rb,
                                float* dev_nchw, void* stream);

/* ---- hyperpose::parser::paf (include/hyperpose/operator/parser/paf.hpp:17-93, src/paf.cpp) -------- */
typedef struct hp_paf hp_paf;

/* paf::paf(conf_thresh, paf_thresh, resolution_size) (paf.hpp:27).  res_w/res_h = -1 keeps the reference's
 * lazy default `cv::Size(dim1*4, dim2*4)` of the first processed tensor (src/paf.cpp:314-315, including its
 * swapped naming: width = 4*rows, height = 4*cols).  max_batch sizes the device scratch. */
int hp_paf_create(hp_paf** out, float conf_thresh, float paf_thresh, int res_w, int res_h, int max_batch);
void hp_paf_destroy(hp_paf* p);
int hp_paf_set_conf_thresh(hp_paf* p, float thresh); /* paf::set_conf_thresh, src/paf.cpp:382 */
int hp_paf_set_paf_thresh(hp_paf* p, float thresh);  /* paf::set_paf_thresh,  src/paf.cpp:377 */

/* paf::process(conf, paf) (src/paf.cpp:300-375) for n frames at once.
 *   conf [n, J, rows, cols], paf [n, 2L, rows, cols], fp32, contiguous; shapes WITHOUT batch dim as in
 *   feature_map_t::shape() (include/hyperpose/utility/data.hpp:22-23).  on_device != 0: device pointers.
 *   out: host array [n * cap_per_frame]; n_out: host array [n] receiving the human count of every frame.
 * Blocks until the result is on the host.  The first call fixes the shapes (reference: lazy one-shot
 * allocation, src/paf.cpp:321-332; a later call with other shapes returns HP_ERR_STATE instead of UB). */
int hp_paf_process_batch(hp_paf* p, int n, const float* conf, const int conf_shape[3], const float* paf,
                         const int paf_shape[3], int on_device, hp_human* out, int cap_per_frame, int* n_out);

/* Asynchronous halves of the same call for pipelines: enqueue launches the kernels and the D2H copy of the
 * humans on `stream` (NULL = the parser's own stream) and returns at once; collect waits for that batch. */
int hp_paf_enqueue(hp_paf* p, int n, const float* dev_conf, const int conf_shape[3], const float* dev_paf,
                   const int paf_shape[3], void* stream);
int hp_paf_collect(hp_paf* p, hp_human* out, int cap_per_frame, int* n_out);

/* Stage-wise parity taps (valid after a completed process/collect): the peak list (post_process.hpp:171-193
 * order) and the per-limb connections (src/paf.cpp:252-270 order) of one frame of the last batch. */
int hp_paf_debug_peaks(hp_paf* p, int frame, hp_peak* out, int cap, int* n);
int hp_paf_debug_conns(hp_paf* p, int frame, hp_conn* out, int cap, int* n);
/* The up-sampled (resize_area) and the smoothed (GaussianBlur) confidence maps [J,res_h,res_w] of ONE frame,
 * host pointers, either output may be NULL (tests only; the production kernels never materialise them). */
int hp_paf_debug_maps(hp_paf* p, const float* host_conf, const int conf_shape[3], float* host_up, float* host_smoothed);

#ifdef __cplusplus
}
#endif
#endif /* HP_HIP_H */
