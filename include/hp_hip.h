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
/* Everything enqueued on `waiter` after this call starts only after everything enqueued on `signaler` before it has
 * finished (hipEventRecord + hipStreamWaitEvent).  Lets a parser run on its own stream behind its engine, the way the
 * reference's stream pipeline hands a batch from its inference thread to its parser thread (stream.hpp:139-190). */
int hp_stream_wait_stream(void* waiter, void* signaler);

/* ---- multi-GPU: frames shard over the GPUs of a node, one process per GPU, no steady-state collective.  The one
 * collective is the start-up broadcast of the weight blob over RCCL / xGMI.  Rendezvous: rank 0 calls hp_dist_unique_id and hands the
 * 128 bytes to the other ranks by whatever channel launched them (a file, an environment variable, MPI, a TCP store);
 * every rank then calls hp_dist_init after hp_init(device).  librccl.so is loaded on first use. */
#define HP_DIST_ID_BYTES 128
typedef struct hp_comm hp_comm;
int hp_dist_unique_id(char id[HP_DIST_ID_BYTES]);
int hp_dist_init(hp_comm** out, int rank, int world, const char id[HP_DIST_ID_BYTES]);
void hp_dist_destroy(hp_comm* c);
/* host_weights [n]: read on `root`, overwritten on every other rank (ncclBroadcast through a device buffer) */
int hp_dist_broadcast_weights(hp_comm* c, float* host_weights, size_t n, int root);
/* contiguous split of a global batch over the ranks: rank r processes frames [start, start + count) */
void hp_dist_shard(int total_frames, int rank, int world, int* start, int* count);

/* ---- pre-processing: replaces hyperpose::nhwc_images_append_nchw_batch (src/data.cpp:21-51) -------
 * u8 HWC (BGR) frames [n,h,w,3] -> f32 CHW [n,3,h,w], value = (float)((double)u8 * factor), channel
 * order {2,1,0} when flip_rb.  Both pointers are device pointers; `stream` is a hipStream_t (NULL = default). */
int hp_preproc_u8hwc_to_f32nchw(const uint8_t* dev_hwc, int n, int h, int w, double factor, int flip_rb,
                                float* dev_nchw, void* stream);

/* ---- stream front-end geometry (reference: cv::resize at src/stream.cpp:93,101 and hyperpose::non_scaling_resize,
 * include/hyperpose/utility/data.hpp:67, src/data.cpp:53-69; resume_ratio, include/hyperpose/utility/human.hpp:44-58).
 * Images are 8-bit BGR HWC in DEVICE memory, row strides in bytes; results equal OpenCV's INTER_LINEAR bit for bit
 * (restated in oracle/resize_oracle.cpp; parity unpinned: no OpenCV in the build image). */
int hp_resize_u8c3(const uint8_t* dev_src, int sw, int sh, int src_stride, uint8_t* dev_dst, int dw, int dh, int dst_stride,
                   void* stream);
/* non_scaling_resize: aspect-preserving resize into the top-left corner, the rest filled with (b, g, r) */
int hp_letterbox_u8c3(const uint8_t* dev_src, int sw, int sh, int src_stride, uint8_t* dev_dst, int dw, int dh, int dst_stride,
                      int b, int g, int r, void* stream);
void hp_letterbox_inner(int sw, int sh, int dw, int dh, int* inner_w, int* inner_h); /* size of the resized region */
/* resume_ratio on n humans in place (host memory): undo the letterbox for (src = frame size, dst = network size) */
void hp_resume_ratio(hp_human* humans, int n, int src_w, int src_h, int dst_w, int dst_h);

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

void* hp_paf_stream(hp_paf* p); /* hipStream_t the parser owns (used when enqueue is given stream = NULL) */
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
/* The device's restatement of libstdc++'s std::sort(first, last, std::greater<connection_candidate>) (src/paf.cpp:249: the order of
 * equal scores is whatever that algorithm leaves) on n scores given in generation order; host_order[i] = index of the element
 * that ends at position i; *used_heap = 1 when the introsort depth limit was hit and the heap-sort fall-back ran (tests only). */
int hp_paf_debug_sort(const float* host_scores, int n, int* host_order, int* used_heap);

/* ---- hyperpose::parser::pose_proposal (include/hyperpose/operator/parser/proposal_network.hpp:17-81,
 * src/pose_proposal.cpp).  GPU: threshold + box decode + per-class NMS + limb-candidate gather; host: the
 * order-dependent tail (limb selection, hash merge, filter) on the compacted lists. */
typedef struct hp_ppn hp_ppn;
/* pose_proposal(net_resolution, point_thresh = 0.10, limb_thresh = 0.05, mns_thresh = 0.3), proposal_network.hpp:26 */
int hp_ppn_create(hp_ppn** out, int net_w, int net_h, float point_thresh, float limb_thresh, float nms_thresh, int max_batch);
void hp_ppn_destroy(hp_ppn* p);
int hp_ppn_set_thresholds(hp_ppn* p, float point_thresh, float limb_thresh, float nms_thresh); /* set_{point,limb,nms}_thresh */
/* pose_proposal::process (src/pose_proposal.cpp:68-337) for n frames: tensors[7] = {conf_point, conf_iou, x, y, w, h,
 * edge}, each batch-major contiguous ([n,K,gh,gw] x6, [n,E,nh,nw,gh,gw]); conf_shape = {K,gh,gw}, edge_shape =
 * {E,nh,nw,gh,gw}; on_device != 0: device pointers.  out: host [n*cap_per_frame]; n_out: host [n]. */
int hp_ppn_process_batch(hp_ppn* p, int n, const float* const tensors[7], const int conf_shape[3], const int edge_shape[5],
                         int on_device, hp_human* out, int cap_per_frame, int* n_out);
/* Asynchronous halves (same contract as hp_paf_enqueue / hp_paf_collect): enqueue launches the extraction kernel on `stream`
 * (NULL = the parser's own), which writes its compacted lists straight into pinned host memory, and returns at once; collect waits
 * for that batch and runs the order-dependent tail of its frames on the library's host worker pool. */
void* hp_ppn_stream(hp_ppn* p);
int hp_ppn_enqueue(hp_ppn* p, int n, const float* const dev_tensors[7], const int conf_shape[3], const int edge_shape[5], void* stream);
int hp_ppn_collect(hp_ppn* p, hp_human* out, int cap_per_frame, int* n_out);
/* Per frame of the last collected batch: 0 = assembled on the device (ppn_assemble_kernel); bits 4 / 8 / 16 = the device tail declined
 * the frame (more than 2048 skeleton fragments / 8192 hash entries / 2048 humans) and the same statements ran on the host; bits 1 / 2 = a
 * list of the extract kernel overflowed (HP_ERR_CAPACITY); -1 = HP_PPN_HOST_TAIL=1 (tests: every frame on the host). */
int hp_ppn_decode_flags(hp_ppn* p, int* flags, int n);

/* ---- hyperpose::parser::pifpaf (include/hyperpose/operator/parser/pifpaf.hpp:8-26, src/pifpaf.cpp,
 * src/pifpaf_decoder/openpifpaf_postprocessor.cpp).  GPU: PIF cell compaction, seed and CAF scoring with the
 * hi-res confidence map evaluated on demand (never materialised), and the decoder itself - seed-ordered greedy grow, occupancy,
 * soft-NMS, the 17 -> 18 key-point remap - as one wavefront per frame (pp_decode_kernel).  A frame the device decoder cannot finish
 * (capacity, or a score too close to a float rounding boundary to be libm-independent) is decoded by the host tail from the packed
 * lists; HP_PIFPAF_HOST_TAIL=1 sends every frame there. */
typedef struct hp_pifpaf hp_pifpaf;
int hp_pifpaf_create(hp_pifpaf** out, int net_h, int net_w, float thresh, int max_batch); /* pifpaf(int h, int w, float thresh = 0.1) */
void hp_pifpaf_destroy(hp_pifpaf* p);
/* pifpaf::process(paf, pif) (src/pifpaf.cpp:7 — the .cpp argument order): paf [n,19,9,fh,fw], pif [n,17,5,fh,fw]. */
int hp_pifpaf_process_batch(hp_pifpaf* p, int n, const float* paf, const float* pif, int fh, int fw, int on_device,
                            hp_human* out, int cap_per_frame, int* n_out);
/* Asynchronous halves: enqueue = the five kernels + the packed lists written into pinned host memory, on `stream` (NULL = the
 * parser's own) and the device decoder behind them, humans written into pinned memory; collect = wait, copy out, and the host tail
 * (worker pool) for the frames the device decoder declined. */
void* hp_pifpaf_stream(hp_pifpaf* p);
int hp_pifpaf_enqueue(hp_pifpaf* p, int n, const float* dev_paf, const float* dev_pif, int fh, int fw, void* stream);
int hp_pifpaf_collect(hp_pifpaf* p, hp_human* out, int cap_per_frame, int* n_out);
/* Per frame of the last collected batch: 0 = decoded on the device, -1 = host tail by configuration, > 0 = why the device decoder
 * handed the frame to the host tail (1 annotations > 256, 2 soft-NMS extent, 4 sort depth, 8 seeds, 16 frontier / more than 256 list
 * entries inside one search box, 32 rounding, 64 declined by the HP_PIFPAF_DECLINE_ODD test hook). */
int hp_pifpaf_decode_flags(const hp_pifpaf* p, int* flags, int n);

/* ---- hyperpose::dnn engine: replaces dnn::tensorrt (include/hyperpose/operator/dnn/tensorrt.hpp:33-141,
 * src/tensorrt.cpp).  The network is a static list of layers over numbered tensors (tensor 0 = the input
 * image); weights are one fp32 blob in the layouts below.  TensorRT's UFF/ONNX parsing is replaced by the
 * built-in topology builders (hp_model_*) that restate hyperpose/Model/<arch>.py, and by hp_model_from_onnx
 * (SURVEY.md 8f-1).  Activations live in HBM as NHWC fp16 (fp32 accumulate on MFMA); network outputs are
 * fp32 NCHW, the layout of feature_map_t. */
enum { HP_OP_CONV = 1, HP_OP_DWCONV = 2, HP_OP_MAXPOOL = 3,
       HP_OP_UPSAMPLE = 4 }; /* integer up-scaling by `stride` (UpSampling2d of the MobilenetSmall backbone, hyperpose/Model/backbones.py:325,339; ONNX
                              * Resize / Upsample): kh = 0 nearest (source = floor(dst / scale)), kh = 1 bilinear with half-pixel centres
                              * (tf.image.resize / align_corners = False); cin == cout, no weights, no activation */
enum { HP_ACT_NONE = 0, HP_ACT_RELU = 1, HP_ACT_RELU6 = 2, HP_ACT_LEAKY = 3, HP_ACT_PRELU = 4, HP_ACT_SIGMOID = 5, HP_ACT_SOFTPLUS = 6 };

typedef struct hp_layer {
    int32_t op;              /* HP_OP_* */
    int32_t in, in_coff;     /* tensor read (0 = network input) and its first channel */
    int32_t res;             /* residual tensor added in the epilogue, or -1 */
    int32_t res_before_act;  /* 1: act(conv + res) (ResNet); 0: act(conv) + res (LW-OpenPose blocks) */
    int32_t out, out_coff;   /* tensor written and its first channel (concat by offset) */
    int32_t cin, cout;
    int32_t kh, kw, stride, dil; /* padding is TF "SAME" (out = ceil(in/stride), extra pad bottom/right) unless pad_explicit */
    int32_t act;             /* HP_ACT_* applied after bias (BatchNorm is folded into w/bias by the caller) */
    float act_param;         /* LeakyReLU slope */
    int64_t w_off;           /* float offset in the blob: CONV [cout][kh][kw][cin]; DWCONV [c][kh][kw] */
    int64_t b_off;           /* bias [cout], or -1 for zeros */
    int64_t alpha_off;       /* PReLU slopes [cout], or -1 */
    int32_t pad_explicit;    /* 1: pad[] = {top, left, bottom, right} as in an ONNX Conv / MaxPool `pads` attribute,
                              * out = floor((in + pad_before + pad_after - ((k-1)*dil+1)) / stride) + 1; 0: TF "SAME" */
    int32_t pad[4];
} hp_layer;

typedef struct hp_output_desc {
    char name[48];           /* outputs are returned sorted by name (src/tensorrt.cpp:405) */
    int32_t tensor, coff, channels; /* channel range of the fp16 tensor that feeds this output */
    int32_t act;             /* element-wise op while converting to fp32 NCHW (HP_ACT_NONE / SIGMOID / SOFTPLUS), all channels */
    /* optional transforms used by the PoseProposal / PifPaf heads (all zero = plain conversion): */
    int32_t shuffle;         /* 2: pixel_shuffle(x, 2) (hyperpose/Model/pifpaf/utils.py:371-379): C/4 channels, 2H x 2W */
    int32_t group;           /* > 0: output channels come in groups of `group` components with per-component ops: */
    uint32_t sigmoid_mask;   /*      bit k set -> sigmoid on component k   (pif conf, paf conf) */
    uint32_t softplus_mask;  /*      bit k set -> softplus on component k  (pif / paf scales)   */
    int32_t out_h, out_w;    /* crop of the (shuffled) map, 0 = keep (PifPaf: 2*25 = 50 -> 49) */
    float scale;             /* y = (op(v) + grid term) * scale; 0 means 1 (PoseProposal restore_coor, model.py:111-119) */
    int32_t grid;            /* 1: add the column index, 2: add the row index before scaling */
} hp_output_desc;

/* Arithmetic of the engine.  HP_DTYPE_F16: fp16 storage, fp16 MFMA products, fp32 accumulation - the fast path (data_type::kHALF).
 * HP_DTYPE_F32: fp32 storage and fp32 matrix-pipe arithmetic, one launch per layer - what data_type::kFLOAT, the reference's default,
 * promises: outputs agree with an fp32 evaluation of the graph to ~1e-5 relative (tests/test_engine_fp32_gpu.py).
 * HP_DTYPE_F32S ("split"): the HP_DTYPE_F32 engine - fp32 storage, fp32 accumulation, same launches - with the products of its dense
 * 1 x 1 / 3 x 3 stride-1 layers formed on the fp16 matrix pipe: x = hi + 2^-11 lo with hi, lo fp16, a b = hi hi + 2^-11 (hi lo + lo hi),
 * every partial product exact in the fp32 accumulator, ~2^-22 relative per product (csrc/conv32_direct.hip).  Opt-in; an activation beyond
 * fp16's range (|x| > 65504) makes the engine re-run the batch on the fp32 pipe and stay there (hp_engine_split_fallbacks counts). */
enum { HP_DTYPE_F16 = 0, HP_DTYPE_F32 = 1, HP_DTYPE_F32S = 2 };

typedef struct hp_engine_desc {
    int32_t in_w, in_h, max_batch;   /* tensorrt(..., cv::Size input_size, int max_batch_size = 8, ...) */
    double factor;                   /* tensorrt.hpp:49: every input element is multiplied by factor (default 1/255) */
    int32_t flip_rb;                 /* BGR -> RGB (default true) */
    float mean[3], inv_std[3];       /* in-graph input normalisation of VGG19 / PifPaf, applied after factor */
    const hp_layer* layers;
    int32_t n_layers;
    const hp_output_desc* outputs;
    int32_t n_outputs;
    const float* weights;            /* host fp32 blob */
    size_t n_weights;
    int32_t dtype;                   /* HP_DTYPE_F16 (0, the zero-initialised default), HP_DTYPE_F32 or HP_DTYPE_F32S: the reference's data_type argument
                                      * (include/hyperpose/operator/dnn/tensorrt.hpp:14-21,48; src/tensorrt.cpp:327,353) */
} hp_engine_desc;

typedef struct hp_engine hp_engine;
int hp_engine_create(hp_engine** out, const hp_engine_desc* desc);
void hp_engine_destroy(hp_engine* e);
int hp_engine_max_batch(const hp_engine* e);                  /* tensorrt::max_batch_size() */
/* The description the engine was created from (topology, outputs, pre-processing, fp32 weights), as pointers INTO the engine, valid
 * while it lives: lets a stream (hp_pipeline_create_ex) replicate an engine that came from a file (ONNX, serialized) - the reference's
 * stream shares ONE engine between its stages (include/hyperpose/stream/stream.hpp:136), the GPU pipeline keeps one per batch in flight. */
int hp_engine_describe(const hp_engine* e, hp_engine_desc* out);
int hp_engine_input_size(const hp_engine* e, int* w, int* h); /* tensorrt::input_size() */

/* tensorrt::inference(std::vector<cv::Mat>) with network-sized frames (src/tensorrt.cpp:436-461): n u8 HWC BGR
 * frames [n,in_h,in_w,3]; the u8->f32 conversion of src/data.cpp:21-51 is fused into the first layer.
 * n > max_batch returns HP_ERR_CAPACITY (the reference throws std::logic_error, :439-443).  The call only
 * ENQUEUES on `stream` (NULL = the engine's own stream); outputs stay in device memory. */
int hp_engine_infer_u8(hp_engine* e, const uint8_t* hwc_bgr, int n, int on_device, void* stream);
/* tensorrt::inference(const std::vector<float>&, size_t) (src/tensorrt.cpp:364-434): n f32 NCHW frames, no
 * scaling / channel swap. */
int hp_engine_infer_f32(hp_engine* e, const float* nchw, int n, int on_device, void* stream);
int hp_engine_synchronize(hp_engine* e);
void* hp_engine_stream(hp_engine* e); /* hipStream_t of the engine */
int hp_engine_set_graph(hp_engine* e, int enable); /* replay the schedule from a captured hipGraph (default on) */
/* parts = 2: every batch of >= 2 frames runs as two half-batches side by side, the second on an internal stream that forks from and joins the
 * call's stream (HP_DTYPE_F32 / F32S engines; an fp16 engine keeps one stream and hp_engine_concurrency() says so) - for a caller that keeps ONE batch in flight, as the reference's synchronous
 * tensorrt::inference does (src/tensorrt.cpp:364-434): the two halves fill each other's idle phases.  Outputs are bit-identical to parts = 1.
 * Callers that overlap several batches on several engines (hp_pipeline_*) keep 1. */
int hp_engine_set_concurrency(hp_engine* e, int parts);
int hp_engine_concurrency(const hp_engine* e);

/* Outputs, sorted by name.  shape[] receives the non-batch dims (C,H,W), dev the fp32 NCHW device buffer
 * [max_batch][C][H][W] of which the first n frames are valid after the last inference completed. */
int hp_engine_num_outputs(const hp_engine* e);
int hp_engine_output(const hp_engine* e, int i, const char** name, int shape[3], const float** dev);
/* Serialized engines: tensorrt::save (include/hyperpose/operator/dnn/tensorrt.hpp:121-123, src/tensorrt.cpp:463-471) and the
 * tensorrt_serialized constructor (include/hyperpose/utility/model.hpp:27-32, src/tensorrt.cpp:225-252).  The file carries the
 * topology, outputs, pre-processing and fp32 weights the engine was created from; max_batch <= 0 keeps the saved one. */
int hp_engine_save(const hp_engine* e, const char* path);
int hp_engine_load(hp_engine** out, const char* path, int max_batch);
int hp_engine_output_to_host(hp_engine* e, int i, int n, float* host); /* synchronises, then D2H */
/* Read back an internal fp16 NHWC tensor as fp32 NCHW [n][C][H][W] (layer-wise parity tests only). */
int hp_engine_debug_tensor(hp_engine* e, int tensor, int n, float* host, int shape[3]);

/* Per-layer device time (ms, averaged over iters) measured with HIP events on the engine stream for batch n:
 * the numbers the roofline report is built from.  flops = 2*MACs of the layer for that batch. */
typedef struct hp_layer_time {
    int32_t layer, op, tile; /* tile = BM*1000+BN for MFMA convs, 0 otherwise */
    float ms;
    double flops, bytes;     /* algorithmic FLOPs and compulsory HBM bytes (inputs + weights + outputs once) */
} hp_layer_time;
int hp_engine_profile(hp_engine* e, int n, int iters, hp_layer_time* out, int cap, int* n_out);
/* Same table, but measured IN SEQUENCE: the whole schedule runs `iters` times in order and every launch records its own begin /
 * end timestamps (hipExtLaunchKernelGGL start / stop events: the numbers rocprofv3's kernel trace reports, no packets added
 * between the kernels), so every kernel sees the cache state it sees in a real inference.  The back-to-back form above
 * re-runs one layer with its weights warm in L2 and reads ~10-15 % faster for the weight-heavy 3x3 layers. */
int hp_engine_profile_sequence(hp_engine* e, int n, int iters, hp_layer_time* out, int cap, int* n_out);
/* Machine time per launch: step k of two engines of the same model launched alternately on their two streams (two instances of the
 * kernel share the GPU as two pipes' kernels do); ms = elapsed / (2 * iters).  tools/profile_layers.py --pair. */
int hp_engine_profile_pair(hp_engine* e, hp_engine* other, int n, int iters, hp_layer_time* out, int cap, int* n_out);

/* ---- built-in topologies (restating hyperpose/Model/<arch>.py) --------------------- */
typedef struct hp_model hp_model;
/* arch: "lw_openpose_mobilenet" (MobilenetDilated + LightWeightOpenPose, backbones.py:177-229,
 * openpose/model/lw_openpose.py), "lw_openpose_vggtiny" (backbones.py:343-391), "openpose_vgg19"
 * (backbones.py:447-509, openpose/model/openpose.py), ... see hp_model_archs(). */
int hp_model_build(hp_model** out, const char* arch, int in_w, int in_h);
void hp_model_destroy(hp_model* m);
const char* hp_model_archs(void);                       /* comma-separated list */
int hp_model_layers(const hp_model* m, const hp_layer** layers, int* n);
int hp_model_outputs(const hp_model* m, const hp_output_desc** outs, int* n);
size_t hp_model_num_weights(const hp_model* m);
int hp_model_preproc(const hp_model* m, float mean[3], float inv_std[3]);
double hp_model_flops_per_frame(const hp_model* m);     /* 2*MACs of all CONV/DWCONV layers */
/* Deterministic synthetic weights (there is no network to fetch the released models): He-normal conv
 * kernels from a counter-based generator keyed by (seed, layer, index), small biases, PReLU slopes 0.25. */
int hp_model_init_weights(const hp_model* m, uint64_t seed, float* blob, size_t n);
/* ONNX import: what nvonnxparser does for dnn::tensorrt(const onnx&, cv::Size, ...) (include/hyperpose/operator/dnn/tensorrt.hpp:53-62,
 * include/hyperpose/utility/model.hpp:23-25, src/tensorrt.cpp:162-223).  The file / buffer is a serialized ONNX ModelProto with ONE
 * input of 3 channels (N,3,H,W, 3,H,W, or N,H,W,3 followed by a Transpose); in_w x in_h is the caller's input size as in the reference
 * (0,0 = take the static size stored in the graph).  Supported operators and how they are lowered onto hp_layer: see the header of
 * hyperpose_amd/csrc/onnx_import.cpp; anything else returns HP_ERR_INVALID with the node and operator named in hp_last_error().
 * The model owns the imported weights: pass weights = NULL to hp_engine_create_from_model, or read them with hp_model_weights. */
int hp_model_from_onnx(hp_model** out, const void* data, size_t size, int in_w, int in_h);
int hp_model_from_onnx_file(hp_model** out, const char* path, int in_w, int in_h);
int hp_model_weights(const hp_model* m, const float** blob, size_t* n); /* blob = NULL for built-in topologies */
int hp_model_input_size(const hp_model* m, int* w, int* h);
/* Convenience: build an engine for a topology with blob weights (NULL = the model's own, imported models only). */
int hp_engine_create_from_model(hp_engine** out, const hp_model* m, int max_batch, double factor, int flip_rb,
                                const float* weights, size_t n_weights);
/* the same with the arithmetic chosen (HP_DTYPE_*); hp_engine_create_from_model is the HP_DTYPE_F16 form */
int hp_engine_create_from_model_dtype(hp_engine** out, const hp_model* m, int max_batch, double factor, int flip_rb,
                                      const float* weights, size_t n_weights, int dtype);
int hp_engine_dtype(const hp_engine* e); /* HP_DTYPE_* of an engine (serialized engines carry theirs) */
/* HP_DTYPE_F32S engines: how many times the engine left the split kernels for the fp32 pipe because an activation did not fit fp16's
 * range (0 or 1: it does not go back); 0 for the other types */
int hp_engine_split_fallbacks(const hp_engine* e);
/* HBM the engine holds for its max_batch, in bytes: bytes[0] activation tensors (with their zero halos), bytes[1] packed weights in every
 * form its kernels read (fragment orders, the Winograd U matrices of an HP_DTYPE_F32 engine ...), bytes[2] the fp32 NCHW network outputs */
int hp_engine_device_bytes(const hp_engine* e, uint64_t bytes[3]);
/* The activation arena of an HP_DTYPE_F32 / F32S engine (tensors of one geometry take turns in the fewest buffers their lifetimes in the schedule
 * allow; HP_NO_ARENA=1 at creation: one allocation per tensor): info = { buffers, tensors living in them, bytes the same tensors would take with one
 * allocation each }.  All zero for engines without an arena. */
int hp_engine_arena_info(const hp_engine* e, uint64_t info[3]);
/* Diagnostic (HP_FIRST_CONV_VERIFY=1 at launch time): the fp32 first-layer kernel re-reads its staged weights and input patch from LDS after computing and
 * compares them with global memory; out = { patch words that differed, weight words that differed, blocks checked, 0 } since the last reset. */
int hp_debug_first_conv_verify(unsigned out[4], int reset);

/* ---- hyperpose::stream on the GPU (reference include/hyperpose/stream/stream.hpp:119-390, src/stream.cpp:60-147): host frames of
 * any size in, humans out, in submission order.  Each submit copies one batch (<= max_batch frames, 8-bit BGR HWC, packed rows) to
 * the device and enqueues resize (keep_ratio = 0: cv::resize; 1: non_scaling_resize + resume_ratio on the way out) -> conv stack ->
 * PAF parser on the stream of the next free engine+parser pair; `n_pipes` batches can be in flight.  Frames in pinned memory
 * (hp_malloc_host) are copied from where they lie, others through a pinned staging buffer.  The network must have the two PAF
 * outputs (conf, paf).  max_frame_bytes bounds width*height*3 of a submitted frame. */
typedef struct hp_pipeline hp_pipeline;
/* which hyperpose::parser the pipeline ends in, with that parser's constructor arguments:
 *   HP_PARSER_PAF     thresh = {conf_thresh, paf_thresh}, res_w / res_h = resolution_size (-1 = default)     paf.hpp:27
 *   HP_PARSER_PPN     thresh = {point_thresh, limb_thresh, nms_thresh}; net_resolution = the engine's input size   proposal_network.hpp:26
 *   HP_PARSER_PIFPAF  thresh = {thresh}; (h, w) = the engine's input size                                    pifpaf.hpp:10 */
enum { HP_PARSER_PAF = 0, HP_PARSER_PPN = 1, HP_PARSER_PIFPAF = 2 };
typedef struct hp_parser_desc {
    int32_t kind;
    float thresh[3];
    int32_t res_w, res_h;
} hp_parser_desc;
int hp_pipeline_create_ex(hp_pipeline** out, const hp_engine_desc* desc, const hp_parser_desc* parser, int n_pipes, int keep_ratio,
                          size_t max_frame_bytes);
/* the PAF form of hp_pipeline_create_ex */
int hp_pipeline_create(hp_pipeline** out, const hp_engine_desc* desc, int n_pipes, int keep_ratio, float conf_thresh,
                       float paf_thresh, size_t max_frame_bytes);
void hp_pipeline_destroy(hp_pipeline* p);
/* HP_ERR_STATE when all pipes are busy (collect first) */
int hp_pipeline_submit(hp_pipeline* p, const uint8_t* const* frames, const int* widths, const int* heights, int n);
/* waits for the OLDEST batch in flight; out[i * cap_per_frame + j], n_out[i] for i < *n_frames */
int hp_pipeline_collect(hp_pipeline* p, hp_human* out, int cap_per_frame, int* n_out, int* n_frames);
int hp_pipeline_in_flight(const hp_pipeline* p);

#ifdef __cplusplus
}
#endif
#endif /* HP_HIP_H */
