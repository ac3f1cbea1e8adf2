// pipeline.cpp — hyperpose::stream on the GPU (reference include/hyperpose/stream/stream.hpp:119-390, src/stream.cpp):
// the reference pushes every frame through four CPU threads and three mutex-guarded queues (resize -> inference ->
// parse -> writer) with a host round trip between each; here a batch of host frames is copied once to the device and
// stays there: H2D copy, cv::resize / non_scaling_resize (resize.hip), the conv stack (engine.cpp) and the PAF parser
// (paf_parser.hip) are enqueued back to back on ONE HIP stream of one of `n_pipes` engine+parser pairs, and only the
// humans come back.  Several batches are in flight (one per pipe); results are returned in submission order, like the
// reference's queues.  resume_ratio is applied on the way out when the aspect ratio was kept.
#include "hp_common.hpp"

#include <cstring>
#include <memory>
#include <vector>

namespace {

struct pipe_t {
    hp_engine* eng = nullptr;
    hp_paf* paf = nullptr;
    hipStream_t s = nullptr;
    hp::host_buf stage;     // pinned staging for frames that arrive in pageable memory
    hp::dev_buf raw, net;   // frames as submitted; frames at network size [max_batch][in_h][in_w][3]
    int n = 0;              // frames in flight (0 = free)
    std::vector<int> w, h;
};

} // namespace

struct hp_pipeline {
    int n_pipes = 0, max_batch = 0, in_w = 0, in_h = 0, keep_ratio = 0;
    size_t max_frame_bytes = 0;
    std::vector<pipe_t> pipes;
    int head = 0, tail = 0, inflight = 0; // ring over the pipes

    ~hp_pipeline()
    {
        for (auto& p : pipes) {
            if (p.s)
                (void)hipStreamSynchronize(p.s);
            if (p.paf)
                hp_paf_destroy(p.paf);
            if (p.eng)
                hp_engine_destroy(p.eng);
        }
    }
};

extern "C" {

int hp_pipeline_create(hp_pipeline** out, const hp_engine_desc* desc, int n_pipes, int keep_ratio, float conf_thresh, float paf_thresh,
    size_t max_frame_bytes)
{
    HP_REQUIRE(out && desc, HP_ERR_INVALID, "hp_pipeline_create: null argument");
    HP_REQUIRE(n_pipes >= 1 && n_pipes <= 16, HP_ERR_INVALID, "hp_pipeline_create: n_pipes %d", n_pipes);
    std::unique_ptr<hp_pipeline> pl(new hp_pipeline());
    pl->n_pipes = n_pipes, pl->max_batch = desc->max_batch, pl->in_w = desc->in_w, pl->in_h = desc->in_h, pl->keep_ratio = keep_ratio;
    const size_t net_frame = (size_t)desc->in_w * desc->in_h * 3;
    pl->max_frame_bytes = std::max(max_frame_bytes, net_frame);
    pl->pipes = std::vector<pipe_t>(n_pipes);
    for (auto& p : pl->pipes) {
        // engine first, parser second: their streams then alternate over the runtime's hardware queues
        HP_TRY(hp_engine_create(&p.eng, desc));
        HP_TRY(hp_paf_create(&p.paf, conf_thresh, paf_thresh, -1, -1, desc->max_batch));
        p.s = (hipStream_t)hp_engine_stream(p.eng);
        HP_TRY(p.stage.alloc((pl->max_frame_bytes + 256) * desc->max_batch)); // frames are packed at 256-byte offsets
        HP_TRY(p.raw.alloc((pl->max_frame_bytes + 256) * desc->max_batch));
        HP_TRY(p.net.alloc(net_frame * desc->max_batch));
        p.w.assign(desc->max_batch, 0), p.h.assign(desc->max_batch, 0);
    }
    // the two feature maps the PAF parser reads, by the reference's convention: outputs sorted by name, "conf" then "paf"
    HP_REQUIRE(hp_engine_num_outputs(pl->pipes[0].eng) == 2, HP_ERR_INVALID,
        "hp_pipeline_create: the PAF pipeline needs a network with exactly two outputs (conf, paf)");
    *out = pl.release();
    return HP_OK;
}

void hp_pipeline_destroy(hp_pipeline* pl) { delete pl; }

int hp_pipeline_in_flight(const hp_pipeline* pl) { return pl ? pl->inflight : 0; }

int hp_pipeline_submit(hp_pipeline* pl, const uint8_t* const* frames, const int* widths, const int* heights, int n)
{
    HP_REQUIRE(pl && frames && widths && heights, HP_ERR_INVALID, "hp_pipeline_submit: null argument");
    HP_REQUIRE(n >= 1 && n <= pl->max_batch, HP_ERR_CAPACITY, "hp_pipeline_submit: batch %d > max_batch %d", n, pl->max_batch);
    HP_REQUIRE(pl->inflight < pl->n_pipes, HP_ERR_STATE, "hp_pipeline_submit: all %d pipes are busy, collect first", pl->n_pipes);
    pipe_t& p = pl->pipes[pl->head];
    size_t off = 0;
    std::vector<size_t> offs(n);
    for (int i = 0; i < n; ++i) {
        HP_REQUIRE(frames[i] && widths[i] > 0 && heights[i] > 0, HP_ERR_INVALID, "hp_pipeline_submit: frame %d is empty", i);
        const size_t bytes = (size_t)widths[i] * heights[i] * 3;
        HP_REQUIRE(bytes <= pl->max_frame_bytes, HP_ERR_CAPACITY, "hp_pipeline_submit: frame %d (%dx%d) exceeds max_frame_bytes %zu", i,
            widths[i], heights[i], pl->max_frame_bytes);
        offs[i] = off;
        // pinned memory (hp_malloc_host / hipHostMalloc) is copied straight from where it lies; anything else through staging
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, frames[i]) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!pinned)
            (void)hipGetLastError();
        const uint8_t* src = frames[i];
        if (!pinned) {
            memcpy(p.stage.as<uint8_t>() + off, frames[i], bytes);
            src = p.stage.as<uint8_t>() + off;
        }
        HP_HIP_TRY(hipMemcpyAsync(p.raw.as<uint8_t>() + off, src, bytes, hipMemcpyHostToDevice, p.s));
        off += (bytes + 255) & ~(size_t)255;
        p.w[i] = widths[i], p.h[i] = heights[i];
    }
    const size_t net_frame = (size_t)pl->in_w * pl->in_h * 3;
    for (int i = 0; i < n; ++i) {
        uint8_t* dst = p.net.as<uint8_t>() + (size_t)i * net_frame;
        const uint8_t* src = p.raw.as<uint8_t>() + offs[i];
        if (pl->keep_ratio)
            HP_TRY(hp_letterbox_u8c3(src, p.w[i], p.h[i], p.w[i] * 3, dst, pl->in_w, pl->in_h, pl->in_w * 3, 0, 0, 0, p.s));
        else
            HP_TRY(hp_resize_u8c3(src, p.w[i], p.h[i], p.w[i] * 3, dst, pl->in_w, pl->in_h, pl->in_w * 3, p.s));
    }
    HP_TRY(hp_engine_infer_u8(p.eng, p.net.as<uint8_t>(), n, 1, p.s));
    const char* name = nullptr;
    int cs[3], ps[3];
    const float *dconf = nullptr, *dpaf = nullptr;
    HP_TRY(hp_engine_output(p.eng, 0, &name, cs, &dconf));
    HP_TRY(hp_engine_output(p.eng, 1, &name, ps, &dpaf));
    HP_TRY(hp_paf_enqueue(p.paf, n, dconf, cs, dpaf, ps, p.s));
    p.n = n;
    pl->head = (pl->head + 1) % pl->n_pipes;
    ++pl->inflight;
    return HP_OK;
}

int hp_pipeline_collect(hp_pipeline* pl, hp_human* out, int cap_per_frame, int* n_out, int* n_frames)
{
    HP_REQUIRE(pl && n_out && n_frames, HP_ERR_INVALID, "hp_pipeline_collect: null argument");
    HP_REQUIRE(pl->inflight > 0, HP_ERR_STATE, "hp_pipeline_collect: nothing in flight");
    pipe_t& p = pl->pipes[pl->tail];
    const int n = p.n;
    const int rc = hp_paf_collect(p.paf, out, cap_per_frame, n_out);
    p.n = 0;
    pl->tail = (pl->tail + 1) % pl->n_pipes;
    --pl->inflight;
    *n_frames = n;
    if (rc == HP_OK && out && pl->keep_ratio)
        for (int i = 0; i < n; ++i) // src/stream.cpp:120-124
            hp_resume_ratio(out + (size_t)i * cap_per_frame, std::min(n_out[i], cap_per_frame), p.w[i], p.h[i], pl->in_w, pl->in_h);
    return rc;
}

} // extern "C"
