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
    hp_ppn* ppn = nullptr;
    hp_pifpaf* pifpaf = nullptr;
    hipStream_t s = nullptr;
    hp::host_buf stage;     // pinned staging for frames that arrive in pageable memory
    hp::dev_buf raw, net;   // frames as submitted; frames at network size [max_batch][in_h][in_w][3]
    int n = 0;              // frames in flight (0 = free)
    std::vector<int> w, h;
};

} // namespace

struct hp_pipeline {
    int n_pipes = 0, max_batch = 0, in_w = 0, in_h = 0, keep_ratio = 0, kind = HP_PARSER_PAF;
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
            if (p.ppn)
                hp_ppn_destroy(p.ppn);
            if (p.pifpaf)
                hp_pifpaf_destroy(p.pifpaf);
            if (p.eng)
                hp_engine_destroy(p.eng);
        }
    }
};

extern "C" {

int hp_pipeline_create(hp_pipeline** out, const hp_engine_desc* desc, int n_pipes, int keep_ratio, float conf_thresh, float paf_thresh,
    size_t max_frame_bytes)
{
    hp_parser_desc pd;
    memset(&pd, 0, sizeof(pd));
    pd.kind = HP_PARSER_PAF;
    pd.thresh[0] = conf_thresh, pd.thresh[1] = paf_thresh;
    pd.res_w = pd.res_h = -1;
    return hp_pipeline_create_ex(out, desc, &pd, n_pipes, keep_ratio, max_frame_bytes);
}

int hp_pipeline_create_ex(hp_pipeline** out, const hp_engine_desc* desc, const hp_parser_desc* parser, int n_pipes, int keep_ratio,
    size_t max_frame_bytes)
{
    HP_REQUIRE(out && desc && parser, HP_ERR_INVALID, "hp_pipeline_create: null argument");
    HP_REQUIRE(n_pipes >= 1 && n_pipes <= 16, HP_ERR_INVALID, "hp_pipeline_create: n_pipes %d", n_pipes);
    HP_REQUIRE(parser->kind == HP_PARSER_PAF || parser->kind == HP_PARSER_PPN || parser->kind == HP_PARSER_PIFPAF, HP_ERR_INVALID,
        "hp_pipeline_create: unknown parser kind %d", parser->kind);
    // HP_DTYPE_F32S is refused here: its guard against values outside fp16's range (|x| > 65504) runs in hp_engine_synchronize /
    // hp_engine_output_to_host, which a pipeline never calls - a batch that overflowed would be parsed and returned as humans with no error
    HP_REQUIRE(desc->dtype != HP_DTYPE_F32S, HP_ERR_INVALID,
        "hp_pipeline_create: HP_DTYPE_F32S engines are for single-stream use (hp_engine_infer_* + hp_engine_synchronize); use HP_DTYPE_F32 or HP_DTYPE_F16 in a pipeline");
    std::unique_ptr<hp_pipeline> pl(new hp_pipeline());
    pl->kind = parser->kind;
    pl->n_pipes = n_pipes, pl->max_batch = desc->max_batch, pl->in_w = desc->in_w, pl->in_h = desc->in_h, pl->keep_ratio = keep_ratio;
    const size_t net_frame = (size_t)desc->in_w * desc->in_h * 3;
    pl->max_frame_bytes = std::max(max_frame_bytes, net_frame);
    pl->pipes = std::vector<pipe_t>(n_pipes);
    for (auto& p : pl->pipes) {
        // engine first, parser second: their streams then alternate over the runtime's hardware queues
        HP_TRY(hp_engine_create(&p.eng, desc));
        if (parser->kind == HP_PARSER_PAF)
            HP_TRY(hp_paf_create(&p.paf, parser->thresh[0], parser->thresh[1], parser->res_w, parser->res_h, desc->max_batch));
        else if (parser->kind == HP_PARSER_PPN) // pose_proposal(net_resolution, point_thresh, limb_thresh, mns_thresh)
            HP_TRY(hp_ppn_create(&p.ppn, desc->in_w, desc->in_h, parser->thresh[0], parser->thresh[1], parser->thresh[2], desc->max_batch));
        else // pifpaf(h, w, thresh)
            HP_TRY(hp_pifpaf_create(&p.pifpaf, desc->in_h, desc->in_w, parser->thresh[0], desc->max_batch));
        p.s = (hipStream_t)hp_engine_stream(p.eng);
        HP_TRY(p.stage.alloc((pl->max_frame_bytes + 256) * desc->max_batch)); // frames are packed at 256-byte offsets
        HP_TRY(p.raw.alloc((pl->max_frame_bytes + 256) * desc->max_batch));
        HP_TRY(p.net.alloc(net_frame * desc->max_batch));
        p.w.assign(desc->max_batch, 0), p.h.assign(desc->max_batch, 0);
    }
    // the feature maps the parsers read, by the reference's convention: outputs sorted by name (src/tensorrt.cpp:405) = the parsers'
    // argument order: PAF (conf, paf), src/paf.cpp:300; PifPaf (paf, pif), src/pifpaf.cpp:7; PoseProposal 7 tensors, src/pose_proposal.cpp:12-20
    const int want = parser->kind == HP_PARSER_PPN ? 7 : 2;
    HP_REQUIRE(hp_engine_num_outputs(pl->pipes[0].eng) == want, HP_ERR_INVALID, "hp_pipeline_create: this parser needs a network with exactly %d outputs, the engine has %d",
        want, hp_engine_num_outputs(pl->pipes[0].eng));
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
    const size_t net_frame = (size_t)pl->in_w * pl->in_h * 3;
    size_t off = 0;
    std::vector<size_t> offs(n);
    std::vector<char> direct(n, 0);
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
        // a frame that already has the network's size needs no geometry: cv::resize to the same size is a copy (src/tensorrt.cpp:446-451)
        // and non_scaling_resize at ratio 1 fills the whole target (src/data.cpp:53-69) -> H2D straight into the network's input slot
        direct[i] = widths[i] == pl->in_w && heights[i] == pl->in_h;
        uint8_t* dst = direct[i] ? p.net.as<uint8_t>() + (size_t)i * net_frame : p.raw.as<uint8_t>() + off;
        // consecutive network-sized frames that are also consecutive in (pinned) host memory go in ONE copy
        int run = 1;
        if (direct[i] && pinned)
            while (i + run < n && widths[i + run] == pl->in_w && heights[i + run] == pl->in_h && frames[i + run] == frames[i] + (size_t)run * net_frame)
                ++run;
        HP_HIP_TRY(hipMemcpyAsync(dst, src, bytes * run, hipMemcpyHostToDevice, p.s));
        for (int r = 0; r < run; ++r) {
            direct[i + r] = direct[i];
            offs[i + r] = off;
            p.w[i + r] = widths[i + r], p.h[i + r] = heights[i + r];
        }
        i += run - 1;
        if (!direct[i] || !pinned) // the slot of the staging / raw buffer is in use until the copy has run
            off += (bytes + 255) & ~(size_t)255;
    }
    for (int i = 0; i < n; ++i) {
        if (direct[i])
            continue;
        uint8_t* dst = p.net.as<uint8_t>() + (size_t)i * net_frame;
        const uint8_t* src = p.raw.as<uint8_t>() + offs[i];
        if (pl->keep_ratio)
            HP_TRY(hp_letterbox_u8c3(src, p.w[i], p.h[i], p.w[i] * 3, dst, pl->in_w, pl->in_h, pl->in_w * 3, 0, 0, 0, p.s));
        else
            HP_TRY(hp_resize_u8c3(src, p.w[i], p.h[i], p.w[i] * 3, dst, pl->in_w, pl->in_h, pl->in_w * 3, p.s));
    }
    HP_TRY(hp_engine_infer_u8(p.eng, p.net.as<uint8_t>(), n, 1, p.s));
    const char* name = nullptr;
    if (pl->kind == HP_PARSER_PAF) {
        int cs[3], ps[3];
        const float *dconf = nullptr, *dpaf = nullptr;
        HP_TRY(hp_engine_output(p.eng, 0, &name, cs, &dconf));
        HP_TRY(hp_engine_output(p.eng, 1, &name, ps, &dpaf));
        HP_TRY(hp_paf_enqueue(p.paf, n, dconf, cs, dpaf, ps, p.s));
    } else if (pl->kind == HP_PARSER_PPN) {
        const float* d[7];
        int sh[7][3];
        for (int t = 0; t < 7; ++t)
            HP_TRY(hp_engine_output(p.eng, t, &name, sh[t], &d[t]));
        // the edge tensor leaves the network as [E*nh*nw, gh, gw] (pose_proposal/model.py:104-109): nh = nw = sqrt(C / E) with E = 17 limbs
        const int E = 17, nn = sh[6][0] / E;
        int nh = 1;
        while (nh * nh < nn)
            ++nh;
        HP_REQUIRE(nh * nh * E == sh[6][0], HP_ERR_INVALID, "hp_pipeline: edge output has %d channels, expected 17 * k * k", sh[6][0]);
        const int es[5] = { E, nh, nh, sh[6][1], sh[6][2] };
        HP_TRY(hp_ppn_enqueue(p.ppn, n, d, sh[0], es, p.s));
    } else {
        int a[3], b[3];
        const float *dpaf = nullptr, *dpif = nullptr;
        HP_TRY(hp_engine_output(p.eng, 0, &name, a, &dpaf));
        HP_TRY(hp_engine_output(p.eng, 1, &name, b, &dpif));
        HP_TRY(hp_pifpaf_enqueue(p.pifpaf, n, dpaf, dpif, b[1], b[2], p.s));
    }
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
    const int rc = pl->kind == HP_PARSER_PAF ? hp_paf_collect(p.paf, out, cap_per_frame, n_out)
        : pl->kind == HP_PARSER_PPN        ? hp_ppn_collect(p.ppn, out, cap_per_frame, n_out)
                                           : hp_pifpaf_collect(p.pifpaf, out, cap_per_frame, n_out);
    p.n = 0;
    pl->tail = (pl->tail + 1) % pl->n_pipes;
    --pl->inflight;
    *n_frames = n;
    if ((rc == HP_OK || rc == HP_ERR_CAPACITY) && out && pl->keep_ratio) // (capacity: that frame's list is cut, everything returned is valid)
        for (int i = 0; i < n; ++i) // src/stream.cpp:120-124
            hp_resume_ratio(out + (size_t)i * cap_per_frame, std::min(n_out[i], cap_per_frame), p.w[i], p.h[i], pl->in_w, pl->in_h);
    return rc;
}

} // extern "C"
