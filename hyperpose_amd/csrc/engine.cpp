// engine.cpp — hyperpose::dnn engine on gfx950 (replaces reference src/tensorrt.cpp).
//
// The reference hands an exported graph to TensorRT (create_{uff,onnx,serialized}_engine, src/tensorrt.cpp:121-252),
// allocates one device buffer per binding for max_batch (:255-316) and, per call, does H2D of an f32 NCHW
// batch, executeV2, and one D2H per (output x image) (:364-434).  Here the graph is a static layer list
// (include/hp_hip.h: hp_layer) executed as a fixed schedule of hand-written HIP kernels (conv_kernels.hip)
// on one stream, optionally replayed from a captured hipGraph:
//   * input stays u8 HWC in HBM; the u8->f32 / BGR->RGB / mean-std step of src/data.cpp:21-51 is folded into the
//     first convolution's load;
//   * activations are NHWC fp16 with fp32 MFMA accumulation, concat is a channel offset into a shared buffer;
//   * heads write fp32 NCHW straight from the conv epilogue into the buffers the parser kernels read —
//     feature maps never cross PCIe.
#include "conv_fp32.hpp"
#include "conv_kernels.hpp"
#include "hp_common.hpp"

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

struct tensor_info {
    bool defined = false;
    bool elided = false;     // produced and consumed inside one fused launch: never materialised in HBM
    bool unwritten = false;  // a network output nobody else reads: only its fp32 NCHW form is produced
    int H = 0, W = 0, C = 0; // C = total channels written
    int cs = 0;              // channel stride of the buffer
    int P = 0;               // zero halo (pixels) around every image: the largest padding any consumer needs
    hp::dev_buf buf;         // the tensor's own allocation, or
    void* shared = nullptr;  // ... a buffer of the engine's activation arena that tensors of the same geometry take turns in (hp_engine::arena)
    int arena_slot = -1;
    template <typename T>
    T* base() const { return static_cast<T*>(shared ? shared : buf.p); }
    hp::tview view(int coff) const
    {
        hp::tview v;
        const int wp = W + 2 * P;
        v.p = base<__half>() + ((size_t)P * wp + P) * cs;
        v.cs = cs, v.coff = coff, v.wp = wp, v.img = (H + 2 * P) * wp;
        return v;
    }
    // rows one image takes in an fp32 buffer: H + 2 P, made EVEN when there is a halo - the images of a batch then form one tall image whose
    // separator rows are zeros and whose 2 x 2 Winograd tiles fall on the same rows in every image (conv32_winograd.hip, "tall" form)
    int rows32() const { return P > 0 ? (H + 2 * P + 1) / 2 * 2 : H; }
    hp::tview32 view32(int coff) const // the same geometry with 4-byte elements (HP_DTYPE_F32 engines)
    {
        hp::tview32 v;
        const int wp = W + 2 * P;
        v.p = base<float>() + ((size_t)P * wp + P) * cs;
        v.cs = cs, v.coff = coff, v.wp = wp, v.img = rows32() * wp;
        return v;
    }
};

struct out_info {
    std::string name;
    int tensor, coff, channels, act;
    int H, W;       // source tensor size
    hp::out_xform x{}; // conversion parameters; x.out_h / x.out_w / (C / shuffle^2) is the reported shape
    int out_c() const { return x.C / (x.shuffle * x.shuffle); }
    bool plain() const { return x.shuffle == 1 && x.group == 0 && x.out_h == H && x.out_w == W && x.scale == 1.f && x.grid == 0; }
    int fused_layer = -1; // layer whose epilogue writes it, or -1 -> conversion kernel
    std::unique_ptr<hp::dev_buf> buf;
};

struct step {
    int layer;
    int op;
    bool first = false; // direct 3-channel conv
    hp::conv_params cp{};
    hp::first_conv_params fp{};
    hp::dw_params dp{};
    hp::pool_params pp{};
    hp::sep_params sp{}; // op == OP_SEPCONV: depthwise layer `layer` fused with the pointwise layer `layer + 1`
    hp::sep_params sp2{}; // ... and, when `sep_pair`, the NEXT separable block (layers `layer + 2`, `layer + 3`) in the same launch
    bool sep_pair = false;
    hp::head_params hp_{}; // op == OP_MLPHEAD: 1x1 conv `layer` (-> 512, relu) fused with the 1x1 conv `layer + 1`
    hp::head_params hp2_{}; // ... and, when `paired`, the sibling head on the same input (layers `layer + 2`, `layer + 3`)
    bool paired = false;
    hp::chain_params ch{}; // op == OP_CHAIN: [1x1 ->] 3x3 -> 3x3 on 128 channels in one launch (conv_chain.hip), layers `layer` ..
    hp::bneck_params bn{}; // op == OP_BNECK: [3x3 ->] expansion 1x1 + shortcut [-> the next block's reduction 1x1] (conv_bottleneck.hip)
    // HP_DTYPE_F32 engines (conv_fp32.hip): one launch per layer, `op` = the layer's op, `f32` set
    bool f32 = false;
    hp::conv32_params cp32{};
    hp::first_conv32_params fp32{};
    hp::dw32_params dp32{};
    hp::pool32_params pp32{};
    bool wino = false;     // HP_DTYPE_F32: a 3 x 3 stride-1 layer on conv32_winograd_kernel (cp32.w_wino)
    bool head32 = false;   // HP_DTYPE_F32: 1 x 1 128 -> HID -> 1 x 1 HID -> C2 in one launch (conv32_head.hip): cp32 = the second layer's epilogue, hh = the first layer
    hp::head32_hidden hh{};
    int cin_split = 0;     // fp32 engines: input channels as conv32_direct_kernel reads them (whole chunks), 0 = the layer stays on conv32_kernel
    int n_layers = 1;      // consecutive layers this step covers
    double flops = 0, bytes = 0; // per frame
};
constexpr int OP_SEPCONV = 100; // schedule-only op codes (not part of the hp_layer ABI)
constexpr int OP_MLPHEAD = 101;
constexpr int OP_CHAIN = 102;
constexpr int OP_BNECK = 103;

void same_pad(int in, int k, int stride, int dil, int& out, int& pad_before)
{
    // TF "SAME": out = ceil(in / stride); pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0); extra at the end
    out = (in + stride - 1) / stride;
    const int total = std::max((out - 1) * stride + (k - 1) * dil + 1 - in, 0);
    pad_before = total / 2;
}

} // namespace

struct hp_engine {
    int in_w = 0, in_h = 0, max_batch = 0;
    int dtype = HP_DTYPE_F16; // HP_DTYPE_F32: fp32 storage and arithmetic (the reference's data_type::kFLOAT), conv_fp32.hip;
                              // HP_DTYPE_F32S: the same engine with the dense layers' products on the fp16 pipe (conv32_direct.hip)
    bool is_f32() const { return dtype == HP_DTYPE_F32 || dtype == HP_DTYPE_F32S; }
    // which engines may run a batch as two half-batches (hp_engine_set_concurrency): the fp32 engines (their steps take a frame offset)
    bool halves_ok() const { return is_f32(); }
    // HP_DTYPE_F32S: a pinned host word the split kernels OR into when an activation exceeds fp16's range (|x| > 65504); once seen
    // (hp_engine_synchronize / hp_engine_inference) the engine runs conv32_kernel instead - exact fp32 products, any range
    hp::host_buf ovf_flag;
    bool split_off = false;
    int split_fallbacks = 0;
    struct { const void* input = nullptr; size_t frame_bytes = 0; int n = 0, on_device = 0, kind = 0; void* stream = nullptr; } last; // the newest hp_engine_infer_* call
    bool split_overflowed() const { return dtype == HP_DTYPE_F32S && !split_off && ovf_flag.p && *static_cast<volatile unsigned*>(ovf_flag.p) != 0; }
    int leave_split(); // stop using conv32_direct_kernel: drop the captured graphs (they hold its launches), count the event
    bool dbg_conv = false, dbg_bn = false, dbg_chain = false, dbg_sep = false; // HP_*_DBG block timelines, read once at creation
    double factor = 1.0 / 255;
    int flip_rb = 1;
    float mean[3] = { 0, 0, 0 }, inv_std[3] = { 1, 1, 1 };
    std::vector<hp_layer> layers;
    std::vector<hp_output_desc> out_descs; // as given (hp_engine_save)
    std::vector<float> weights_blob;       // as given (hp_engine_save)
    std::vector<std::unique_ptr<tensor_info>> tensors;
    std::vector<out_info> outputs; // sorted by name
    std::vector<step> steps;
    std::vector<std::unique_ptr<hp::dev_buf>> weight_bufs;
    // Activation arena (HP_DTYPE_F32 / F32S engines; HP_NO_ARENA=1 gives every tensor its own allocation, as before round 6).  TensorRT re-uses
    // activation memory between layers (the reference's engine owns only the binding buffers, src/tensorrt.cpp:255-316); here tensors of ONE
    // geometry class - same H x W, halo, channel stride and channel count, so that the zero halo and the zero pad channels of a buffer stay valid
    // for every tenant - take turns in the minimum number of buffers their lifetimes in the schedule need (interval colouring in layer order).
    struct arena_buf {
        std::unique_ptr<hp::dev_buf> mem;
        int tenants = 0;
    };
    std::vector<arena_buf> arena;
    size_t arena_private_bytes = 0; // what the same tensors took with one allocation each
    hp::dev_buf in_stage; // staging for host inputs
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // hp_engine_set_concurrency(2): a batch runs as two half-batches, the second on `stream2` (fork / join by events; inside a captured graph:
    // two parallel branches) - what a caller with ONE batch in flight gets instead of a second pipe (fp32 engines; frames are independent)
    int parts = 1;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // captured graphs keyed by (n, input pointer, kind)
    struct graph_key {
        int n;
        const void* ptr;
        int kind;
        int b0; // first frame of the range the graph covers (hp_engine_set_concurrency(2): one graph per half-batch, launched on its own stream)
        bool operator<(const graph_key& o) const { return std::tie(n, ptr, kind, b0) < std::tie(o.n, o.ptr, o.kind, o.b0); }
    };
    std::map<graph_key, hipGraphExec_t> graphs;
    bool use_graph = true;

    ~hp_engine()
    {
        for (auto& g : graphs)
            (void)hipGraphExecDestroy(g.second);
        if (ev0)
            (void)hipEventDestroy(ev0);
        if (ev1)
            (void)hipEventDestroy(ev1);
        if (ev_fork)
            (void)hipEventDestroy(ev_fork);
        if (ev_join)
            (void)hipEventDestroy(ev_join);
        if (stream2)
            (void)hipStreamDestroy(stream2);
        if (stream)
            (void)hipStreamDestroy(stream);
    }

    int build(const hp_engine_desc* d);
    // host_src != nullptr (eager launches of a host batch): every range copies ITS frames to the device on ITS stream first (frame_bytes each)
    int enqueue(const uint8_t* u8, const float* f32, int n, hipStream_t s, const void* host_src = nullptr, size_t frame_bytes = 0);
    int run_step(step& st, const uint8_t* u8, const float* f32, int n, hipStream_t s, int b0 = 0);
    int enqueue_range(const uint8_t* u8, const float* f32, int b0, int n, hipStream_t s);
};

int hp_engine::build(const hp_engine_desc* d)
{
    HP_REQUIRE(d->in_w > 0 && d->in_h > 0 && d->max_batch >= 1, HP_ERR_INVALID, "engine: bad input size / batch");
    HP_REQUIRE(d->layers && d->n_layers > 0 && d->weights, HP_ERR_INVALID, "engine: no layers / weights");
    in_w = d->in_w, in_h = d->in_h, max_batch = d->max_batch, factor = d->factor, flip_rb = d->flip_rb;
    HP_REQUIRE(d->dtype == HP_DTYPE_F16 || d->dtype == HP_DTYPE_F32 || d->dtype == HP_DTYPE_F32S, HP_ERR_INVALID,
        "engine: dtype %d is none of HP_DTYPE_F16 / HP_DTYPE_F32 / HP_DTYPE_F32S", d->dtype);
    dtype = d->dtype;
    const bool f32 = is_f32();
    unsigned* ovf_dev = nullptr;
    if (dtype == HP_DTYPE_F32S) {
        HP_TRY(ovf_flag.alloc(64));
        memset(ovf_flag.p, 0, 64);
        HP_HIP_TRY(hipHostGetDevicePointer((void**)&ovf_dev, ovf_flag.p, 0));
    }
    dbg_conv = getenv("HP_CONV_DBG") != nullptr, dbg_bn = getenv("HP_BN_DBG") != nullptr;
    dbg_chain = getenv("HP_CHAIN_DBG") != nullptr, dbg_sep = getenv("HP_SEP_DBG") != nullptr;
    for (int c = 0; c < 3; ++c)
        mean[c] = d->mean[c], inv_std[c] = d->inv_std[c];
    layers.assign(d->layers, d->layers + d->n_layers);
    out_descs.assign(d->outputs, d->outputs + (d->outputs ? d->n_outputs : 0));
    weights_blob.assign(d->weights, d->weights + d->n_weights);

    // ---- pass 1: tensor shapes
    int max_id = 0;
    for (const auto& L : layers)
        max_id = std::max({ max_id, L.in, L.out, L.res });
    HP_REQUIRE(max_id < (1 << 16) && d->n_layers < (1 << 16), HP_ERR_INVALID, "engine: %d layers / tensor id %d: not a plausible network", d->n_layers, max_id);
    for (size_t i = 0; i < layers.size(); ++i) {
        const hp_layer& L = layers[i];
        HP_REQUIRE(L.cin > 0 && L.cout > 0 && L.cin <= (1 << 16) && L.cout <= (1 << 16) && L.kh >= 0 && L.kh <= 31 && L.kw >= 0 && L.kw <= 31
                && L.stride >= 1 && L.stride <= 64 && L.dil >= 0 && L.dil <= 64,
            HP_ERR_INVALID, "layer %zu: implausible channel count / geometry", i);
        // channel offsets index device buffers: a negative one walks out of the front of the tensor
        HP_REQUIRE(L.in_coff >= 0 && L.in_coff <= (1 << 16) && L.out_coff >= 0 && L.out_coff + L.cout <= (1 << 16), HP_ERR_INVALID,
            "layer %zu: channel offsets in %d / out %d outside [0, 65536)", i, L.in_coff, L.out_coff);
        HP_REQUIRE(L.res >= -1 && L.res_before_act >= 0 && L.res_before_act <= 1, HP_ERR_INVALID, "layer %zu: bad residual fields", i);
        if (L.pad_explicit)
            for (int k = 0; k < 4; ++k)
                HP_REQUIRE(L.pad[k] >= 0 && L.pad[k] <= 64, HP_ERR_INVALID, "layer %zu: pad[%d] = %d outside [0, 64]", i, k, L.pad[k]);
        // a layer may read one channel slice of a concat buffer and write another, but never the channels it reads (the
        // blocks of one launch would race) - neither through its input nor through its residual
        if (L.in == L.out)
            HP_REQUIRE(L.in_coff + L.cin <= L.out_coff || L.out_coff + L.cout <= L.in_coff, HP_ERR_INVALID,
                "layer %zu reads channels [%d,%d) and writes [%d,%d) of the same tensor %d", i, L.in_coff, L.in_coff + L.cin, L.out_coff,
                L.out_coff + L.cout, L.out);
        if (L.res >= 0 && L.res == L.out)
            HP_REQUIRE(L.cout <= L.out_coff, HP_ERR_INVALID, "layer %zu adds channels [0,%d) of tensor %d while writing [%d,%d) of it", i, L.cout, L.res,
                L.out_coff, L.out_coff + L.cout);
    }
    tensors.resize(max_id + 1);
    for (auto& t : tensors)
        t = std::make_unique<tensor_info>();
    tensors[0]->defined = true, tensors[0]->H = in_h, tensors[0]->W = in_w, tensors[0]->C = 3;
    struct geo {
        int OH, OW, pt, pl;
    };
    std::vector<geo> geos(layers.size());
    for (size_t i = 0; i < layers.size(); ++i) {
        const hp_layer& L = layers[i];
        HP_REQUIRE(L.in >= 0 && L.in <= max_id && tensors[L.in]->defined, HP_ERR_INVALID, "layer %zu reads undefined tensor %d", i, L.in);
        HP_REQUIRE(L.out > 0, HP_ERR_INVALID, "layer %zu: bad output tensor %d", i, L.out);
        HP_REQUIRE(L.op == HP_OP_CONV || L.op == HP_OP_DWCONV || L.op == HP_OP_MAXPOOL || L.op == HP_OP_UPSAMPLE, HP_ERR_INVALID, "layer %zu: unknown op %d", i, L.op);
        if (L.op == HP_OP_UPSAMPLE)
            HP_REQUIRE(L.stride >= 1 && L.stride <= 16 && (L.kh == 0 || L.kh == 1) && L.in != 0 && L.res < 0 && L.act == HP_ACT_NONE, HP_ERR_INVALID,
                "layer %zu: up-sampling takes an integer scale 1..16 in `stride`, kh = 0 (nearest) or 1 (bilinear), a feature-map input and no activation / residual", i);
        else
            HP_REQUIRE(L.stride >= 1 && L.kh >= 1 && L.kw >= 1 && L.dil >= 1, HP_ERR_INVALID, "layer %zu: bad geometry", i);
        const tensor_info& ti = *tensors[L.in];
        HP_REQUIRE(L.in_coff >= 0 && L.in_coff + L.cin <= ti.C, HP_ERR_INVALID, "layer %zu reads channels [%d,%d) of a %d-channel tensor", i, L.in_coff, L.in_coff + L.cin, ti.C);
        geo g;
        if (L.op == HP_OP_UPSAMPLE) {
            g.OH = ti.H * L.stride, g.OW = ti.W * L.stride, g.pt = g.pl = 0;
        } else if (L.pad_explicit) {
            HP_REQUIRE(L.pad[0] >= 0 && L.pad[1] >= 0 && L.pad[2] >= 0 && L.pad[3] >= 0, HP_ERR_INVALID, "layer %zu: negative padding", i);
            g.pt = L.pad[0], g.pl = L.pad[1];
            g.OH = (ti.H + L.pad[0] + L.pad[2] - ((L.kh - 1) * L.dil + 1)) / L.stride + 1;
            g.OW = (ti.W + L.pad[1] + L.pad[3] - ((L.kw - 1) * L.dil + 1)) / L.stride + 1;
            HP_REQUIRE(g.OH >= 1 && g.OW >= 1, HP_ERR_INVALID, "layer %zu: empty output", i);
        } else {
            same_pad(ti.H, L.kh, L.stride, L.dil, g.OH, g.pt);
            same_pad(ti.W, L.kw, L.stride, L.dil, g.OW, g.pl);
        }
        geos[i] = g;
        tensor_info& to = *tensors[L.out];
        if (!to.defined)
            to.defined = true, to.H = g.OH, to.W = g.OW;
        HP_REQUIRE(to.H == g.OH && to.W == g.OW, HP_ERR_INVALID, "layer %zu: writers of tensor %d disagree on its size", i, L.out);
        to.C = std::max(to.C, L.out_coff + L.cout);
        if (L.res >= 0) {
            HP_REQUIRE(tensors[L.res]->defined && tensors[L.res]->H == g.OH && tensors[L.res]->W == g.OW && tensors[L.res]->C >= L.cout,
                HP_ERR_INVALID, "layer %zu: residual tensor %d does not match the output", i, L.res);
        }
        if (L.op != HP_OP_CONV)
            HP_REQUIRE(L.cin == L.cout, HP_ERR_INVALID, "layer %zu: depthwise/pool need cin == cout", i);
        if (L.op != HP_OP_MAXPOOL && L.op != HP_OP_UPSAMPLE && L.in != 0) {
            // halo this consumer needs on its input: SAME padding before / after in both dimensions
            const int pb_y = std::max((g.OH - 1) * L.stride + (L.kh - 1) * L.dil + 1 - ti.H - g.pt, 0);
            const int pb_x = std::max((g.OW - 1) * L.stride + (L.kw - 1) * L.dil + 1 - ti.W - g.pl, 0);
            tensors[L.in]->P = std::max({ tensors[L.in]->P, g.pt, g.pl, pb_y, pb_x });
        }
    }
    // ---- separable blocks: a depthwise layer whose only consumer is the next layer, a plain 1x1 convolution, runs
    // as ONE launch (sepconv_kernel) and its output tensor is never materialised.  HP_NO_FUSE=1 keeps the two launches.
    std::vector<char> fuse_with_next(layers.size(), 0);
    if (!getenv("HP_NO_FUSE") && !f32) {
        for (size_t i = 0; i + 1 < layers.size(); ++i) {
            const hp_layer &A = layers[i], &Bn = layers[i + 1];
            if (A.op != HP_OP_DWCONV || A.kh != 3 || A.kw != 3 || A.in == 0 || A.out_coff != 0 || A.in_coff % 8)
                continue;
            if (Bn.op != HP_OP_CONV || Bn.kh != 1 || Bn.kw != 1 || Bn.stride != 1 || Bn.in != A.out || Bn.in_coff != 0 || Bn.cin != A.cout
                || Bn.res >= 0 || Bn.out == A.out || Bn.cout % 8 || Bn.out_coff % 8)
                continue;
            if ((A.act != HP_ACT_RELU && A.act != HP_ACT_RELU6) || Bn.act == HP_ACT_SIGMOID || Bn.act == HP_ACT_SOFTPLUS)
                continue; // the fused kernel evaluates the depthwise activation as one clamp
            if (tensors[A.out]->C != A.cout)
                continue;
            {   // the fused kernel is configured for TF-SAME geometry: explicit pads must reproduce it, the 1x1 half must not pad at all
                int oh, ow, pt, pl;
                same_pad(tensors[A.in]->H, 3, A.stride, A.dil, oh, pt);
                same_pad(tensors[A.in]->W, 3, A.stride, A.dil, ow, pl);
                if (geos[i].OH != oh || geos[i].OW != ow || geos[i].pt != pt || geos[i].pl != pl)
                    continue;
                if (geos[i + 1].OH != oh || geos[i + 1].OW != ow || geos[i + 1].pt != 0 || geos[i + 1].pl != 0)
                    continue;
            }
            bool sole = true;
            for (size_t j = 0; j < layers.size(); ++j)
                if (j != i + 1 && (layers[j].in == A.out || layers[j].res == A.out || (j != i && layers[j].out == A.out)))
                    sole = false;
            for (int o = 0; o < d->n_outputs; ++o)
                if (d->outputs[o].tensor == A.out || d->outputs[o].tensor == Bn.out)
                    sole = false; // network outputs keep the generic epilogue (fp32 NCHW copy)
            if (!sole)
                continue;
            const int cout_pad = round_up(Bn.cout, 128);
            const int variant = hp::sepconv_variant_for(A.cin, cout_pad, A.stride, A.dil, Bn.cout);
            if (!variant || (Bn.cout <= 64 && variant != 7))
                continue; // (<= 64 output channels idle half of the general fused kernels' wavefronts - slower than two launches - except in the dedicated 32-channel form)
            fuse_with_next[i] = 1;
            tensors[A.out]->elided = true;
        }
    }
    // ---- fp32 engines: the same pairing for conv32_direct_kernel's depthwise-fused forms (stride 1, dilation 1 | 2, whole 64-channel chunks;
    // whether the 1 x 1 half takes them is decided where its parameters are known: pass 2).  HP_NO_FUSE=1 / HP_NO_FUSE32=1 keep two launches.
    // Default: HP_DTYPE_F32S only (HP_FUSE32=1 fuses on the fp32 pipe too: tests).  Measured on LW-OpenPose @ 8 x 46 x 54, us per batch alone |
    // with a second stream, fused -> two launches: split 1528 | 1077 -> 1514 | 1079 (nothing lost, one 40 MB tensor per block not allocated);
    // fp32 pipe 2880 | 2294 -> 2671 | 1984: conv32_direct_kernel's fused forms compute the depthwise tile once per 128-channel block of the 1 x 1
    // layer (4 x at 512 outputs) between two barriers, the MFMA pipe idle meanwhile - 194 us for dw + 512 -> 512 against 24.5 + 100.  A second
    // attempt in round 5 - one block per pixel tile owning ALL 512 outputs (eight wavefronts, 16-channel chunks double-buffered, the depthwise
    // reads hidden under the chunk's 32 MFMAs, one barrier per chunk) - computed the depthwise tile once and still ran 161 us alone / 108 with
    // a second stream against 125 / 107 for the two launches (profiles/r05_ab_layers_f32_sep_kernel.txt): one block of eight wavefronts per CU
    // moves in step with its own barrier, where conv32_kernel's four independent blocks per CU fill each other's gaps.  Not adopted, removed.
    std::vector<char> fuse32_with_next(layers.size(), 0);
    if (f32 && !getenv("HP_NO_FUSE") && !getenv("HP_NO_FUSE32") && (dtype == HP_DTYPE_F32S || getenv("HP_FUSE32"))) {
        for (size_t i = 0; i + 1 < layers.size(); ++i) {
            const hp_layer &A = layers[i], &Bn = layers[i + 1];
            if (A.op != HP_OP_DWCONV || A.kh != 3 || A.kw != 3 || A.in == 0 || A.out_coff != 0 || A.in_coff % 4 || A.stride != 1 || (A.dil != 1 && A.dil != 2)
                || A.cin % 64 || A.res >= 0)
                continue;
            if (A.act != HP_ACT_NONE && A.act != HP_ACT_RELU && A.act != HP_ACT_RELU6 && A.act != HP_ACT_LEAKY)
                continue;
            if (Bn.op != HP_OP_CONV || Bn.kh != 1 || Bn.kw != 1 || Bn.stride != 1 || Bn.in != A.out || Bn.in_coff != 0 || Bn.cin != A.cout || Bn.out == A.out
                || Bn.res == A.out)
                continue;
            if (tensors[A.out]->C != A.cout || geos[i].OH != tensors[A.in]->H || geos[i].OW != tensors[A.in]->W || geos[i].pt != A.dil || geos[i].pl != A.dil
                || geos[i + 1].OH != geos[i].OH || geos[i + 1].OW != geos[i].OW || geos[i + 1].pt || geos[i + 1].pl)
                continue; // TF-SAME geometry of a stride-1 3 x 3: the map keeps its size, the padding is the dilation; the 1 x 1 half does not pad
            bool sole = true;
            for (size_t j = 0; j < layers.size(); ++j)
                if (j != i + 1 && (layers[j].in == A.out || layers[j].res == A.out || (j != i && layers[j].out == A.out)))
                    sole = false;
            for (int o = 0; o < d->n_outputs; ++o)
                if (d->outputs[o].tensor == A.out)
                    sole = false;
            if (!sole)
                continue;
            // the 1 x 1 half as pass 2 will describe it, to ask the kernel family whether a fused form exists for its size
            hp::conv32_params q{};
            q.B = max_batch, q.H = q.OH = geos[i].OH, q.W = q.OW = geos[i].OW, q.Cin = Bn.cin, q.Cout = Bn.cout, q.Cout_pad = round_up(Bn.cout, 64);
            q.KH = q.KW = 1, q.stride = 1, q.dil = 1, q.pad_t = q.pad_l = 0, q.npix = max_batch * q.OH * q.OW;
            // (an HP_DTYPE_F32S engine must also be able to run the block on the fp32 pipe: where it goes when a value leaves fp16's range)
            if (!hp::conv32_dw_fusable(q, false, A.dil) || (dtype == HP_DTYPE_F32S && !hp::conv32_dw_fusable(q, true, A.dil)))
                continue;
            fuse32_with_next[i] = 1;
            tensors[A.out]->elided = true;
        }
    }
    // ---- two-layer heads: 1x1 K1 -> 512 (relu) whose only consumer is the next layer, a 1x1 512 -> <= 64 channels
    std::vector<char> head_with_next(layers.size(), 0);
    if (!getenv("HP_NO_FUSE") && !f32) {
        for (size_t i = 0; i + 1 < layers.size(); ++i) {
            const hp_layer &A = layers[i], &Bn = layers[i + 1];
            if (A.op != HP_OP_CONV || A.kh != 1 || A.kw != 1 || A.stride != 1 || A.in == 0 || A.res >= 0 || A.out_coff != 0 || A.in_coff % 8
                || (A.act != HP_ACT_RELU && A.act != HP_ACT_RELU6))
                continue;
            if (Bn.op != HP_OP_CONV || Bn.kh != 1 || Bn.kw != 1 || Bn.stride != 1 || Bn.in != A.out || Bn.in_coff != 0 || Bn.cin != A.cout
                || Bn.res >= 0 || Bn.out == A.out || Bn.act == HP_ACT_SIGMOID || Bn.act == HP_ACT_SOFTPLUS)
                continue;
            if (tensors[A.out]->C != A.cout || !hp::mlp_head_variant(A.cin, A.cout, Bn.cout))
                continue;
            if (geos[i].OH != tensors[A.in]->H || geos[i].OW != tensors[A.in]->W || geos[i].pt || geos[i].pl || geos[i + 1].OH != geos[i].OH
                || geos[i + 1].OW != geos[i].OW || geos[i + 1].pt || geos[i + 1].pl)
                continue; // a padded 1x1 (ONNX pads) changes the map size: the fused head assumes it does not
            if (i > 0 && fuse_with_next[i - 1])
                continue; // already the pointwise half of a separable block
            bool sole = true;
            for (size_t j = 0; j < layers.size(); ++j)
                if (j != i + 1 && (layers[j].in == A.out || layers[j].res == A.out || (j != i && layers[j].out == A.out)))
                    sole = false;
            for (int o = 0; o < d->n_outputs; ++o)
                if (d->outputs[o].tensor == A.out)
                    sole = false;
            if (!sole)
                continue;
            head_with_next[i] = 1;
            tensors[A.out]->elided = true;
        }
    }
    // ---- HP_DTYPE_F32: the same two-layer heads for conv32_head_kernel (1 x 1 128 -> HID, ReLU family, sole consumer a 1 x 1 HID -> <= 64
    // channels; the hidden tensor stays in registers).  HP_NO_FUSE=1 / HP_NO_HEAD32=1 keep two launches.
    std::vector<char> head32_with_next(layers.size(), 0);
    if (dtype == HP_DTYPE_F32 && !getenv("HP_NO_FUSE") && !getenv("HP_NO_HEAD32")) {
        for (size_t i = 0; i + 1 < layers.size(); ++i) {
            const hp_layer &A = layers[i], &Bn = layers[i + 1];
            if (A.op != HP_OP_CONV || A.kh != 1 || A.kw != 1 || A.stride != 1 || A.in == 0 || A.res >= 0 || A.out_coff != 0 || A.in_coff % 4
                || (A.act != HP_ACT_NONE && A.act != HP_ACT_RELU && A.act != HP_ACT_RELU6 && A.act != HP_ACT_LEAKY))
                continue;
            if (Bn.op != HP_OP_CONV || Bn.kh != 1 || Bn.kw != 1 || Bn.stride != 1 || Bn.in != A.out || Bn.in_coff != 0 || Bn.cin != A.cout || Bn.res >= 0
                || Bn.out == A.out || Bn.act == HP_ACT_SIGMOID || Bn.act == HP_ACT_SOFTPLUS)
                continue;
            if (tensors[A.out]->C != A.cout || !hp::conv32_head_ok(A.cin, A.cout, Bn.cout) || A.in_coff + A.cin > round_up(tensors[A.in]->C, 32))
                continue;
            if (geos[i].OH != tensors[A.in]->H || geos[i].OW != tensors[A.in]->W || geos[i].pt || geos[i].pl || geos[i + 1].OH != geos[i].OH
                || geos[i + 1].OW != geos[i].OW || geos[i + 1].pt || geos[i + 1].pl)
                continue; // a padded 1x1 (ONNX pads) changes the map size: the fused head assumes it does not
            if ((i > 0 && fuse32_with_next[i - 1]) || fuse32_with_next[i])
                continue;
            bool sole = true;
            for (size_t j = 0; j < layers.size(); ++j)
                if (j != i + 1 && (layers[j].in == A.out || layers[j].res == A.out || (j != i && layers[j].out == A.out)))
                    sole = false;
            for (int o = 0; o < d->n_outputs; ++o)
                if (d->outputs[o].tensor == A.out)
                    sole = false;
            if (!sole)
                continue;
            head32_with_next[i] = 1;
            tensors[A.out]->elided = true;
        }
    }
    // ---- outputs
    HP_REQUIRE(d->n_outputs >= 1 && d->outputs, HP_ERR_INVALID, "engine: no outputs");
    for (int i = 0; i < d->n_outputs; ++i) {
        const hp_output_desc& o = d->outputs[i];
        HP_REQUIRE(o.tensor > 0 && o.tensor <= max_id && tensors[o.tensor]->defined, HP_ERR_INVALID, "output %d: bad tensor", i);
        const tensor_info& ti = *tensors[o.tensor];
        HP_REQUIRE(o.coff >= 0 && o.channels > 0 && o.coff + o.channels <= ti.C, HP_ERR_INVALID, "output %d: bad channel range", i);
        out_info oi;
        oi.name.assign(o.name, strnlen(o.name, sizeof(o.name)));
        oi.tensor = o.tensor, oi.coff = o.coff, oi.channels = o.channels, oi.act = o.act, oi.H = ti.H, oi.W = ti.W;
        oi.x.C = o.channels, oi.x.act = o.act, oi.x.shuffle = o.shuffle == 2 ? 2 : 1, oi.x.group = o.group;
        oi.x.sigmoid_mask = o.sigmoid_mask, oi.x.softplus_mask = o.softplus_mask;
        HP_REQUIRE(o.shuffle == 0 || o.shuffle == 1 || o.shuffle == 2, HP_ERR_INVALID, "output %d: shuffle must be 0, 1 or 2", i);
        HP_REQUIRE(o.group >= 0 && o.group <= 32 && (o.group == 0 || (o.channels / (oi.x.shuffle * oi.x.shuffle)) % o.group == 0), HP_ERR_INVALID,
            "output %d: group %d must be 0 or divide the channel count (<= 32 components)", i, o.group);
        HP_REQUIRE(o.grid >= 0 && o.grid <= 2 && (o.act == HP_ACT_NONE || o.act == HP_ACT_SIGMOID || o.act == HP_ACT_SOFTPLUS) && o.out_h >= 0
                && o.out_w >= 0 && std::isfinite(o.scale),
            HP_ERR_INVALID, "output %d: bad grid / act / crop / scale", i);
        if (o.group > 0 && o.group < 32)
            HP_REQUIRE((o.sigmoid_mask >> o.group) == 0 && (o.softplus_mask >> o.group) == 0, HP_ERR_INVALID,
                "output %d: sigmoid / softplus masks name components beyond group %d", i, o.group);
        HP_REQUIRE(oi.x.C % (oi.x.shuffle * oi.x.shuffle) == 0, HP_ERR_INVALID, "output %d: channels not divisible by shuffle^2", i);
        oi.x.out_h = o.out_h > 0 ? o.out_h : ti.H * oi.x.shuffle, oi.x.out_w = o.out_w > 0 ? o.out_w : ti.W * oi.x.shuffle;
        HP_REQUIRE(oi.x.out_h <= ti.H * oi.x.shuffle && oi.x.out_w <= ti.W * oi.x.shuffle, HP_ERR_INVALID, "output %d: crop larger than the map", i);
        oi.x.scale = o.scale == 0.f ? 1.f : o.scale, oi.x.grid = o.grid;
        outputs.push_back(std::move(oi));
    }
    std::stable_sort(outputs.begin(), outputs.end(), [](const out_info& a, const out_info& b) { return a.name < b.name; }); // tensorrt.cpp:405
    for (auto& o : outputs) {
        o.buf = std::make_unique<hp::dev_buf>();
        HP_TRY(o.buf->alloc((size_t)max_batch * o.out_c() * o.x.out_h * o.x.out_w * sizeof(float)));
        // fuse into the producing conv when one MFMA conv writes exactly this channel range and no post-op is needed
        int writers = 0, last = -1;
        for (size_t i = 0; i < layers.size(); ++i)
            if (layers[i].out == o.tensor && layers[i].out_coff < o.coff + o.channels && layers[i].out_coff + layers[i].cout > o.coff)
                ++writers, last = (int)i;
        if (writers == 1 && layers[last].op == HP_OP_CONV && layers[last].in != 0 && layers[last].out_coff == o.coff
            && layers[last].cout == o.channels && o.act == HP_ACT_NONE && o.plain())
            o.fused_layer = last;
    }

    // ---- activation memory.  fp16 engines: one zero-filled allocation per tensor.  fp32 engines: the arena (see hp_engine::arena).
    {
        const bool use_arena = f32 && !getenv("HP_NO_ARENA");
        const int NL = (int)layers.size(), INF = NL + 1;
        std::vector<int> first(tensors.size(), INF), last(tensors.size(), -1);
        for (int i = 0; i < NL; ++i) {
            const hp_layer& L = layers[i];
            // the step a layer's reads happen in: the first half of a fused pair (depthwise / hidden head layer) runs inside the NEXT layer's launch
            const int at = (fuse32_with_next[i] || head32_with_next[i]) ? i + 1 : i;
            first[L.out] = std::min(first[L.out], i);
            last[L.out] = std::max(last[L.out], i);
            if (L.in > 0)
                last[L.in] = std::max(last[L.in], at);
            if (L.res >= 0)
                last[L.res] = std::max(last[L.res], at);
        }
        for (const auto& o : outputs)
            if (o.fused_layer < 0)
                last[o.tensor] = INF; // read by the conversion kernel behind the last layer
        struct cls_key {
            int H, W, P, cs, C;
            bool operator<(const cls_key& o) const { return std::tie(H, W, P, cs, C) < std::tie(o.H, o.W, o.P, o.cs, o.C); }
        };
        std::map<cls_key, std::vector<int>> free_slots; // arena slots of a class nobody lives in at the moment
        std::vector<std::vector<int>> dying(NL + 2);
        for (size_t t = 1; t < tensors.size(); ++t) {
            tensor_info& ti = *tensors[t];
            if (!ti.defined || ti.elided)
                continue;
            ti.cs = round_up(ti.C, 32);
        }
        auto bytes_of = [&](const tensor_info& ti) { return (size_t)max_batch * (f32 ? ti.rows32() : ti.H + 2 * ti.P) * (ti.W + 2 * ti.P) * ti.cs * (f32 ? sizeof(float) : sizeof(__half)); };
        if (!use_arena) {
            for (size_t t = 1; t < tensors.size(); ++t) {
                tensor_info& ti = *tensors[t];
                if (!ti.defined || ti.elided)
                    continue;
                HP_TRY(ti.buf.alloc(bytes_of(ti)));
                HP_HIP_TRY(hipMemset(ti.buf.p, 0, bytes_of(ti))); // the halo and the pad channels must read as zero, forever
            }
        } else {
            for (int i = 0; i < NL; ++i) {
                const int t = layers[i].out;
                tensor_info& ti = *tensors[t];
                if (first[t] == i && !ti.elided && ti.arena_slot < 0) {
                    const cls_key key{ ti.H, ti.W, ti.P, ti.cs, ti.C };
                    auto& fl = free_slots[key];
                    int slot;
                    if (!fl.empty()) {
                        slot = fl.back();
                        fl.pop_back();
                    } else {
                        slot = (int)arena.size();
                        arena.emplace_back();
                        arena.back().mem = std::make_unique<hp::dev_buf>();
                        HP_TRY(arena.back().mem->alloc(bytes_of(ti)));
                        HP_HIP_TRY(hipMemset(arena.back().mem->p, 0, bytes_of(ti))); // halo and pad channels: zero for every tenant, forever
                    }
                    ti.arena_slot = slot, ti.shared = arena[slot].mem->p;
                    ++arena[slot].tenants;
                    arena_private_bytes += bytes_of(ti);
                    dying[std::min(std::max(last[t], i), NL + 1)].push_back(t);
                }
                for (int d : dying[i]) { // tensors whose last reader is this step: their buffers are free from the next step on
                    const tensor_info& td = *tensors[d];
                    free_slots[cls_key{ td.H, td.W, td.P, td.cs, td.C }].push_back(td.arena_slot);
                }
            }
        }
    }

    // a tensor some layer reads (input or residual), or that an un-fused output conversion will read
    auto tensor_is_read = [&](int t) {
        for (const auto& L2 : layers)
            if (L2.in == t || L2.res == t)
                return true;
        for (const auto& o : outputs)
            if (o.tensor == t && o.fused_layer < 0)
                return true;
        return false;
    };

    // ---- pass 2: pack weights, build the schedule
    auto blob = [&](int64_t off, size_t n, const char* what, size_t layer) -> const float* {
        if (off < 0 || (size_t)off + n > d->n_weights) {
            hp::set_error("layer %zu: %s range [%lld,+%zu) outside the %zu-float weight blob", layer, what, (long long)off, n, d->n_weights);
            return nullptr;
        }
        return d->weights + off;
    };
    auto upload = [&](const void* src, size_t bytes, void** dst) -> int {
        weight_bufs.push_back(std::make_unique<hp::dev_buf>());
        HP_TRY(weight_bufs.back()->alloc(bytes));
        HP_HIP_TRY(hipMemcpy(weight_bufs.back()->p, src, bytes, hipMemcpyHostToDevice));
        *dst = weight_bufs.back()->p;
        return HP_OK;
    };

    for (size_t i = 0; i < layers.size(); ++i) {
        const hp_layer& L = layers[i];
        const tensor_info& ti = *tensors[L.in];
        tensor_info& to = *tensors[L.out];
        const geo& g = geos[i];
        step st;
        st.layer = (int)i, st.op = L.op;
        const double opix = (double)g.OH * g.OW;
        if (f32) {
            // ---- HP_DTYPE_F32: one fp32 launch per layer (conv_fp32.hip), weights uploaded as they are
            st.f32 = true;
            if (fuse32_with_next[i] || head32_with_next[i]) // the first half of a fused pair: described at the 1 x 1 layer that follows
                continue;
            const bool dw_in_front = i > 0 && fuse32_with_next[i - 1];
            const bool head_in_front = i > 0 && head32_with_next[i - 1];
            auto padded = [&](int64_t off, int n, int n_pad, const char* what, std::vector<float>& v) -> bool {
                v.assign(n_pad, 0.f);
                if (off < 0)
                    return true;
                const float* src = blob(off, n, what, i);
                if (!src)
                    return false;
                std::copy(src, src + n, v.begin());
                return true;
            };
            if (L.op == HP_OP_CONV && L.in == 0) {
                HP_REQUIRE(L.cin == 3 && L.in_coff == 0, HP_ERR_INVALID, "layer %zu: the network input has 3 channels", i);
                HP_REQUIRE(L.dil == 1 && L.res < 0 && L.act != HP_ACT_PRELU, HP_ERR_INVALID, "layer %zu: unsupported first-layer options", i);
                const size_t nw = (size_t)L.cout * L.kh * L.kw * 3;
                const float* w = blob(L.w_off, nw, "weights", i);
                std::vector<float> bias;
                if (!w || !padded(L.b_off, L.cout, round_up(L.cout, 8), "bias", bias))
                    return HP_ERR_INVALID;
                st.first = true;
                auto& p = st.fp32;
                p.factor = factor, p.flip_rb = flip_rb;
                for (int c = 0; c < 3; ++c)
                    p.mean[c] = mean[c], p.inv_std[c] = inv_std[c];
                p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.Cout = L.cout, p.KH = L.kh, p.KW = L.kw, p.stride = L.stride;
                p.pad_t = g.pt, p.pad_l = g.pl, p.act = L.act, p.act_param = L.act_param;
                void *dw = nullptr, *db = nullptr;
                HP_TRY(upload(w, nw * sizeof(float), &dw));
                HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
                p.w = (const float*)dw, p.bias = (const float*)db;
                p.out = to.view32(L.out_coff);
                st.flops = 2.0 * opix * L.cout * L.kh * L.kw * 3;
                st.bytes = (double)ti.H * ti.W * 3 + opix * L.cout * 4 + nw * 4;
            } else if (L.op == HP_OP_CONV) {
                const int cin_pad = round_up(L.cin, 16), cout_pad = round_up(L.cout, 64), taps = L.kh * L.kw;
                // (behind a fused depthwise layer the tensor this layer "reads" is never materialised: what the kernel reads is the depthwise input)
                const tensor_info& tsrc = (dw_in_front || head_in_front) ? *tensors[layers[i - 1].in] : ti;
                const int src_coff = (dw_in_front || head_in_front) ? layers[i - 1].in_coff : L.in_coff;
                HP_REQUIRE(src_coff % 4 == 0 && src_coff + (head_in_front ? layers[i - 1].cin : cin_pad) <= tsrc.cs, HP_ERR_INVALID,
                    "layer %zu: channel slice [%d,+%d) not 4-aligned / exceeds the padded stride %d", i, src_coff, cin_pad, tsrc.cs);
                const size_t nw = (size_t)L.cout * taps * L.cin;
                const float* w = blob(L.w_off, nw, "weights", i);
                std::vector<float> bias, alpha;
                if (!w || !padded(L.b_off, L.cout, cout_pad, "bias", bias))
                    return HP_ERR_INVALID;
                auto& p = st.cp32;
                p.alpha = nullptr;
                if (L.act == HP_ACT_PRELU) {
                    HP_REQUIRE(L.alpha_off >= 0, HP_ERR_INVALID, "layer %zu: PReLU without slopes", i);
                    if (!padded(L.alpha_off, L.cout, cout_pad, "prelu slopes", alpha))
                        return HP_ERR_INVALID;
                    void* da = nullptr;
                    HP_TRY(upload(alpha.data(), alpha.size() * sizeof(float), &da));
                    p.alpha = (const float*)da;
                }
                std::vector<float> packed((size_t)taps * cout_pad * cin_pad, 0.f);
                for (int co = 0; co < L.cout; ++co)
                    for (int t = 0; t < taps; ++t)
                        std::copy(w + ((size_t)co * taps + t) * L.cin, w + ((size_t)co * taps + t + 1) * L.cin, packed.begin() + ((size_t)t * cout_pad + co) * cin_pad);
                void *dw = nullptr, *db = nullptr;
                HP_TRY(upload(packed.data(), packed.size() * sizeof(float), &dw));
                HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
                p.w = (const float*)dw, p.bias = (const float*)db;
                p.in = ti.view32(L.in_coff);
                p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.Cin = cin_pad, p.Cout = L.cout, p.Cout_pad = cout_pad;
                p.KH = L.kh, p.KW = L.kw, p.stride = L.stride, p.dil = L.dil, p.pad_t = g.pt, p.pad_l = g.pl;
                p.dw_w = nullptr, p.dw_dil = 0, p.dw_slope = 1.f, p.dw_hi = __builtin_huge_valf();
                if (dw_in_front) { // conv32_direct_kernel's fused form: `in` is the depthwise layer's input, its weights ride along as [9][C] + [C]
                    const hp_layer& D = layers[i - 1];
                    const float* dwf = blob(D.w_off, (size_t)D.cin * 9, "weights", i - 1);
                    std::vector<float> dbias;
                    if (!dwf || !padded(D.b_off, D.cin, D.cin, "bias", dbias))
                        return HP_ERR_INVALID;
                    std::vector<float> dpack((size_t)10 * D.cin);
                    for (int c = 0; c < D.cin; ++c) {
                        for (int t = 0; t < 9; ++t)
                            dpack[(size_t)t * D.cin + c] = dwf[(size_t)c * 9 + t];
                        dpack[(size_t)9 * D.cin + c] = dbias[c];
                    }
                    void* ddw = nullptr;
                    HP_TRY(upload(dpack.data(), dpack.size() * sizeof(float), &ddw));
                    p.dw_w = (const float*)ddw, p.dw_dil = D.dil;
                    p.dw_slope = D.act == HP_ACT_NONE ? 1.f : D.act == HP_ACT_LEAKY ? D.act_param : 0.f;
                    p.dw_hi = D.act == HP_ACT_RELU6 ? 6.f : __builtin_huge_valf();
                    p.in = tensors[D.in]->view32(D.in_coff);
                    st.layer = (int)i - 1, st.n_layers = 2;
                }
                p.act = L.act, p.act_param = L.act_param;
                p.res = hp::tview32{ nullptr, 0, 0, 0, 0 }, p.res_before_act = L.res_before_act;
                if (L.res >= 0)
                    p.res = tensors[L.res]->view32(0);
                p.out = to.view32(L.out_coff);
                p.out_f32 = nullptr;
                for (auto& o : outputs)
                    if (o.fused_layer == (int)i)
                        p.out_f32 = o.buf->as<float>();
                if (p.out_f32 && !tensor_is_read(L.out) && tensors[L.out]->C == L.cout)
                    p.out.p = nullptr, to.unwritten = true; // only the fp32 network output is wanted
                HP_REQUIRE(hp::set_act32(p), HP_ERR_INVALID, "layer %zu: activation %d cannot be fused into a dense conv (use an output post-op)", i, L.act);
                p.B = max_batch, p.npix = p.pick_npix = max_batch * g.OH * g.OW;
                p.w_split = nullptr, p.w_frag = nullptr, p.w_wino = nullptr, p.w_wino3 = nullptr, p.ovf = ovf_dev, p.dbg = nullptr;
                p.lane_epilogue = getenv("HP_LANE_EPILOGUE") ? atoi(getenv("HP_LANE_EPILOGUE")) : 0; // (A/B switches are read per build: an in-process A/B compares two engines)
                // The layers conv32_direct_kernel covers (square 1 x 1 / 3 x 3, stride 1, whole 32- / 64-channel chunks inside the buffer's
                // channel stride) get their weights in fragment order as well: fp32 for HP_DTYPE_F32 (HP_NO_DIRECT32=1: the A/B switch back
                // to conv32_kernel), fp16 (hi, lo) pairs for HP_DTYPE_F32S; the others stay on conv32_kernel
                // fp32 pipe, measured per layer of LW-OpenPose @ 8 x 46 x 54 (conv32_kernel -> direct, us alone | with a second stream): 3 x 3 128 -> 128
                // 72.2 -> 58.7 | 52.6 -> 48.9; the 1 x 1 heads 512 -> 19 / 38 27.9 / 29.3 -> 26.1 / 27.4 | 18.8 / 19.6 -> 17.9 / 18.2; every wider 1 x 1
                // layer within +-3 % or worse (64 -> 128 22.9 -> 29.0 paired, 512 -> 512 100 -> 100 alone): 1 x 1 layers wider than 64 padded outputs
                // stay on conv32_kernel (HP_DIRECT32_MAX_1X1 moves the limit, HP_NO_DIRECT32=1 is the A/B switch)
                const bool no_direct = getenv("HP_NO_DIRECT32") != nullptr;
                const int direct_max_1x1 = getenv("HP_DIRECT32_MAX_1X1") ? atoi(getenv("HP_DIRECT32_MAX_1X1")) : 64;
                // (a 3 x 3 layer the Winograd kernel will take needs no direct-form fragments: they doubled its weight bytes in HBM)
                const bool wino_takes_it = dtype == HP_DTYPE_F32 && !dw_in_front && !getenv("HP_NO_WINOGRAD32") && hp::conv32_winograd_ok(p) && !head_in_front;
                // HP_SPLIT_LAYERS="lo hi" (diagnostic, section 7B.8): an HP_DTYPE_F32S engine forms split products in layers lo .. hi only, every other
                // dense layer runs conv32_kernel
                bool split_here = dtype == HP_DTYPE_F32S;
                if (const char* sl = getenv("HP_SPLIT_LAYERS")) {
                    int lo = 0, hi = 1 << 30;
                    sscanf(sl, "%d %d", &lo, &hi);
                    split_here = split_here && ((int)i >= lo && (int)i <= hi);
                }
                if (split_here || dw_in_front || (dtype != HP_DTYPE_F32S && !no_direct && !wino_takes_it && (taps > 1 || cout_pad <= direct_max_1x1))) {
                    const int ck = taps == 1 ? 64 : 32, cin_s = round_up(L.cin, ck);
                    hp::conv32_params q = p;
                    q.Cin = cin_s;
                    HP_REQUIRE(!dw_in_front || (src_coff + cin_s <= tsrc.cs && hp::conv32_direct_ok(q)), HP_ERR_STATE,
                        "layer %zu: fused behind a depthwise layer but not a direct-kernel layer (engine bug)", i);
                    if (src_coff + cin_s <= tsrc.cs && hp::conv32_direct_ok(q)) {
                        std::vector<float> wide((size_t)taps * cout_pad * cin_s, 0.f);
                        for (int t = 0; t < taps; ++t)
                            for (int co = 0; co < cout_pad; ++co)
                                std::copy(packed.begin() + ((size_t)t * cout_pad + co) * cin_pad, packed.begin() + ((size_t)t * cout_pad + co) * cin_pad + cin_pad,
                                    wide.begin() + ((size_t)t * cout_pad + co) * cin_s);
                        void* dws = nullptr;
                        if (dtype == HP_DTYPE_F32S) {
                            // (a BN-folded weight beyond fp16's range has no (hi, lo) split - hi would be inf - and the run-time flag only watches
                            // activations: such an engine starts on the fp32 pipe.  ADVICE r5)
                            for (float v : wide)
                                if (!(std::fabs(v) <= 65504.f) && !split_off)
                                    split_off = true, ++split_fallbacks;
                            std::vector<_Float16> ws(wide.size() * 2);
                            hp::conv32_split_pack(wide.data(), taps, cout_pad, cin_s, ws.data());
                            HP_TRY(upload(ws.data(), ws.size() * sizeof(_Float16), &dws));
                            p.w_split = (const _Float16*)dws;
                        }
                        // (an HP_DTYPE_F32S engine keeps the fp32 fragments too: what it runs after a value left fp16's range)
                        std::vector<float> wf(wide.size());
                        hp::conv32_frag_pack(wide.data(), taps, cout_pad, cin_s, wf.data());
                        HP_TRY(upload(wf.data(), wf.size() * sizeof(float), &dws));
                        p.w_frag = (const float*)dws;
                        st.cin_split = cin_s;
                    }
                }
                st.flops = 2.0 * opix * L.cout * taps * L.cin;
                st.bytes = (double)ti.H * ti.W * L.cin * 4 + opix * L.cout * 4 + (double)nw * 4;
                // HP_DTYPE_F32: 3 x 3 stride-1 layers in Winograd's F(2 x 2, 3 x 3) form - 16 MFMA products per 2 x 2 output tile and channel pair
                // instead of 36 (conv32_winograd.hip; HP_NO_WINOGRAD32=1 is the A/B switch back to the direct kernel).  The step's `flops` stay the
                // layer's ALGORITHMIC count (the direct form's 2 * 9 * Cin * Cout per pixel); the MFMA work issued is 16 / 36 of it (bench.py reports both).
                if (dtype == HP_DTYPE_F32 && !dw_in_front && !getenv("HP_NO_WINOGRAD32") && hp::conv32_winograd_ok(p)) {
                    std::vector<float> wu((size_t)16 * cout_pad * cin_pad);
                    hp::conv32_winograd_pack(packed.data(), cout_pad, cin_pad, wu.data());
                    void* dwu = nullptr;
                    HP_TRY(upload(wu.data(), wu.size() * sizeof(float), &dwu));
                    p.w_wino = (const float*)dwu;
                    st.wino = true;
                    st.bytes += (double)nw * 4 * (16.0 / 9 - 1);
                    // ... or, opt-in (HP_WINO_F33=1, read per engine), F(3 x 3, 3 x 3): 25 products per 3 x 3 tile - 1.44 x fewer MFMA cycles, same accuracy
                    // (conv32_winograd3.hip).  Measured: it wins where its 24 x 6-pixel blocks tile the map exactly and the layer is mid-sized (256 -> 256
                    // at 32 x 24 x 24: 121 -> 87 us) and loses everywhere else - LW-OpenPose's 128 -> 128 at 8 x 46 x 54 48.6 | 28.6 us alone | paired against
                    // 45.0 (35.9 in 8 x 8 blocks) | 27.0, every layer of PifPaf's 97 / 49 / 25-row maps - because one 16-tile column per wavefront streams
                    // 3.1 x the U bytes per output pixel.  Not the default: taken by itself only on PoseProposal's exactly tiled 48 x 48 / 24 x 24
                    // stages it made configs[3] 0.6 % faster (4 353 -> 4 380 frames/s resident, two runs each) - not worth a second numeric path.
                    const bool f33 = getenv("HP_WINO_F33") && atoi(getenv("HP_WINO_F33")) == 1;
                    if (f33 && hp::conv32_winograd3_ok(p)) {
                        std::vector<float> wu3((size_t)25 * cout_pad * cin_pad);
                        hp::conv32_winograd3_pack(packed.data(), cout_pad, cin_pad, wu3.data());
                        void* dwu3 = nullptr;
                        HP_TRY(upload(wu3.data(), wu3.size() * sizeof(float), &dwu3));
                        p.w_wino3 = (const float*)dwu3;
                        st.bytes += (double)nw * 4 * (25.0 / 9 - 16.0 / 9);
                    }
                }
                if (dw_in_front) // + the depthwise taps; the tensor between the two layers costs no bytes any more
                    st.flops += 2.0 * opix * L.cin * 9, st.bytes += (double)L.cin * 10 * 4;
                if (head_in_front) { // conv32_head_kernel: this layer's epilogue + the hidden layer in front of it
                    const hp_layer& A = layers[i - 1];
                    const size_t nwa = (size_t)A.cout * A.cin;
                    const float* wa = blob(A.w_off, nwa, "weights", i - 1);
                    std::vector<float> biasa;
                    if (!wa || !padded(A.b_off, A.cout, A.cout, "bias", biasa))
                        return HP_ERR_INVALID;
                    std::vector<float> w1f(nwa), w2p((size_t)cout_pad * L.cin, 0.f), w2f;
                    hp::conv32_frag_pack(wa, 1, A.cout, A.cin, w1f.data()); // ([1][HID][128] is the blob's own layout: HID rows of 128)
                    const int tm2 = L.cout <= 32 ? 1 : 2;
                    std::copy(w, w + nw, w2p.begin()); // [cout][HID] rows, zero rows up to 32 tm2
                    w2f.resize((size_t)32 * tm2 * L.cin);
                    hp::conv32_head_pack(w2p.data(), tm2, L.cin, w2f.data());
                    void *d1 = nullptr, *db1 = nullptr, *d2 = nullptr;
                    HP_TRY(upload(w1f.data(), w1f.size() * sizeof(float), &d1));
                    HP_TRY(upload(biasa.data(), biasa.size() * sizeof(float), &db1));
                    HP_TRY(upload(w2f.data(), w2f.size() * sizeof(float), &d2));
                    st.hh.w1_frag = (const float*)d1, st.hh.bias1 = (const float*)db1, st.hh.w2_frag = (const float*)d2, st.hh.HID = A.cout;
                    st.hh.slope1 = A.act == HP_ACT_NONE ? 1.f : A.act == HP_ACT_LEAKY ? A.act_param : 0.f;
                    st.hh.hi1 = A.act == HP_ACT_RELU6 ? 6.f : __builtin_huge_valf();
                    p.in = tsrc.view32(src_coff);
                    st.head32 = true, st.wino = false, st.cin_split = 0;
                    st.layer = (int)i - 1, st.n_layers = 2;
                    st.flops = 2.0 * opix * ((double)A.cout * A.cin + (double)L.cout * L.cin);
                    st.bytes = (double)tsrc.H * tsrc.W * A.cin * 4 + opix * L.cout * 4 + (double)(nwa + nw) * 4;
                }
            } else if (L.op == HP_OP_DWCONV) {
                HP_REQUIRE(L.kh == 3 && L.kw == 3, HP_ERR_INVALID, "layer %zu: depthwise kernels are 3x3", i);
                HP_REQUIRE(L.cin % 4 == 0 && L.in_coff % 4 == 0 && L.out_coff % 4 == 0, HP_ERR_INVALID, "layer %zu: depthwise needs 4-aligned channels", i);
                // (the fp16 path refuses the same set, set_act: a PReLU here would silently run as ReLU - there are no slopes in dw32_params)
                HP_REQUIRE(L.act != HP_ACT_PRELU && L.act != HP_ACT_SIGMOID && L.act != HP_ACT_SOFTPLUS && L.res < 0, HP_ERR_INVALID,
                    "layer %zu: activation %d / a residual on a depthwise layer is not supported", i, L.act);
                const float* w = blob(L.w_off, (size_t)L.cin * 9, "weights", i);
                std::vector<float> bias;
                if (!w || !padded(L.b_off, L.cin, L.cin, "bias", bias))
                    return HP_ERR_INVALID;
                std::vector<float> packed((size_t)9 * L.cin);
                for (int c = 0; c < L.cin; ++c)
                    for (int t = 0; t < 9; ++t)
                        packed[(size_t)t * L.cin + c] = w[(size_t)c * 9 + t];
                auto& p = st.dp32;
                void *dw = nullptr, *db = nullptr;
                HP_TRY(upload(packed.data(), packed.size() * sizeof(float), &dw));
                HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
                p.w = (const float*)dw, p.bias = (const float*)db;
                p.in = ti.view32(L.in_coff);
                p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = L.cin, p.stride = L.stride, p.dil = L.dil;
                p.pad_t = g.pt, p.pad_l = g.pl, p.act = L.act, p.act_param = L.act_param;
                p.out = to.view32(L.out_coff);
                st.flops = 2.0 * opix * L.cin * 9;
                st.bytes = (double)ti.H * ti.W * L.cin * 4 + opix * L.cin * 4;
            } else { // max-pool / up-sampling
                HP_REQUIRE(L.cin % 4 == 0 && L.in_coff % 4 == 0 && L.out_coff % 4 == 0 && (L.op == HP_OP_UPSAMPLE || L.kh == L.kw), HP_ERR_INVALID,
                    "layer %zu: pooling / up-sampling needs 4-aligned channels and a square window", i);
                auto& p = st.pp32;
                p.in = ti.view32(L.in_coff);
                p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = L.cin, p.k = L.kh, p.stride = L.stride;
                p.pad_t = L.op == HP_OP_UPSAMPLE ? 0 : g.pt, p.pad_l = L.op == HP_OP_UPSAMPLE ? 0 : g.pl;
                p.out = to.view32(L.out_coff);
                st.flops = 0;
                st.bytes = (double)ti.H * ti.W * L.cin * 4 + opix * L.cin * 4;
            }
            steps.push_back(st);
            continue;
        }
        if (L.op == HP_OP_CONV && L.in == 0) {
            HP_REQUIRE(L.cin == 3 && L.in_coff == 0, HP_ERR_INVALID, "layer %zu: the network input has 3 channels", i);
            HP_REQUIRE(L.dil == 1 && L.res < 0 && L.act != HP_ACT_PRELU, HP_ERR_INVALID, "layer %zu: unsupported first-layer options", i);
            const size_t nw = (size_t)L.cout * L.kh * L.kw * 3;
            const float* w = blob(L.w_off, nw, "weights", i);
            if (!w)
                return HP_ERR_INVALID;
            std::vector<float> bias(round_up(L.cout, 8), 0.f);
            if (L.b_off >= 0) {
                const float* b = blob(L.b_off, L.cout, "bias", i);
                if (!b)
                    return HP_ERR_INVALID;
                std::copy(b, b + L.cout, bias.begin());
            }
            st.first = true;
            auto& p = st.fp;
            p.factor = factor, p.flip_rb = flip_rb;
            for (int c = 0; c < 3; ++c)
                p.mean[c] = mean[c], p.inv_std[c] = inv_std[c];
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.Cout = L.cout, p.KH = L.kh, p.KW = L.kw, p.stride = L.stride;
            p.pad_t = g.pt, p.pad_l = g.pl, p.act = L.act, p.act_param = L.act_param;
            void* dw = nullptr;
            void* db = nullptr;
            HP_TRY(upload(w, nw * sizeof(float), &dw));
            HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
            p.w = (const float*)dw, p.bias = (const float*)db;
            p.w16 = nullptr;
            if (L.kh == L.kw && (L.kh == 3 || L.kh == 7) && L.cout % 8 == 0 && L.cout <= 64) {
                // fragment order of first_conv_f16_kernel: kernel rows padded to ROWP (a multiple of 8), K' = KS * ROWP in steps of 16
                const int KS = L.kh, ROWP = (KS * 3 + 7) / 8 * 8, KP = KS * ROWP, STEPS = (KP + 15) / 16, MT = L.cout <= 32 ? 1 : 2;
                std::vector<__half> w16((size_t)MT * STEPS * 64 * 8, __float2half(0.f));
                for (int co = 0; co < L.cout; ++co)
                    for (int ky = 0; ky < KS; ++ky)
                        for (int r = 0; r < KS * 3; ++r) {
                            const int k = ky * ROWP + r, st = k / 16, hh = (k % 16) / 8, e = k % 8;
                            w16[((((size_t)(co / 32) * STEPS + st) * 64) + hh * 32 + co % 32) * 8 + e] = __float2half(w[((size_t)co * KS + ky) * KS * 3 + r]);
                        }
                void* d16 = nullptr;
                HP_TRY(upload(w16.data(), w16.size() * sizeof(__half), &d16));
                p.w16 = (const __half*)d16;
            }
            p.out = to.view(L.out_coff);
            st.flops = 2.0 * opix * L.cout * L.kh * L.kw * 3;
            st.bytes = (double)ti.H * ti.W * 3 + opix * L.cout * 2 + nw * 4;
        } else if (L.op == HP_OP_CONV && head_with_next[i]) {
            const hp_layer& Pn = layers[i + 1];
            tensor_info& tp = *tensors[Pn.out];
            const int K1 = L.cin, KQ1 = K1 / 16, HID = L.cout;
            const float* w1f = blob(L.w_off, (size_t)HID * K1, "weights", i);
            const float* w2f = blob(Pn.w_off, (size_t)Pn.cout * HID, "weights", i + 1);
            if (!w1f || !w2f)
                return HP_ERR_INVALID;
            std::vector<__half> w1p((size_t)HID * K1), w2p((size_t)64 * HID, __float2half(0.f));
            for (int m = 0; m < HID; ++m)
                for (int k = 0; k < K1; ++k)
                    w1p[((((size_t)(m / 32) * KQ1 + k / 16) * 64) + (k % 16 / 8) * 32 + m % 32) * 8 + k % 8] = __float2half(w1f[(size_t)m * K1 + k]);
            for (int m = 0; m < Pn.cout; ++m)
                for (int c = 0; c < HID; ++c) {
                    const int wv = c / 128, ii = (c % 128) / 32, r32 = c % 32;
                    const int hh = (r32 >> 2) & 1, r = (r32 & 3) + 4 * (r32 >> 3), ss = r >> 3, ee = r & 7;
                    w2p[((((size_t)(m / 32) * 32 + 8 * wv + 2 * ii + ss) * 64) + hh * 32 + m % 32) * 8 + ee] = __float2half(w2f[(size_t)m * HID + c]);
                }
            std::vector<float> b1(HID, 0.f), b2(64, 0.f), alpha;
            if (L.b_off >= 0) {
                const float* bb = blob(L.b_off, HID, "bias", i);
                if (!bb)
                    return HP_ERR_INVALID;
                std::copy(bb, bb + HID, b1.begin());
            }
            if (Pn.b_off >= 0) {
                const float* bb = blob(Pn.b_off, Pn.cout, "bias", i + 1);
                if (!bb)
                    return HP_ERR_INVALID;
                std::copy(bb, bb + Pn.cout, b2.begin());
            }
            st.op = OP_MLPHEAD, st.n_layers = 2; // (covers layers i and i + 1: the passes below must see every layer a step touches)
            auto& p = st.hp_;
            void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
            HP_TRY(upload(w1p.data(), w1p.size() * sizeof(__half), &d0));
            HP_TRY(upload(b1.data(), b1.size() * sizeof(float), &d1));
            HP_TRY(upload(w2p.data(), w2p.size() * sizeof(__half), &d2));
            HP_TRY(upload(b2.data(), b2.size() * sizeof(float), &d3));
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.K1 = K1;
            p.w1 = (const __half*)d0, p.b1 = (const float*)d1, p.w2 = (const __half*)d2;
            p.hi1 = L.act == HP_ACT_RELU6 ? 6.f : __builtin_huge_valf();
            auto& q = p.pw;
            q.w = nullptr, q.w_layout = 0, q.bias = (const float*)d3, q.alpha = nullptr;
            if (Pn.act == HP_ACT_PRELU) {
                alpha.assign(64, 0.f);
                const float* a = blob(Pn.alpha_off, Pn.cout, "prelu slopes", i + 1);
                if (!a)
                    return HP_ERR_INVALID;
                std::copy(a, a + Pn.cout, alpha.begin());
                void* da = nullptr;
                HP_TRY(upload(alpha.data(), alpha.size() * sizeof(float), &da));
                q.alpha = (const float*)da;
            }
            q.H = ti.H, q.W = ti.W, q.OH = ti.H, q.OW = ti.W, q.Cin = HID, q.Cout = Pn.cout, q.Cout_pad = 64;
            q.KH = q.KW = 1, q.stride = 1, q.dil = 1, q.pad_t = q.pad_l = 0;
            q.act = Pn.act, q.act_param = Pn.act_param;
            q.res = hp::tview{ nullptr, 0, 0, 0, 0 }, q.res_before_act = 0;
            q.out = tp.view(Pn.out_coff);
            q.out_f32 = nullptr, q.dbg = nullptr;
            for (auto& o : outputs)
                if (o.fused_layer == (int)i + 1)
                    q.out_f32 = o.buf->as<float>();
            if (q.out_f32 && !tensor_is_read(Pn.out))
                q.out.p = nullptr, tp.unwritten = true; // only the fp32 network output is wanted
            HP_REQUIRE(hp::set_act(q), HP_ERR_INVALID, "layer %zu: activation %d cannot be fused into a dense conv", i + 1, Pn.act);
            st.flops = 2.0 * opix * HID * K1 + 2.0 * opix * Pn.cout * HID;
            st.bytes = (double)ti.H * ti.W * K1 * 2 + opix * Pn.cout * 2 + ((double)HID * K1 + (double)Pn.cout * HID) * 2;
            steps.push_back(st);
            ++i; // the second convolution is part of this step
            continue;
        } else if (L.op == HP_OP_CONV) {
            const int cin_pad = round_up(L.cin, 32);
            HP_REQUIRE(L.in_coff % 8 == 0 && L.in_coff + cin_pad <= ti.cs, HP_ERR_INVALID,
                "layer %zu: channel slice [%d,+%d) not 8-aligned / exceeds the padded stride %d", i, L.in_coff, cin_pad, ti.cs);
            const int cout_pad = L.cout > 64 ? round_up(L.cout, 128) : 64;
            const int taps = L.kh * L.kw;
            const size_t nw = (size_t)L.cout * taps * L.cin;
            const float* w = blob(L.w_off, nw, "weights", i);
            if (!w)
                return HP_ERR_INVALID;
            std::vector<float> bias(cout_pad, 0.f), alpha;
            if (L.b_off >= 0) {
                const float* b = blob(L.b_off, L.cout, "bias", i);
                if (!b)
                    return HP_ERR_INVALID;
                std::copy(b, b + L.cout, bias.begin());
            }
            auto& p = st.cp;
            void* dw = nullptr;
            void* db = nullptr;
            HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
            p.w = nullptr, p.w_layout = 0, p.bias = (const float*)db, p.alpha = nullptr;
            if (L.act == HP_ACT_PRELU) {
                alpha.assign(cout_pad, 0.f);
                const float* a = blob(L.alpha_off, L.cout, "prelu slopes", i);
                if (!a)
                    return HP_ERR_INVALID;
                std::copy(a, a + L.cout, alpha.begin());
                void* da = nullptr;
                HP_TRY(upload(alpha.data(), alpha.size() * sizeof(float), &da));
                p.alpha = (const float*)da;
            }
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.Cin = cin_pad, p.Cout = L.cout, p.Cout_pad = cout_pad;
            p.KH = L.kh, p.KW = L.kw, p.stride = L.stride, p.dil = L.dil, p.pad_t = g.pt, p.pad_l = g.pl;
            p.act = L.act, p.act_param = L.act_param;
            p.res = hp::tview{ nullptr, 0, 0, 0, 0 }, p.res_before_act = L.res_before_act;
            if (L.res >= 0)
                p.res = tensors[L.res]->view(0);
            p.out = to.view(L.out_coff);
            p.out_f32 = nullptr;
            p.dbg = nullptr;
            for (auto& o : outputs)
                if (o.fused_layer == (int)i)
                    p.out_f32 = o.buf->as<float>();
            if (p.out_f32 && !tensor_is_read(L.out) && tensors[L.out]->C == L.cout)
                p.out.p = nullptr, to.unwritten = true; // only the fp32 network output is wanted
            HP_REQUIRE(hp::set_act(p), HP_ERR_INVALID, "layer %zu: activation %d cannot be fused into a dense conv (use an output post-op)", i, L.act);
            // weights: the launcher says which packing its kernel for this shape reads
            p.B = max_batch, p.npix = max_batch * g.OH * g.OW;
            p.w_layout = hp::conv_weight_layout(p);
            std::vector<__half> packed((size_t)taps * cout_pad * cin_pad, __float2half(0.f));
            const int KQ = cin_pad / 16;
            for (int co = 0; co < L.cout; ++co)
                for (int t = 0; t < taps; ++t)
                    for (int ci = 0; ci < L.cin; ++ci) {
                        const size_t at = p.w_layout == 1
                            ? ((((size_t)t * (cout_pad / 32) + co / 32) * KQ + ci / 16) * 64 + (ci % 16 / 8) * 32 + co % 32) * 8 + ci % 8
                            : ((size_t)t * cout_pad + co) * cin_pad + ci;
                        packed[at] = __float2half(w[((size_t)co * taps + t) * L.cin + ci]);
                    }
            HP_TRY(upload(packed.data(), packed.size() * sizeof(__half), &dw));
            p.w = (const __half*)dw;
            st.flops = 2.0 * opix * L.cout * taps * L.cin;
            st.bytes = (double)ti.H * ti.W * L.cin * 2 + opix * L.cout * 2 + (double)nw * 2;
        } else if (L.op == HP_OP_DWCONV && fuse_with_next[i]) {
            const hp_layer& Pn = layers[i + 1];
            tensor_info& tp = *tensors[Pn.out];
            const float* w = blob(L.w_off, (size_t)L.cin * 9, "weights", i);
            if (!w)
                return HP_ERR_INVALID;
            std::vector<__half> dpacked((size_t)9 * L.cin);
            for (int c = 0; c < L.cin; ++c)
                for (int t = 0; t < 9; ++t)
                    dpacked[(size_t)t * L.cin + c] = __float2half(w[(size_t)c * 9 + t]);
            std::vector<float> dbias(L.cin, 0.f);
            if (L.b_off >= 0) {
                const float* b = blob(L.b_off, L.cin, "bias", i);
                if (!b)
                    return HP_ERR_INVALID;
                std::copy(b, b + L.cin, dbias.begin());
            }
            const int C = L.cin, KQ = C / 16, cout_pad = round_up(Pn.cout, 128);
            const float* pwf = blob(Pn.w_off, (size_t)Pn.cout * C, "weights", i + 1);
            if (!pwf)
                return HP_ERR_INVALID;
            // MFMA-fragment order: [32-row tile][k16 step][lane = (k % 16 / 8) * 32 + m % 32][k % 8]
            std::vector<__half> ppacked((size_t)cout_pad * C, __float2half(0.f));
            for (int m = 0; m < Pn.cout; ++m)
                for (int k = 0; k < C; ++k)
                    ppacked[((((size_t)(m / 32) * KQ + k / 16) * 64) + (k % 16 / 8) * 32 + m % 32) * 8 + k % 8] = __float2half(pwf[(size_t)m * C + k]);
            std::vector<float> pbias(cout_pad, 0.f), alpha;
            if (Pn.b_off >= 0) {
                const float* b = blob(Pn.b_off, Pn.cout, "bias", i + 1);
                if (!b)
                    return HP_ERR_INVALID;
                std::copy(b, b + Pn.cout, pbias.begin());
            }
            st.op = OP_SEPCONV, st.n_layers = 2;
            auto& p = st.sp;
            void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
            HP_TRY(upload(dpacked.data(), dpacked.size() * sizeof(__half), &d0));
            HP_TRY(upload(dbias.data(), dbias.size() * sizeof(float), &d1));
            HP_TRY(upload(ppacked.data(), ppacked.size() * sizeof(__half), &d2));
            HP_TRY(upload(pbias.data(), pbias.size() * sizeof(float), &d3));
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = C, p.stride = L.stride, p.dil = L.dil;
            p.pad_t = g.pt, p.pad_l = g.pl, p.halo = ti.P;
            p.dw_w = (const __half*)d0, p.dw_bias = (const float*)d1;
            {
                hp::conv_params tmp{};
                tmp.act = L.act, tmp.act_param = L.act_param, tmp.alpha = nullptr;
                HP_REQUIRE(hp::set_act(tmp), HP_ERR_INVALID, "layer %zu: unsupported depthwise activation %d", i, L.act);
                p.dw_slope = tmp.act_slope, p.dw_hi = tmp.act_hi;
            }
            auto& q = p.pw;
            q.w = (const __half*)d2, q.bias = (const float*)d3, q.alpha = nullptr;
            if (Pn.act == HP_ACT_PRELU) {
                alpha.assign(cout_pad, 0.f);
                const float* a = blob(Pn.alpha_off, Pn.cout, "prelu slopes", i + 1);
                if (!a)
                    return HP_ERR_INVALID;
                std::copy(a, a + Pn.cout, alpha.begin());
                void* da = nullptr;
                HP_TRY(upload(alpha.data(), alpha.size() * sizeof(float), &da));
                q.alpha = (const float*)da;
            }
            q.H = g.OH, q.W = g.OW, q.OH = g.OH, q.OW = g.OW, q.Cin = C, q.Cout = Pn.cout, q.Cout_pad = cout_pad;
            q.KH = q.KW = 1, q.stride = 1, q.dil = 1, q.pad_t = q.pad_l = 0;
            q.act = Pn.act, q.act_param = Pn.act_param;
            q.res = hp::tview{ nullptr, 0, 0, 0, 0 }, q.res_before_act = 0;
            q.out = tp.view(Pn.out_coff);
            q.out_f32 = nullptr, q.dbg = nullptr;
            HP_REQUIRE(hp::set_act(q), HP_ERR_INVALID, "layer %zu: activation %d cannot be fused into a dense conv", i + 1, Pn.act);
            HP_REQUIRE(hp::sepconv_variant(p) != 0, HP_ERR_INVALID, "layer %zu: no fused separable kernel for this block (set HP_NO_FUSE=1)", i);
            st.flops = 2.0 * opix * L.cin * 9 + 2.0 * opix * Pn.cout * C;
            st.bytes = (double)ti.H * ti.W * L.cin * 2 + opix * Pn.cout * 2 + (double)Pn.cout * C * 2;
            steps.push_back(st);
            ++i; // the pointwise layer is part of this step
            continue;
        } else if (L.op == HP_OP_DWCONV) {
            HP_REQUIRE(L.kh == 3 && L.kw == 3, HP_ERR_INVALID, "layer %zu: depthwise kernels are 3x3", i);
            HP_REQUIRE(L.cin % 8 == 0 && L.in_coff % 8 == 0 && L.out_coff % 8 == 0, HP_ERR_INVALID, "layer %zu: depthwise needs 8-aligned channels", i);
            const float* w = blob(L.w_off, (size_t)L.cin * 9, "weights", i);
            if (!w)
                return HP_ERR_INVALID;
            std::vector<__half> packed((size_t)9 * L.cin);
            for (int c = 0; c < L.cin; ++c)
                for (int t = 0; t < 9; ++t)
                    packed[(size_t)t * L.cin + c] = __float2half(w[(size_t)c * 9 + t]);
            std::vector<float> bias(L.cin, 0.f);
            if (L.b_off >= 0) {
                const float* b = blob(L.b_off, L.cin, "bias", i);
                if (!b)
                    return HP_ERR_INVALID;
                std::copy(b, b + L.cin, bias.begin());
            }
            auto& p = st.dp;
            void* dw = nullptr;
            void* db = nullptr;
            HP_TRY(upload(packed.data(), packed.size() * sizeof(__half), &dw));
            HP_TRY(upload(bias.data(), bias.size() * sizeof(float), &db));
            p.w = (const __half*)dw, p.bias = (const float*)db;
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = L.cin, p.stride = L.stride, p.dil = L.dil;
            p.pad_t = g.pt, p.pad_l = g.pl, p.act = L.act, p.act_param = L.act_param, p.halo = ti.P;
            p.out = to.view(L.out_coff);
            st.flops = 2.0 * opix * L.cin * 9;
            st.bytes = (double)ti.H * ti.W * L.cin * 2 + opix * L.cin * 2;
        } else if (L.op == HP_OP_UPSAMPLE) {
            HP_REQUIRE(L.cin % 8 == 0 && L.in_coff % 8 == 0 && L.out_coff % 8 == 0, HP_ERR_INVALID, "layer %zu: up-sampling needs 8-aligned channels", i);
            auto& p = st.pp;
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = L.cin, p.k = L.kh, p.stride = L.stride, p.pad_t = p.pad_l = 0;
            p.out = to.view(L.out_coff);
            st.flops = 0;
            st.bytes = (double)ti.H * ti.W * L.cin * 2 + opix * L.cin * 2;
        } else { // max-pool
            HP_REQUIRE(L.cin % 8 == 0 && L.in_coff % 8 == 0 && L.out_coff % 8 == 0 && L.kh == L.kw, HP_ERR_INVALID, "layer %zu: bad pool", i);
            auto& p = st.pp;
            p.in = ti.view(L.in_coff);
            p.H = ti.H, p.W = ti.W, p.OH = g.OH, p.OW = g.OW, p.C = L.cin, p.k = L.kh, p.stride = L.stride;
            p.pad_t = g.pt, p.pad_l = g.pl;
            p.out = to.view(L.out_coff);
            st.flops = 0;
            st.bytes = (double)ti.H * ti.W * L.cin * 2 + opix * L.cin * 2;
        }
        steps.push_back(st);
    }
    // ---- split-K scratch: convolutions whose tiles leave CUs idle at this engine's batch (conv_splitk) share one buffer - the
    // schedule is serial on one stream, a convolution's partial sums are consumed by its own second launch
    {
        size_t need = 0;
        for (auto& st : steps) {
            size_t bytes = 0;
            if (st.op == HP_OP_CONV && !st.first && !st.f32 && !getenv("HP_NO_SPLITK") && hp::conv_splitk(st.cp, &bytes) > 1)
                need = std::max(need, bytes);
        }
        if (need) {
            weight_bufs.push_back(std::make_unique<hp::dev_buf>());
            HP_TRY(weight_bufs.back()->alloc(need));
            void* const d = weight_bufs.back()->p;
            for (auto& st : steps)
                if (st.op == HP_OP_CONV && !st.first && !st.f32) {
                    const int ks = hp::conv_splitk(st.cp, nullptr);
                    if (ks > 1)
                        st.cp.ksplit = ks, st.cp.splitk = (float*)d;
                }
        }
    }
    // ---- chains of 128-channel convolutions (LW-OpenPose's CPM / initial / refinement stages, lw_openpose.py:106-191): consecutive
    // steps [1x1 ->] 3x3 -> 3x3 whose intermediates nobody else reads run as ONE launch with the intermediates in LDS
    // (conv_chain.hip).  HP_NO_CHAIN=1 keeps one launch per layer (A/B measurements, and hp_engine_debug_tensor on an intermediate).
    if (!getenv("HP_NO_CHAIN") && !getenv("HP_NO_FUSE") && !f32) {
        auto plain_conv = [&](const step& st) { return st.op == HP_OP_CONV && !st.first && st.n_layers == 1; };
        // the tensor a step writes is read only by the given layers (as input or residual) and is no network output
        auto only_read_by = [&](int tensor, int la, int lb) {
            for (size_t j = 0; j < layers.size(); ++j)
                if ((int)j != la && (int)j != lb && (layers[j].in == tensor || layers[j].res == tensor))
                    return false;
            for (const auto& o : outputs)
                if (o.tensor == tensor)
                    return false;
            int writers = 0;
            for (const auto& L2 : layers)
                writers += L2.out == tensor;
            return writers == 1;
        };
        for (size_t k = 0; k + 1 < steps.size(); ++k) {
            if (!plain_conv(steps[k]) || !plain_conv(steps[k + 1]))
                continue;
            hp::chain_params ch{};
            int n = 0;
            const bool three = k + 2 < steps.size() && plain_conv(steps[k + 2]) && steps[k].cp.KH == 1;
            if (three) {
                const int l0 = steps[k].layer, l1 = steps[k + 1].layer, l2 = steps[k + 2].layer;
                const hp_layer &A = layers[l0], &Bn = layers[l1], &Cn = layers[l2];
                if (Bn.in == A.out && Bn.in_coff == A.out_coff && Cn.in == Bn.out && Cn.in_coff == Bn.out_coff && A.res < 0 && Bn.res < 0
                    && (Cn.res < 0 || Cn.res == A.out) && A.out_coff == 0 && only_read_by(A.out, l1, l2) && only_read_by(Bn.out, l2, l2)) {
                    ch.c0 = steps[k].cp, ch.c1 = steps[k + 1].cp, ch.c2 = steps[k + 2].cp;
                    ch.has_c0 = 1, ch.res_mode = Cn.res >= 0 ? 3 : 0;
                    n = 3;
                }
            }
            if (!n && steps[k].cp.KH == 3) {
                const int l1 = steps[k].layer, l2 = steps[k + 1].layer;
                const hp_layer &Bn = layers[l1], &Cn = layers[l2];
                if (Cn.in == Bn.out && Cn.in_coff == Bn.out_coff && !(Bn.res >= 0 && Cn.res >= 0) && only_read_by(Bn.out, l2, l2)
                    && Cn.res != Bn.out) {
                    ch.c1 = steps[k].cp, ch.c2 = steps[k + 1].cp;
                    ch.has_c0 = 0, ch.res_mode = Bn.res >= 0 ? 1 : Cn.res >= 0 ? 2 : 0;
                    n = 2;
                }
            }
            if (!n || !hp::conv_chain_variant(ch))
                continue;
            step& a = steps[k];
            a.op = OP_CHAIN, a.ch = ch, a.n_layers = n;
            for (int q = 1; q < n; ++q) {
                a.flops += steps[k + q].flops;
                tensors[layers[steps[k + q].layer].in]->elided = true; // allocated (pass 1) but never written
            }
            a.bytes = steps[k].bytes + steps[k + n - 1].bytes; // first layer's input + weights ... last layer's output (approximate)
            steps.erase(steps.begin() + k + 1, steps.begin() + k + n);
        }
    }
    // ---- ResNet bottlenecks (configs[3] / [4]): [3x3 ->] expansion 1x1 (+ shortcut) [-> the NEXT block's reduction 1x1] as one launch
    // (conv_bottleneck.hip).  The reduction may sit one or two steps further down the schedule (behind the next stage's projection
    // shortcut, which reads the same tensor): it has no other input, so it can run here.  HP_NO_BNECK=1 keeps one launch per layer.
    if (!getenv("HP_NO_BNECK") && !getenv("HP_NO_FUSE") && !f32) {
        auto plain_conv = [&](const step& st) { return st.op == HP_OP_CONV && !st.first && st.n_layers == 1; };
        auto readers_other_than = [&](int tensor, int la) { // some layer other than la reads the tensor, or it is a network output
            for (size_t j = 0; j < layers.size(); ++j)
                if ((int)j != la && (layers[j].in == tensor || layers[j].res == tensor))
                    return true;
            for (const auto& o : outputs)
                if (o.tensor == tensor)
                    return true;
            return false;
        };
        for (size_t k = 0; k < steps.size(); ++k) {
            if (!plain_conv(steps[k]))
                continue;
            hp::bneck_params bn{};
            size_t ke = k; // the expansion's step
            if (steps[k].cp.KH == 3 && k + 1 < steps.size() && plain_conv(steps[k + 1]) && steps[k + 1].cp.KH == 1) {
                const hp_layer &A = layers[steps[k].layer], &E = layers[steps[k + 1].layer];
                int writers = 0;
                for (const auto& L2 : layers)
                    writers += L2.out == A.out;
                if (E.in == A.out && E.in_coff == A.out_coff && A.res < 0 && writers == 1 && !readers_other_than(A.out, steps[k + 1].layer)) {
                    bn.c3 = steps[k].cp, bn.has_c3 = 1;
                    ke = k + 1;
                }
            }
            if (steps[ke].cp.KH != 1 || steps[ke].cp.Cout != 4 * steps[ke].cp.Cin)
                continue;
            bn.ce = steps[ke].cp;
            const hp_layer& E = layers[steps[ke].layer];
            size_t kr = 0; // the reduction's step
            for (size_t j = ke + 1; j < steps.size() && j <= ke + 2 && !kr; ++j) {
                if (!plain_conv(steps[j]) || steps[j].cp.KH != 1 || steps[j].cp.stride != 1)
                    continue;
                const hp_layer& R = layers[steps[j].layer];
                if (R.in != E.out || R.in_coff != E.out_coff || R.res >= 0)
                    continue;
                bool clash = false; // a step in between must not touch the reduction's output tensor
                for (size_t q = ke + 1; q < j; ++q) {
                    const hp_layer& Q = layers[steps[q].layer];
                    clash |= Q.in == R.out || Q.res == R.out || Q.out == R.out || steps[q].n_layers != 1;
                }
                if (!clash)
                    kr = j;
            }
            if (kr)
                bn.cr = steps[kr].cp, bn.has_cr = 1;
            // the block's shortcut is a projection of the block input that nobody else reads (the first block of a stage at stride 1):
            // computed inside the launch instead of written by one launch and read by the next
            size_t kp = steps.size();
            if (E.res >= 0 && bn.ce.Cin == 64 && E.res_before_act) {
                int writers = 0;
                for (const auto& L2 : layers)
                    writers += L2.out == E.res;
                for (size_t j = 0; j < k && kp == steps.size(); ++j) {
                    if (!plain_conv(steps[j]) || layers[steps[j].layer].out != E.res)
                        continue;
                    const hp_layer& P = layers[steps[j].layer];
                    if (steps[j].cp.KH == 1 && steps[j].cp.stride == 1 && P.res < 0 && P.out_coff == 0 && P.act == HP_ACT_NONE && writers == 1
                        && !readers_other_than(E.res, steps[ke].layer))
                        kp = j;
                }
            }
            const hp::tview res_view = bn.ce.res;
            if (kp < steps.size())
                bn.cp = steps[kp].cp, bn.has_cp = 1, bn.ce.res = hp::tview{ nullptr, 0, 0, 0, 0 };
            // ... and the block's own reduction reads the same block input and feeds only this 3x3: computed on the 3x3's halo tile
            size_t k0 = steps.size();
            if (kp < steps.size() && bn.has_c3) {
                const hp_layer &A3 = layers[steps[k].layer], &P = layers[steps[kp].layer];
                int writers = 0;
                for (const auto& L2 : layers)
                    writers += L2.out == A3.in;
                for (size_t j = 0; j < k && k0 == steps.size(); ++j) {
                    if (!plain_conv(steps[j]) || layers[steps[j].layer].out != A3.in)
                        continue;
                    const hp_layer& R = layers[steps[j].layer];
                    if (steps[j].cp.KH == 1 && steps[j].cp.stride == 1 && R.res < 0 && R.out_coff == 0 && A3.in_coff == 0 && R.in == P.in && R.in_coff == P.in_coff
                        && writers == 1 && !readers_other_than(A3.in, steps[k].layer))
                        k0 = j;
                }
                if (k0 < steps.size())
                    bn.c0 = steps[k0].cp, bn.has_c0 = 1;
            }
            // the stand-alone kernels of some of these shapes (256 -> 64 on the generic implicit GEMM) read row-major weights: the
            // fused kernel wants them in fragment order
            auto want_layout1 = [&](hp::conv_params& c) { c.w_layout = 1; };
            const int lay3 = bn.c3.w_layout, laye = bn.ce.w_layout, layr = bn.cr.w_layout, layp = bn.cp.w_layout, lay0 = bn.c0.w_layout;
            if (bn.has_c0)
                want_layout1(bn.c0);
            if (bn.has_c3)
                want_layout1(bn.c3);
            want_layout1(bn.ce);
            if (bn.has_cr)
                want_layout1(bn.cr);
            if (bn.has_cp)
                want_layout1(bn.cp);
            // the widest combination the kernel has an instance for: without the reduction (e.g. one to a width it does not serve), then
            // without the projection, then without both
            {
                const int had_cr = bn.has_cr, had_cp = bn.has_cp, had_c0 = bn.has_c0;
                bool ok = false;
                for (int drop = 0; drop < 8 && !ok; ++drop) { // bit 0: without the next reduction, bit 1: without the own reduction, bit 2: without the projection
                    if (((drop & 1) && !had_cr) || ((drop & 2) && !had_c0) || ((drop & 4) && !had_cp))
                        continue;
                    bn.has_cr = had_cr && !(drop & 1), bn.has_cp = had_cp && !(drop & 4), bn.has_c0 = had_c0 && !(drop & 2) && bn.has_cp;
                    bn.ce.res = bn.has_cp ? hp::tview{ nullptr, 0, 0, 0, 0 } : res_view;
                    ok = hp::bottleneck_variant(bn) != 0;
                }
                if (!ok)
                    continue;
                if (!bn.has_cr)
                    kr = 0;
                if (!bn.has_cp)
                    kp = steps.size();
                if (!bn.has_c0)
                    k0 = steps.size();
            }
            // repack what was uploaded row-major
            auto repack = [&](hp::conv_params& c, int had, int layer) -> int {
                if (had == 1)
                    return HP_OK;
                const hp_layer& L = layers[layer];
                const int taps = L.kh * L.kw, KQ = c.Cin / 16;
                const float* w = blob(L.w_off, (size_t)L.cout * taps * L.cin, "weights", (size_t)layer);
                if (!w)
                    return HP_ERR_INVALID;
                std::vector<__half> packed((size_t)taps * c.Cout_pad * c.Cin, __float2half(0.f));
                for (int co = 0; co < L.cout; ++co)
                    for (int t = 0; t < taps; ++t)
                        for (int ci = 0; ci < L.cin; ++ci)
                            packed[((((size_t)t * (c.Cout_pad / 32) + co / 32) * KQ + ci / 16) * 64 + (ci % 16 / 8) * 32 + co % 32) * 8 + ci % 8]
                                = __float2half(w[((size_t)co * taps + t) * L.cin + ci]);
                void* dw = nullptr;
                HP_TRY(upload(packed.data(), packed.size() * sizeof(__half), &dw));
                c.w = (const __half*)dw;
                return HP_OK;
            };
            if (bn.has_c3)
                HP_TRY(repack(bn.c3, lay3, steps[k].layer));
            HP_TRY(repack(bn.ce, laye, steps[ke].layer));
            if (kr)
                HP_TRY(repack(bn.cr, layr, steps[kr].layer));
            if (bn.has_cp)
                HP_TRY(repack(bn.cp, layp, steps[kp].layer));
            if (bn.has_c0)
                HP_TRY(repack(bn.c0, lay0, steps[k0].layer));
            step& a = steps[k];
            const double fl = (bn.has_c3 ? steps[ke].flops : 0) + (kr ? steps[kr].flops : 0) + (bn.has_cp ? steps[kp].flops : 0) + (bn.has_c0 ? steps[k0].flops : 0);
            const double by = (bn.has_c3 ? steps[ke].bytes : 0) + (kr ? steps[kr].bytes : 0);
            a.op = OP_BNECK, a.bn = bn, a.n_layers = 1 + bn.has_c3 + bn.has_cr + bn.has_cp + bn.has_c0;
            a.flops += fl, a.bytes += by;
            if (bn.has_c3)
                tensors[layers[steps[ke].layer].in]->elided = true; // the 3x3's output: allocated (pass 1) but never written
            if (bn.has_cp)
                tensors[E.res]->elided = true; // the projection's output
            if (kr)
                steps.erase(steps.begin() + kr);
            if (bn.has_c3)
                steps.erase(steps.begin() + ke);
            if (bn.has_c0)
                tensors[layers[steps[k].layer].in]->elided = true; // the own reduction's output
            // (the projection and the own reduction sit in front of this step: erase the later one first)
            size_t front[2] = { bn.has_cp ? kp : steps.size(), bn.has_c0 ? k0 : steps.size() };
            if (front[0] < front[1])
                std::swap(front[0], front[1]);
            for (size_t q : front)
                if (q < steps.size()) {
                    steps.erase(steps.begin() + q);
                    --k;
                }
        }
    }
    // ---- the MobileNet stem's separable blocks 32 -> 64 and 64 -> 128 (stride 2) as ONE launch, the 64-channel tensor between them (the
    // largest of the network) in LDS only (sepconv_pair_kernel).  HP_NO_SEPPAIR=1 keeps one launch per block.
    if (!getenv("HP_NO_SEPPAIR") && !getenv("HP_NO_FUSE") && !f32) {
        for (size_t k = 0; k + 1 < steps.size(); ++k) {
            step &a = steps[k], &b = steps[k + 1];
            if (a.op != OP_SEPCONV || b.op != OP_SEPCONV || a.sep_pair || b.sep_pair)
                continue;
            const hp_layer &Pa = layers[a.layer + 1], &Db = layers[b.layer];
            if (Db.in != Pa.out || Db.in_coff != Pa.out_coff)
                continue;
            bool private_tensor = true; // nobody but block b's depthwise layer reads it and it is no network output
            for (size_t j = 0; j < layers.size(); ++j)
                if ((int)j != b.layer && (layers[j].in == Pa.out || layers[j].res == Pa.out))
                    private_tensor = false;
            for (const auto& o : outputs)
                private_tensor = private_tensor && o.tensor != Pa.out;
            int writers = 0;
            for (const auto& L2 : layers)
                writers += L2.out == Pa.out;
            hp::seppair_params pp{ a.sp, b.sp };
            if (!private_tensor || writers != 1 || !hp::seppair_variant(pp))
                continue;
            a.sp2 = b.sp, a.sep_pair = true, a.n_layers += b.n_layers;
            a.flops += b.flops;
            a.bytes = a.bytes + b.bytes - 2.0 * (double)a.sp.OH * a.sp.OW * Pa.cout * 2; // the tensor in between is neither written nor read
            tensors[Pa.out]->elided = true; // allocated (pass 1) but never written
            steps.erase(steps.begin() + k + 1);
        }
    }
    // sibling heads (conf / paf branch of one stage: same input, same geometry, neither reads the other) share a launch
    if (!getenv("HP_NO_PAIR_HEADS") && !f32) {
        for (size_t k = 0; k + 1 < steps.size(); ++k) {
            step &a = steps[k], &b = steps[k + 1];
            if (a.op != OP_MLPHEAD || b.op != OP_MLPHEAD || a.paired)
                continue;
            const auto &x = a.hp_, &y = b.hp_;
            if (x.in.p != y.in.p || x.in.coff != y.in.coff || x.K1 != y.K1 || x.H != y.H || x.W != y.W || x.pw.Cout > 64 || y.pw.Cout > 64)
                continue;
            a.hp2_ = b.hp_, a.paired = true, a.n_layers += b.n_layers;
            a.flops += b.flops, a.bytes += b.bytes;
            steps.erase(steps.begin() + k + 1);
        }
    }
    HP_REQUIRE(!steps.empty() && steps[0].first, HP_ERR_INVALID, "engine: the first layer must be a CONV reading tensor 0");

    HP_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HP_HIP_TRY(hipEventCreate(&ev0));
    HP_HIP_TRY(hipEventCreate(&ev1));
    HP_HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HP_HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return HP_OK;
}

namespace {
// frames [b0, ..) of a tensor view / of an fp32 NCHW output
inline hp::tview32 at_frame(hp::tview32 v, int b0)
{
    if (v.p)
        v.p += (long)b0 * v.img * v.cs;
    return v;
}
} // namespace

int hp_engine::run_step(step& st, const uint8_t* u8, const float* f32, int n, hipStream_t s, int b0)
{
    if (st.f32 && b0 > 0) {
        // the second half-batch (hp_engine_set_concurrency): the same step on frames [b0, b0 + n) - every view moved by b0 images
        step t = st;
        if (t.first) {
            t.fp32.out = at_frame(st.fp32.out, b0);
            return run_step(t, u8 ? u8 + (size_t)b0 * in_h * in_w * 3 : nullptr, f32 ? f32 + (size_t)b0 * in_h * in_w * 3 : nullptr, n, s, 0);
        }
        if (t.op == HP_OP_CONV) {
            t.cp32.in = at_frame(st.cp32.in, b0), t.cp32.out = at_frame(st.cp32.out, b0), t.cp32.res = at_frame(st.cp32.res, b0);
            if (t.cp32.out_f32)
                t.cp32.out_f32 += (size_t)b0 * t.cp32.Cout * t.cp32.OH * t.cp32.OW;
        } else if (t.op == HP_OP_DWCONV)
            t.dp32.in = at_frame(st.dp32.in, b0), t.dp32.out = at_frame(st.dp32.out, b0);
        else
            t.pp32.in = at_frame(st.pp32.in, b0), t.pp32.out = at_frame(st.pp32.out, b0);
        return run_step(t, u8, f32, n, s, 0);
    }
    if (st.f32) {
        if (st.first) {
            st.fp32.in_u8 = u8, st.fp32.in_f32 = f32, st.fp32.B = n;
            HP_HIP_TRY(hp::launch_first_conv32(st.fp32, s));
        } else if (st.op == HP_OP_CONV) {
            st.cp32.B = n, st.cp32.npix = n * st.cp32.OH * st.cp32.OW;
            if (st.head32) {
                HP_HIP_TRY(hp::launch_conv32_head(st.cp32, st.hh, s));
            } else if (st.wino) {
                st.cp32.latency = parts > 1;
                if (st.cp32.w_wino3) {
                    HP_HIP_TRY(hp::launch_conv32_winograd3(st.cp32, s));
                    static const bool dbg_w3 = getenv("HP_DIRECT_DBG") != nullptr;
                    if (dbg_w3) { // block timeline (s_memtime, block 9, thread 0): start | chunk 0 transformed | per chunk: patch stored, multiplied | transformed out | stored
                        unsigned long long* dbg = nullptr;
                        HP_HIP_TRY(hipMalloc(&dbg, 64 * 8));
                        HP_HIP_TRY(hipMemset(dbg, 0, 64 * 8));
                        hp::conv32_params q = st.cp32;
                        q.dbg = dbg;
                        HP_HIP_TRY(hp::launch_conv32_winograd3(q, s));
                        HP_HIP_TRY(hipStreamSynchronize(s));
                        unsigned long long h[64];
                        HP_HIP_TRY(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                        fprintf(stderr, "winograd3 layer %d %d->%d cycles [start | chunk 0 transformed | per chunk: patch stored, multiplied | output transformed | stored]:", st.layer, q.Cin, q.Cout);
                        for (int i = 1; i < 60 && h[i]; ++i)
                            fprintf(stderr, " %llu", h[i] - h[i - 1]);
                        fprintf(stderr, "\n");
                        (void)hipFree(dbg);
                    }
                } else
                    HP_HIP_TRY(hp::launch_conv32_winograd(st.cp32, s));
                static const bool dbg_wino = getenv("HP_DIRECT_DBG") != nullptr;
                if (dbg_wino && !st.cp32.w_wino3) { // block timeline (s_memtime = shader cycles, block (1, 0), thread 0), printed per launch
                    unsigned long long* dbg = nullptr;
                    HP_HIP_TRY(hipMalloc(&dbg, 128 * 8));
                    HP_HIP_TRY(hipMemset(dbg, 0, 128 * 8));
                    hp::conv32_params q = st.cp32;
                    q.dbg = dbg;
                    HP_HIP_TRY(hp::launch_conv32_winograd(q, s));
                    HP_HIP_TRY(hipStreamSynchronize(s));
                    unsigned long long h[128];
                    HP_HIP_TRY(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                    fprintf(stderr, "winograd layer %d %d->%d tile %d cycles [start | staged, transformed, multiplied per chunk | stored]:", st.layer, q.Cin, q.Cout,
                        hp::conv32_winograd_tile(q));
                    for (int i = 1; i < 119 && h[i]; ++i)
                        fprintf(stderr, " %llu", h[i] - h[i - 1]);
                    int occ = 0;
                    (void)hp::conv32_winograd_occupancy(q, &occ);
                    fprintf(stderr, " | s_memtime ticks %llu in %llu ticks of the 100 MHz clock; blocks per CU %d\n", h[121] - h[119], h[122] - h[120], occ);
                    (void)hipFree(dbg);
                }
            } else if (st.cin_split) { // the direct kernel: on the fp16 pipe (HP_DTYPE_F32S until a value left fp16's range) or on the fp32 pipe
                hp::conv32_params q = st.cp32;
                q.Cin = st.cin_split;
                HP_HIP_TRY(hp::launch_conv32_direct(q, dtype == HP_DTYPE_F32S && !split_off, s));
                static const bool dbg_direct = getenv("HP_DIRECT_DBG") != nullptr;
                if (dbg_direct) { // block timeline (s_memtime = shader cycles, block (1, 0), thread 0), printed per launch
                    unsigned long long* dbg = nullptr;
                    HP_HIP_TRY(hipMalloc(&dbg, 64 * 8));
                    HP_HIP_TRY(hipMemset(dbg, 0, 64 * 8));
                    q.dbg = dbg;
                    HP_HIP_TRY(hp::launch_conv32_direct(q, dtype == HP_DTYPE_F32S && !split_off, s));
                    HP_HIP_TRY(hipStreamSynchronize(s));
                    unsigned long long h[64];
                    HP_HIP_TRY(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                    fprintf(stderr, "direct layer %d %dx%d %d->%d tile %d cycles [start | staged, multiplied per chunk | stored]:", st.layer, q.KH, q.KW, q.Cin, q.Cout,
                        hp::conv32_direct_tile(q, dtype == HP_DTYPE_F32S && !split_off));
                    for (int i = 1; i < 64 && h[i]; ++i)
                        fprintf(stderr, " %llu", h[i] - h[i - 1]);
                    fprintf(stderr, "\n");
                    (void)hipFree(dbg);
                }
            } else {
                HP_HIP_TRY(hp::launch_conv32(st.cp32, s));
                static const bool dbg_c32 = getenv("HP_DIRECT_DBG") != nullptr;
                static const int dbg_min_cin = getenv("HP_DIRECT_DBG_MINCIN") ? atoi(getenv("HP_DIRECT_DBG_MINCIN")) : 256;
                if (dbg_c32 && st.cp32.Cin >= dbg_min_cin) { // block timeline (s_memtime, block 9, thread 0): start | first tile staged | every 8 K-steps | stored
                    constexpr int NDBG = 128 + 3 * 4096;
                    unsigned long long* dbg = nullptr;
                    HP_HIP_TRY(hipMalloc(&dbg, NDBG * 8));
                    HP_HIP_TRY(hipMemset(dbg, 0, NDBG * 8));
                    hp::conv32_params q = st.cp32;
                    q.dbg = dbg;
                    HP_HIP_TRY(hp::launch_conv32(q, s));
                    HP_HIP_TRY(hipStreamSynchronize(s));
                    std::vector<unsigned long long> hv(NDBG);
                    unsigned long long* h = hv.data();
                    HP_HIP_TRY(hipMemcpy(h, dbg, NDBG * 8, hipMemcpyDeviceToHost));
                    fprintf(stderr, "conv32 layer %d %dx%d %d->%d tile %d cycles [start | staged | per 8 K-steps | stored]:", st.layer, q.KH, q.KW, q.Cin, q.Cout, hp::conv32_tile(q));
                    for (int i = 1; i < 128 && h[i]; ++i)
                        fprintf(stderr, " %llu", h[i] - h[i - 1]);
                    fprintf(stderr, "\n");
                    // residency: every block's (start, end) on the 100 MHz clock and the CU it ran on (XCC_ID, HW_ID: se_id [15:13], sh_id [12], cu_id [11:8])
                    struct blk { unsigned long long t0, t1; unsigned cu; };
                    std::vector<blk> bl;
                    unsigned long long tmin = ~0ull, tmax = 0;
                    for (int b = 0; b < 4096; ++b) {
                        const unsigned long long t0 = h[128 + 3 * b], t1 = h[128 + 3 * b + 1], id = h[128 + 3 * b + 2];
                        if (!t0 || !t1)
                            continue;
                        bl.push_back({ t0, t1, (unsigned)(((id >> 32) & 0xf) << 8 | ((id >> 8) & 0xff)) });
                        tmin = std::min(tmin, t0), tmax = std::max(tmax, t1);
                    }
                    if (!bl.empty()) {
                        std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev; // per CU: (time, +1 / -1)
                        double dsum = 0, dmin = 1e30, dmax = 0;
                        for (const auto& b : bl) {
                            ev[b.cu].push_back({ b.t0, +1 }), ev[b.cu].push_back({ b.t1, -1 });
                            const double d = (b.t1 - b.t0) * 0.01;
                            dsum += d, dmin = std::min(dmin, d), dmax = std::max(dmax, d);
                        }
                        int peak = 0;
                        std::map<int, int> blocks_per_cu, peak_hist;
                        for (auto& kv : ev) {
                            std::sort(kv.second.begin(), kv.second.end());
                            int cur = 0, pk = 0;
                            for (auto& e2 : kv.second)
                                cur += e2.second, pk = std::max(pk, cur);
                            peak = std::max(peak, pk), ++peak_hist[pk], ++blocks_per_cu[(int)kv.second.size() / 2];
                        }
                        fprintf(stderr, "  residency: %zu blocks on %zu CUs in %.2f us (first start -> last end); block duration %.2f .. %.2f us, mean %.2f; peak resident blocks per CU:",
                            bl.size(), ev.size(), (tmax - tmin) * 0.01, dmin, dmax, dsum / bl.size());
                        for (auto& kv : peak_hist)
                            fprintf(stderr, " %d x%d", kv.first, kv.second);
                        fprintf(stderr, "; blocks run per CU:");
                        for (auto& kv : blocks_per_cu)
                            fprintf(stderr, " %d x%d", kv.first, kv.second);
                        fprintf(stderr, "; active blocks at 10 %% .. 90 %% of the launch:");
                        for (int k = 1; k < 10; ++k) {
                            const unsigned long long t = tmin + (tmax - tmin) * k / 10;
                            int a = 0;
                            for (const auto& b : bl)
                                a += b.t0 <= t && t < b.t1;
                            fprintf(stderr, " %d", a);
                        }
                        // block 9 (the one with the s_memtime stamps) on the 100 MHz clock; duration histogram; mean duration per XCD; starts of the late blocks
                        fprintf(stderr, "; block 9: %.2f us", (h[128 + 3 * 9 + 1] - h[128 + 3 * 9]) * 0.01);
                        fprintf(stderr, "; durations (10 bins from min to max):");
                        int hist[10] = { 0 };
                        for (const auto& b : bl)
                            ++hist[std::min(9, (int)(((b.t1 - b.t0) * 0.01 - dmin) / std::max(1e-9, dmax - dmin) * 10))];
                        for (int k = 0; k < 10; ++k)
                            fprintf(stderr, " %d", hist[k]);
                        double xs[16] = { 0 };
                        int xn[16] = { 0 };
                        for (const auto& b : bl)
                            xs[(b.cu >> 8) & 15] += (b.t1 - b.t0) * 0.01, ++xn[(b.cu >> 8) & 15];
                        fprintf(stderr, "; mean duration per XCD:");
                        for (int k = 0; k < 16; ++k)
                            if (xn[k])
                                fprintf(stderr, " %.1f", xs[k] / xn[k]);
                        double late0 = 1e30, late_d = 0;
                        int nlate = 0;
                        for (const auto& b : bl)
                            if ((b.t0 - tmin) * 0.01 > 5.0)
                                late0 = std::min(late0, (b.t0 - tmin) * 0.01), late_d += (b.t1 - b.t0) * 0.01, ++nlate;
                        if (nlate)
                            fprintf(stderr, "; %d blocks started later than 5 us after the first (earliest at %.1f us), their mean duration %.2f us", nlate, late0, late_d / nlate);
                        fprintf(stderr, "\n");
                    }
                    (void)hipFree(dbg);
                }
            }
        } else if (st.op == HP_OP_DWCONV) {
            st.dp32.B = n;
            HP_HIP_TRY(hp::launch_dwconv32(st.dp32, s));
        } else if (st.op == HP_OP_UPSAMPLE) {
            st.pp32.B = n;
            HP_HIP_TRY(hp::launch_upsample32(st.pp32, s));
        } else {
            st.pp32.B = n;
            HP_HIP_TRY(hp::launch_maxpool32(st.pp32, s));
        }
        return HP_OK;
    }
    if (st.first) {
        st.fp.in_u8 = u8, st.fp.in_f32 = f32, st.fp.B = n;
        HP_HIP_TRY(hp::launch_first_conv(st.fp, s));
    } else if (st.op == HP_OP_CONV) {
        st.cp.B = n, st.cp.npix = n * st.cp.OH * st.cp.OW;
        HP_HIP_TRY(hp::launch_conv_mfma(st.cp, s));
        if (dbg_conv && hp::conv_mfma_tile(st.cp) / 100000 == 52) { // block timeline of the pixel-block GEMM
            unsigned long long* dbg = nullptr;
            HP_HIP_TRY(hipMalloc(&dbg, 64 * 8));
            HP_HIP_TRY(hipMemset(dbg, 0, 64 * 8));
            st.cp.dbg = dbg;
            HP_HIP_TRY(hp::launch_conv_mfma(st.cp, s));
            HP_HIP_TRY(hipStreamSynchronize(s));
            unsigned long long h[64];
            HP_HIP_TRY(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "conv layer %d %d->%d tile %d consumer:", st.layer, st.cp.Cin, st.cp.Cout, hp::conv_mfma_tile(st.cp));
            for (int i = 1; i < 32 && h[i]; ++i)
                fprintf(stderr, " %llu", h[i] - h[i - 1]);
            fprintf(stderr, " | producer (from consumer start %lld):", (long long)(h[32] - h[0]));
            for (int i = 33; i < 64 && h[i]; ++i)
                fprintf(stderr, " %llu", h[i] - h[i - 1]);
            fprintf(stderr, "\n");
            st.cp.dbg = nullptr;
            (void)hipFree(dbg);
        }
    } else if (st.op == OP_BNECK) {
        st.bn.c3.B = st.bn.ce.B = st.bn.cr.B = n;
        HP_HIP_TRY(hp::launch_bottleneck(st.bn, s));
        if (dbg_bn) { // block 0's phase timeline (s_memtime deltas) and the start / end of the first 1024 blocks (100 MHz clock)
            constexpr int NDBG = 64 + 2 * 1024;
            unsigned long long* dbg = nullptr;
            HP_HIP_TRY(hipMalloc(&dbg, NDBG * 8));
            HP_HIP_TRY(hipMemset(dbg, 0, NDBG * 8));
            st.bn.ce.dbg = dbg;
            HP_HIP_TRY(hp::launch_bottleneck(st.bn, s));
            HP_HIP_TRY(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(NDBG);
            HP_HIP_TRY(hipMemcpy(h.data(), dbg, NDBG * 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "bottleneck layer %d variant %d timeline:", st.layer, hp::bottleneck_variant(st.bn));
            for (int i = 1; i < 60 && h[i]; ++i)
                fprintf(stderr, " %llu", h[i] - h[i - 1]);
            unsigned long long t0 = ~0ull, t1 = 0, dmin = ~0ull, dmax = 0, dsum = 0;
            int nb = 0;
            for (int i = 0; i < 1024 && h[64 + 2 * i]; ++i, ++nb) {
                const unsigned long long d = h[65 + 2 * i] - h[64 + 2 * i];
                t0 = std::min(t0, h[64 + 2 * i]), t1 = std::max(t1, h[65 + 2 * i]), dmin = std::min(dmin, d), dmax = std::max(dmax, d), dsum += d;
            }
            if (nb)
                fprintf(stderr, "\n  first %d blocks: %.2f us from first start to last end; block duration %.2f .. %.2f us, mean %.2f", nb, (t1 - t0) * 0.01,
                    dmin * 0.01, dmax * 0.01, dsum * 0.01 / nb);
            fprintf(stderr, "\n");
            st.bn.ce.dbg = nullptr;
            (void)hipFree(dbg);
        }
    } else if (st.op == OP_CHAIN) {
        st.ch.c0.B = st.ch.c1.B = st.ch.c2.B = n;
        HP_HIP_TRY(hp::launch_conv_chain(st.ch, s));
        if (dbg_chain) { // block timeline (s_memtime deltas of block 0, thread 0), printed per launch
            unsigned long long* dbg = nullptr;
            HP_HIP_TRY(hipMalloc(&dbg, 32 * 8));
            HP_HIP_TRY(hipMemset(dbg, 0, 32 * 8));
            st.ch.c2.dbg = dbg;
            HP_HIP_TRY(hp::launch_conv_chain(st.ch, s));
            HP_HIP_TRY(hipStreamSynchronize(s));
            unsigned long long h[32];
            HP_HIP_TRY(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "chain layer %d variant %d timeline:", st.layer, hp::conv_chain_variant(st.ch));
            for (int i = 1; i < 32 && h[i]; ++i)
                fprintf(stderr, " %llu", h[i] - h[i - 1]);
            fprintf(stderr, "\n");
            st.ch.c2.dbg = nullptr;
            (void)hipFree(dbg);
        }
    } else if (st.op == OP_SEPCONV) {
        st.sp.B = n, st.sp.pw.B = n, st.sp.pw.npix = n * st.sp.OH * st.sp.OW;
        if (st.sep_pair) {
            st.sp2.B = n, st.sp2.pw.B = n, st.sp2.pw.npix = n * st.sp2.OH * st.sp2.OW;
            HP_HIP_TRY(hp::launch_seppair(hp::seppair_params{ st.sp, st.sp2 }, s));
            return HP_OK;
        }
        HP_HIP_TRY(hp::launch_sepconv(st.sp, s));
        if (dbg_sep) { // block timeline (s_memtime deltas of block 0, thread 0) of every separable block, printed per launch
            unsigned long long* dbg = nullptr;
            constexpr int NDBG = 64 + 2 * 1024 + 64; // [0, 64) block 0's stamps, then (start, end) of the first 1024 blocks (100 MHz clock)
            HP_HIP_TRY(hipMalloc(&dbg, NDBG * 8));
            HP_HIP_TRY(hipMemset(dbg, 0, NDBG * 8));
            st.sp.pw.dbg = dbg;
            HP_HIP_TRY(hp::launch_sepconv(st.sp, s));
            HP_HIP_TRY(hipStreamSynchronize(s));
            std::vector<unsigned long long> hbuf(NDBG);
            unsigned long long* h = hbuf.data();
            HP_HIP_TRY(hipMemcpy(h, dbg, NDBG * 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "sep layer %d C=%d timeline:", st.layer, st.sp.C);
            for (int i = 1; i < 40 && h[i]; ++i)
                fprintf(stderr, " %llu", h[i] - h[i - 1]);
            if (h[41])
                fprintf(stderr, " | total-to-epi0 %llu pass1 %llu epi1 %llu | total %llu", h[41] - h[0], h[42] - h[41], h[43] - h[42], h[43] - h[0]);
            fprintf(stderr, "\n");
            if (h[2112]) {
                fprintf(stderr, "  wavefront 4 of block 1:");
                for (int i = 1; i < 40 && h[2112 + i]; ++i)
                    fprintf(stderr, " %llu", h[2112 + i] - h[2112 + i - 1]);
                fprintf(stderr, "\n");
            }
            if (h[64]) {
                unsigned long long t0 = ~0ull, t1 = 0, dmin = ~0ull, dmax = 0, smax = 0;
                int nb = 0;
                for (int i = 0; i < 1024 && h[64 + 2 * i]; ++i, ++nb) {
                    t0 = std::min(t0, h[64 + 2 * i]), t1 = std::max(t1, h[65 + 2 * i]), smax = std::max(smax, h[64 + 2 * i]);
                    dmin = std::min(dmin, h[65 + 2 * i] - h[64 + 2 * i]), dmax = std::max(dmax, h[65 + 2 * i] - h[64 + 2 * i]);
                }
                fprintf(stderr, "  %d blocks: first start -> last end %.2f us, starts spread over %.2f us, block duration %.2f .. %.2f us\n", nb,
                    (t1 - t0) * 0.01, (smax - t0) * 0.01, dmin * 0.01, dmax * 0.01);
            }
            st.sp.pw.dbg = nullptr;
            (void)hipFree(dbg);
        }
    } else if (st.op == OP_MLPHEAD) {
        st.hp_.B = n, st.hp_.pw.B = n;
        if (st.paired) {
            st.hp2_.B = n, st.hp2_.pw.B = n;
            HP_HIP_TRY(hp::launch_mlp_head_pair(st.hp_, st.hp2_, s));
        } else
            HP_HIP_TRY(hp::launch_mlp_head(st.hp_, s));
    } else if (st.op == HP_OP_DWCONV) {
        st.dp.B = n;
        HP_HIP_TRY(hp::launch_dwconv3x3(st.dp, s));
    } else if (st.op == HP_OP_UPSAMPLE) {
        st.pp.B = n;
        HP_HIP_TRY(hp::launch_upsample(st.pp, s));
    } else {
        st.pp.B = n;
        HP_HIP_TRY(hp::launch_maxpool(st.pp, s));
    }
    return HP_OK;
}

int hp_engine::enqueue_range(const uint8_t* u8, const float* f32, int b0, int n, hipStream_t s)
{
    for (size_t i = 0; i < steps.size(); ++i) {
        step& st = steps[i];
        // two fused heads in a row that read the same tensor (LW-OpenPose's heat-map and PAF heads of a stage): one grid (conv32_head.hip)
        if (st.f32 && st.head32 && i + 1 < steps.size() && steps[i + 1].f32 && steps[i + 1].head32) {
            step a = st, b = steps[i + 1];
            for (step* t : { &a, &b }) {
                t->cp32.B = n, t->cp32.npix = n * t->cp32.OH * t->cp32.OW;
                if (b0 > 0) {
                    t->cp32.in = at_frame(t->cp32.in, b0), t->cp32.out = at_frame(t->cp32.out, b0), t->cp32.res = at_frame(t->cp32.res, b0);
                    if (t->cp32.out_f32)
                        t->cp32.out_f32 += (size_t)b0 * t->cp32.Cout * t->cp32.OH * t->cp32.OW;
                }
            }
            if (hp::conv32_head_pair_ok(a.cp32, b.cp32)) {
                HP_HIP_TRY(hp::launch_conv32_head_pair(a.cp32, a.hh, b.cp32, b.hh, s));
                ++i;
                continue;
            }
        }
        HP_TRY(run_step(st, u8, f32, n, s, b0));
    }
    for (auto& o : outputs)
        if (o.fused_layer < 0) {
            const tensor_info& ti = *tensors[o.tensor];
            float* const dst = o.buf->as<float>() + (size_t)b0 * o.out_c() * o.x.out_h * o.x.out_w;
            if (is_f32())
                HP_HIP_TRY(hp::launch_output_transform32(at_frame(ti.view32(o.coff), b0), n, o.H, o.W, o.x, dst, s));
            else
                HP_HIP_TRY(hp::launch_output_transform(ti.view(o.coff), n, o.H, o.W, o.x, dst, s));
        }
    return HP_OK;
}

int hp_engine::enqueue(const uint8_t* u8, const float* f32, int n, hipStream_t s, const void* host_src, size_t frame_bytes)
{
    unsigned char* const dev_in = u8 ? (unsigned char*)u8 : (unsigned char*)f32;
    auto h2d = [&](int b0, int cnt, hipStream_t st) -> int {
        if (host_src)
            HP_HIP_TRY(hipMemcpyAsync(dev_in + (size_t)b0 * frame_bytes, (const unsigned char*)host_src + (size_t)b0 * frame_bytes, (size_t)cnt * frame_bytes, hipMemcpyHostToDevice, st));
        return HP_OK;
    };
    if (parts < 2 || !halves_ok() || n < 2) {
        HP_TRY(h2d(0, n, s));
        return enqueue_range(u8, f32, 0, n, s);
    }
    // two half-batches side by side: frames [0, n0) on `s`, frames [n0, n) on stream2, which joins `s` again.  The kernels of one half are
    // half as many blocks - what fills the chip is that the two halves are never in the same phase (a store-bound launch of one runs under an
    // MFMA-bound launch of the other): measured 2490 -> 1962 us per synchronous batch of 8 (profiles/r06_half_batch_probe.txt).  Frames are
    // independent and every kernel is batch-invariant bit for bit (tests), so the outputs are those of the one-stream schedule.
    // (A host batch goes up in two copies, each on the stream that reads it: an event recorded behind ONE hipMemcpyAsync from pageable memory
    // does not cover all of that copy's staged chunks - measured: stream2's first layer now and then read a few stale input rows of frames 4 / 5
    // of 8, tools/r6_halves_debug.py - while a kernel behind the copy on the same stream always saw it complete.)
    const int n0 = (n + 1) / 2;
    HP_HIP_TRY(hipEventRecord(ev_fork, s));
    HP_HIP_TRY(hipStreamWaitEvent(stream2, ev_fork, 0));
    HP_TRY(h2d(0, n0, s));
    HP_TRY(h2d(n0, n - n0, stream2));
    HP_TRY(enqueue_range(u8, f32, 0, n0, s));
    HP_TRY(enqueue_range(u8, f32, n0, n - n0, stream2));
    HP_HIP_TRY(hipEventRecord(ev_join, stream2));
    HP_HIP_TRY(hipStreamWaitEvent(s, ev_join, 0));
    return HP_OK;
}

extern "C" {

int hp_engine_create(hp_engine** out, const hp_engine_desc* desc)
{
    HP_REQUIRE(out && desc, HP_ERR_INVALID, "hp_engine_create: null argument");
    std::unique_ptr<hp_engine> e(new hp_engine());
    try { // descriptions come from files (hp_engine_load, ONNX import): absurd sizes must come back as error codes
        HP_TRY(e->build(desc));
    } catch (const std::exception& ex) {
        HP_REQUIRE(false, HP_ERR_INVALID, "hp_engine_create: %s", ex.what());
    }
    *out = e.release();
    return HP_OK;
}

// ---- serialized engines (reference: tensorrt::save, src/tensorrt.cpp:463-471; tensorrt_serialized + the deserialising
// constructor, include/hyperpose/utility/model.hpp:27-32, src/tensorrt.cpp:225-252).  The file holds what hp_engine_create
// was given - topology, outputs, pre-processing, fp32 weights - so loading rebuilds the identical engine without the model
// source; the packing into kernel layouts happens at load (tens of ms), there is no per-device tuning cache to carry.
namespace {
constexpr char ENGINE_MAGIC[8] = { 'H', 'P', 'E', 'N', 'G', '0', '0', '2' }; // 002: + dtype
struct engine_file_header {
    char magic[8];
    int32_t layer_size, output_size; // sizeof(hp_layer) / sizeof(hp_output_desc): ABI guard
    int32_t in_w, in_h, max_batch, flip_rb;
    double factor;
    float mean[3], inv_std[3];
    int32_t n_layers, n_outputs;
    uint64_t n_weights;
    int32_t dtype, reserved; // HP_DTYPE_*: TensorRT bakes the builder's precision into the plan it serializes, so does this file
};
} // namespace

int hp_engine_save(const hp_engine* e, const char* path)
{
    HP_REQUIRE(e && path, HP_ERR_INVALID, "hp_engine_save: null argument");
    FILE* f = fopen(path, "wb");
    HP_REQUIRE(f, HP_ERR_INVALID, "hp_engine_save: cannot open %s for writing", path);
    engine_file_header h{};
    memcpy(h.magic, ENGINE_MAGIC, 8);
    h.layer_size = (int32_t)sizeof(hp_layer), h.output_size = (int32_t)sizeof(hp_output_desc);
    h.in_w = e->in_w, h.in_h = e->in_h, h.max_batch = e->max_batch, h.flip_rb = e->flip_rb, h.factor = e->factor;
    for (int c = 0; c < 3; ++c)
        h.mean[c] = e->mean[c], h.inv_std[c] = e->inv_std[c];
    h.n_layers = (int32_t)e->layers.size(), h.n_outputs = (int32_t)e->out_descs.size(), h.n_weights = e->weights_blob.size();
    h.dtype = e->dtype;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fwrite(e->layers.data(), sizeof(hp_layer), e->layers.size(), f) == e->layers.size();
    ok = ok && fwrite(e->out_descs.data(), sizeof(hp_output_desc), e->out_descs.size(), f) == e->out_descs.size();
    ok = ok && fwrite(e->weights_blob.data(), sizeof(float), e->weights_blob.size(), f) == e->weights_blob.size();
    ok = (fclose(f) == 0) && ok;
    HP_REQUIRE(ok, HP_ERR_INVALID, "hp_engine_save: short write to %s", path);
    return HP_OK;
}

int hp_engine_load(hp_engine** out, const char* path, int max_batch)
{
    HP_REQUIRE(out && path, HP_ERR_INVALID, "hp_engine_load: null argument");
    FILE* f = fopen(path, "rb");
    HP_REQUIRE(f, HP_ERR_INVALID, "hp_engine_load: cannot open %s", path);
    engine_file_header h{};
    std::vector<hp_layer> layers;
    std::vector<hp_output_desc> outs;
    std::vector<float> w;
    // an HPENG001 file (before data_type was honoured) is an 002 file without the trailing (dtype, reserved) pair, and its engine was fp16
    constexpr size_t HDR_001 = offsetof(engine_file_header, dtype);
    bool ok = fread(&h, HDR_001, 1, f) == 1;
    if (ok && memcmp(h.magic, "HPENG001", 8) == 0)
        memcpy(h.magic, ENGINE_MAGIC, 8), h.dtype = HP_DTYPE_F16, h.reserved = 0;
    else
        ok = ok && fread(reinterpret_cast<char*>(&h) + HDR_001, sizeof(h) - HDR_001, 1, f) == 1;
    ok = ok && memcmp(h.magic, ENGINE_MAGIC, 8) == 0 && h.layer_size == (int32_t)sizeof(hp_layer)
        && h.output_size == (int32_t)sizeof(hp_output_desc) && h.n_layers > 0 && h.n_layers < (1 << 20) && h.n_outputs > 0
        && h.n_outputs < 4096 && h.n_weights < ((uint64_t)1 << 34) && (h.dtype == HP_DTYPE_F16 || h.dtype == HP_DTYPE_F32 || h.dtype == HP_DTYPE_F32S);
    if (ok) { // the counts must account for the file exactly before anything is allocated from them
        const long at = ftell(f);
        ok = fseek(f, 0, SEEK_END) == 0;
        const long size = ftell(f);
        ok = ok && fseek(f, at, SEEK_SET) == 0
            && (uint64_t)size == (uint64_t)at + (uint64_t)h.n_layers * sizeof(hp_layer) + (uint64_t)h.n_outputs * sizeof(hp_output_desc) + h.n_weights * sizeof(float);
    }
    if (ok) {
        layers.resize(h.n_layers), outs.resize(h.n_outputs), w.resize(h.n_weights);
        ok = fread(layers.data(), sizeof(hp_layer), layers.size(), f) == layers.size()
            && fread(outs.data(), sizeof(hp_output_desc), outs.size(), f) == outs.size() && fread(w.data(), sizeof(float), w.size(), f) == w.size();
    }
    fclose(f);
    HP_REQUIRE(ok, HP_ERR_INVALID, "hp_engine_load: %s is not an engine file written by this library version", path);
    hp_engine_desc d{};
    d.in_w = h.in_w, d.in_h = h.in_h, d.max_batch = max_batch > 0 ? max_batch : h.max_batch, d.factor = h.factor, d.flip_rb = h.flip_rb;
    for (int c = 0; c < 3; ++c)
        d.mean[c] = h.mean[c], d.inv_std[c] = h.inv_std[c];
    d.layers = layers.data(), d.n_layers = h.n_layers, d.outputs = outs.data(), d.n_outputs = h.n_outputs, d.weights = w.data(), d.n_weights = w.size();
    d.dtype = h.dtype;
    return hp_engine_create(out, &d);
}

void hp_engine_destroy(hp_engine* e)
{
    if (!e)
        return;
    if (e->stream)
        (void)hipStreamSynchronize(e->stream);
    delete e;
}

int hp_engine_max_batch(const hp_engine* e) { return e ? e->max_batch : HP_ERR_INVALID; }

int hp_engine_describe(const hp_engine* e, hp_engine_desc* d)
{
    HP_REQUIRE(e && d, HP_ERR_INVALID, "hp_engine_describe: null argument");
    d->in_w = e->in_w, d->in_h = e->in_h, d->max_batch = e->max_batch, d->factor = e->factor, d->flip_rb = e->flip_rb;
    for (int c = 0; c < 3; ++c)
        d->mean[c] = e->mean[c], d->inv_std[c] = e->inv_std[c];
    d->layers = e->layers.data(), d->n_layers = (int32_t)e->layers.size();
    d->outputs = e->out_descs.data(), d->n_outputs = (int32_t)e->out_descs.size();
    d->weights = e->weights_blob.data(), d->n_weights = e->weights_blob.size();
    d->dtype = e->dtype;
    return HP_OK;
}

int hp_engine_dtype(const hp_engine* e) { return e ? e->dtype : HP_ERR_INVALID; }

int hp_engine_input_size(const hp_engine* e, int* w, int* h)
{
    HP_REQUIRE(e && w && h, HP_ERR_INVALID, "hp_engine_input_size: null argument");
    *w = e->in_w, *h = e->in_h;
    return HP_OK;
}

int hp_engine::leave_split()
{
    HP_HIP_TRY(hipStreamSynchronize(stream));
    if (last.stream)
        HP_HIP_TRY(hipStreamSynchronize((hipStream_t)last.stream));
    for (auto& g : graphs)
        (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    split_off = true, ++split_fallbacks;
    *static_cast<volatile unsigned*>(ovf_flag.p) = 0;
    return HP_OK;
}

static int infer_common(hp_engine* e, const void* input, size_t frame_bytes, int n, int on_device, void* stream, int kind)
{
    HP_REQUIRE(e && input, HP_ERR_INVALID, "hp_engine_infer: null argument");
    if (e->split_overflowed()) // an earlier batch held a value outside fp16's range: from here on the fp32 matrix pipe
        HP_TRY(e->leave_split());
    e->last.input = input, e->last.frame_bytes = frame_bytes, e->last.n = n, e->last.on_device = on_device, e->last.kind = kind, e->last.stream = stream;
    HP_REQUIRE(n >= 1, HP_ERR_INVALID, "hp_engine_infer: empty batch");
    // src/tensorrt.cpp:439-443 throws std::logic_error here
    HP_REQUIRE(n <= e->max_batch, HP_ERR_CAPACITY, "Input batch size overflow: Yours@%d Max@%d", n, e->max_batch);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const void* dev_in = input;
    if (!on_device) {
        const size_t need = (size_t)e->max_batch * e->in_h * e->in_w * 3 * sizeof(float);
        if (e->in_stage.bytes < need)
            HP_TRY(e->in_stage.alloc(need));
        // (the copy is not part of the captured schedule; two half-batches: each half goes up on the stream that reads it, below)
        if (e->use_graph && !(e->parts == 2 && e->halves_ok() && n >= 2))
            HP_HIP_TRY(hipMemcpyAsync(e->in_stage.p, input, frame_bytes * n, hipMemcpyHostToDevice, s));
        dev_in = e->in_stage.p;
    }
    const uint8_t* u8 = kind == 0 ? (const uint8_t*)dev_in : nullptr;
    const float* f32 = kind == 1 ? (const float*)dev_in : nullptr;
    if (!e->use_graph)
        return e->enqueue(u8, f32, n, s, on_device ? nullptr : input, frame_bytes);

    // capture a range's schedule once per (frames, input buffer, first frame); replays cost one launch
    auto graph_for = [&](int b0, int cnt, hipStream_t cs, hipGraphExec_t* out) -> int {
        const hp_engine::graph_key key{ cnt, dev_in, kind, b0 };
        auto it = e->graphs.find(key);
        if (it == e->graphs.end()) {
            hipGraph_t graph = nullptr;
            HP_HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            const int rc = e->enqueue_range(u8, f32, b0, cnt, cs);
            const hipError_t ee = hipStreamEndCapture(cs, &graph);
            if (rc != HP_OK) {
                if (graph)
                    (void)hipGraphDestroy(graph);
                return rc;
            }
            HP_HIP_TRY(ee);
            hipGraphExec_t exec = nullptr;
            const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HP_HIP_TRY(ie);
            if (e->graphs.size() > 64) { // bound the cache for callers that pass a fresh pointer every time
                (void)hipStreamSynchronize(s); // earlier launches of these executables may still be running on this stream ...
                (void)hipStreamSynchronize(e->stream); // ... or on the engine's own
                if (e->stream2)
                    (void)hipStreamSynchronize(e->stream2);
                for (auto& g : e->graphs)
                    (void)hipGraphExecDestroy(g.second);
                e->graphs.clear();
            }
            it = e->graphs.emplace(key, exec).first;
        }
        *out = it->second;
        return HP_OK;
    };
    if (e->parts == 2 && e->halves_ok() && n >= 2) {
        // two half-batches side by side (hp_engine::enqueue has the reasoning): ONE graph with two branches is replayed branch after branch by
        // the runtime (measured: 2.27 -> 2.22 ms per call), two graphs on two streams run side by side (the probe's 2.49 -> 1.96 ms)
        const int n0 = (n + 1) / 2;
        hipGraphExec_t ga = nullptr, gb = nullptr;
        HP_TRY(graph_for(0, n0, s, &ga));
        HP_TRY(graph_for(n0, n - n0, e->stream2, &gb));
        HP_HIP_TRY(hipEventRecord(e->ev_fork, s));
        HP_HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
        if (!on_device) { // the second half's frames cross PCIe under the first half's first layers
            HP_HIP_TRY(hipMemcpyAsync(e->in_stage.p, input, frame_bytes * n0, hipMemcpyHostToDevice, s));
            HP_HIP_TRY(hipMemcpyAsync(e->in_stage.as<unsigned char>() + frame_bytes * n0, (const unsigned char*)input + frame_bytes * n0, frame_bytes * (n - n0),
                hipMemcpyHostToDevice, e->stream2));
        }
        HP_HIP_TRY(hipGraphLaunch(ga, s));
        HP_HIP_TRY(hipGraphLaunch(gb, e->stream2));
        HP_HIP_TRY(hipEventRecord(e->ev_join, e->stream2));
        HP_HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
        return HP_OK;
    }
    hipGraphExec_t g = nullptr;
    HP_TRY(graph_for(0, n, s, &g));
    HP_HIP_TRY(hipGraphLaunch(g, s));
    return HP_OK;
}

int hp_engine_infer_u8(hp_engine* e, const uint8_t* hwc_bgr, int n, int on_device, void* stream)
{
    HP_REQUIRE(e, HP_ERR_INVALID, "hp_engine_infer_u8: null engine");
    return infer_common(e, hwc_bgr, (size_t)e->in_h * e->in_w * 3, n, on_device, stream, 0);
}

int hp_engine_infer_f32(hp_engine* e, const float* nchw, int n, int on_device, void* stream)
{
    HP_REQUIRE(e, HP_ERR_INVALID, "hp_engine_infer_f32: null engine");
    return infer_common(e, nchw, (size_t)e->in_h * e->in_w * 3 * sizeof(float), n, on_device, stream, 1);
}

int hp_engine_synchronize(hp_engine* e)
{
    HP_REQUIRE(e, HP_ERR_INVALID, "null engine");
    HP_HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->last.stream && (hipStream_t)e->last.stream != e->stream) // the newest call ran on the caller's stream: the flag below is its kernels'
        HP_HIP_TRY(hipStreamSynchronize((hipStream_t)e->last.stream));
    if (e->split_overflowed() && e->last.input) {
        // HP_DTYPE_F32S: the batch just finished saw |x| > 65504 somewhere - its outputs are not trustworthy.  Run it again on the fp32
        // pipe before the caller reads them (a host input is copied again from the caller's buffer, which the API keeps valid until the
        // outputs are read)
        HP_TRY(e->leave_split());
        HP_TRY(infer_common(e, e->last.input, e->last.frame_bytes, e->last.n, e->last.on_device, e->last.stream, e->last.kind));
        HP_HIP_TRY(hipStreamSynchronize(e->last.stream ? (hipStream_t)e->last.stream : e->stream));
    }
    return HP_OK;
}

int hp_engine_split_fallbacks(const hp_engine* e) { return e ? e->split_fallbacks : HP_ERR_INVALID; }

int hp_engine_device_bytes(const hp_engine* e, uint64_t bytes[3])
{
    HP_REQUIRE(e && bytes, HP_ERR_INVALID, "hp_engine_device_bytes: null argument");
    bytes[0] = bytes[1] = bytes[2] = 0;
    for (const auto& t : e->tensors)
        if (t)
            bytes[0] += t->buf.bytes;
    for (const auto& a : e->arena)
        bytes[0] += a.mem->bytes;
    for (const auto& w : e->weight_bufs)
        bytes[1] += w->bytes;
    for (const auto& o : e->outputs)
        if (o.buf)
            bytes[2] += o.buf->bytes;
    return HP_OK;
}

int hp_debug_first_conv_verify(unsigned out[4], int reset)
{
    HP_REQUIRE(out, HP_ERR_INVALID, "hp_debug_first_conv_verify: null argument");
    hp::first_conv32_verify_counts(out, reset != 0);
    return HP_OK;
}

int hp_engine_arena_info(const hp_engine* e, uint64_t info[3])
{
    HP_REQUIRE(e && info, HP_ERR_INVALID, "hp_engine_arena_info: null argument");
    info[0] = e->arena.size(), info[1] = 0, info[2] = e->arena_private_bytes;
    for (const auto& a : e->arena)
        info[1] += a.tenants;
    return HP_OK;
}

void* hp_engine_stream(hp_engine* e) { return e ? (void*)e->stream : nullptr; }

int hp_engine_set_concurrency(hp_engine* e, int parts)
{
    HP_REQUIRE(e && (parts == 1 || parts == 2), HP_ERR_INVALID, "hp_engine_set_concurrency: parts must be 1 or 2");
    // HP_DTYPE_F32 / F32S only: the fp16 engine's fused launches take no frame offset.  (A split engine failed this in round 6 until the cause was
    // found in first_conv32_kernel's packed FMAs: DESIGN.md section 7B.8, hyperpose_amd/build.py.)
    if (!e->halves_ok())
        parts = 1;
    if (parts == e->parts)
        return HP_OK;
    HP_HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->last.stream)
        HP_HIP_TRY(hipStreamSynchronize((hipStream_t)e->last.stream));
    if (parts == 2 && !e->stream2)
        HP_HIP_TRY(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    for (auto& g : e->graphs) // (the captured schedules are of the other form)
        (void)hipGraphExecDestroy(g.second);
    e->graphs.clear();
    e->parts = parts;
    return HP_OK;
}

int hp_engine_concurrency(const hp_engine* e) { return e ? e->parts : HP_ERR_INVALID; }

int hp_engine_set_graph(hp_engine* e, int enable)
{
    HP_REQUIRE(e, HP_ERR_INVALID, "null engine");
    e->use_graph = enable != 0;
    return HP_OK;
}

int hp_engine_num_outputs(const hp_engine* e) { return e ? (int)e->outputs.size() : HP_ERR_INVALID; }

int hp_engine_output(const hp_engine* e, int i, const char** name, int shape[3], const float** dev)
{
    HP_REQUIRE(e && i >= 0 && i < (int)e->outputs.size(), HP_ERR_INVALID, "hp_engine_output: index %d", i);
    const out_info& o = e->outputs[i];
    if (name)
        *name = o.name.c_str();
    if (shape)
        shape[0] = o.out_c(), shape[1] = o.x.out_h, shape[2] = o.x.out_w;
    if (dev)
        *dev = o.buf->as<float>();
    return HP_OK;
}

int hp_engine_output_to_host(hp_engine* e, int i, int n, float* host)
{
    HP_REQUIRE(e && host && i >= 0 && i < (int)e->outputs.size() && n >= 1 && n <= e->max_batch, HP_ERR_INVALID, "hp_engine_output_to_host: bad argument");
    const out_info& o = e->outputs[i];
    HP_TRY(hp_engine_synchronize(e)); // (an HP_DTYPE_F32S engine re-runs the batch here if a value left fp16's range)
    HP_HIP_TRY(hipMemcpy(host, o.buf->p, (size_t)n * o.out_c() * o.x.out_h * o.x.out_w * sizeof(float), hipMemcpyDeviceToHost));
    return HP_OK;
}

int hp_engine_debug_tensor(hp_engine* e, int tensor, int n, float* host, int shape[3])
{
    HP_REQUIRE(e && tensor > 0 && tensor < (int)e->tensors.size() && e->tensors[tensor]->defined, HP_ERR_INVALID, "hp_engine_debug_tensor: bad tensor %d", tensor);
    HP_REQUIRE(!e->tensors[tensor]->unwritten, HP_ERR_STATE, "hp_engine_debug_tensor: tensor %d exists only as the fp32 network output", tensor);
    HP_REQUIRE(!e->tensors[tensor]->elided, HP_ERR_STATE, "hp_engine_debug_tensor: tensor %d lives only inside a fused separable block (HP_NO_FUSE=1 materialises it)", tensor);
    HP_REQUIRE(e->tensors[tensor]->arena_slot < 0 || e->arena[e->tensors[tensor]->arena_slot].tenants == 1, HP_ERR_STATE,
        "hp_engine_debug_tensor: tensor %d shares its buffer with later tensors of the same geometry (the activation arena); build the engine with HP_NO_ARENA=1 to look at it", tensor);
    const tensor_info& ti = *e->tensors[tensor];
    if (shape)
        shape[0] = ti.C, shape[1] = ti.H, shape[2] = ti.W;
    if (!host)
        return HP_OK;
    HP_REQUIRE(n >= 1 && n <= e->max_batch, HP_ERR_INVALID, "hp_engine_debug_tensor: bad batch");
    hp::dev_buf tmp;
    HP_TRY(tmp.alloc((size_t)n * ti.C * ti.H * ti.W * sizeof(float)));
    hp::out_xform px{};
    px.C = ti.C, px.act = 0, px.shuffle = 1, px.group = 0, px.out_h = ti.H, px.out_w = ti.W, px.scale = 1.f, px.grid = 0;
    if (e->is_f32())
        HP_HIP_TRY(hp::launch_output_transform32(ti.view32(0), n, ti.H, ti.W, px, tmp.as<float>(), e->stream));
    else
        HP_HIP_TRY(hp::launch_output_transform(ti.view(0), n, ti.H, ti.W, px, tmp.as<float>(), e->stream));
    HP_HIP_TRY(hipStreamSynchronize(e->stream));
    HP_HIP_TRY(hipMemcpy(host, tmp.p, tmp.bytes, hipMemcpyDeviceToHost));
    return HP_OK;
}

int hp_engine_profile(hp_engine* e, int n, int iters, hp_layer_time* out, int cap, int* n_out)
{
    HP_REQUIRE(e && n >= 1 && n <= e->max_batch && iters >= 1 && n_out, HP_ERR_INVALID, "hp_engine_profile: bad argument");
    // a zero-filled synthetic input: only timing matters here
    const size_t need = (size_t)e->max_batch * e->in_h * e->in_w * 3 * sizeof(float);
    if (e->in_stage.bytes < need)
        HP_TRY(e->in_stage.alloc(need));
    const uint8_t* u8 = e->in_stage.as<uint8_t>();
    int k = 0;
    for (auto& st : e->steps) {
        HP_TRY(e->run_step(st, u8, nullptr, n, e->stream)); // warm
        HP_HIP_TRY(hipEventRecord(e->ev0, e->stream));
        for (int it = 0; it < iters; ++it)
            HP_TRY(e->run_step(st, u8, nullptr, n, e->stream));
        HP_HIP_TRY(hipEventRecord(e->ev1, e->stream));
        HP_HIP_TRY(hipEventSynchronize(e->ev1));
        float ms = 0;
        HP_HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        if (out && k < cap) {
            out[k].layer = st.layer, out[k].op = st.op;
            out[k].tile = st.op == OP_SEPCONV ? 4000000 + (st.sep_pair ? 20 : hp::sepconv_variant(st.sp))
                : st.op == OP_MLPHEAD        ? 6000000 + st.hp_.K1
                : st.op == OP_CHAIN          ? 7000000 + hp::conv_chain_variant(st.ch)
                : st.op == OP_BNECK          ? 9000000 + hp::bottleneck_variant(st.bn)
                : (st.op == HP_OP_CONV && !st.first) ? (st.f32 ? (st.head32 ? hp::conv32_head_tile(st.hh.HID, st.cp32.Cout) : st.wino ? hp::conv32_winograd_tile(st.cp32) : st.cin_split ? hp::conv32_direct_tile(st.cp32, e->dtype == HP_DTYPE_F32S && !e->split_off) : hp::conv32_tile(st.cp32)) : hp::conv_mfma_tile(st.cp))
                                                    : 0;
            out[k].ms = ms / iters;
            out[k].flops = st.flops * n, out[k].bytes = st.bytes * n;
        }
        ++k;
    }
    *n_out = k;
    return HP_OK;
}

// Machine time per launch: step k of two engines of the same model launched alternately on their two streams, so that two instances of
// the kernel share the GPU the way two pipes' kernels do; reported = elapsed / (2 * iters).  (A kernel that fills the chip gains nothing
// from the second stream; one that leaves CUs, slots or pipes idle does - the end-to-end rate with several pipes follows this figure.)
int hp_engine_profile_pair(hp_engine* e, hp_engine* f, int n, int iters, hp_layer_time* out, int cap, int* n_out)
{
    HP_REQUIRE(e && f && e != f && n >= 1 && n <= e->max_batch && n <= f->max_batch && iters >= 1 && n_out && e->steps.size() == f->steps.size(),
        HP_ERR_INVALID, "hp_engine_profile_pair: bad argument");
    for (size_t i = 0; i < e->steps.size(); ++i) // the same schedule: step by step the same operation on the same amount of work
        HP_REQUIRE(e->steps[i].op == f->steps[i].op && e->steps[i].layer == f->steps[i].layer && e->steps[i].flops == f->steps[i].flops
                && e->steps[i].bytes == f->steps[i].bytes,
            HP_ERR_INVALID, "hp_engine_profile_pair: the engines' schedules differ at step %zu", i);
    const size_t need = (size_t)e->max_batch * e->in_h * e->in_w * 3 * sizeof(float);
    for (hp_engine* g : { e, f })
        if (g->in_stage.bytes < need)
            HP_TRY(g->in_stage.alloc(need));
    int k = 0;
    for (size_t i = 0; i < e->steps.size(); ++i) {
        auto &sa = e->steps[i], &sb = f->steps[i];
        HP_TRY(e->run_step(sa, e->in_stage.as<uint8_t>(), nullptr, n, e->stream));
        HP_TRY(f->run_step(sb, f->in_stage.as<uint8_t>(), nullptr, n, f->stream));
        HP_HIP_TRY(hipStreamSynchronize(e->stream));
        HP_HIP_TRY(hipStreamSynchronize(f->stream));
        HP_HIP_TRY(hipEventRecord(e->ev0, e->stream));
        HP_HIP_TRY(hipEventRecord(f->ev0, f->stream));
        for (int it = 0; it < iters; ++it) {
            HP_TRY(e->run_step(sa, e->in_stage.as<uint8_t>(), nullptr, n, e->stream));
            HP_TRY(f->run_step(sb, f->in_stage.as<uint8_t>(), nullptr, n, f->stream));
        }
        HP_HIP_TRY(hipEventRecord(e->ev1, e->stream));
        HP_HIP_TRY(hipEventRecord(f->ev1, f->stream));
        HP_HIP_TRY(hipEventSynchronize(e->ev1));
        HP_HIP_TRY(hipEventSynchronize(f->ev1));
        float m0 = 0, m1 = 0;
        HP_HIP_TRY(hipEventElapsedTime(&m0, e->ev0, e->ev1));
        HP_HIP_TRY(hipEventElapsedTime(&m1, f->ev0, f->ev1));
        if (out && k < cap) {
            auto& st = sa;
            out[k].layer = st.layer, out[k].op = st.op;
            out[k].tile = st.op == OP_SEPCONV ? 4000000 + (st.sep_pair ? 20 : hp::sepconv_variant(st.sp))
                : st.op == OP_MLPHEAD        ? 6000000 + st.hp_.K1
                : st.op == OP_CHAIN          ? 7000000 + hp::conv_chain_variant(st.ch)
                : st.op == OP_BNECK          ? 9000000 + hp::bottleneck_variant(st.bn)
                : (st.op == HP_OP_CONV && !st.first) ? (st.f32 ? (st.head32 ? hp::conv32_head_tile(st.hh.HID, st.cp32.Cout) : st.wino ? hp::conv32_winograd_tile(st.cp32) : st.cin_split ? hp::conv32_direct_tile(st.cp32, e->dtype == HP_DTYPE_F32S && !e->split_off) : hp::conv32_tile(st.cp32)) : hp::conv_mfma_tile(st.cp))
                                                    : 0;
            out[k].ms = std::max(m0, m1) / (2 * iters);
            out[k].flops = st.flops * n, out[k].bytes = st.bytes * n;
        }
        ++k;
    }
    *n_out = k;
    return HP_OK;
}

int hp_engine_profile_sequence(hp_engine* e, int n, int iters, hp_layer_time* out, int cap, int* n_out)
{
    HP_REQUIRE(e && n >= 1 && n <= e->max_batch && iters >= 1 && n_out, HP_ERR_INVALID, "hp_engine_profile_sequence: bad argument");
    const size_t need = (size_t)e->max_batch * e->in_h * e->in_w * 3 * sizeof(float);
    if (e->in_stage.bytes < need)
        HP_TRY(e->in_stage.alloc(need));
    const uint8_t* u8 = e->in_stage.as<uint8_t>();
    const size_t ns = e->steps.size();
    // one (start, stop) event pair per step; the launch itself records them (hipExtLaunchKernelGGL), so they bracket the
    // kernel's own execution and no packet is inserted between consecutive kernels
    std::vector<hipEvent_t> ev0(ns, nullptr), ev1(ns, nullptr);
    std::vector<double> acc(ns, 0.0);
    int rc = HP_OK;
    for (size_t k = 0; k < ns; ++k)
        if (hipEventCreate(&ev0[k]) != hipSuccess || hipEventCreate(&ev1[k]) != hipSuccess)
            rc = HP_ERR_HIP;
    for (int it = -1; it < iters && rc == HP_OK; ++it) { // it == -1: warm-up pass
        for (size_t k = 0; k < ns && rc == HP_OK; ++k) {
            hp::prof_start = ev0[k], hp::prof_stop = ev1[k];
            rc = e->run_step(e->steps[k], u8, nullptr, n, e->stream);
            hp::prof_start = hp::prof_stop = nullptr;
        }
        if (rc == HP_OK && hipStreamSynchronize(e->stream) != hipSuccess)
            rc = HP_ERR_HIP;
        for (size_t k = 0; k < ns && rc == HP_OK && it >= 0; ++k) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev0[k], ev1[k]) != hipSuccess)
                rc = HP_ERR_HIP;
            acc[k] += ms;
        }
    }
    hp::prof_start = hp::prof_stop = nullptr;
    for (size_t k = 0; k < ns; ++k) {
        if (ev0[k])
            (void)hipEventDestroy(ev0[k]);
        if (ev1[k])
            (void)hipEventDestroy(ev1[k]);
    }
    HP_REQUIRE(rc == HP_OK, rc, "hp_engine_profile_sequence: launch or event failure");
    int k = 0;
    for (auto& st : e->steps) {
        if (out && k < cap) {
            out[k].layer = st.layer, out[k].op = st.op;
            out[k].tile = st.op == OP_SEPCONV ? 4000000 + (st.sep_pair ? 20 : hp::sepconv_variant(st.sp))
                : st.op == OP_MLPHEAD        ? 6000000 + st.hp_.K1
                : st.op == OP_CHAIN          ? 7000000 + hp::conv_chain_variant(st.ch)
                : st.op == OP_BNECK          ? 9000000 + hp::bottleneck_variant(st.bn)
                : (st.op == HP_OP_CONV && !st.first) ? (st.f32 ? (st.head32 ? hp::conv32_head_tile(st.hh.HID, st.cp32.Cout) : st.wino ? hp::conv32_winograd_tile(st.cp32) : st.cin_split ? hp::conv32_direct_tile(st.cp32, e->dtype == HP_DTYPE_F32S && !e->split_off) : hp::conv32_tile(st.cp32)) : hp::conv_mfma_tile(st.cp))
                                                    : 0;
            out[k].ms = (float)(acc[k] / iters);
            out[k].flops = st.flops * n, out[k].bytes = st.bytes * n;
        }
        ++k;
    }
    *n_out = k;
    return HP_OK;
}

} // extern "C"
