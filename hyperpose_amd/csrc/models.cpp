// models.cpp — built-in network topologies as hp_layer lists (include/hp_hip.h) + deterministic synthetic
// weights.  Each builder restates one exported model of the reference's training library; BatchNorm layers
// are inference-folded into the preceding convolution (so every conv_block = conv + bias + activation):
//   MobilenetDilated backbone   hyperpose/Model/backbones.py:177-229 (dw_conv_block :190-197)
//   vggtiny backbone            hyperpose/Model/backbones.py:343-391
//   vgg19 backbone              hyperpose/Model/backbones.py:447-509 (subtracts vgg_mean/255, :455,501)
//   LightWeightOpenPose head    hyperpose/Model/openpose/model/lw_openpose.py:12-191
//   OpenPose (CMU) head         hyperpose/Model/openpose/model/openpose.py:13-198
// The released weights are Google-Drive downloads (scripts/downloader.py:12-21) and there is no network, so
// weights are synthetic: He-normal kernels from a counter-based generator (same numbers from C++ and Python).
#include "model.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>


namespace {

int backbone_mobilenet_dilated(hp_model& m, int& out_c)
{
    int t = m.conv(0, 3, 32, 3, HP_ACT_RELU, 2); // conv_block(32, strides 2)
    t = m.dw_block(t, 32, 64);
    t = m.dw_block(t, 64, 128, 2);
    t = m.dw_block(t, 128, 128);
    t = m.dw_block(t, 128, 256, 2);
    t = m.dw_block(t, 256, 256);
    t = m.dw_block(t, 256, 512);
    t = m.dw_block(t, 512, 512, 1, 2); // dilation 2, scale_size == 8 -> strides (1,1)
    t = m.dw_block(t, 512, 512);
    t = m.dw_block(t, 512, 512);
    t = m.dw_block(t, 512, 512);
    t = m.dw_block(t, 512, 512);
    out_c = 512;
    return t;
}

int backbone_vggtiny(hp_model& m, int& out_c)
{
    int t = m.conv(0, 3, 32, 3, HP_ACT_RELU);
    t = m.conv(t, 32, 64, 3, HP_ACT_RELU);
    t = m.pool(t, 64, 2, 2);
    t = m.conv(t, 64, 128, 3, HP_ACT_RELU);
    t = m.conv(t, 128, 128, 3, HP_ACT_RELU);
    t = m.pool(t, 128, 2, 2);
    t = m.conv(t, 128, 200, 3, HP_ACT_RELU);
    t = m.conv(t, 200, 200, 3, HP_ACT_RELU);
    t = m.conv(t, 200, 200, 3, HP_ACT_RELU);
    t = m.pool(t, 200, 2, 2);
    t = m.conv(t, 200, 384, 3, HP_ACT_RELU);
    t = m.conv(t, 384, 384, 3, HP_ACT_RELU);
    out_c = 384;
    return t;
}

int backbone_vgg19(hp_model& m, int& out_c)
{
    // forward: x = x - vgg_mean (/255), backbones.py:455,501
    m.mean[0] = 103.939f / 255, m.mean[1] = 116.779f / 255, m.mean[2] = 123.68f / 255;
    int t = m.conv(0, 3, 64, 3, HP_ACT_RELU);
    t = m.conv(t, 64, 64, 3, HP_ACT_RELU);
    t = m.pool(t, 64, 2, 2);
    t = m.conv(t, 64, 128, 3, HP_ACT_RELU);
    t = m.conv(t, 128, 128, 3, HP_ACT_RELU);
    t = m.pool(t, 128, 2, 2);
    t = m.conv(t, 128, 256, 3, HP_ACT_RELU);
    for (int i = 0; i < 3; ++i)
        t = m.conv(t, 256, 256, 3, HP_ACT_RELU);
    t = m.pool(t, 256, 2, 2);
    t = m.conv(t, 256, 512, 3, HP_ACT_RELU);
    t = m.conv(t, 512, 512, 3, HP_ACT_RELU);
    out_c = 512;
    return t;
}

// LightWeightOpenPose head (lw_openpose.py:38-75): cpm -> init stage -> concat(185) -> one refinement stage.
void head_lw_openpose(hp_model& m, int feat, int feat_c)
{
    const int NC = 128, NCONF = 19, NPAF = 38;
    // Cpm_stage (:106-121): 1x1 relu; x + 3 x conv_block(3x3 +BN+relu); 3x3 relu
    const int x = m.conv(feat, feat_c, NC, 1, HP_ACT_RELU);
    int y = m.conv(x, NC, NC, 3, HP_ACT_RELU);
    y = m.conv(y, NC, NC, 3, HP_ACT_RELU);
    y = m.add(HP_OP_CONV, y, 0, NC, NC, 3, 1, 1, HP_ACT_RELU, true, -1, 0, x, 0);
    const int cat = m.new_tensor(); // [cpm 128 | init conf 19 | init paf 38]  (tf.concat, :58)
    m.add(HP_OP_CONV, y, 0, NC, NC, 3, 1, 1, HP_ACT_RELU, true, cat, 0);
    // Init_stage (:123-148)
    int a = m.conv(cat, NC, NC, 3, HP_ACT_RELU, 1, 1, 0);
    a = m.conv(a, NC, NC, 3, HP_ACT_RELU);
    a = m.conv(a, NC, NC, 3, HP_ACT_RELU);
    const int hc = m.conv(a, NC, 512, 1, HP_ACT_RELU);
    m.add(HP_OP_CONV, hc, 0, 512, NCONF, 1, 1, 1, HP_ACT_NONE, true, cat, NC, -1, 0, 0.02f);
    const int hp_ = m.conv(a, NC, 512, 1, HP_ACT_RELU);
    m.add(HP_OP_CONV, hp_, 0, 512, NPAF, 1, 1, 1, HP_ACT_NONE, true, cat, NC + NCONF, -1, 0, 0.02f);
    // Refinement_stage (:150-191): 5 blocks {1x1 relu; 2 x conv_block; residual}
    int r = cat, rc = NC + NCONF + NPAF;
    for (int b = 0; b < 5; ++b) {
        const int u = m.conv(r, rc, NC, 1, HP_ACT_RELU);
        const int v = m.conv(u, NC, NC, 3, HP_ACT_RELU);
        r = m.add(HP_OP_CONV, v, 0, NC, NC, 3, 1, 1, HP_ACT_RELU, true, -1, 0, u, 0);
        rc = NC;
    }
    const int rcf = m.conv(r, NC, 512, 1, HP_ACT_RELU);
    const int conf = m.add(HP_OP_CONV, rcf, 0, 512, NCONF, 1, 1, 1, HP_ACT_NONE, true, -1, 0, -1, 0, 0.02f);
    const int rpf = m.conv(r, NC, 512, 1, HP_ACT_RELU);
    const int paf = m.add(HP_OP_CONV, rpf, 0, 512, NPAF, 1, 1, 1, HP_ACT_NONE, true, -1, 0, -1, 0, 0.02f);
    m.output("conf", conf, 0, NCONF); // infer() returns (conf_map, paf_map), :71-75
    m.output("paf", paf, 0, NPAF);
}

// OpenPose (CMU) head (openpose.py): CPM 3x3 512->256, 256->128 (ReLU); init stage per branch 3 x (3x3 128 + PReLU),
// 1x1 128->512 PReLU, 1x1 512->{19,38} PReLU; 5 refinement stages on concat(128+19+38): 7x7 185->128, 4 x 7x7 128->128,
// 1x1 128->128, 1x1 128->{19,38}, every conv followed by PReLU.
void head_openpose(hp_model& m, int feat, int feat_c)
{
    const int NC = 128, NCONF = 19, NPAF = 38, NCAT = NC + NCONF + NPAF;
    const int t = m.conv(feat, feat_c, 256, 3, HP_ACT_RELU);
    // Stage k reads concat([cpm, conf_{k-1}, paf_{k-1}]) (openpose.py:62) and both of its branches read that SAME
    // concat, so the two branches' outputs must not overwrite their own input: two concat buffers are used
    // alternately; the (cheap) last CPM conv is issued twice with shared weights to put the cpm slice in both.
    int cat[2] = { m.new_tensor(), m.new_tensor() };
    m.add(HP_OP_CONV, t, 0, 256, NC, 3, 1, 1, HP_ACT_RELU, true, cat[0], 0);
    {
        hp_layer dup = m.layers.back(); // same w_off / b_off
        dup.out = cat[1];
        m.layers.push_back(dup);
        m.init_scale.push_back(1.f);
        m.init_bias.push_back(0.f);
    }
    auto branch_init = [&](int n_out, int dst, int dst_off) {
        int a = m.conv(cat[0], NC, NC, 3, HP_ACT_PRELU, 1, 1, 0);
        a = m.conv(a, NC, NC, 3, HP_ACT_PRELU);
        a = m.conv(a, NC, NC, 3, HP_ACT_PRELU);
        a = m.conv(a, NC, 512, 1, HP_ACT_PRELU);
        m.add(HP_OP_CONV, a, 0, 512, n_out, 1, 1, 1, HP_ACT_PRELU, true, dst, dst_off, -1, 0, 0.05f);
    };
    // the init stage reads only the cpm slice of cat[0] and writes conf/paf into cat[1]
    branch_init(NCONF, cat[1], NC);
    branch_init(NPAF, cat[1], NC + NCONF);
    int conf_t = -1, paf_t = -1;
    for (int s = 0; s < 5; ++s) {
        const bool last = (s == 4);
        const int src = cat[(s + 1) & 1], dst = cat[s & 1];
        auto branch = [&](int n_out, int out_t, int out_off) {
            int a = m.conv(src, NCAT, NC, 7, HP_ACT_PRELU, 1, 1, 0);
            for (int i = 0; i < 4; ++i)
                a = m.conv(a, NC, NC, 7, HP_ACT_PRELU);
            a = m.conv(a, NC, NC, 1, HP_ACT_PRELU);
            return m.add(HP_OP_CONV, a, 0, NC, n_out, 1, 1, 1, HP_ACT_PRELU, true, out_t, out_off, -1, 0, 0.05f);
        };
        if (last) {
            conf_t = branch(NCONF, -1, 0);
            paf_t = branch(NPAF, -1, 0);
        } else {
            branch(NCONF, dst, NC);
            branch(NPAF, dst, NC + NCONF);
        }
    }
    m.output("conf", conf_t, 0, NCONF);
    m.output("paf", paf_t, 0, NPAF);
}

// Resnet50_backbone (backbones.py:587-697): conv 7x7 s2 (+BN+ReLU), optional max-pool 3x3 s2, bottlenecks [3,4,6,3]
// = {1x1, 3x3 (stride), 1x1 x4} + BN, projection shortcut when the shape changes, relu(x + res) (:697);
// scale_size == 32 -> stride 2 in stages 3 and 4 (:598-601).
int backbone_resnet50(hp_model& m, int& out_c, bool use_pool, bool stride32)
{
    int t = m.conv(0, 3, 64, 7, HP_ACT_RELU, 2);
    if (use_pool)
        t = m.pool(t, 64, 3, 2);
    int cin = 64;
    auto block = [&](int nf, int stride) {
        int res = t;
        if (stride != 1 || cin != 4 * nf)
            res = m.add(HP_OP_CONV, t, 0, cin, 4 * nf, 1, stride, 1, HP_ACT_NONE, true); // downsample conv + BN
        int x = m.conv(t, cin, nf, 1, HP_ACT_RELU);
        x = m.conv(x, nf, nf, 3, HP_ACT_RELU, stride);
        t = m.add(HP_OP_CONV, x, 0, nf, 4 * nf, 1, 1, 1, HP_ACT_RELU, true, -1, 0, res, 1, 0.5f); // relu(bn3(conv3) + res)
        cin = 4 * nf;
    };
    const int s34 = stride32 ? 2 : 1;
    for (int i = 0; i < 3; ++i)
        block(64, 1);
    for (int i = 0; i < 4; ++i)
        block(128, i == 0 ? 2 : 1);
    for (int i = 0; i < 6; ++i)
        block(256, i == 0 ? s34 : 1);
    for (int i = 0; i < 3; ++i)
        block(512, i == 0 ? s34 : 1);
    out_c = 2048;
    return t;
}

// PoseProposal head (pose_proposal/model.py:13-119): 3x3 C->512 +BN+LeakyReLU(0.1), 3x3 512->512 (same), 1x1 512 ->
// 6K + 9*9*L, sigmoid, split into pc, pi, px, py, pw, ph [K,h,w] and pe [L,9,9,h,w], restore_coor on x/y/w/h (:111-119).
// Outputs are named so that their name order is the parser's argument order (src/pose_proposal.cpp:12-20).
void head_pose_proposal(hp_model& m, int feat, int feat_c, int in_w, int in_h)
{
    const int K = 18, L = 17, NB = 9;
    int t = m.add(HP_OP_CONV, feat, 0, feat_c, 512, 3, 1, 1, HP_ACT_LEAKY, true, -1, 0, -1, 0, 1.f, 0.1f);
    t = m.add(HP_OP_CONV, t, 0, 512, 512, 3, 1, 1, HP_ACT_LEAKY, true, -1, 0, -1, 0, 1.f, 0.1f);
    const int o = m.add(HP_OP_CONV, t, 0, 512, 6 * K + NB * NB * L, 1, 1, 1, HP_ACT_NONE, true, -1, 0, -1, 0, 0.08f);
    m.init_bias.back() = -4.f; // synthetic weights: sigmoid(-4 + N(0,~1)) keeps a few % of the cells above the thresholds
    const int gh = (in_h + 31) / 32, gw = (in_w + 31) / 32;
    auto out = [&](const char* name, int coff, int ch, float scale, int grid) {
        m.output(name, o, coff, ch, HP_ACT_SIGMOID);
        m.outputs.back().scale = scale, m.outputs.back().grid = grid;
    };
    out("0_conf_point", 0, K, 1.f, 0);
    out("1_conf_iou", K, K, 1.f, 0);
    out("2_x", 2 * K, K, (float)in_w / gw, 1); // rx = (x + grid_x) * grid_size_x
    out("3_y", 3 * K, K, (float)in_h / gh, 2);
    out("4_w", 4 * K, K, (float)in_w, 0);      // rw = w * win
    out("5_h", 5 * K, K, (float)in_h, 0);
    out("6_edge", 6 * K, NB * NB * L, 1.f, 0);  // [L*9*9, h, w] == [L, 9, 9, h, w] in memory
}

// Pifpaf heads (pifpaf/model.py:215-281): 1x1 C -> 17*5*4 and 1x1 C -> 19*9*4, pixel_shuffle(2), sigmoid on the
// confidences, softplus on the scales; the last row/column is cropped so that an (8k+1)-pixel input yields the
// (k+1)-cell fields the decoder's H_hr = (H-1)*8+1 convention assumes.  Outputs in the argument
// order of parser::pifpaf::process(paf, pif) (src/pifpaf.cpp:7).
void head_pifpaf(hp_model& m, int feat, int feat_c, int in_w, int in_h)
{
    // synthetic weights: small logits around -3 keep the random-weight fields sparse (the backbone output has std ~8)
    const int pif = m.add(HP_OP_CONV, feat, 0, feat_c, 17 * 5 * 4, 1, 1, 1, HP_ACT_NONE, true, -1, 0, -1, 0, 0.05f);
    m.init_bias.back() = -3.f;
    const int paf = m.add(HP_OP_CONV, feat, 0, feat_c, 19 * 9 * 4, 1, 1, 1, HP_ACT_NONE, true, -1, 0, -1, 0, 0.05f);
    m.init_bias.back() = -3.f;
    const int fh = (in_h - 1) / 8 + 1, fw = (in_w - 1) / 8 + 1;
    m.output("0_paf", paf, 0, 19 * 9 * 4);
    hp_output_desc& a = m.outputs.back();
    a.shuffle = 2, a.group = 9, a.sigmoid_mask = 1u, a.softplus_mask = (1u << 7) | (1u << 8), a.out_h = fh, a.out_w = fw;
    m.output("1_pif", pif, 0, 17 * 5 * 4);
    hp_output_desc& b = m.outputs.back();
    b.shuffle = 2, b.group = 5, b.sigmoid_mask = 1u, b.softplus_mask = 1u << 4, b.out_h = fh, b.out_w = fw;
}

// counter-based generator: splitmix64 hash -> two uniforms -> Box-Muller
inline uint64_t mix(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
inline float normal_at(uint64_t seed, uint64_t layer, uint64_t idx)
{
    const uint64_t h = mix(mix(seed ^ (layer * 0x100000001b3ull)) + idx);
    const uint64_t h2 = mix(h);
    const double u1 = ((h >> 11) + 1.0) / 9007199254740993.0; // (0,1]
    const double u2 = (h2 >> 11) / 9007199254740992.0;
    return (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
}

} // namespace

extern "C" {

const char* hp_model_archs(void) { return "lw_openpose_mobilenet,lw_openpose_vggtiny,openpose_vgg19,pose_proposal_resnet50,pifpaf_resnet50"; }

int hp_model_build(hp_model** out, const char* arch, int in_w, int in_h)
{
    HP_REQUIRE(out && arch, HP_ERR_INVALID, "hp_model_build: null argument");
    HP_REQUIRE(in_w >= 32 && in_h >= 32, HP_ERR_INVALID, "hp_model_build: input %dx%d too small", in_w, in_h);
    std::unique_ptr<hp_model> m(new hp_model());
    m->arch = arch, m->in_w = in_w, m->in_h = in_h;
    int c = 0;
    const std::string a = arch;
    if (a == "lw_openpose_mobilenet") {
        const int f = backbone_mobilenet_dilated(*m, c);
        head_lw_openpose(*m, f, c);
    } else if (a == "lw_openpose_vggtiny") {
        const int f = backbone_vggtiny(*m, c);
        head_lw_openpose(*m, f, c);
    } else if (a == "openpose_vgg19") {
        const int f = backbone_vgg19(*m, c);
        head_openpose(*m, f, c);
    } else if (a == "pose_proposal_resnet50") {
        const int f = backbone_resnet50(*m, c, true, true);
        head_pose_proposal(*m, f, c, in_w, in_h);
    } else if (a == "pifpaf_resnet50") {
        // x = (x - mean) / std with the ImageNet statistics (pifpaf/model.py:38-39,56)
        const float mean[3] = { 0.485f, 0.456f, 0.406f }, stdv[3] = { 0.229f, 0.224f, 0.225f };
        for (int k = 0; k < 3; ++k)
            m->mean[k] = mean[k], m->inv_std[k] = 1.f / stdv[k];
        const int f = backbone_resnet50(*m, c, false, true);
        head_pifpaf(*m, f, c, in_w, in_h);
    } else {
        hp::set_error("hp_model_build: unknown arch '%s' (have: %s)", arch, hp_model_archs());
        return HP_ERR_INVALID;
    }
    *out = m.release();
    return HP_OK;
}

void hp_model_destroy(hp_model* m) { delete m; }

int hp_model_layers(const hp_model* m, const hp_layer** layers, int* n)
{
    HP_REQUIRE(m && layers && n, HP_ERR_INVALID, "hp_model_layers: null argument");
    *layers = m->layers.data(), *n = (int)m->layers.size();
    return HP_OK;
}

int hp_model_outputs(const hp_model* m, const hp_output_desc** outs, int* n)
{
    HP_REQUIRE(m && outs && n, HP_ERR_INVALID, "hp_model_outputs: null argument");
    *outs = m->outputs.data(), *n = (int)m->outputs.size();
    return HP_OK;
}

size_t hp_model_num_weights(const hp_model* m) { return m ? (size_t)m->n_weights : 0; }

int hp_model_preproc(const hp_model* m, float mean[3], float inv_std[3])
{
    HP_REQUIRE(m && mean && inv_std, HP_ERR_INVALID, "hp_model_preproc: null argument");
    for (int c = 0; c < 3; ++c)
        mean[c] = m->mean[c], inv_std[c] = m->inv_std[c];
    return HP_OK;
}

double hp_model_flops_per_frame(const hp_model* m)
{
    if (!m)
        return 0;
    // replay the shape propagation of engine.cpp
    std::vector<int> H(m->next_tensor, 0), W(m->next_tensor, 0);
    H[0] = m->in_h, W[0] = m->in_w;
    double flops = 0;
    for (const hp_layer& L : m->layers) {
        int oh, ow;
        hp_layer_out_size(L, H[L.in], W[L.in], oh, ow);
        H[L.out] = oh, W[L.out] = ow;
        if (L.op == HP_OP_CONV)
            flops += 2.0 * oh * ow * L.cout * L.kh * L.kw * L.cin;
        else if (L.op == HP_OP_DWCONV)
            flops += 2.0 * oh * ow * L.cin * L.kh * L.kw;
    }
    return flops;
}

int hp_model_init_weights(const hp_model* m, uint64_t seed, float* blob, size_t n)
{
    HP_REQUIRE(m && blob, HP_ERR_INVALID, "hp_model_init_weights: null argument");
    HP_REQUIRE(n >= (size_t)m->n_weights, HP_ERR_INVALID, "hp_model_init_weights: blob too small (%zu < %lld)", n, (long long)m->n_weights);
    for (size_t li = 0; li < m->layers.size(); ++li) {
        const hp_layer& L = m->layers[li];
        if (L.op == HP_OP_MAXPOOL || L.op == HP_OP_UPSAMPLE)
            continue;
        const int fan_in = L.op == HP_OP_CONV ? L.kh * L.kw * L.cin : L.kh * L.kw;
        const size_t nw = L.op == HP_OP_CONV ? (size_t)L.cout * L.kh * L.kw * L.cin : (size_t)L.cin * L.kh * L.kw;
        const float stdv = m->init_scale[li] * std::sqrt(2.0f / fan_in);
        for (size_t i = 0; i < nw; ++i)
            blob[L.w_off + i] = stdv * normal_at(seed, li, i);
        if (L.b_off >= 0)
            for (int i = 0; i < L.cout; ++i)
                blob[L.b_off + i] = m->init_bias[li] + 0.02f * m->init_scale[li] * normal_at(seed, li, nw + i);
        if (L.alpha_off >= 0)
            for (int i = 0; i < L.cout; ++i)
                blob[L.alpha_off + i] = 0.25f;
    }
    return HP_OK;
}

int hp_engine_create_from_model(hp_engine** out, const hp_model* m, int max_batch, double factor, int flip_rb,
    const float* weights, size_t n_weights)
{
    return hp_engine_create_from_model_dtype(out, m, max_batch, factor, flip_rb, weights, n_weights, HP_DTYPE_F16);
}

int hp_engine_create_from_model_dtype(hp_engine** out, const hp_model* m, int max_batch, double factor, int flip_rb,
    const float* weights, size_t n_weights, int dtype)
{
    HP_REQUIRE(out && m, HP_ERR_INVALID, "hp_engine_create_from_model: null argument");
    if (!weights) // imported models carry their own
        weights = m->weights.data(), n_weights = m->weights.size();
    HP_REQUIRE(weights && n_weights, HP_ERR_INVALID, "hp_engine_create_from_model: a built-in topology needs a weight blob");
    hp_engine_desc d;
    memset(&d, 0, sizeof(d));
    d.in_w = m->in_w, d.in_h = m->in_h, d.max_batch = max_batch, d.factor = factor, d.flip_rb = flip_rb;
    for (int c = 0; c < 3; ++c)
        d.mean[c] = m->mean[c], d.inv_std[c] = m->inv_std[c];
    d.layers = m->layers.data(), d.n_layers = (int)m->layers.size();
    d.outputs = m->outputs.data(), d.n_outputs = (int)m->outputs.size();
    d.weights = weights, d.n_weights = n_weights;
    d.dtype = dtype;
    return hp_engine_create(out, &d);
}

} // extern "C"
