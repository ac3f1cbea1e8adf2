// conv_kernels.hpp — launch interface of the gfx950 convolution kernels (conv_kernels.hip) used by the
// engine (engine.cpp).  These replace the TensorRT-built network of the reference
// (src/tensorrt.cpp:393 `executeV2`): every layer of the exported graphs (hyperpose/Model/backbones.py,
// openpose/model/*.py, pose_proposal/model.py, pifpaf/model.py) maps onto one of the launches below.
//
// Activation layout in HBM: NHWC fp16 with a ZERO HALO: a tensor [B][H][W][C] lives in a buffer
// [B][H+2P][W+2P][cs] (cs = channel stride, multiple of 32; P = the largest padding any consumer needs) whose
// border and pad channels are zeroed once and never written, so convolution taps that fall into the padding are
// plain in-bounds loads of zeros — no per-tap bounds checks, no clamping, uniform per-tap address offsets.
// A tensor may be a channel slice [coff, coff+C) of a wider buffer (concat by offset).
// Network outputs are written fp32 NCHW [B][C][H][W], the layout feature_map_t / the parsers consume.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hp {

enum act_t : int { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LEAKY = 3, ACT_PRELU = 4, ACT_SIGMOID = 5, ACT_SOFTPLUS = 6 };

// fp16 NHWC view with halo: element (b, y, x, c) is p[((long)b * img + (long)y * wp + x) * cs + coff + c],
// valid for y in [-P, H+P), x in [-P, W+P).
struct tview {
    __half* p;   // interior origin: image 0, pixel (0,0), channel 0 of the underlying buffer
    int cs;      // channel stride (elements per pixel)
    int coff;    // first channel of this tensor inside the buffer
    int wp;      // padded row length in pixels (W + 2P)
    int img;     // padded pixels per image ((H + 2P) * (W + 2P))
};

struct conv_params {
    tview in;
    int B, H, W;
    int OH, OW;
    int Cin;      // channels read per tap; multiple of 32 for the MFMA kernels
    int Cout;     // real output channels
    int Cout_pad; // rows of the packed weight matrix (multiple of the M tile)
    int KH, KW, stride, dil, pad_t, pad_l;
    const __half* w;    // packed [KH*KW][Cout_pad][Cin] (w_layout 0) or MFMA-fragment order (w_layout 1, see conv_weight_layout)
    int w_layout;
    const float* bias;  // [Cout_pad]
    const float* alpha; // PReLU slopes [Cout_pad] or nullptr
    int act;            // ACT_NONE / RELU / RELU6 / LEAKY / PRELU (the piecewise-linear ones)
    float act_param;
    float act_slope, act_hi; // y = v > 0 ? min(v, act_hi) : v * act_slope, filled by set_act()
    // optional residual (same shape as the output), added after (res_before_act = 0) or before the activation
    tview res; // res.p == nullptr: none
    int res_before_act;
    // outputs (either may be null)
    tview out;      // out.p may be null
    float* out_f32; // fp32 NCHW [B][Cout][OH][OW]
    int npix;       // B*OH*OW
    unsigned long long* dbg; // optional s_memtime timeline of block 0 / wave 0 (tools/microbench), nullptr in production
    // split-K (conv_direct_kernel on maps with fewer tiles than CUs): ksplit >= 2 blocks share a tile's channel chunks, park their fp32
    // partial sums in `splitk` (conv_splitk() bytes) and a second launch adds them and runs the epilogue.  0 / nullptr: off.
    int ksplit;
    float* splitk;
};

// how many ways the launcher would split the K of this convolution (1 = not at all) and the scratch bytes that needs; the engine
// allocates the scratch and sets ksplit / splitk (a convolution without scratch runs unsplit)
int conv_splitk(const conv_params& p, size_t* scratch_bytes);

// fills act_slope / act_hi from act / act_param; false for activations the MFMA epilogue does not fuse
bool set_act(conv_params& p);
// Dense k x k convolution as an implicit GEMM on MFMA (v_mfma_f32_32x32x16_f16).  Picks between the generic
// implicit-GEMM kernel and the LDS-halo 3x3 kernel.  Returns hipError_t.
hipError_t launch_conv_mfma(const conv_params& p, hipStream_t s);
// which kernel/tile the launcher picks (profile rows): BM*1000+BN for the generic implicit GEMM, 5064192 conv3x3_direct_kernel,
// 51xxxxx conv1x1_small_kernel, 52xxxxx conv1x1_big_kernel<TM, NTP>, 6xxxxxx conv_direct_kernel
int conv_mfma_tile(const conv_params& p);
// Weight layout the launcher wants for this convolution (fill every other field of p first):
//   0: [tap][Cout_pad][Cin] rows;   1: MFMA-fragment order for the barrier-free 3x3 kernel, half index
//   ((((tap * (Cout_pad / 32) + m / 32) * (Cin / 16) + k / 16) * 64) + (k % 16 / 8) * 32 + m % 32) * 8 + k % 8
int conv_weight_layout(const conv_params& p);

// set by hp_engine_profile_sequence around one step: the next launch records its own begin / end into these events
extern thread_local hipEvent_t prof_start, prof_stop;

struct first_conv_params {
    const uint8_t* in_u8; // [B][H][W][3] or nullptr
    const float* in_f32;  // [B][3][H][W] (already scaled / ordered) or nullptr
    double factor;
    int flip_rb;
    float mean[3]; // subtracted after scaling (VGG19 / PifPaf style pre-processing folded into the load)
    float inv_std[3];
    int B, H, W, OH, OW;
    int Cout, KH, KW, stride, pad_t, pad_l;
    const float* w; // fp32 [Cout][KH][KW][3]
    const __half* w16; // the same weights for first_conv_f16_kernel: fp16, [Cout / 32][step][64 lanes][8], k' = ky * ROWP + kx * 3 + c (or nullptr)
    const float* bias;
    int act;
    float act_param;
    tview out;
};
// Direct convolution for the 3-channel network input (u8 HWC or f32 NCHW), pre-processing fused into the load.
hipError_t launch_first_conv(const first_conv_params& p, hipStream_t s);

struct dw_params {
    tview in;
    int B, H, W, OH, OW, C; // C multiple of 8
    int stride, dil, pad_t, pad_l;
    int halo; // zero-halo width of the input tensor in HBM (pixels)
    float act_slope, act_hi; // filled by launch_dwconv3x3
    const __half* w;   // packed [9][C]
    const float* bias; // [C]
    int act;
    float act_param;
    tview out;
};
// Depthwise 3x3 (VALU, HBM/L2-bound): one thread = one output pixel x 8 channels.
hipError_t launch_dwconv3x3(const dw_params& p, hipStream_t s);

// Depthwise 3x3 + pointwise 1x1 fused (sepconv_kernel).  `pw` describes the pointwise half exactly like a 1x1
// conv_params (Cin = C, bias, activation, out, OH/OW/npix) except that pw.w holds the weights in MFMA-fragment order:
// half index (((m / 32) * (C / 16) + k / 16) * 64 + (k % 16 / 8) * 32 + m % 32) * 8 + k % 8 for row m, input channel k.
constexpr int SEP_CMAX = 512;
struct sep_params {
    tview in;
    int B, H, W, OH, OW, C;
    int stride, dil, pad_t, pad_l;
    int halo;            // zero-halo width of the input tensor in HBM (pixels)
    const __half* dw_w;  // [9][C]
    const float* dw_bias; // [C]
    float dw_slope, dw_hi; // depthwise activation as y = v > 0 ? min(v, hi) : v * slope
    conv_params pw;
};
// 0 when no fused instantiation serves this pair (the caller keeps dwconv3x3 + conv_mfma)
int sepconv_variant(const sep_params& p);
int sepconv_variant_for(int C, int cout_pad, int stride, int dil, int cout = 0); // the pointer-free part of the same decision (cout: true output channels, 0 = unknown)
hipError_t launch_sepconv(const sep_params& p, hipStream_t s);

// Two consecutive separable blocks (32 -> 64 stride 1, 64 -> 128 stride 2: the MobileNet stem) in one launch, the tensor between them
// in LDS only (sepconv_pair_kernel).  `b.in` must be `a.pw.out`.
struct seppair_params {
    sep_params a, b;
};
int seppair_variant(const seppair_params& p); // 0 = none
hipError_t launch_seppair(const seppair_params& p, hipStream_t s);

// Two chained 1x1 convolutions K1 -> 512 (relu family) -> Cout2 <= 64 in one launch (mlp_head_kernel).
//   w1: MFMA-fragment order, half index (((m / 32) * (K1 / 16) + k / 16) * 64 + (k % 16 / 8) * 32 + m % 32) * 8 + k % 8
//   w2: rows padded to 64; the hidden channel c = 128 w + 32 i + r32 sits at K-step 8 w + 2 i + s, lane half h, element e
//       with h = (r32 >> 2) & 1, r = (r32 & 3) + 4 (r32 >> 3), s = r >> 3, e = r & 7:
//       half index (((m / 32) * 32 + 8 w + 2 i + s) * 64 + h * 32 + m % 32) * 8 + e
//   pw: bias (padded to 64), activation, out / out_f32, Cout, OH, OW of the SECOND convolution.
struct head_params {
    tview in;
    int B, H, W, K1;
    const __half* w1;
    const float* b1; // [512]
    float hi1;       // hidden activation = clamp(x, 0, hi1)
    const __half* w2;
    conv_params pw;
};
int mlp_head_variant(int k1, int hidden, int cout2);
hipError_t launch_mlp_head(const head_params& p, hipStream_t s);
// two heads on the same input, same K1 / geometry, in one launch (blockIdx.y = head)
hipError_t launch_mlp_head_pair(const head_params& p0, const head_params& p1, hipStream_t s);


// A chain of 128 -> 128 convolutions in one launch (conv_chain.hip): [c0: 1x1 ->] c1: 3x3 -> c2: 3x3, relu-family activations,
// intermediates in LDS.  c0 / c1 / c2 are filled exactly like stand-alone convolutions (weights in fragment order, w_layout 1);
// only c0.in (or c1.in without c0), c2.out and the residual views are touched in HBM.
//   res_mode 0: no residual; 1: c1.res added to c1's output; 2: c2.res added to c2's output; 3: c0's output added to c2's output
struct chain_params {
    conv_params c0, c1, c2;
    int has_c0;
    int res_mode;
};
// 0 when the chain kernel does not take this combination, otherwise its variant code (profile rows: tile = 7000000 + variant)
int conv_chain_variant(const chain_params& p);
hipError_t launch_conv_chain(const chain_params& p, hipStream_t s);

// A ResNet bottleneck's tail and the next block's head in one launch (conv_bottleneck.hip): [3x3 M -> M] -> 1x1 M -> 4M + shortcut
// [-> 1x1 4M -> M' of the next block].  c3 / ce / cr are filled exactly like stand-alone convolutions (w_layout 1); c3's output and
// (as an input) ce's output never leave the block - ce.out is still written, the next block's shortcut reads it.
struct bneck_params {
    conv_params c3, ce, cr;
    int has_c3, has_cr;
    conv_params cp; // has_cp: the block's projection shortcut (1x1 M -> 4M of the block input, no activation), computed in the block
    int has_cp;     // instead of being read: ce.res is then empty
    conv_params c0; // has_c0 (with has_cp and has_c3): the block's OWN reduction (1x1 M -> M of the same block input), computed on the
    int has_c0;     // 3x3's halo tile: the block reads nothing but its input
};
// 0 when the kernel does not take this combination, otherwise 1000 * (M / 64) + 100 * (has_cp + 2 * has_c0) + 10 * (M' / 64) + has_c3 (profile rows: tile = 9000000 + variant)
int bottleneck_variant(const bneck_params& p);
hipError_t launch_bottleneck(const bneck_params& p, hipStream_t s);

struct pool_params {
    tview in;
    int B, H, W, OH, OW, C;
    int k, stride, pad_t, pad_l;
    tview out;
};
hipError_t launch_maxpool(const pool_params& p, hipStream_t s);
// HP_OP_UPSAMPLE with the same parameter block: stride = integer scale, k = 0 nearest / 1 bilinear (half-pixel centres)
hipError_t launch_upsample(const pool_params& p, hipStream_t s);

// fp16 NHWC view -> fp32 NCHW network output (for outputs not produced by a conv epilogue): optional pixel shuffle x2,
// crop, element-wise / per-component sigmoid / softplus, and the PoseProposal restore_coor affine map.
struct out_xform {
    int C;        // channels read from the view
    int act;      // applied to every channel when group == 0
    int shuffle;  // 1 or 2
    int group;    // 0 or components per group
    unsigned sigmoid_mask, softplus_mask;
    int out_h, out_w; // output spatial size (after shuffle / crop)
    float scale;
    int grid;
};
hipError_t launch_output_transform(tview in, int B, int H, int W, const out_xform& x, float* out, hipStream_t s);

} // namespace hp
