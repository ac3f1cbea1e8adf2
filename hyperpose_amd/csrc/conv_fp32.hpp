// conv_fp32.hpp — launch interface of the fp32-faithful kernel family (conv_fp32.hip): what the engine runs when it is created with
// HP_DTYPE_F32, i.e. behind `data_type::kFLOAT` of the reference's engine (include/hyperpose/operator/dnn/tensorrt.hpp:14-21,48: fp32 is
// the reference's default precision; docs/markdown/quick_start/prediction.md:145).  Storage AND arithmetic are fp32: activations NHWC fp32
// with the same zero halo as the fp16 path (conv_kernels.hpp), weights fp32, products and sums on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32 multiply-add, no TF32-style truncation) or the fp32 vector pipe.  The generic
// kernels are in conv_fp32.hip; since round 5 the 3 x 3 stride-1 layers run in Winograd's F(2 x 2, 3 x 3) form (conv32_winograd.hip), the two-layer
// heads in one launch (conv32_head.hip), narrow 1 x 1 layers on the barrier-free direct kernel (conv32_direct.hip) - all with exact fp32
// products; HP_DTYPE_F32S is the same engine with the dense products formed on the fp16 pipe (conv32_direct.hip, SPLIT).
#pragma once
#include "conv_kernels.hpp"

namespace hp {

// fp32 NHWC view with halo: element (b, y, x, c) is p[((long)b * img + (long)y * wp + x) * cs + coff + c]
struct tview32 {
    float* p;
    int cs, coff, wp, img;
};

struct conv32_params {
    tview32 in;
    int B, H, W, OH, OW;
    int Cin;      // channels read per tap, multiple of 16 (the slice [coff, coff + Cin) must lie inside the buffer's channel stride)
    int Cout;     // real output channels
    int Cout_pad; // rows of the packed weight matrix, multiple of 64
    int KH, KW, stride, dil, pad_t, pad_l;
    const float* w;     // [KH*KW][Cout_pad][Cin] fp32, zero rows / columns in the padding
    const float* bias;  // [Cout_pad]
    const float* alpha; // PReLU slopes [Cout_pad] or nullptr
    int act; // ACT_NONE / RELU / RELU6 / LEAKY / PRELU
    float act_param;
    float act_slope, act_hi; // y = v > 0 ? min(v, act_hi) : v * act_slope (PReLU: alpha[] per channel), filled by set_act32()
    tview32 res; // res.p == nullptr: none
    int res_before_act;
    tview32 out;    // out.p may be null
    float* out_f32; // fp32 NCHW [B][Cout][OH][OW] network output, or nullptr
    int npix;       // B * OH * OW
    int pick_npix;  // max_batch * OH * OW (0 = npix): what the choice between conv32_kernel and conv32_t16_kernel looks at - the two sum a K-step's
                    // products in different groupings (32 x 32 x 2 / 16 x 16 x 4), so the choice must not depend on how many frames a call brings:
                    // a frame's outputs are the same bits in every batch (and in both halves of hp_engine_set_concurrency(2))
    // conv32_direct_kernel (conv32_direct.hip): the same weights in MFMA-fragment order - fp32 (w_frag, HP_DTYPE_F32) or split into fp16
    // (hi, lo) pairs (w_split, HP_DTYPE_F32S) - and the sticky flag the split kernel raises when an activation does not fit fp16's range
    // (then the engine falls back to the fp32 pipe)
    const float* w_frag;
    const _Float16* w_split;
    unsigned* ovf;
    // conv32_winograd_kernel (conv32_winograd.hip, HP_DTYPE_F32 only): U = G g Gt of a 3 x 3 stride-1 layer in MFMA-fragment order, or nullptr
    const float* w_wino;
    // conv32_winograd3_kernel (conv32_winograd3.hip): U = G g Gt of the F(3 x 3, 3 x 3) form (25 positions) in the same fragment order, or nullptr
    const float* w_wino3;
    // a depthwise 3 x 3 (stride 1, dilation dw_dil = 1 | 2, SAME padding) fused in front of this 1 x 1 convolution (conv32_direct_kernel's DWD
    // forms): `in` is then the DEPTHWISE layer's input; dw_w = [9][Cin] tap-major weights followed by [Cin] biases; y = v > 0 ? min(v, dw_hi) : v * dw_slope
    const float* dw_w;
    int dw_dil;
    float dw_slope, dw_hi;
    int latency;       // 1 = the caller has ONE batch in flight (hp_engine_set_concurrency(e, 2)): kernels that have a form for a launch that runs alone take it
                       // (conv32_winograd_kernel: 8 x 8-pixel blocks); the forms give the same bits
    int lane_epilogue; // HP_LANE_EPILOGUE (A/B switch): 0 = per kernel (launch_conv32: rows on the 64-pixel tile; direct kernels: rows), 1 = the accumulators' lane = pixel layout everywhere, -1 = whole pixel rows everywhere (conv32_epilogue.hpp)
    unsigned long long* dbg; // HP_DIRECT_DBG: s_memtime stamps (shader cycles) of block (0, 0)'s thread 0 - start, chunk staged, chunk multiplied, ..., stored
};
// fills act_slope / act_hi from act / act_param; false for activations the epilogue does not evaluate (sigmoid / softplus are output post-ops)
bool set_act32(conv32_params& p);
// Dense KH x KW convolution (any stride / dilation) as an implicit GEMM on v_mfma_f32_32x32x2_f32.
hipError_t launch_conv32(const conv32_params& p, hipStream_t s);
int conv32_tile(const conv32_params& p); // profile rows: 32000000 + 400000 (row-major epilogue) + BM * 1000 + BN; conv32_wk_kernel (1 x 1, 32 / 64 input channels, large maps): 39000000 + Cin * 1000 + pixels per block
// The barrier-free direct form for square 1 x 1 / 3 x 3, stride 1, dilation 1, SAME padding (conv32_direct.hip): a chunk's halo tile staged in LDS once
// for all taps, weights in fragment order straight from L2.  split = false: exact fp32 products on v_mfma_f32_32x32x2_f32 (needs w_frag);
// split = true: every fp32 product formed as three exact fp16 x fp16 products on the fp16 matrix pipe, x = hi + 2^-11 lo,
// hi-hi + 2^-11 (hi-lo + lo-hi), fp32 accumulation (needs w_split).  Cin must be whole chunks (64 channels at 1 x 1, 32 at 3 x 3).
bool conv32_direct_ok(const conv32_params& p);
hipError_t launch_conv32_direct(const conv32_params& p, bool split, hipStream_t s);
int conv32_direct_tile(const conv32_params& p, bool split); // profile rows: 33000000 (split) / 34000000 (fp32) + 100000 * fused depthwise dilation + KS * 1000 + wavefront groups per block
// whether the 1 x 1 layer p can take a depthwise 3 x 3 of dilation `dil` in front of it in the same launch (p.dw_w etc. not yet set)
bool conv32_dw_fusable(const conv32_params& p, bool split, int dil);
// packed = [taps][cout_pad][cin] fp32 (conv32_params::w's layout) -> the kernels' fragment order: 2 * taps * cout_pad * cin halves / taps * cout_pad * cin floats
void conv32_split_pack(const float* packed, int taps, int cout_pad, int cin, _Float16* out);
void conv32_frag_pack(const float* packed, int taps, int cout_pad, int cin, float* out);

// Winograd F(2 x 2, 3 x 3) on the fp32 matrix pipe for 3 x 3, stride 1, dilation 1, SAME-padded layers with an NHWC output (conv32_winograd.hip):
// 16 instead of 36 MFMA products per output tile and channel pair.  Needs w_wino (conv32_winograd_pack of the packed matrix).
bool conv32_winograd_ok(const conv32_params& p);
hipError_t launch_conv32_winograd(const conv32_params& p, hipStream_t s);
hipError_t conv32_winograd_occupancy(const conv32_params& p, int* blocks_per_cu);
// Winograd F(3 x 3, 3 x 3) for the same layers (conv32_winograd3.hip): 25 products per 3 x 3 output tile and channel pair - 2.78 per pixel against 4.
// Needs w_wino3 (conv32_winograd3_pack of the packed matrix: 25 * cout_pad * cin floats).
bool conv32_winograd3_ok(const conv32_params& p);
hipError_t launch_conv32_winograd3(const conv32_params& p, hipStream_t s);
int conv32_winograd3_tile(const conv32_params& p);
void conv32_winograd3_pack(const float* packed, int cout_pad, int cin, float* out); // what the runtime grants this launch's kernel
int conv32_winograd_tile(const conv32_params& p);     // profile rows: 35000000 + 3000 + wavefronts per block
double conv32_winograd_flops(const conv32_params& p); // the MFMA work of one launch: 2 * 16 * tiles * Cout * Cin
void conv32_winograd_pack(const float* packed, int cout_pad, int cin, float* out); // [9][cout_pad][cin] -> 16 * cout_pad * cin floats

// The two-layer heads of an HP_DTYPE_F32 engine in one launch (conv32_head.hip): 1 x 1 128 -> HID (ReLU family) -> 1 x 1 HID -> C2 <= 64, the
// hidden tensor in registers.  `q` = the SECOND layer's conv32_params (bias, activation, out, out_f32, Cout = C2, OH / OW / npix) with q.in = the
// FIRST layer's input view; `h` = what the first layer adds.
struct head32_hidden {
    const float* w1_frag; // conv32_frag_pack of the first layer's [1][HID][128] matrix
    const float* bias1;   // [HID]
    const float* w2_frag; // conv32_head_pack of the second layer's [32 tm2][HID] matrix (zero rows in the padding), tm2 = C2 <= 32 ? 1 : 2
    float slope1, hi1;    // hidden activation: y = v > 0 ? min(v, hi1) : v * slope1
    int HID;
};
bool conv32_head_ok(int k1, int hid, int c2);
hipError_t launch_conv32_head(const conv32_params& q, const head32_hidden& h, hipStream_t s);
// two heads that read the same tensor in one grid (LW-OpenPose's heat-map and PAF heads of a stage): same results, the blocks of both side by side
bool conv32_head_pair_ok(const conv32_params& a, const conv32_params& b);
hipError_t launch_conv32_head_pair(const conv32_params& qa, const head32_hidden& ha, const conv32_params& qb, const head32_hidden& hb, hipStream_t s);
int conv32_head_tile(int hid, int c2); // profile rows: 37000000 + 100 * HID + C2
void conv32_head_pack(const float* w2, int tm2, int hid, float* out);

struct first_conv32_params {
    const uint8_t* in_u8; // [B][H][W][3] or nullptr
    const float* in_f32;  // [B][3][H][W] or nullptr
    double factor;
    int flip_rb;
    float mean[3], inv_std[3];
    int B, H, W, OH, OW;
    int Cout, KH, KW, stride, pad_t, pad_l;
    const float* w; // fp32 [Cout][KH][KW][3]
    const float* bias;
    int act;
    float act_param;
    tview32 out;
};
// The 3-channel network input (u8 HWC or f32 NCHW), pre-processing of src/data.cpp:21-51 folded into the load; all-fp32.
hipError_t launch_first_conv32(const first_conv32_params& p, hipStream_t s);
void first_conv32_verify_counts(unsigned out[4], bool reset); // HP_FIRST_CONV_VERIFY=1: LDS words that differed from global memory (patch, weights), blocks checked

struct dw32_params {
    tview32 in;
    int B, H, W, OH, OW, C; // C multiple of 4
    int stride, dil, pad_t, pad_l;
    const float* w;    // [9][C]
    const float* bias; // [C]
    int act;
    float act_param;
    tview32 out;
};
hipError_t launch_dwconv32(const dw32_params& p, hipStream_t s);

struct pool32_params {
    tview32 in;
    int B, H, W, OH, OW, C; // C multiple of 4
    int k, stride, pad_t, pad_l;
    tview32 out;
};
hipError_t launch_maxpool32(const pool32_params& p, hipStream_t s);
hipError_t launch_upsample32(const pool32_params& p, hipStream_t s); // stride = integer scale, k = 0 nearest / 1 bilinear
hipError_t launch_output_transform32(tview32 in, int B, int H, int W, const out_xform& x, float* out, hipStream_t s);

} // namespace hp
