// conv_kernels.hpp — launch interface of the gfx950 convolution kernels (conv_kernels.hip) used by the
// engine (engine.cpp).  These replace the TensorRT-built network of the reference
// (src/tensorrt.cpp:393 `executeV2`): every layer of the exported graphs (hyperpose/Model/backbones.py,
// openpose/model/*.py, pose_proposal/model.py, pifpaf/model.py) maps onto one of the launches below.
//
// Activation layout in HBM: NHWC fp16, channel count padded to a multiple of 8 (16-byte vectors), a tensor
// may be a channel slice [coff, coff+C) of a wider buffer with pixel stride `cs` (concat by offset).
// Network outputs are written fp32 NCHW [B][C][H][W], the layout feature_map_t / the parsers consume.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hp {

enum act_t : int { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LEAKY = 3, ACT_PRELU = 4, ACT_SIGMOID = 5, ACT_SOFTPLUS = 6 };

struct conv_params {
    // input: fp16 NHWC view
    const __half* in;
    int in_cs, in_coff;
    int B, H, W;
    int OH, OW;
    int Cin;      // channels read per tap; multiple of 32 for the MFMA kernel
    int Cout;     // real output channels
    int Cout_pad; // rows of the packed weight matrix (multiple of the M tile)
    int KH, KW, stride, dil, pad_t, pad_l;
    const __half* w;    // packed [KH*KW][Cout_pad][Cin]
    const float* bias;  // [Cout_pad]
    const float* alpha; // PReLU slopes [Cout_pad] or nullptr
    int act;          // ACT_NONE / RELU / RELU6 / LEAKY / PRELU (the piecewise-linear ones)
    float act_param;
    float act_slope, act_hi; // y = v > 0 ? min(v, act_hi) : v * act_slope, filled by set_act()
    // optional residual (same shape as the output), added after (res_before_act = 0) or before the activation
    const __half* res;
    int res_cs, res_coff, res_before_act;
    // outputs (either may be null)
    __half* out;
    int out_cs, out_coff;
    float* out_f32; // fp32 NCHW [B][Cout][OH][OW]
    int npix;       // B*OH*OW
};

// fills act_slope / act_hi from act / act_param; false for activations the MFMA epilogue does not fuse
bool set_act(conv_params& p);
// Dense k x k convolution as an implicit GEMM on MFMA (v_mfma_f32_32x32x16_f16).  Returns hipError_t.
hipError_t launch_conv_mfma(const conv_params& p, hipStream_t s);
// which tile the launcher picks (for reporting): returns BM*1000+BN
int conv_mfma_tile(const conv_params& p);

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
    const float* bias;
    int act;
    float act_param;
    __half* out;
    int out_cs, out_coff;
};
// Direct convolution for the 3-channel network input (u8 HWC or f32 NCHW), pre-processing fused into the load.
hipError_t launch_first_conv(const first_conv_params& p, hipStream_t s);

struct dw_params {
    const __half* in;
    int in_cs, in_coff;
    int B, H, W, OH, OW, C; // C multiple of 8
    int stride, dil, pad_t, pad_l;
    const __half* w;   // packed [9][C]
    const float* bias; // [C]
    int act;
    float act_param;
    __half* out;
    int out_cs, out_coff;
};
// Depthwise 3x3 (VALU, HBM/L2-bound): one thread = one output pixel x 8 channels.
hipError_t launch_dwconv3x3(const dw_params& p, hipStream_t s);

struct pool_params {
    const __half* in;
    int in_cs, in_coff;
    int B, H, W, OH, OW, C;
    int k, stride, pad_t, pad_l;
    __half* out;
    int out_cs, out_coff;
};
hipError_t launch_maxpool(const pool_params& p, hipStream_t s);

// fp16 NHWC view -> fp32 NCHW (for outputs not produced by a conv epilogue) with an optional element-wise op.
hipError_t launch_nhwc_to_nchw_f32(const __half* in, int in_cs, int in_coff, int B, int H, int W, int C, int act, float* out,
    hipStream_t s);

} // namespace hp
