// conv_kernels.hip — hand-written gfx950 kernels for the backbone + head convolution stack.
//
// Replaces the TensorRT engine the reference builds and runs at src/tensorrt.cpp:121-252 / :393; layer
// semantics (TF "SAME" padding, folded BatchNorm, activation placement) follow the Python model
// definitions the reference exports from (hyperpose/Model/backbones.py, openpose/model/lw_openpose.py, ...).
//
//   conv_mfma_kernel   dense k x k conv as implicit GEMM:  D[cout][pixel] = sum_{tap,cin} W[tap][cout][cin] * X[pixel@tap][cin]
//                      v_mfma_f32_32x32x16_f16, A = weights, B = activations (both K-contiguous in HBM: packed
//                      weights [tap][cout][cin], activations NHWC), fp32 accumulate.  256 threads = 2x2 wavefronts,
//                      block tile BM x BN x BK, global->register->LDS double buffering with ONE barrier per
//                      K-step, XOR-swizzled LDS rows so that ds_read_b128 fragment reads are bank-conflict free.
//                      The accumulator layout gives every lane 4 consecutive output channels of one pixel, so
//                      the NHWC fp16 store is an 8-byte vector; bias / activation / residual / the fp32 NCHW
//                      copy for the parsers are fused into the epilogue.
//   first_conv_kernel  3-channel network input (u8 HWC or f32 NCHW): pre-processing (x factor, BGR->RGB, mean/std)
//                      fused into the load, fp32 math, HBM-bound.
//   dwconv3x3_kernel   depthwise 3x3, one thread = one pixel x 8 channels (16-byte loads/stores), HBM/L2-bound.
#include "conv_kernels.hpp"

namespace hp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4))); // native 16-byte vector (HIP's uint4 struct defeats SROA here)

__device__ __forceinline__ float apply_act(float v, int act, float param, float alpha)
{
    switch (act) {
    case ACT_RELU:
        return fmaxf(v, 0.f);
    case ACT_RELU6:
        return fminf(fmaxf(v, 0.f), 6.f);
    case ACT_LEAKY:
        return v > 0.f ? v : v * param;
    case ACT_PRELU:
        return v > 0.f ? v : v * alpha;
    case ACT_SIGMOID:
        return 1.f / (1.f + __expf(-v));
    case ACT_SOFTPLUS:
        return v > 20.f ? v : log1pf(__expf(v));
    default:
        return v;
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS tile: ROWS x BK halves, row = BK*2 bytes, 16-byte chunks XOR-swizzled by the row index so that the
// 16-lane service groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table) hit 16 distinct 16-byte slots.
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk)
{
    if (BK == 32)
        return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    else
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BM, int BN, int BK, int EPI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const conv_params p)
{
    constexpr int CH = BK / 8;            // 16-byte chunks per tile row
    constexpr int RPP = 256 / CH;         // tile rows covered by one pass of the 256 threads
    constexpr int A_LD = BM / RPP;        // 16-byte global loads per thread for the weight tile
    constexpr int B_LD = BN / RPP;        // ... for the activation tile
    constexpr int TM = BM / 64, TN = BN / 64; // 32x32 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int TILE_BYTES = (BM + BN) * BK * 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const int ld_row = tid / CH, ld_chunk = tid % CH;
    const int KC = p.Cin / BK;
    const int OHW = p.OH * p.OW;

    // activation rows (pixels) this thread stages
    int pb[B_LD], iy0[B_LD], ix0[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int n = n0 + ld_row + i * RPP;
        if (n < p.npix) {
            const int b = n / OHW, rem = n - b * OHW;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            pb[i] = b * p.H;
            iy0[i] = oy * p.stride - p.pad_t;
            ix0[i] = ox * p.stride - p.pad_l;
        } else {
            pb[i] = 0;
            iy0[i] = -(1 << 20); // clamps to row 0 and is flagged invalid for every tap
            ix0[i] = 0;
        }
    }

    // Global loads of the NEXT K-step are issued before the MFMA phase of the current one and consumed after
    // it; every load is unconditional (clamped address, zeroed at the LDS store when the tap falls in the
    // padding) so that nothing forces an early s_waitcnt.  (Written without lambdas: hipcc keeps lambda-captured
    // register arrays in scratch.)
    u32x4 ra[A_LD], rb[B_LD];
    unsigned bvalid = 0;
    int l_ky = 0, l_kx = 0, l_kc = 0;
#define HP_GLOAD()                                                                                               \
    {                                                                                                            \
        const int tap_ = l_ky * p.KW + l_kx;                                                                     \
        const __half* wbase_ = p.w + ((size_t)tap_ * p.Cout_pad + m0) * p.Cin + l_kc * BK + ld_chunk * 8;        \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                         \
            ra[i] = *reinterpret_cast<const u32x4*>(wbase_ + (size_t)(ld_row + i * RPP) * p.Cin);                \
        bvalid = 0;                                                                                              \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                                         \
        {                                                                                                        \
            const int iy_ = iy0[i] + l_ky * p.dil, ix_ = ix0[i] + l_kx * p.dil;                                  \
            const bool ok_ = iy_ >= 0 && iy_ < p.H && ix_ >= 0 && ix_ < p.W;                                     \
            bvalid |= (ok_ ? 1u : 0u) << i;                                                                      \
            const int cy_ = min(max(iy_, 0), p.H - 1), cx_ = min(max(ix_, 0), p.W - 1);                          \
            rb[i] = *reinterpret_cast<const u32x4*>(                                                             \
                p.in + ((size_t)(pb[i] + cy_) * p.W + cx_) * p.in_cs + p.in_coff + l_kc * BK + ld_chunk * 8);    \
        }                                                                                                        \
        if (++l_kc == KC) {                                                                                      \
            l_kc = 0;                                                                                            \
            if (++l_kx == p.KW) {                                                                                \
                l_kx = 0;                                                                                        \
                ++l_ky;                                                                                          \
            }                                                                                                    \
        }                                                                                                        \
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    const int steps = p.KH * p.KW * KC;
    const int frow = lane & 31, fk = lane >> 5;
    HP_GLOAD();
    for (int s = 0; s < steps; ++s) {
        unsigned char* a = lds + (s & 1) * TILE_BYTES;
        unsigned char* b = a + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<u32x4*>(a + lds_off<BK>(ld_row + i * RPP, ld_chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const unsigned keep = ((bvalid >> i) & 1u) ? 0xffffffffu : 0u;
            *reinterpret_cast<u32x4*>(b + lds_off<BK>(ld_row + i * RPP, ld_chunk)) = rb[i] & keep;
        }
        __syncthreads();
        if (s + 1 < steps)
            HP_GLOAD();
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch ABOVE the MFMA phase (hipcc otherwise sinks it to its use)
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const half8*>(a + lds_off<BK>(wm * (BM / 2) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const half8*>(b + lds_off<BK>(wn * (BN / 2) + j * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef HP_GLOAD

    // epilogue: lane holds pixel n = (lane & 31) of each 32-wide tile and channels 8g + 4*(lane>>5) + {0..3}.
    // Activations are piecewise linear: y = v > 0 ? min(v, hi) : v * slope  (none/relu/relu6/leaky/prelu).
    const float hi = p.act_hi;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        const bool nvalid = n < p.npix;
        int b = 0, rem = 0;
        if (EPI == 1) {
            b = n / OHW;
            rem = n - b * OHW;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = m0 + wm * (BM / 2) + i * 32 + 8 * g + 4 * (lane >> 5);
                if (nvalid && m < p.Cout) {
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + m);
                    float4 sl = make_float4(p.act_slope, p.act_slope, p.act_slope, p.act_slope);
                    if (p.alpha)
                        sl = *reinterpret_cast<const float4*>(p.alpha + m);
                    float v0 = acc[i][j][4 * g + 0] + bs.x, v1 = acc[i][j][4 * g + 1] + bs.y;
                    float v2 = acc[i][j][4 * g + 2] + bs.z, v3 = acc[i][j][4 * g + 3] + bs.w;
                    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
                    if (p.res) {
                        const __half* rp = p.res + (size_t)n * p.res_cs + p.res_coff + m;
                        if (EPI == 0) {
                            const half4 h = *reinterpret_cast<const half4*>(rp);
                            r0 = (float)h[0], r1 = (float)h[1], r2 = (float)h[2], r3 = (float)h[3];
                        } else {
                            r0 = __half2float(rp[0]);
                            r1 = m + 1 < p.Cout ? __half2float(rp[1]) : 0.f;
                            r2 = m + 2 < p.Cout ? __half2float(rp[2]) : 0.f;
                            r3 = m + 3 < p.Cout ? __half2float(rp[3]) : 0.f;
                        }
                        if (p.res_before_act)
                            v0 += r0, v1 += r1, v2 += r2, v3 += r3, r0 = r1 = r2 = r3 = 0.f;
                    }
                    v0 = (v0 > 0.f ? fminf(v0, hi) : v0 * sl.x) + r0;
                    v1 = (v1 > 0.f ? fminf(v1, hi) : v1 * sl.y) + r1;
                    v2 = (v2 > 0.f ? fminf(v2, hi) : v2 * sl.z) + r2;
                    v3 = (v3 > 0.f ? fminf(v3, hi) : v3 * sl.w) + r3;
                    if (EPI == 0) {
                        half4 h;
                        h[0] = (_Float16)v0, h[1] = (_Float16)v1, h[2] = (_Float16)v2, h[3] = (_Float16)v3;
                        *reinterpret_cast<half4*>(p.out + (size_t)n * p.out_cs + p.out_coff + m) = h;
                    } else {
                        const bool c1 = m + 1 < p.Cout, c2 = m + 2 < p.Cout, c3 = m + 3 < p.Cout;
                        if (p.out) {
                            __half* op = p.out + (size_t)n * p.out_cs + p.out_coff + m;
                            op[0] = __float2half(v0);
                            if (c1)
                                op[1] = __float2half(v1);
                            if (c2)
                                op[2] = __float2half(v2);
                            if (c3)
                                op[3] = __float2half(v3);
                        }
                        if (p.out_f32) {
                            float* fp = p.out_f32 + ((size_t)b * p.Cout + m) * OHW + rem;
                            fp[0] = v0;
                            if (c1)
                                fp[OHW] = v1;
                            if (c2)
                                fp[2 * (size_t)OHW] = v2;
                            if (c3)
                                fp[3 * (size_t)OHW] = v3;
                        }
                    }
                }
            }
        }
    }
}

// fast epilogue (aligned fp16 NHWC vectors) when every 4-channel group is whole and 8-byte aligned
static bool fast_epilogue(const conv_params& p)
{
    return p.out && !p.out_f32 && p.Cout % 4 == 0 && p.out_coff % 4 == 0 && p.out_cs % 4 == 0
        && (!p.res || (p.res_coff % 4 == 0 && p.res_cs % 4 == 0));
}

template <int BM, int BN, int BK>
static hipError_t launch_tile(const conv_params& p, hipStream_t s)
{
    dim3 grid((p.npix + BN - 1) / BN, p.Cout_pad / BM);
    if (fast_epilogue(p))
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, 0>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, 1>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

bool set_act(conv_params& p)
{
    const float inf = __builtin_huge_valf();
    switch (p.act) {
    case ACT_NONE:
        p.act_slope = 1.f, p.act_hi = inf;
        return true;
    case ACT_RELU:
        p.act_slope = 0.f, p.act_hi = inf;
        return true;
    case ACT_RELU6:
        p.act_slope = 0.f, p.act_hi = 6.f;
        return true;
    case ACT_LEAKY:
        p.act_slope = p.act_param, p.act_hi = inf;
        return true;
    case ACT_PRELU:
        p.act_slope = 0.f, p.act_hi = inf;
        return p.alpha != nullptr;
    default:
        return false;
    }
}

int conv_mfma_tile(const conv_params& p)
{
    const int BM = (p.Cout_pad % 128 == 0) ? 128 : 64;
    // prefer the 128-pixel tile only when it still fills the 256 CUs at least once
    const long blocks128 = (long)((p.npix + 127) / 128) * (p.Cout_pad / BM);
    const int BN = blocks128 >= 256 ? 128 : 64;
    return BM * 1000 + BN;
}

hipError_t launch_conv_mfma(const conv_params& p, hipStream_t s)
{
    const int t = conv_mfma_tile(p);
    const int BM = t / 1000, BN = t % 1000;
    const bool k64 = (p.Cin % 64 == 0);
    if (BM == 128 && BN == 128)
        return k64 ? launch_tile<128, 128, 64>(p, s) : launch_tile<128, 128, 32>(p, s);
    if (BM == 128 && BN == 64)
        return k64 ? launch_tile<128, 64, 64>(p, s) : launch_tile<128, 64, 32>(p, s);
    if (BM == 64 && BN == 128)
        return k64 ? launch_tile<64, 128, 64>(p, s) : launch_tile<64, 128, 32>(p, s);
    return k64 ? launch_tile<64, 64, 64>(p, s) : launch_tile<64, 64, 32>(p, s);
}

// ---------------------------------------------------------------------------------------------------
// First layer: Cin = 3.  Block = 256 threads = (256 / G) pixels x G groups of 8 output channels.
__global__ __launch_bounds__(256) void first_conv_kernel(const first_conv_params p)
{
    extern __shared__ __attribute__((aligned(16))) float s_w[]; // [KH*KW*3][Cout_pad8]
    const int G = (p.Cout + 7) / 8;
    const int CP = G * 8;
    const int taps = p.KH * p.KW;
    for (int i = threadIdx.x; i < taps * 3 * CP; i += 256) {
        const int co = i % CP, t = i / CP; // t = tap*3 + c
        s_w[i] = co < p.Cout ? p.w[(size_t)co * taps * 3 + t] : 0.f;
    }
    __syncthreads();
    const int ppb = 256 / G;
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    if (pl >= ppb)
        return;
    const int OHW = p.OH * p.OW;
    const long npix = (long)p.B * OHW;
    for (long n = (long)blockIdx.x * ppb + pl; n < npix; n += (long)gridDim.x * ppb) {
        const int b = (int)(n / OHW), rem = (int)(n - (long)b * OHW);
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            acc[r] = (g * 8 + r < p.Cout) ? p.bias[g * 8 + r] : 0.f;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.stride - p.pad_t + ky;
            if (iy < 0 || iy >= p.H)
                continue;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx;
                if (ix < 0 || ix >= p.W)
                    continue;
                float x[3];
                if (p.in_u8) {
                    const uint8_t* px = p.in_u8 + (((size_t)b * p.H + iy) * p.W + ix) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int sc = p.flip_rb ? 2 - c : c;
                        x[c] = (float)((double)px[sc] * p.factor); // src/data.cpp:48
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        x[c] = p.in_f32[(((size_t)b * 3 + c) * p.H + iy) * p.W + ix];
                }
                const float* wt = s_w + (size_t)((ky * p.KW + kx) * 3) * CP + g * 8;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xv = (x[c] - p.mean[c]) * p.inv_std[c];
                    const float4 w0 = *reinterpret_cast<const float4*>(wt + c * CP);
                    const float4 w1 = *reinterpret_cast<const float4*>(wt + c * CP + 4);
                    acc[0] += xv * w0.x, acc[1] += xv * w0.y, acc[2] += xv * w0.z, acc[3] += xv * w0.w;
                    acc[4] += xv * w1.x, acc[5] += xv * w1.y, acc[6] += xv * w1.z, acc[7] += xv * w1.w;
                }
            }
        }
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            h[r] = (_Float16)apply_act(acc[r], p.act, p.act_param, 0.f);
        __half* op = p.out + (size_t)n * p.out_cs + p.out_coff + g * 8;
        if (g * 8 + 7 < p.Cout && ((p.out_coff & 7) == 0) && ((p.out_cs & 7) == 0))
            *reinterpret_cast<half8*>(op) = h;
        else
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (g * 8 + r < p.Cout)
                    reinterpret_cast<_Float16*>(op)[r] = h[r];
    }
}

hipError_t launch_first_conv(const first_conv_params& p, hipStream_t s)
{
    const int G = (p.Cout + 7) / 8;
    if (G > 256)
        return hipErrorInvalidValue;
    const int ppb = 256 / G;
    const long npix = (long)p.B * p.OH * p.OW;
    const int blocks = (int)std::min<long>((npix + ppb - 1) / ppb, 256 * 16);
    const size_t lds = (size_t)p.KH * p.KW * 3 * G * 8 * sizeof(float);
    hipLaunchKernelGGL(first_conv_kernel, dim3(blocks), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const dw_params p)
{
    const int CG = p.C / 8;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        float acc[8];
        {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cg * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cg * 8 + 4);
            acc[0] = b0.x, acc[1] = b0.y, acc[2] = b0.z, acc[3] = b0.w, acc[4] = b1.x, acc[5] = b1.y, acc[6] = b1.z, acc[7] = b1.w;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * p.stride - p.pad_t + ky * p.dil;
            if (iy < 0 || iy >= p.H)
                continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx * p.dil;
                if (ix < 0 || ix >= p.W)
                    continue;
                const half8 x = *reinterpret_cast<const half8*>(p.in + (((size_t)b * p.H + iy) * p.W + ix) * p.in_cs + p.in_coff + cg * 8);
                const half8 w = *reinterpret_cast<const half8*>(p.w + (size_t)(ky * 3 + kx) * p.C + cg * 8);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    acc[r] += (float)x[r] * (float)w[r];
            }
        }
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            h[r] = (_Float16)apply_act(acc[r], p.act, p.act_param, 0.f);
        *reinterpret_cast<half8*>(p.out + (size_t)n * p.out_cs + p.out_coff + cg * 8) = h;
    }
}

hipError_t launch_dwconv3x3(const dw_params& p, hipStream_t s)
{
    const long total = (long)p.B * p.OH * p.OW * (p.C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const pool_params p)
{
    const int CG = p.C / 8;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        float m[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            m[r] = -65504.f;
        for (int ky = 0; ky < p.k; ++ky) {
            const int iy = oy * p.stride - p.pad_t + ky;
            if (iy < 0 || iy >= p.H)
                continue;
            for (int kx = 0; kx < p.k; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx;
                if (ix < 0 || ix >= p.W)
                    continue;
                const half8 x = *reinterpret_cast<const half8*>(p.in + (((size_t)b * p.H + iy) * p.W + ix) * p.in_cs + p.in_coff + cg * 8);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    m[r] = fmaxf(m[r], (float)x[r]);
            }
        }
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            h[r] = (_Float16)m[r];
        *reinterpret_cast<half8*>(p.out + (size_t)n * p.out_cs + p.out_coff + cg * 8) = h;
    }
}

hipError_t launch_maxpool(const pool_params& p, hipStream_t s)
{
    const long total = (long)p.B * p.OH * p.OW * (p.C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const __half* __restrict__ in, int in_cs, int in_coff, int B, int HW,
    int C, int act, float* __restrict__ out)
{
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), b = (int)(t / C);
        const float v = __half2float(in[((size_t)b * HW + pix) * in_cs + in_coff + c]);
        out[i] = apply_act(v, act, 0.f, 0.f);
    }
}

hipError_t launch_nhwc_to_nchw_f32(const __half* in, int in_cs, int in_coff, int B, int H, int W, int C, int act, float* out,
    hipStream_t s)
{
    const long total = (long)B * C * H * W;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, in, in_cs, in_coff, B, H * W, C, act, out);
    return hipGetLastError();
}

} // namespace hp
