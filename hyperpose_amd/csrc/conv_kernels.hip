// conv_kernels.hip — hand-written gfx950 kernels for the backbone + head convolution stack.
//
// Replaces the TensorRT engine the reference builds and runs at src/tensorrt.cpp:121-252 / :393; layer
// semantics (TF "SAME" padding, folded BatchNorm, activation placement) follow the Python model
// definitions the reference exports from (hyperpose/Model/backbones.py, openpose/model/lw_openpose.py, ...).
//
//   conv_mfma_kernel    dense k x k conv as implicit GEMM:  D[cout][pixel] = sum_{tap,cin} W[tap][cout][cin] * X[pixel@tap][cin]
//                       v_mfma_f32_32x32x16_f16, A = weights, B = activations (both K-contiguous in HBM: packed
//                       weights [tap][cout][cin], activations NHWC), fp32 accumulate.  256 threads = 2x2 wavefronts,
//                       block tile BM x BN x BK, global->register->LDS staging with the loads of K-step s+2 in
//                       flight while step s computes, ONE barrier per K-step, XOR-swizzled LDS rows so that
//                       ds_read_b128 fragment reads are bank-conflict free, XCD-aware block->tile mapping.
//   conv3x3_halo_kernel the 3x3 / stride 1 case (44 % of LW-OpenPose's conv time): an 8x16-pixel output tile and its
//                       1-pixel halo are staged in LDS ONCE and re-used by all 9 taps (L2 traffic per block drops
//                       from 9 activation tiles to 1.4), only the weights stream through a prefetched LDS ring.
//   Both share one epilogue: each lane owns 4 consecutive output channels of a pixel -> 8-byte NHWC stores; bias,
//   piecewise-linear activation, residual add and the fp32 NCHW copy for the parsers are fused.
//   first_conv_kernel   3-channel network input (u8 HWC or f32 NCHW): pre-processing (x factor, BGR->RGB, mean/std)
//                       fused into the load, fp32 math, HBM-bound.
//   dwconv3x3_kernel    depthwise 3x3, one thread = one pixel x 8 channels (16-byte loads/stores), HBM/L2-bound.
// Activations carry a zero halo in HBM (conv_kernels.hpp), so taps in the padding are ordinary loads.
#include "conv_device.hpp"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace hp {
thread_local hipEvent_t prof_start = nullptr, prof_stop = nullptr;
}

namespace hp {

// One MFMA, then one LDS read, four times: a single wavefront per SIMD issues in order, so the next fragments' reads
// must sit INSIDE the 32-cycle shadows of the current MFMAs (cdna_hip_programming.md T19) instead of after them.
#define HP_INTERLEAVE4()                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);

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
    else if (BK == 64)
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    else
        return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// Shared epilogue.  Lane holds, for MFMA tile (i, j), pixel j-th "column" (given by pb/py/px/pv) and channels
// m_wave + i*32 + 8g + 4*(lane>>5) + {0..3}, g = 0..3.  Activations are piecewise linear:
// y = v > 0 ? min(v, hi) : v * slope  (none / relu / relu6 / leaky / prelu).
template <int TM, int TN, int EPI>
__device__ __forceinline__ void conv_epilogue(const conv_params& p, const floatx16 (&acc)[TM][TN], int m_wave, int lane,
    const int (&pb)[TN], const int (&py)[TN], const int (&px)[TN], const bool (&pv)[TN])
{
    const float hi = p.act_hi;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const bool nvalid = pv[j];
        long o_off = 0, r_off = 0, f_off = 0;
        if (nvalid) {
            if (p.out.p)
                o_off = tv_off(p.out, pb[j], py[j], px[j]);
            if (p.res.p)
                r_off = tv_off(p.res, pb[j], py[j], px[j]);
            if (EPI == 1)
                f_off = ((long)pb[j] * p.Cout * p.OH + py[j]) * p.OW + px[j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = m_wave + i * 32 + 8 * g + 4 * (lane >> 5);
                if (nvalid && m < p.Cout) {
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + m);
                    float4 sl = make_float4(p.act_slope, p.act_slope, p.act_slope, p.act_slope);
                    if (p.alpha)
                        sl = *reinterpret_cast<const float4*>(p.alpha + m);
                    float v0 = acc[i][j][4 * g + 0] + bs.x, v1 = acc[i][j][4 * g + 1] + bs.y;
                    float v2 = acc[i][j][4 * g + 2] + bs.z, v3 = acc[i][j][4 * g + 3] + bs.w;
                    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
                    if (p.res.p) {
                        const __half* rp = p.res.p + r_off + m;
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
                        *reinterpret_cast<half4*>(p.out.p + o_off + m) = h;
                    } else {
                        const bool c1 = m + 1 < p.Cout, c2 = m + 2 < p.Cout, c3 = m + 3 < p.Cout;
                        if (p.out.p) {
                            __half* op = p.out.p + o_off + m;
                            op[0] = __float2half(v0);
                            if (c1)
                                op[1] = __float2half(v1);
                            if (c2)
                                op[2] = __float2half(v2);
                            if (c3)
                                op[3] = __float2half(v3);
                        }
                        if (p.out_f32) {
                            const long plane = (long)p.OH * p.OW;
                            float* fp = p.out_f32 + f_off + (long)m * plane;
                            fp[0] = v0;
                            if (c1)
                                fp[plane] = v1;
                            if (c2)
                                fp[2 * plane] = v2;
                            if (c3)
                                fp[3 * plane] = v3;
                        }
                    }
                }
            }
        }
    }
}

// Fast epilogue (EPI == 0: aligned fp16 NHWC output, whole 4-channel groups): every wavefront transposes its
// accumulator tile through a private LDS slab ([32 pixels][32*TM channels] fp32 per pass, rows padded by 16 B so that
// both the ds_write_b128 of the MFMA layout and the ds_read_b128 of the store layout are conflict-free) and then
// stores 16 bytes per lane with 4*TM consecutive lanes covering one pixel's contiguous channel run — 8 cache lines
// per store instruction instead of 64 with the raw MFMA layout (cdna_hip_programming.md T21).  Bias, activation and
// the residual add happen on the store side in fp32, i.e. the same arithmetic as the direct epilogue.
template <int TM>
struct stage_geom {
    static constexpr int ROW = TM * 128 + 16;          // bytes per staged pixel row
    static constexpr int SLAB = 32 * ROW + 32 * 8 * 2; // + per-pixel output / residual offsets (long)
    static constexpr int CPP = TM * 4;                 // 8-channel chunks per pixel
    static constexpr int PPP = 64 / CPP;               // pixels per store pass
    static constexpr int PASSES = 32 / PPP;
};

#define HP_ESTAMP()                                                                                               \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                                           \
        p.dbg[40 + (edbg++)] = __builtin_amdgcn_s_memtime();
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_staged(const conv_params& p, const floatx16 (&acc)[TM][TN], int m_wave, int lane,
    unsigned char* slab, const int (&pb)[TN], const int (&py)[TN], const int (&px)[TN], const bool (&pv)[TN])
{
    using G = stage_geom<TM>;
    int edbg = 0;
    HP_ESTAMP();
    long* s_ooff = reinterpret_cast<long*>(slab + 32 * G::ROW);
    long* s_roff = s_ooff + 32;
    const int chunk = lane % G::CPP, prow = lane / G::CPP;
    const int mc = m_wave + chunk * 8; // first of this lane's 8 output channels on the store side (Cout % 8 == 0 here)
    const bool mvalid = mc < p.Cout;
    float bs[8], sl[8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + mc), b1 = *reinterpret_cast<const float4*>(p.bias + mc + 4);
        bs[0] = b0.x, bs[1] = b0.y, bs[2] = b0.z, bs[3] = b0.w, bs[4] = b1.x, bs[5] = b1.y, bs[6] = b1.z, bs[7] = b1.w;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            sl[r] = p.act_slope;
        if (p.alpha) { // uniform
            const float4 a0 = *reinterpret_cast<const float4*>(p.alpha + mc), a1 = *reinterpret_cast<const float4*>(p.alpha + mc + 4);
            sl[0] = a0.x, sl[1] = a0.y, sl[2] = a0.z, sl[3] = a0.w, sl[4] = a1.x, sl[5] = a1.y, sl[6] = a1.z, sl[7] = a1.w;
        }
    }
    const float hi = p.act_hi;
    const bool has_res = p.res.p != nullptr; // uniform
    const bool clamp_only = !has_res && !p.alpha && p.act_slope == 0.f; // uniform: v > 0 ? min(v, hi) : v * 0 == med3(v, 0, hi) up to the sign of zero
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HP_ESTAMP();
    // The uniform choice is made ONCE, around the loop over the pixel tiles: with the residual requests and the clamp-only arithmetic in
    // one loop body the compiler put `s_waitcnt vmcnt(0)` in front of every store pass of the clamp-only path (a request of the other
    // path might be pending) - and vmcnt counts stores on gfx9: every pass waited for the previous pass's stores to reach L2.
    auto tile = [&](int j, auto clamp_) {
        constexpr bool CLAMP = decltype(clamp_)::value;
        // MFMA layout -> LDS: lane owns pixel (lane & 31), channels i*32 + 8g + 4*(lane>>5) + {0..3}
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc[i][j][4 * g + 0], v.y = acc[i][j][4 * g + 1], v.z = acc[i][j][4 * g + 2], v.w = acc[i][j][4 * g + 3];
                *reinterpret_cast<float4*>(slab + (lane & 31) * G::ROW + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
            }
        if (lane < 32) {
            s_ooff[lane] = pv[j] ? tv_off(p.out, pb[j], py[j], px[j]) : -1;
            if constexpr (!CLAMP)
                s_roff[lane] = (pv[j] && has_res) ? tv_off(p.res, pb[j], py[j], px[j]) : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes have landed (DS ops retire in order)
        __builtin_amdgcn_wave_barrier();
        // store side: offsets, then ALL residual loads (unconditional, straight-line), then every slab row, then math + 16-byte stores
        long oo[G::PASSES];
        half8 rs[G::PASSES];
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps)
            oo[ps] = s_ooff[ps * G::PPP + prow];
        if constexpr (!CLAMP) {
            if (has_res) {
#pragma unroll
                for (int ps = 0; ps < G::PASSES; ++ps) // invalid pixels read offset 0: in bounds, result unused
                    rs[ps] = *reinterpret_cast<const half8*>(p.res.p + s_roff[ps * G::PPP + prow] + (mvalid ? mc : 0));
            } else {
#pragma unroll
                for (int ps = 0; ps < G::PASSES; ++ps)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        rs[ps][r] = (_Float16)0.f;
            }
        }
        float4 a0[G::PASSES], a1[G::PASSES];
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps) {
            const int pix = ps * G::PPP + prow;
            a0[ps] = *reinterpret_cast<const float4*>(slab + pix * G::ROW + chunk * 32);
            a1[ps] = *reinterpret_cast<const float4*>(slab + pix * G::ROW + chunk * 32 + 16);
        }
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps) {
            const float v[8] = { a0[ps].x, a0[ps].y, a0[ps].z, a0[ps].w, a1[ps].x, a1[ps].y, a1[ps].z, a1[ps].w };
            half8 h;
            if constexpr (CLAMP) { // relu / relu6 family without a residual: bias add + one v_med3_f32 per value
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    h[r] = (_Float16)__builtin_amdgcn_fmed3f(v[r] + bs[r], 0.f, hi);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float x = v[r] + bs[r];
                    const float rr = (float)rs[ps][r];
                    if (p.res_before_act)
                        x += rr;
                    x = x > 0.f ? fminf(x, hi) : x * sl[r];
                    if (!p.res_before_act)
                        x += rr;
                    h[r] = (_Float16)x;
                }
            }
            if (oo[ps] >= 0 && mvalid)
                *reinterpret_cast<half8*>(p.out.p + oo[ps] + mc) = h;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // slab reads done before the next pass overwrites it
        __builtin_amdgcn_wave_barrier();
    };
    if (clamp_only) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
            tile(j, std::true_type{});
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
            tile(j, std::false_type{});
    }
    HP_ESTAMP();
}
#undef HP_ESTAMP

// The same epilogue with ALL TN pixel tiles of a wavefront staged at once (slab = TN x stage_geom<TM>::SLAB per wavefront): one
// write phase, one wait, then every read / residual load / store of the wavefront in flight together.  conv_epilogue_staged runs
// write -> wait -> (read -> math -> store) x passes once per pixel tile, a chain of LDS and store latencies that measured ~2.5 k
// cycles per tile in the pixel-block GEMM (8 of a block's 27 k cycles at TM = 2, TN = 3); same arithmetic, same stores.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_wide(const conv_params& p, const floatx16 (&acc)[TM][TN], int m_wave, int lane,
    unsigned char* slab, const int (&pb)[TN], const int (&py)[TN], const int (&px)[TN], const bool (&pv)[TN])
{
    using G = stage_geom<TM>;
    const int chunk = lane % G::CPP, prow = lane / G::CPP;
    const int mc = m_wave + chunk * 8; // first of this lane's 8 output channels on the store side (Cout % 8 == 0 here)
    const bool mvalid = mc < p.Cout;
    const bool has_res = p.res.p != nullptr; // uniform
    float bs[8], sl[8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + mc), b1 = *reinterpret_cast<const float4*>(p.bias + mc + 4);
        bs[0] = b0.x, bs[1] = b0.y, bs[2] = b0.z, bs[3] = b0.w, bs[4] = b1.x, bs[5] = b1.y, bs[6] = b1.z, bs[7] = b1.w;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            sl[r] = p.act_slope;
        if (p.alpha) { // uniform
            const float4 a0 = *reinterpret_cast<const float4*>(p.alpha + mc), a1 = *reinterpret_cast<const float4*>(p.alpha + mc + 4);
            sl[0] = a0.x, sl[1] = a0.y, sl[2] = a0.z, sl[3] = a0.w, sl[4] = a1.x, sl[5] = a1.y, sl[6] = a1.z, sl[7] = a1.w;
        }
    }
    const float hi = p.act_hi;
    const bool clamp_only = !has_res && !p.alpha && p.act_slope == 0.f; // uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        unsigned char* const sj = slab + j * G::SLAB;
        long* const s_ooff = reinterpret_cast<long*>(sj + 32 * G::ROW);
        // MFMA layout -> LDS: lane owns pixel (lane & 31), channels i*32 + 8g + 4*(lane>>5) + {0..3}
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc[i][j][4 * g + 0], v.y = acc[i][j][4 * g + 1], v.z = acc[i][j][4 * g + 2], v.w = acc[i][j][4 * g + 3];
                *reinterpret_cast<float4*>(sj + (lane & 31) * G::ROW + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) = v;
            }
        if (lane < 32) {
            s_ooff[lane] = pv[j] ? tv_off(p.out, pb[j], py[j], px[j]) : -1;
            s_ooff[32 + lane] = (pv[j] && has_res) ? tv_off(p.res, pb[j], py[j], px[j]) : 0;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes have landed (DS ops retire in order)
    __builtin_amdgcn_wave_barrier();
    long oo[TN][G::PASSES];
    half8 rs[TN][G::PASSES];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long* const s_ooff = reinterpret_cast<const long*>(slab + j * G::SLAB + 32 * G::ROW);
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps) {
            oo[j][ps] = s_ooff[ps * G::PPP + prow];
            if (has_res) // invalid pixels read offset 0: in bounds, result unused
                rs[j][ps] = *reinterpret_cast<const half8*>(p.res.p + s_ooff[32 + ps * G::PPP + prow] + (mvalid ? mc : 0));
            else
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    rs[j][ps][r] = (_Float16)0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const unsigned char* const sj = slab + j * G::SLAB;
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps) {
            const int pix = ps * G::PPP + prow;
            const float4 a0 = *reinterpret_cast<const float4*>(sj + pix * G::ROW + chunk * 32);
            const float4 a1 = *reinterpret_cast<const float4*>(sj + pix * G::ROW + chunk * 32 + 16);
            const float v[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
            half8 h;
            if (clamp_only) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    h[r] = (_Float16)__builtin_amdgcn_fmed3f(v[r] + bs[r], 0.f, hi);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float x = v[r] + bs[r];
                    const float rr = (float)rs[j][ps][r];
                    if (p.res_before_act)
                        x += rr;
                    x = x > 0.f ? fminf(x, hi) : x * sl[r];
                    if (!p.res_before_act)
                        x += rr;
                    h[r] = (_Float16)x;
                }
            }
            if (oo[j][ps] >= 0 && mvalid)
                *reinterpret_cast<half8*>(p.out.p + oo[j][ps] + mc) = h;
        }
    }
}

// (Round 4 tried this epilogue with the uniform choices hoisted around whole loop nests and every slab read of a pixel tile ahead of its
// stores, as in conv_epilogue_staged / _packed: conv1x1_big_kernel<2,2> went from 158 to 197 registers and ResNet-50's 256 -> 1024
// expansions from 203 to 300 us (tools/profile_layers.py, configs[4]).  Kept as it was.)

// The staged epilogue with the arithmetic moved to the MFMA side of the transpose: bias, activation and the rounding to fp16 happen
// on the accumulator registers (same operations in the same order as above, so the same bits), and the slab holds HALVES: half the
// LDS bytes in both directions and nothing but ds_read_b128 -> global store on the store side.  A slab is 4.9 KB per (wavefront,
// pixel tile) at TM = 2, so all TN tiles of a wavefront are staged at once even with eight wavefronts (the fp32 slabs of
// conv_epilogue_wide would need 221 KB there).  A residual is added in fp32 BEFORE the rounding, so a convolution with one takes
// conv_epilogue_staged (the slab area is sized for either).
template <int TM, int TN>
struct packed_geom {
    static constexpr int ROW = TM * 64 + 16;      // bytes per staged pixel row (halves)
    static constexpr int SLAB = 32 * ROW + 32 * 8; // + per-pixel output offsets
    static constexpr int CPP = TM * 4, PPP = 64 / CPP, PASSES = 32 / PPP;
    static constexpr int WAVE_BYTES = TN * SLAB > stage_geom<TM>::SLAB ? TN * SLAB : stage_geom<TM>::SLAB;
};

// this lane's bias vectors in the order conv_epilogue_packed wants them: a caller with idle registers before its last MFMAs requests
// them there and passes them in, so that the epilogue does not start with a memory latency
template <int TM>
__device__ __forceinline__ void packed_bias(const conv_params& p, int m_wave, int lane, float4 (&bs)[TM][4])
{
    const int mq = m_wave + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bs[i][g] = *reinterpret_cast<const float4*>(p.bias + mq + i * 32 + 8 * g);
}

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_packed(const conv_params& p, const floatx16 (&acc)[TM][TN], int m_wave, int lane,
    unsigned char* slab, const int (&pb)[TN], const int (&py)[TN], const int (&px)[TN], const bool (&pv)[TN], const float4 (&bs)[TM][4])
{
    using G = packed_geom<TM, TN>;
    if (p.res.p) { // uniform
        conv_epilogue_staged<TM, TN>(p, acc, m_wave, lane, slab, pb, py, px, pv);
        return;
    }
    const float hi = p.act_hi;
    const bool clamp_only = !p.alpha && p.act_slope == 0.f; // uniform
    const int mq = m_wave + 4 * (lane >> 5);
    // every bias request goes out before the first use, and the (uniform) choice of the activation is made ONCE around the whole
    // loop nest: with the load and the choice inside it the compiler waited for each of the TM x 4 loads in turn - eight memory
    // latencies, 2.4 of the 3.6 us the 512-channel separable block spent after its last MFMA (round 4)
    auto put = [&](int i, int g, int j, float v0, float v1, float v2, float v3) {
        half4 h;
        h[0] = (_Float16)v0, h[1] = (_Float16)v1, h[2] = (_Float16)v2, h[3] = (_Float16)v3;
        *reinterpret_cast<half4*>(slab + j * G::SLAB + (lane & 31) * G::ROW + (i * 32 + 8 * g + 4 * (lane >> 5)) * 2) = h;
    };
    if (clamp_only) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    put(i, g, j, __builtin_amdgcn_fmed3f(acc[i][j][4 * g + 0] + bs[i][g].x, 0.f, hi),
                        __builtin_amdgcn_fmed3f(acc[i][j][4 * g + 1] + bs[i][g].y, 0.f, hi),
                        __builtin_amdgcn_fmed3f(acc[i][j][4 * g + 2] + bs[i][g].z, 0.f, hi),
                        __builtin_amdgcn_fmed3f(acc[i][j][4 * g + 3] + bs[i][g].w, 0.f, hi));
    } else {
        float4 sl[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                sl[i][g] = make_float4(p.act_slope, p.act_slope, p.act_slope, p.act_slope);
        if (p.alpha) { // uniform
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    sl[i][g] = *reinterpret_cast<const float4*>(p.alpha + mq + i * 32 + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v0 = acc[i][j][4 * g + 0] + bs[i][g].x, v1 = acc[i][j][4 * g + 1] + bs[i][g].y;
                    const float v2 = acc[i][j][4 * g + 2] + bs[i][g].z, v3 = acc[i][j][4 * g + 3] + bs[i][g].w;
                    put(i, g, j, v0 > 0.f ? fminf(v0, hi) : v0 * sl[i][g].x, v1 > 0.f ? fminf(v1, hi) : v1 * sl[i][g].y,
                        v2 > 0.f ? fminf(v2, hi) : v2 * sl[i][g].z, v3 > 0.f ? fminf(v3, hi) : v3 * sl[i][g].w);
                }
    }
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
            reinterpret_cast<long*>(slab + j * G::SLAB + 32 * G::ROW)[lane] = pv[j] ? tv_off(p.out, pb[j], py[j], px[j]) : -1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes have landed (DS ops retire in order)
    __builtin_amdgcn_wave_barrier();
    const int chunk = lane % G::CPP, prow = lane / G::CPP;
    const int mc = m_wave + chunk * 8;
    const bool mvalid = mc < p.Cout;
    // (all LDS reads of a pixel tile first: a read next to the store that needs it costs its LDS latency once per pass)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const unsigned char* const sj = slab + j * G::SLAB;
        half8 h[G::PASSES];
        long oo[G::PASSES];
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps) {
            const int pix = ps * G::PPP + prow;
            h[ps] = *reinterpret_cast<const half8*>(sj + pix * G::ROW + chunk * 16);
            oo[ps] = reinterpret_cast<const long*>(sj + 32 * G::ROW)[pix];
        }
#pragma unroll
        for (int ps = 0; ps < G::PASSES; ++ps)
            if (oo[ps] >= 0 && mvalid)
                *reinterpret_cast<half8*>(p.out.p + oo[ps] + mc) = h[ps];
    }
}

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_packed(const conv_params& p, const floatx16 (&acc)[TM][TN], int m_wave, int lane,
    unsigned char* slab, const int (&pb)[TN], const int (&py)[TN], const int (&px)[TN], const bool (&pv)[TN])
{
    float4 bs[TM][4];
    packed_bias<TM>(p, m_wave, lane, bs);
    conv_epilogue_packed<TM, TN>(p, acc, m_wave, lane, slab, pb, py, px, pv, bs);
}

// ---------------------------------------------------------------------------------------------------
// Generic implicit GEMM.  1-D grid; block id -> (pixel tile, cout tile) with the cout tiles of one pixel tile
// adjacent and consecutive logical ids on the same XCD (blocks are dispatched round-robin over the 8 XCDs).
template <int BM, int BN, int BK, int EPI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const conv_params p)
{
    constexpr int CH = BK / 8;            // 16-byte chunks per tile row
    constexpr int RPP = 256 / CH;         // tile rows covered by one pass of the 256 threads
    constexpr int A_LD = BM / RPP;        // 16-byte global loads per thread for the weight tile
    constexpr int B_LD = BN / RPP;        // ... for the activation tile
    constexpr int TM = BM / 64, TN = BN / 64; // 32x32 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int TILE_BYTES = (BM + BN) * BK * 2;
    constexpr int EPI_BYTES = EPI == 0 ? 4 * stage_geom<TM>::SLAB : 0;
    constexpr int LDS_BYTES = 2 * TILE_BYTES > EPI_BYTES ? 2 * TILE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int MB = p.Cout_pad / BM;
    int m0, n0;
    {
        const int total = gridDim.x, P = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = P & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (P >> 3);
        m0 = (L % MB) * BM;
        n0 = (L / MB) * BN;
    }

    const int ld_row = tid / CH, ld_chunk = tid % CH;
    const int KC = p.Cin / BK;
    const int steps = p.KH * p.KW * KC;
    const int OHW = p.OH * p.OW;

    // per-thread element offsets of the activation rows (pixels) it stages; rows past the end alias the last pixel
    long rowoff[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int n = min(n0 + ld_row + i * RPP, p.npix - 1);
        const int b = n / OHW, rem = n - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        rowoff[i] = tv_off(p.in, b, oy * p.stride - p.pad_t, ox * p.stride - p.pad_l) + ld_chunk * 8;
    }
    const __half* wrow = p.w + (size_t)(m0 + ld_row) * p.Cin + ld_chunk * 8;
    const long w_tap_stride = (long)p.Cout_pad * p.Cin;

    // two register sets: the loads of K-step s+2 are in flight while step s computes
    u32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];
    int l_ky = 0, l_kx = 0, l_kc = 0, l_step = 0;
#define HP_GLOAD(RA, RB)                                                                                          \
    {                                                                                                             \
        const long toff_ = ((long)(l_ky * p.dil) * p.in.wp + l_kx * p.dil) * p.in.cs + l_kc * BK;                 \
        const __half* wb_ = wrow + (long)(l_ky * p.KW + l_kx) * w_tap_stride + l_kc * BK;                         \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                          \
            RA[i] = *reinterpret_cast<const u32x4*>(wb_ + (size_t)(i * RPP) * p.Cin);                             \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                                          \
            RB[i] = *reinterpret_cast<const u32x4*>(p.in.p + rowoff[i] + toff_);                                  \
        if (++l_step < steps) { /* the loads past the last K-step repeat it: issued unconditionally, never used */ \
            if (++l_kc == KC) {                                                                                   \
                l_kc = 0;                                                                                         \
                if (++l_kx == p.KW) {                                                                             \
                    l_kx = 0;                                                                                     \
                    ++l_ky;                                                                                       \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
#define HP_LSTORE(RA, RB, BUF)                                                                                    \
    {                                                                                                             \
        unsigned char* a_ = lds + (BUF) * TILE_BYTES;                                                             \
        unsigned char* b_ = a_ + BM * BK * 2;                                                                     \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                          \
            *reinterpret_cast<u32x4*>(a_ + lds_off<BK>(ld_row + i * RPP, ld_chunk)) = RA[i];                      \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                                          \
            *reinterpret_cast<u32x4*>(b_ + lds_off<BK>(ld_row + i * RPP, ld_chunk)) = RB[i];                      \
    }
#define HP_FRAGS(FA, FB, KS)                                                                                      \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                            \
            FA[i] = *reinterpret_cast<const half8*>(a_ + lds_off<BK>(wm * (BM / 2) + i * 32 + frow, (KS) * 2 + fk)); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
            FB[j] = *reinterpret_cast<const half8*>(b_ + lds_off<BK>(wn * (BN / 2) + j * 32 + frow, (KS) * 2 + fk)); \
    }
#define HP_MMA(FA, FB)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);
// fragments of k-substep ks+1 are read from LDS while the MFMAs of substep ks execute
#define HP_COMPUTE(BUF)                                                                                           \
    {                                                                                                             \
        const unsigned char* a_ = lds + (BUF) * TILE_BYTES;                                                       \
        const unsigned char* b_ = a_ + BM * BK * 2;                                                               \
        half8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];                                                                 \
        HP_FRAGS(fa0, fb0, 0);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < BK / 16; ks += 2)                                                 \
        {                                                                                                         \
            HP_FRAGS(fa1, fb1, ks + 1);                                                                           \
            HP_MMA(fa0, fb0);                                                                                     \
            HP_INTERLEAVE4();                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (ks + 2 < BK / 16) {                                                                               \
                HP_FRAGS(fa0, fb0, ks + 2);                                                                       \
                HP_MMA(fa1, fb1);                                                                                 \
                HP_INTERLEAVE4();                                                                                 \
            } else {                                                                                              \
                HP_MMA(fa1, fb1);                                                                                 \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    const int frow = lane & 31, fk = lane >> 5;
    int dbg_i = 0;
#define HP_STAMP()                                                                                                \
    if (p.dbg && blockIdx.x == 0 && tid == 0)                                                                     \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    HP_GLOAD(ra0, rb0);
    HP_GLOAD(ra1, rb1);
    for (int s = 0; s < steps; s += 2) {
        HP_LSTORE(ra0, rb0, 0);
        HP_STAMP();
        lds_barrier(); // not __syncthreads(): the prefetch of the other register set stays in flight
        HP_STAMP();
        HP_GLOAD(ra0, rb0);
        __builtin_amdgcn_sched_barrier(0); // keep the prefetch ABOVE the MFMA phase (hipcc otherwise sinks it to its use)
        HP_COMPUTE(0);
        __builtin_amdgcn_sched_barrier(0);
        HP_STAMP();
        if (s + 1 < steps) {
            HP_LSTORE(ra1, rb1, 1);
            HP_STAMP();
            lds_barrier();
            HP_STAMP();
            HP_GLOAD(ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            HP_COMPUTE(1);
            __builtin_amdgcn_sched_barrier(0);
            HP_STAMP();
        }
    }
#undef HP_STAMP
#undef HP_GLOAD
#undef HP_LSTORE
#undef HP_COMPUTE
#undef HP_FRAGS
#undef HP_MMA

    int pb[TN], py[TN], px[TN];
    bool pv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        pv[j] = n < p.npix;
        const int nn = min(n, p.npix - 1);
        pb[j] = nn / OHW;
        const int rem = nn - pb[j] * OHW;
        py[j] = rem / p.OW;
        px[j] = rem - py[j] * p.OW;
    }
    if (EPI == 0) {
        __syncthreads(); // every wave is done with the main-loop tiles before the slabs overwrite them
        conv_epilogue_staged<TM, TN>(p, acc, m0 + wm * (BM / 2), lane, lds + wave * stage_geom<TM>::SLAB, pb, py, px, pv);
    } else
        conv_epilogue<TM, TN, EPI>(p, acc, m0 + wm * (BM / 2), lane, pb, py, px, pv);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / dilation 1, barrier-free form (p.w_layout == 1).  Block = 64 output channels x 16 x 12 output pixels, the
// input tile + halo ((16 + 2) x (12 + 2) pixels x CIN halves, 16-byte chunks XOR-swizzled by the pixel position) staged in LDS once;
// the weights never touch LDS: they are packed in MFMA-fragment order [tap][32-row tile][k16 step][lane][8 halves], so every
// A fragment is one coalesced 1 KB load that exactly one wavefront needs.  The four wavefronts are 2 (32-row tiles) x
// 2 (halves of every tap's CIN): no two of them want the same weights, all four read their B fragments from the one
// static halo tile in LDS, and nothing synchronises them between the prologue and the final exchange of the K-halves
// (each wavefront keeps the three pixel tiles it will store and hands the other three to its partner).
// LDS = the halo tile only (64.5 KB at CIN = 128): two blocks, or one block and any other conv kernel, share a CU.
template <int CIN, int TH>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_kernel(const conv_params p, int tiles_x, int tiles_y)
{
    // TH = 16: 192 pixels (6 column tiles of 32) per block; TH = 8: 96 pixels (3) - twice the blocks, two per CU
    constexpr int TW = 12, HPH = TH + 2, HPW = TW + 2, NT = TH * TW / 32;
    constexpr int K0 = (NT + 1) / 2, K1 = NT / 2; // column tiles the K-half 0 / 1 wavefront finishes (3 + 3, or 2 + 1)
    constexpr int CHP = CIN / 8;       // 16-byte chunks per halo pixel
    constexpr int KQ = CIN / 16;       // k16 steps per tap
    constexpr int NS = KQ / 2;         // ... per tap and K-half
    constexpr int HALO_BYTES = HPH * HPW * CIN * 2;
    constexpr int RED_BYTES = 4 * K0 * 16 * 64 * 4; // each wave parks up to K0 accumulator tiles
    constexpr int EPI_BYTES = 4 * stage_geom<1>::SLAB;
    constexpr int LDS_BYTES = HALO_BYTES > RED_BYTES + EPI_BYTES ? HALO_BYTES : RED_BYTES + EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    auto hkey = [](int hy, int hx) { return CHP == 16 ? ((hy * TW + hx) & 15) : (((hy * TW + hx) >> 1) & 7); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, kg = wave >> 1;
    const int m0 = blockIdx.y * 64;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    int dbg_i = 0;
#define HP_STAMP()                                                                                                \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)                                                  \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    // ---- A fragments of taps 0 and 1 (this wave's 32 rows, its half of CIN): 2 x NS loads in flight from the start
    const long tap_stride = (long)(p.Cout_pad / 32) * KQ * 512; // halves per tap
    const __half* wfrag = p.w + ((size_t)((m0 / 32 + wm) * KQ + kg * NS) * 64 + lane) * 8;
    u32x4 a0[NS], a1[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        a0[ks] = *reinterpret_cast<const u32x4*>(wfrag + (size_t)ks * 512);
        a1[ks] = *reinterpret_cast<const u32x4*>(wfrag + tap_stride + (size_t)ks * 512);
    }

    // ---- halo tile: all loads first (one L2 round trip), then the LDS stores
    {
        constexpr int NIT = (HPH * HPW * CHP + 255) / 256;
        u32x4 hv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int hp = min(i, HPH * HPW * CHP - 1) / CHP, c = i % CHP;
            const int hy = hp / HPW, hx = hp - hy * HPW;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = y <= p.H && x <= p.W; // y, x >= -1 always: inside the zero halo of the HBM tensor
            const u32x4 v = *reinterpret_cast<const u32x4*>(p.in.p + tv_off(p.in, b, min(y, p.H), min(x, p.W)) + c * 8);
            hv[it] = v & (ok ? 0xffffffffu : 0u);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < HPH * HPW * CHP) {
                const int hp = i / CHP, c = i - hp * CHP;
                const int hy = hp / HPW, hx = hp - hy * HPW;
                *reinterpret_cast<u32x4*>(lds + hp * (CIN * 2) + ((c ^ hkey(hy, hx)) << 4)) = hv[it];
            }
        }
    }

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[j][r] = 0.f;

    const int fk = lane >> 5;
    int brow[NT], bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + (lane & 31);
        brow[j] = n / TW;
        bcol[j] = n - brow[j] * TW;
    }
    HP_STAMP();
    lds_barrier(); // the halo tile is complete; the only barrier before the K-halves meet
    HP_STAMP();

    // one tap: NS k16 steps of 6 MFMAs; the B fragments of step ks+1 are read while step ks multiplies, and the A
    // fragments of tap + 2 are requested as soon as this tap's are consumed
    // per-lane constants of the B-fragment addresses: pixel offset and swizzle key at tap (0,0), this lane's chunk
    int hpo0[NT], key0[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        hpo0[j] = (brow[j] * HPW + bcol[j]) * (CIN * 2);
        key0[j] = brow[j] * TW + bcol[j];
    }
    const int cb16 = ((kg * NS) * 2 + fk) << 4;
    // (B fragments are read one whole k16 step = NT MFMAs ahead of their use, across tap boundaries too, and the order is pinned:
    // hipcc otherwise sinks every ds_read to just before its MFMA and the wavefront eats one LDS latency per MFMA)
    auto tap_geom = [&](int tap, int (&base)[NT], int (&k16)[NT]) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = (ky * HPW + kx) * (CIN * 2), tkey = ky * TW + kx; // uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            base[j] = hpo0[j] + toff;
            k16[j] = (CHP == 16 ? ((key0[j] + tkey) & 15) : (((key0[j] + tkey) >> 1) & 7)) << 4;
        }
    };
    half8 fb[2][NT];
    {
        int base[NT], k16[NT];
        tap_geom(0, base, k16);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[0][j] = *reinterpret_cast<const half8*>(lds + base[j] + (cb16 ^ k16[j]));
    }
#define HP_TAP(A, TAP)                                                                                            \
    {                                                                                                             \
        int base_[NT], k16_[NT];                                                                                  \
        tap_geom((TAP), base_, k16_);                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < NS; ++ks)                                                         \
        {                                                                                                         \
            if (ks + 1 < NS) {                                                                                    \
                _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(lds + base_[j] + ((cb16 + (ks + 1) * 32) ^ k16_[j])); \
            } else {                                                                                              \
                int basen_[NT], k16n_[NT];                                                                        \
                tap_geom(min((TAP) + 1, 8), basen_, k16n_);                                                       \
                _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(lds + basen_[j] + (cb16 ^ k16n_[j]));    \
            }                                                                                                     \
            half8 fa;                                                                                             \
            __builtin_memcpy(&fa, &A[ks], 16);                                                                    \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 1][j], acc[j], 0, 0, 0);              \
            A[ks] = *reinterpret_cast<const u32x4*>(wfrag + (long)min((TAP) + 2, 8) * tap_stride + (size_t)ks * 512); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
            {                                                                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
            }                                                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                    \
        }                                                                                                         \
    }
#pragma unroll 1
    for (int tap = 0; tap < 8; tap += 2) {
        HP_TAP(a0, tap);
        HP_STAMP();
        HP_TAP(a1, tap + 1);
        HP_STAMP();
    }
    HP_TAP(a0, 8);
    HP_STAMP();
#undef HP_TAP

    // ---- the K-halves meet: wave (wm, 0) finishes column tiles 0 .. K0-1, wave (wm, 1) tiles K0 .. NT-1; each parks the
    // tiles the other one finishes (slot j of a wave's parking area = the j-th tile of its partner)
    __syncthreads(); // every wave is done with the halo tile
    HP_STAMP();
    float4* const park = reinterpret_cast<float4*>(lds) + (size_t)wave * (K0 * 4 * 64) + lane;
    const float4* const take = reinterpret_cast<const float4*>(lds) + (size_t)(wave ^ 2) * (K0 * 4 * 64) + lane;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        constexpr int dummy = 0;
        (void)dummy;
        const floatx16& give = kg ? acc[j] : acc[K0 + j < NT ? K0 + j : NT - 1];
        if (kg || j < K1) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                park[(j * 4 + g4) * 64] = make_float4(give[4 * g4], give[4 * g4 + 1], give[4 * g4 + 2], give[4 * g4 + 3]);
        }
    }
    __syncthreads();
    floatx16 mine[1][K0];
    int pb[K0], py[K0], px[K0];
    bool pv[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        const int jt = K0 + j < NT ? K0 + j : NT - 1; // (kg = 1 has no K0-th tile when NT is odd: masked below)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 o = take[(j * 4 + g4) * 64];
            const floatx16& keep = kg ? acc[jt] : acc[j];
            mine[0][j][4 * g4] = keep[4 * g4] + o.x, mine[0][j][4 * g4 + 1] = keep[4 * g4 + 1] + o.y;
            mine[0][j][4 * g4 + 2] = keep[4 * g4 + 2] + o.z, mine[0][j][4 * g4 + 3] = keep[4 * g4 + 3] + o.w;
        }
        pb[j] = b;
        py[j] = y0 + (kg ? brow[jt] : brow[j]);
        px[j] = x0 + (kg ? bcol[jt] : bcol[j]);
        pv[j] = py[j] < p.OH && px[j] < p.OW && (!kg || j < K1);
    }
    HP_STAMP();
#undef HP_STAMP
    // the slabs live behind the parking area: no wave can still be reading what another overwrites
    conv_epilogue_staged<1, K0>(p, mine, m0 + wm * 32, lane, lds + RED_BYTES + wave * stage_geom<1>::SLAB, pb, py, px, pv);
}

// ---------------------------------------------------------------------------------------------------
// The same barrier-free design for ANY square kernel (KS = 3, 5, 7: the 7x7 stage convolutions are 68 % of OpenPose-VGG19's FLOPs)
// and ANY input width that is a multiple of CK channels, with 128 output channels per block:
//   * 8 wavefronts = 4 (32-row tiles of the 128 output channels) x 2 (halves of every tap's CK channels); two wavefronts per SIMD,
//     so one's LDS / L2 round trips sit under the other's MFMAs.  Per wavefront and k16 step: ONE A fragment (coalesced 1 KB
//     straight from L2, fragment order, requested two taps ahead) feeds SIX MFMAs whose B fragments come from the halo tile.
//   * the input is consumed in chunks of CK channels; a chunk's halo tile ((TH + KS - 1) x (12 + KS - 1) pixels x CK) is staged in
//     LDS once and serves all KS*KS taps (49 at 7x7: 2.6 halo pixels loaded per output pixel, against 49 in an im2col GEMM).
//     NBUF = 2: the next chunk is requested into registers when a chunk starts and written to the other LDS buffer when it
//     ends - one LDS-only barrier per chunk, no global round trip on the critical path; NBUF = 1 (a 7x7 tile of 128 channels
//     fills 101 KB): single chunk only.
//   * weights never touch LDS; nothing synchronises the wavefronts inside a chunk; the K-halves meet once through LDS.
template <int KS, int CK, int TH, int NBUF>
struct direct_geom {
    static constexpr int TW = 12, HPH = TH + KS - 1, HPW = TW + KS - 1, NT = TH * TW / 32;
    static constexpr int K0 = (NT + 1) / 2;
    static constexpr int HALO_BYTES = HPH * HPW * CK * 2;
    static constexpr int RED_BYTES = 8 * K0 * 16 * 64 * 4; // each wave parks up to K0 accumulator tiles
    static constexpr int EPI_BYTES = 8 * stage_geom<1>::SLAB;
    static constexpr int LDS_BYTES = NBUF * HALO_BYTES > RED_BYTES + EPI_BYTES ? NBUF * HALO_BYTES : RED_BYTES + EPI_BYTES;
};

// one block's work on the TH x 12 pixel tile at (b, y0, x0); TH = 16 is the full tile, TH = 8 serves tile rows of which at most 8 rows
// exist (the last tile row of a 54-row map has 6): half the MFMAs and halo rows instead of multiplying rows that are thrown away
// chunk_base / nchunks: the channel chunks this block multiplies (split-K: a slice of them - the partial sums then go to p.splitk
// instead of through the epilogue, slot = this block's index among the (tile, output-channel group) pairs)
template <int KS, int CK, int TH, int NBUF>
__device__ __forceinline__ void conv_direct_body(const conv_params& p, unsigned char* lds, int b, int y0, int x0, int nchunks, int chunk_base = 0,
    bool park_partial = false)
{
    using G = direct_geom<KS, CK, TH, NBUF>;
    constexpr int TW = 12, HPH = G::HPH, HPW = G::HPW, NT = G::NT, PAD = KS / 2, TAPS = KS * KS;
    constexpr int K0 = (NT + 1) / 2, K1 = NT / 2;
    constexpr int CHP = CK / 8;   // 16-byte chunks per halo pixel
    constexpr int KQC = CK / 16;  // k16 steps per tap and chunk
    constexpr int NS = KQC / 2;   // ... per tap, chunk and K-half
    constexpr int HALO_BYTES = G::HALO_BYTES, RED_BYTES = G::RED_BYTES;
    constexpr int NIT = (HPH * HPW * CHP + 511) / 512;
    static_assert(HPW % 2 == 0 && (CHP == 16 || CHP == 8) && NT * 32 == TH * TW, "tile geometry");
    auto hkey = [](int hy, int hx) { return CHP == 16 ? ((hy * TW + hx) & 15) : (((hy * TW + hx) >> 1) & 7); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, kg = wave >> 2;
    const int m0 = blockIdx.y * 128;
    const int KQ = p.Cin / 16; // k16 steps per tap over all chunks
    const int total = nchunks * TAPS;
    int dbg_i = 0; // tools/direct_timeline.hip: s_memtime stamps of block (0, 0) / wave 0 and of the last wave
#define HP_DSTAMP()                                                                                               \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 448))                                  \
        p.dbg[(tid ? 32 : 0) + (dbg_i++)] = __builtin_amdgcn_s_memtime();
    HP_DSTAMP();

    // ---- A fragments of steps 0 and 1 (step q = chunk * TAPS + tap): 2 x NS loads in flight from the start
    const long tap_stride = (long)(p.Cout_pad / 32) * KQ * 512; // halves per tap
    const __half* wfrag = p.w + ((size_t)((m0 / 32 + wm) * KQ + kg * NS) * 64 + lane) * 8;
    auto a_off = [&](int q) -> long { // element offset of step q's first A fragment of this wave
        const int ch = q / TAPS, tp = q - ch * TAPS;
        return (long)tp * tap_stride + (long)(chunk_base + ch) * (KQC * 512);
    };
    u32x4 a0[NS], a1[NS];
    {
        const long o0 = a_off(0), o1 = a_off(min(1, total - 1));
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            a0[ks] = *reinterpret_cast<const u32x4*>(wfrag + o0 + (size_t)ks * 512);
            a1[ks] = *reinterpret_cast<const u32x4*>(wfrag + o1 + (size_t)ks * 512);
        }
    }

    // ---- halo tile of one chunk: global -> registers (one L2 round trip per pass of <= 8 loads per thread; one pass of 13 is slower: 9.0 k instead of 8.5 k cycles) -> LDS
    constexpr int NPASS = NBUF == 2 ? 1 : (NIT + 7) / 8, PIT = (NIT + NPASS - 1) / NPASS;
    static_assert(NBUF == 1 || NIT <= 8, "the prefetched chunk lives in registers across a whole chunk");
    u32x4 hv[PIT];
    auto halo_load = [&](int chunk, int pass) {
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = tid + (pass * PIT + it) * 512;
            const int hp = min(i, HPH * HPW * CHP - 1) / CHP, c = i % CHP;
            const int hy = hp / HPW, hx = hp - hy * HPW;
            const int y = y0 + hy - PAD, x = x0 + hx - PAD;
            // y, x >= -PAD always; rows / columns up to H + PAD - 1 lie in the zero halo of the HBM tensor, beyond that clamp + zero
            const bool ok = y < p.H + PAD && x < p.W + PAD;
            const u32x4 v = *reinterpret_cast<const u32x4*>(
                p.in.p + tv_off(p.in, b, min(y, p.H + PAD - 1), min(x, p.W + PAD - 1)) + (chunk_base + chunk) * CK + c * 8);
            hv[it] = v & (ok ? 0xffffffffu : 0u);
        }
    };
    auto halo_store = [&](int buf, int pass) {
        unsigned char* dst = lds + buf * HALO_BYTES;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = tid + (pass * PIT + it) * 512;
            if (i < HPH * HPW * CHP) {
                const int hp = i / CHP, c = i - hp * CHP;
                const int hy = hp / HPW, hx = hp - hy * HPW;
                *reinterpret_cast<u32x4*>(dst + hp * (CK * 2) + ((c ^ hkey(hy, hx)) << 4)) = hv[it];
            }
        }
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        halo_load(0, pass);
        halo_store(0, pass);
    }
    if (NBUF == 2 && nchunks > 1)
        halo_load(1, 0);

    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[j][r] = 0.f;

    const int fk = lane >> 5;
    int hpo0[NT], key0[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + (lane & 31);
        const int br = n / TW, bc = n - br * TW;
        hpo0[j] = (br * HPW + bc) * (CK * 2);
        key0[j] = br * TW + bc;
    }
    HP_DSTAMP();
    lds_barrier(); // chunk 0 is complete
    HP_DSTAMP();
    const int cb16 = ((kg * NS) * 2 + fk) << 4;

    // one step = one tap of one chunk: NS k16 steps of NT MFMAs.  The B fragments are read ONE WHOLE k16 STEP (NT MFMAs = 192
    // matrix-pipe cycles) before their use - across the step boundary too: the last k16 step of a step reads the first fragments
    // of the next one - and the A fragments of step q + 2 are requested as soon as this step's are consumed.  The
    // sched_group_barriers pin that order (hipcc otherwise sinks every read to just before its MFMA, one LDS latency each).
    auto step_geom = [&](int q, const unsigned char*& hb, int (&base)[NT], int (&k16)[NT]) {
        const int ch = q / TAPS, tap = q - ch * TAPS;
        hb = lds + (NBUF == 2 ? (ch & 1) * HALO_BYTES : 0);
        const int ky = tap / KS, kx = tap - ky * KS;
        const int toff = (ky * HPW + kx) * (CK * 2), tkey = ky * TW + kx; // uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            base[j] = hpo0[j] + toff;
            k16[j] = (CHP == 16 ? ((key0[j] + tkey) & 15) : (((key0[j] + tkey) >> 1) & 7)) << 4;
        }
    };
    half8 fb[2][NT];
    {
        const unsigned char* hb;
        int base[NT], k16[NT];
        step_geom(0, hb, base, k16);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[0][j] = *reinterpret_cast<const half8*>(hb + base[j] + (cb16 ^ k16[j]));
    }
#define HP_STEP(A, Q)                                                                                             \
    {                                                                                                             \
        const int q_ = (Q);                                                                                       \
        const int ch_ = q_ / TAPS, tap_ = q_ - ch_ * TAPS;                                                        \
        const unsigned char* hb_;                                                                                 \
        int base_[NT], k16_[NT];                                                                                  \
        step_geom(q_, hb_, base_, k16_);                                                                          \
        if (NBUF == 2 && tap_ == 0 && ch_ > 0) { /* uniform: chunk boundary (NBUF == 1 is only ever launched with one chunk) */ \
            halo_store(ch_ & 1, 0); /* its last readers passed the previous boundary's barrier */                 \
            lds_barrier();                                                                                        \
            if (ch_ + 1 < nchunks)                                                                                \
                halo_load(ch_ + 1, 0);                                                                            \
            /* what the previous step pre-read from this buffer was the chunk before last: read again */          \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
                fb[0][j] = *reinterpret_cast<const half8*>(hb_ + base_[j] + (cb16 ^ k16_[j]));                    \
        }                                                                                                         \
        const long nxt_ = a_off(min(q_ + 2, total - 1));                                                          \
        _Pragma("unroll") for (int ks = 0; ks < NS; ++ks)                                                         \
        {                                                                                                         \
            if (ks + 1 < NS) {                                                                                    \
                _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(hb_ + base_[j] + ((cb16 + (ks + 1) * 32) ^ k16_[j])); \
            } else {                                                                                              \
                const unsigned char* hbn_;                                                                        \
                int basen_[NT], k16n_[NT];                                                                        \
                step_geom(min(q_ + 1, total - 1), hbn_, basen_, k16n_);                                           \
                _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                    fb[0][j] = *reinterpret_cast<const half8*>(hbn_ + basen_[j] + (cb16 ^ k16n_[j]));             \
            }                                                                                                     \
            half8 fa;                                                                                             \
            __builtin_memcpy(&fa, &A[ks], 16);                                                                    \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 1][j], acc[j], 0, 0, 0);              \
            A[ks] = *reinterpret_cast<const u32x4*>(wfrag + nxt_ + (size_t)ks * 512);                             \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                        \
            {                                                                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
            }                                                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                    \
        }                                                                                                         \
    }
#pragma unroll 1
    for (int q = 0; q + 1 < total; q += 2) {
        HP_STEP(a0, q);
        HP_STEP(a1, q + 1);
        if (p.dbg && q % 12 == 10)
            HP_DSTAMP();
    }
    if (total & 1)
        HP_STEP(a0, total - 1);
    HP_DSTAMP();
#undef HP_STEP

    // ---- the K-halves meet: wave (wm, 0) finishes column tiles 0 .. K0-1, wave (wm, 1) tiles K0 .. NT-1; each parks the
    // tiles the other one finishes (slot j of a wave's parking area = the j-th tile of its partner)
    __syncthreads(); // every wave is done with the halo tiles
    float4* const park = reinterpret_cast<float4*>(lds) + (size_t)wave * (K0 * 4 * 64) + lane;
    const float4* const take = reinterpret_cast<const float4*>(lds) + (size_t)(wave ^ 4) * (K0 * 4 * 64) + lane;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        const floatx16& give = kg ? acc[j] : acc[K0 + j < NT ? K0 + j : NT - 1];
        if (kg || j < K1) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                park[(j * 4 + g4) * 64] = make_float4(give[4 * g4], give[4 * g4 + 1], give[4 * g4 + 2], give[4 * g4 + 3]);
        }
    }
    __syncthreads();
    floatx16 mine[1][K0];
    int pb[K0], py[K0], px[K0];
    bool pv[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        const int jt = K0 + j < NT ? K0 + j : NT - 1; // (kg = 1 has no K0-th tile when NT is odd: masked below)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 o = take[(j * 4 + g4) * 64];
            const floatx16& keep = kg ? acc[jt] : acc[j];
            mine[0][j][4 * g4] = keep[4 * g4] + o.x, mine[0][j][4 * g4 + 1] = keep[4 * g4 + 1] + o.y;
            mine[0][j][4 * g4 + 2] = keep[4 * g4 + 2] + o.z, mine[0][j][4 * g4 + 3] = keep[4 * g4 + 3] + o.w;
        }
        const int n = (kg ? jt : j) * 32 + (lane & 31);
        const int br = n / TW, bc = n - br * TW;
        pb[j] = b;
        py[j] = y0 + br;
        px[j] = x0 + bc;
        pv[j] = py[j] < p.OH && px[j] < p.OW && (!kg || j < K1);
    }
    HP_DSTAMP();
    if (park_partial) { // uniform: split-K - this block's sums wait in HBM for conv_direct_finish_kernel, in this lane layout
        const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x, nslots = (size_t)gridDim.x * gridDim.y;
        float4* const dst = reinterpret_cast<float4*>(p.splitk) + ((blockIdx.z * nslots + slot) * 8 + wave) * (K0 * 4 * 64) + lane;
#pragma unroll
        for (int j = 0; j < K0; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                dst[(j * 4 + g4) * 64] = make_float4(mine[0][j][4 * g4], mine[0][j][4 * g4 + 1], mine[0][j][4 * g4 + 2], mine[0][j][4 * g4 + 3]);
        return;
    }
    // the slabs live behind the parking area: no wave can still be reading what another overwrites
    conv_epilogue_staged<1, K0>(p, mine, m0 + wm * 32, lane, lds + RED_BYTES + wave * stage_geom<1>::SLAB, pb, py, px, pv);
    HP_DSTAMP();
#undef HP_DSTAMP
}

// The single-chunk form (7 x 7 / 5 x 5 x 128: OpenPose-VGG19's stage convolutions) with the wavefronts cut the other way: wavefront =
// (pair of 32-row tiles rp, half of the six pixel tiles ph, K-half kg) - 2 x 3 MFMA tiles instead of 1 x 6.  Per k16 step it reads THREE
// B fragments from LDS and two A fragments from L2 for its six MFMAs; the 1 x 6 cut reads six from LDS - 49 KB per step and CU, as much as
// the LDS delivers in the 384 cycles the step's MFMAs take.  Measured (round 4, VGG19's 7 x 7 128 -> 128 at 16 x 54 x 96): 125.9 -> 123.6 us
// per launch - halving the LDS reads buys 2 %, so the 1 x 6 cut was NOT bound by them (0.43 of the MFMA peak either way; what the loop
// waits for is still open: DESIGN.md section 7.5).  Kept for the 2 %.
// Same K order per output as conv_direct_body (taps ascending, this K-half's k16 steps ascending, the two halves added once): same bits.
template <int KS, int CK>
__device__ __forceinline__ void conv_direct_body_b(const conv_params& p, unsigned char* lds, int b, int y0, int x0)
{
    using G = direct_geom<KS, CK, 16, 1>;
    constexpr int TW = 12, HPH = G::HPH, HPW = G::HPW, NT = G::NT, TM = 2, TN = NT / 2, PAD = KS / 2, TAPS = KS * KS;
    constexpr int CHP = CK / 8, KQC = CK / 16, NS = KQC / 2;
    constexpr int HALO_BYTES = G::HALO_BYTES, RED_BYTES = G::RED_BYTES;
    constexpr int NIT = (HPH * HPW * CHP + 511) / 512, NPASS = (NIT + 7) / 8, PIT = (NIT + NPASS - 1) / NPASS;
    static_assert(NT == 6 && G::K0 == TN && HPW % 2 == 0 && (CHP == 16 || CHP == 8), "tile geometry");
    auto hkey = [](int hy, int hx) { return CHP == 16 ? ((hy * TW + hx) & 15) : (((hy * TW + hx) >> 1) & 7); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rp = wave & 1, ph = (wave >> 1) & 1, kg = wave >> 2;
    const int m0 = blockIdx.y * 128;
    const int KQ = p.Cin / 16;
    int dbg_i = 0; // tools/direct_timeline.hip: s_memtime stamps of block (0, 0) / wave 0 and of the last wave
#define HP_DSTAMP()                                                                                               \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 448))                                  \
        p.dbg[(tid ? 32 : 0) + (dbg_i++)] = __builtin_amdgcn_s_memtime();
    HP_DSTAMP();
    const long tap_stride = (long)(p.Cout_pad / 32) * KQ * 512, row_stride = (long)KQ * 512; // halves per tap / per 32-row tile
    const __half* wfrag = p.w + ((size_t)((m0 / 32 + rp * TM) * KQ + kg * NS) * 64 + lane) * 8;
    u32x4 a0[TM][NS], a1[TM][NS];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            a0[i][ks] = *reinterpret_cast<const u32x4*>(wfrag + i * row_stride + (size_t)ks * 512);
            a1[i][ks] = *reinterpret_cast<const u32x4*>(wfrag + tap_stride + i * row_stride + (size_t)ks * 512);
        }
    {   // halo tile: global -> registers (passes of <= 8 loads per thread) -> LDS
        u32x4 hv[PIT];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = tid + (pass * PIT + it) * 512;
                const int hp = min(i, HPH * HPW * CHP - 1) / CHP, c = i % CHP;
                const int hy = hp / HPW, hx = hp - hy * HPW;
                const int y = y0 + hy - PAD, x = x0 + hx - PAD;
                const bool ok = y < p.H + PAD && x < p.W + PAD; // (y, x >= -PAD: inside the zero halo of the HBM tensor)
                const u32x4 v = *reinterpret_cast<const u32x4*>(p.in.p + tv_off(p.in, b, min(y, p.H + PAD - 1), min(x, p.W + PAD - 1)) + c * 8);
                hv[it] = v & (ok ? 0xffffffffu : 0u);
            }
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int i = tid + (pass * PIT + it) * 512;
                if (i < HPH * HPW * CHP) {
                    const int hp = i / CHP, c = i - hp * CHP;
                    const int hy = hp / HPW, hx = hp - hy * HPW;
                    *reinterpret_cast<u32x4*>(lds + hp * (CK * 2) + ((c ^ hkey(hy, hx)) << 4)) = hv[it];
                }
            }
        }
    }
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    const int fk = lane >> 5;
    int hpo0[TN], key0[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = (ph * TN + j) * 32 + (lane & 31);
        const int br = n / TW, bc = n - br * TW;
        hpo0[j] = (br * HPW + bc) * (CK * 2);
        key0[j] = br * TW + bc;
    }
    HP_DSTAMP();
    lds_barrier(); // the halo tile is complete
    HP_DSTAMP();
    const int cb16 = ((kg * NS) * 2 + fk) << 4;
    auto step_geom = [&](int tap, int (&base)[TN], int (&k16)[TN]) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const int toff = (ky * HPW + kx) * (CK * 2), tkey = ky * TW + kx; // uniform
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            base[j] = hpo0[j] + toff;
            k16[j] = (CHP == 16 ? ((key0[j] + tkey) & 15) : (((key0[j] + tkey) >> 1) & 7)) << 4;
        }
    };
    half8 fb[2][TN];
    {
        int base[TN], k16[TN];
        step_geom(0, base, k16);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[0][j] = *reinterpret_cast<const half8*>(lds + base[j] + (cb16 ^ k16[j]));
    }
    // one step = one tap: NS k16 steps of TM x TN MFMAs; B fragments one whole k16 step ahead (across the tap boundary too), the A
    // fragments of tap q + 2 requested as this tap's are consumed
#define HP_STEPB(A, Q)                                                                                            \
    {                                                                                                             \
        const int q_ = (Q);                                                                                       \
        int base_[TN], k16_[TN];                                                                                  \
        step_geom(q_, base_, k16_);                                                                               \
        const long nxt_ = (long)min(q_ + 2, TAPS - 1) * tap_stride;                                               \
        _Pragma("unroll") for (int ks = 0; ks < NS; ++ks)                                                         \
        {                                                                                                         \
            if (ks + 1 < NS) {                                                                                    \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                    \
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(lds + base_[j] + ((cb16 + (ks + 1) * 32) ^ k16_[j])); \
            } else {                                                                                              \
                int basen_[TN], k16n_[TN];                                                                        \
                step_geom(min(q_ + 1, TAPS - 1), basen_, k16n_);                                                  \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                    \
                    fb[0][j] = *reinterpret_cast<const half8*>(lds + basen_[j] + (cb16 ^ k16n_[j]));              \
            }                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                        \
            {                                                                                                     \
                half8 fa;                                                                                         \
                __builtin_memcpy(&fa, &A[i][ks], 16);                                                             \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                    \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 1][j], acc[i][j], 0, 0, 0);    \
                A[i][ks] = *reinterpret_cast<const u32x4*>(wfrag + nxt_ + i * row_stride + (size_t)ks * 512);     \
            }                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
            {                                                                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
            }                                                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                    \
        }                                                                                                         \
    }
#pragma unroll 1
    for (int q = 0; q + 1 < TAPS; q += 2) {
        HP_STEPB(a0, q);
        HP_STEPB(a1, q + 1);
    }
    if (TAPS & 1)
        HP_STEPB(a0, TAPS - 1);
#undef HP_STEPB
    HP_DSTAMP();

    // ---- the K-halves meet: K-half kg finishes row tile kg of the pair and parks the other one's three tiles for its partner
    __syncthreads(); // every wave is done with the halo tile
    float4* const park = reinterpret_cast<float4*>(lds) + (size_t)wave * (TN * 4 * 64) + lane;
    const float4* const take = reinterpret_cast<const float4*>(lds) + (size_t)(wave ^ 4) * (TN * 4 * 64) + lane;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const floatx16& give = kg ? acc[0][j] : acc[1][j];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            park[(j * 4 + g4) * 64] = make_float4(give[4 * g4], give[4 * g4 + 1], give[4 * g4 + 2], give[4 * g4 + 3]);
    }
    __syncthreads();
    HP_DSTAMP();
    floatx16 mine[1][TN];
    int pb[TN], py[TN], px[TN];
    bool pv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const floatx16& keep = kg ? acc[1][j] : acc[0][j];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 o = take[(j * 4 + g4) * 64];
            mine[0][j][4 * g4] = keep[4 * g4] + o.x, mine[0][j][4 * g4 + 1] = keep[4 * g4 + 1] + o.y;
            mine[0][j][4 * g4 + 2] = keep[4 * g4 + 2] + o.z, mine[0][j][4 * g4 + 3] = keep[4 * g4 + 3] + o.w;
        }
        const int n = (ph * TN + j) * 32 + (lane & 31);
        const int br = n / TW, bc = n - br * TW;
        pb[j] = b, py[j] = y0 + br, px[j] = x0 + bc;
        pv[j] = py[j] < p.OH && px[j] < p.OW;
    }
    // the slabs live behind the parking area: no wave can still be reading what another overwrites
    conv_epilogue_staged<1, TN>(p, mine, m0 + (rp * TM + kg) * 32, lane, lds + RED_BYTES + wave * stage_geom<1>::SLAB, pb, py, px, pv);
    HP_DSTAMP();
#undef HP_DSTAMP
}

template <int KS, int CK, int NBUF>
__global__ __launch_bounds__(512) void conv_direct_kernel(const conv_params p, int tiles_x, int tiles_y, int nchunks)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[direct_geom<KS, CK, 16, NBUF>::LDS_BYTES];
    // block -> tile: the tiles of the last tile row come LAST in dispatch order - when that row is a half tile they are the short
    // blocks, and short jobs at the end even out the CUs' finishing times
    int tx, ty, b;
    {
        const int t = blockIdx.x, per_full = tiles_x * (tiles_y - 1), full = per_full * p.B;
        if (t < full) {
            b = t / per_full;
            const int r = t - b * per_full;
            ty = r / tiles_x, tx = r - ty * tiles_x;
        } else {
            const int r = t - full;
            b = r / tiles_x, tx = r - b * tiles_x, ty = tiles_y - 1;
        }
    }
    const int y0 = ty * 16, x0 = tx * 12;
    // (the two-path form of the chunk-pipelined kernels needs more than 256 registers: they always take the full tile)
    if (NBUF == 1 && p.OH - y0 <= 8) // uniform
        conv_direct_body<KS, CK, 8, NBUF>(p, lds, b, y0, x0, nchunks);
    else if constexpr (NBUF == 1)
        conv_direct_body_b<KS, CK>(p, lds, b, y0, x0);
    else if (NBUF == 2 && gridDim.z > 1) // split-K: gridDim.z blocks per tile, nchunks / gridDim.z chunks each
        conv_direct_body<KS, CK, 16, NBUF>(p, lds, b, y0, x0, nchunks / gridDim.z, blockIdx.z * (nchunks / gridDim.z), true);
    else
        conv_direct_body<KS, CK, 16, NBUF>(p, lds, b, y0, x0, nchunks);
}

// (Round 4 paired the k-th 7x7 of the conf and of the paf branch of an OpenPose stage in ONE launch - 768 full + 256 half tiles: 3.5 rounds
// for two layers instead of 2 x 2 - to recover the 1.75-of-2 round quantisation of these layers.  Measured: 240 - 249 us per pair against
// 2 x 125 us, 1 331 vs 1 340 frames/s end to end: the CUs that idle in the half-empty second round are not lost time, the busy ones run
// that much faster (clock / power and L2 headroom).  Removed; profiles/r04_layer_times_config2_paired_branches.txt)

// split-K, second launch: the same grid without z; every wavefront adds the ksplit partial sums of the tiles it finishes (parked in its
// own lane layout) and runs the kernel's epilogue.  A kernel boundary between the two makes the sums visible across the XCDs' L2s.
template <int KS, int CK>
__global__ __launch_bounds__(512) void conv_direct_finish_kernel(const conv_params p, int tiles_x, int tiles_y)
{
    using G = direct_geom<KS, CK, 16, 2>;
    constexpr int TW = 12, NT = G::NT, K0 = (NT + 1) / 2, K1 = NT / 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * stage_geom<1>::SLAB];
    int tx, ty, b;
    {
        const int t = blockIdx.x, per_full = tiles_x * (tiles_y - 1), full = per_full * p.B;
        if (t < full) {
            b = t / per_full;
            const int r = t - b * per_full;
            ty = r / tiles_x, tx = r - ty * tiles_x;
        } else {
            const int r = t - full;
            b = r / tiles_x, tx = r - b * tiles_x, ty = tiles_y - 1;
        }
    }
    const int y0 = ty * 16, x0 = tx * 12;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, kg = wave >> 2;
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x, nslots = (size_t)gridDim.x * gridDim.y;
    floatx16 mine[1][K0];
    int pb[K0], py[K0], px[K0];
    bool pv[K0];
#pragma unroll
    for (int j = 0; j < K0; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mine[0][j][r] = 0.f;
        for (int z = 0; z < p.ksplit; ++z) { // (ascending K: the order does not depend on which block finished first)
            const float4* src = reinterpret_cast<const float4*>(p.splitk) + ((z * nslots + slot) * 8 + wave) * (K0 * 4 * 64) + lane;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 o = src[(j * 4 + g4) * 64];
                mine[0][j][4 * g4] += o.x, mine[0][j][4 * g4 + 1] += o.y, mine[0][j][4 * g4 + 2] += o.z, mine[0][j][4 * g4 + 3] += o.w;
            }
        }
        const int jt = K0 + j < NT ? K0 + j : NT - 1;
        const int n = (kg ? jt : j) * 32 + (lane & 31);
        const int br = n / TW, bc = n - br * TW;
        pb[j] = b, py[j] = y0 + br, px[j] = x0 + bc;
        pv[j] = py[j] < p.OH && px[j] < p.OW && (!kg || j < K1);
    }
    conv_epilogue_staged<1, K0>(p, mine, blockIdx.y * 128 + wm * 32, lane, lds + wave * stage_geom<1>::SLAB, pb, py, px, pv);
}

// fast epilogue (aligned fp16 NHWC vectors) when every 8-channel chunk is whole and 16-byte aligned
static bool fast_epilogue(const conv_params& p)
{
    return p.out.p && !p.out_f32 && p.Cout % 8 == 0 && p.out.coff % 8 == 0 && p.out.cs % 8 == 0
        && (!p.res.p || (p.res.coff % 8 == 0 && p.res.cs % 8 == 0));
}

static bool use_halo(const conv_params& p);
static bool use_small1x1(const conv_params& p);
static int big1x1_variant(const conv_params& p);
static bool fast_epilogue(const conv_params& p);
// conv_direct_kernel (8 wavefronts, 128 output channels x 16x12 pixels per block, any square kernel / chunked Cin) serves this layer:
// 0 = no, otherwise the channel chunk CK (128 or 64).
static int use_gdirect(const conv_params& p)
{
    if (p.KH != p.KW || (p.KH != 3 && p.KH != 5 && p.KH != 7) || p.stride != 1 || p.dil != 1 || p.pad_t != p.KH / 2
        || p.pad_l != p.KH / 2 || p.OH != p.H || p.OW != p.W || p.Cin % 64 || p.Cout_pad % 128 || p.in.coff % 8 || !fast_epilogue(p))
        return 0;
    if (p.KH == 3 && p.Cin <= 128)
        return 0; // (these stay with conv3x3_direct_kernel, whose half-size blocks share a CU at batch 8)
    // maps smaller than two tiles: the generic implicit GEMM packs pixels of several images into one tile - worth more than the halo
    // re-use unless K is long (measured at 12 x 12: 512 -> 512 57 -> 45 us, 2048 -> 512 212 -> 163 us on this kernel)
    if ((long)p.OH * p.OW < 256 && p.Cin < 256)
        return 0;
    // 3x3 on maps the 16 x 12 tiles cover badly (49 x 49: 20 tiles for 12.5 tiles of pixels, 25 x 25: 6 for 3.3): the generic kernel has no
    // tiles to round up to (measured at batch 64: 256 channels at 49 x 49 253 -> 223 us, 512 channels at 25 x 25 276 -> 220 us)
    if (p.KH == 3 && (double)p.OH * p.OW < 0.68 * ((p.OH + 15) / 16 * 16) * ((p.OW + 11) / 12 * 12))
        return 0;
    // 128-channel chunks only where ONE chunk is the whole input (7x7 / 5x5 x 128: a 101 / 82 KB tile, single-buffered); everything
    // else runs on double-buffered 64-channel chunks (the 128-channel form of that pipeline needs more than 256 registers)
    // (measured: 7x7 x 128 as two pipelined 64-channel chunks is 10 % slower than as one 128-channel chunk - the chunk barrier waits for
    // the wavefronts that lose the matrix-pipe arbitration)
    return p.Cin == 128 ? 128 : 64;
}
// 1: the weights of this convolution are to be packed in MFMA-fragment order for conv3x3_direct_kernel / conv_direct_kernel
int conv_weight_layout(const conv_params& p)
{
    if (use_small1x1(p) && fast_epilogue(p))
        return 1;
    if (big1x1_variant(p) && fast_epilogue(p))
        return 1;
    if (use_gdirect(p))
        return 1;
    return use_halo(p) && fast_epilogue(p) ? 1 : 0;
}

static bool use_halo(const conv_params& p)
{
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.dil == 1 && p.pad_t == 1 && p.pad_l == 1 && (p.Cin == 128 || p.Cin == 64)
        && p.Cout_pad % 64 == 0 && p.in.coff % 8 == 0;
}

template <int BM, int BN, int BK>
static hipError_t launch_tile(const conv_params& p, hipStream_t s)
{
    dim3 grid(((p.npix + BN - 1) / BN) * (p.Cout_pad / BM));
    if (fast_epilogue(p))
        HP_LAUNCH((conv_mfma_kernel<BM, BN, BK, 0>), grid, dim3(256), 0, s, p);
    else
        HP_LAUNCH((conv_mfma_kernel<BM, BN, BK, 1>), grid, dim3(256), 0, s, p);
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

// ---------------------------------------------------------------------------------------------------
// Small 1x1 convolutions (<= 256 input channels -> <= 128 outputs, stride 1: the entry convolutions of the refinement blocks):
// one block = 64 consecutive pixels, their activations go global -> LDS once (all channels, rows padded by 16 bytes instead of
// swizzled), the weights come from L2 in MFMA-fragment order (w_layout 1), one wavefront per 32-row tile of output channels,
// two column tiles each.  One barrier, < 128 registers, <= 53 KB of LDS: three blocks per CU.  The generic implicit-GEMM
// kernel stages A and B through LDS in K steps with a barrier each; at 0.65 GFLOP per launch that structure is all overhead.
template <int KP>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const conv_params p)
{
    constexpr int NPX = 64, NT = 2, CG = KP / 8, KQ = KP / 16, ROW = KP * 2 + 16, NLD = NPX * CG / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPX * ROW + 4 * stage_geom<1>::SLAB];
    unsigned char* const s_b = lds;
    unsigned char* const s_slab = lds + NPX * ROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * NPX, HW = p.OH * p.OW;
    const int groups = p.Cout_pad / 128; // 128 output channels at a time against the SAME staged pixels (ResNet's 64 -> 256, 128 -> 512 ...)

    u32x4 a[KQ];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks)
        a[ks] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)(wave * KQ + ks) * 64 + lane) * 8);
    {
        u32x4 hv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * 256, pix = i / CG, c = i - pix * CG;
            const int n = min(n0 + pix, p.npix - 1);
            const int b = n / HW, r = n - b * HW, y = r / p.OW, x = r - y * p.OW;
            hv[k] = *reinterpret_cast<const u32x4*>(p.in.p + tv_off(p.in, b, y, x) + c * 8);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * 256, pix = i / CG, c = i - pix * CG;
            *reinterpret_cast<u32x4*>(s_b + pix * ROW + c * 16) = hv[k];
        }
    }
    lds_barrier();
    const int frow = lane & 31, fk = lane >> 5;
    int pb[NT], py[NT], px[NT];
    bool pv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + frow, nc = min(n, p.npix - 1);
        pb[j] = nc / HW;
        const int r = nc - pb[j] * HW;
        py[j] = r / p.OW, px[j] = r - py[j] * p.OW;
        pv[j] = n < p.npix;
    }
#pragma unroll 1
    for (int mg = 0; mg < groups; ++mg) {
        floatx16 acc[1][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[0][j][r] = 0.f;
        const int nxt = min(mg + 1, groups - 1); // the next group's weights are requested as this group's are consumed
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            half8 fa;
            __builtin_memcpy(&fa, &a[ks], 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const half8 fb = *reinterpret_cast<const half8*>(s_b + (j * 32 + frow) * ROW + (ks * 2 + fk) * 16);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0][j], 0, 0, 0);
            }
            a[ks] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)((nxt * 4 + wave) * KQ + ks) * 64 + lane) * 8);
        }
        conv_epilogue_staged<1, NT>(p, acc, mg * 128 + wave * 32, lane, s_slab + wave * stage_geom<1>::SLAB, pb, py, px, pv);
    }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 convolutions with >= 256 input channels (the pointwise halves of the 512-channel MobileNet blocks when they run un-fused, the
// reductions / expansions of the ResNet bottlenecks) as a pixel-block GEMM without any LDS traffic for the weights:
//   * block = 32 NTP consecutive pixels x 128 TM output channels; wavefront w owns TM 32-row tiles (rows (4 by + w) TM 32 ..) over the
//     FULL K: TM x NTP accumulator tiles in registers (up to 256 of the 512 registers one wavefront per SIMD may use), no split-K;
//   * A (weights) straight from L2 in MFMA-fragment order (w_layout 1): one coalesced 1 KB load per fragment, each feeding NTP MFMAs,
//     re-requested one 64-channel chunk ahead as they are consumed;
//   * B (activations) in 64-channel chunks through a double-buffered swizzled LDS tile (global -> registers one chunk ahead -> LDS,
//     ONE LDS-only barrier per chunk); every B fragment read from LDS feeds TM MFMAs;
//   * shared staged epilogue (bias, activation, residual, 64-byte runs per pixel).
// Per k16 step a wavefront issues TM loads, NTP LDS reads and TM x NTP MFMAs: at TM = NTP = 4 that is 8 memory operations for 512
// matrix-pipe cycles, against 1 + 1 per 32 cycles in the 128 x 128 LDS-staged implicit GEMM this replaces for these layers.
template <int TM, int NTP>
__global__ __launch_bounds__(512) void conv1x1_big_kernel(const conv_params p)
{
    constexpr int NPX = 32 * NTP, CK = 64, KS = 4, BUF = NPX * CK * 2;
    constexpr int SLABS = 4 * NTP * stage_geom<TM>::SLAB; // conv_epilogue_wide / _packed: all NTP pixel tiles of a wavefront staged at once
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF > SLABS ? 2 * BUF : SLABS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD-aware: blocks are dealt round-robin to the eight XCDs, whose L2s do not share.  The CG output-channel groups of one
    // pixel tile read the same activations, so they get ids that are congruent mod 8 (same XCD, same L2) and within 8 * CG of each other
    // (in flight together): id = (t / 8) * 8 * CG + cg * 8 + t % 8.  (As a 2-D grid the four groups of ResNet's 256 -> 1024 expansion
    // landed on four XCDs and the input was fetched from HBM four times: 945 MB per launch instead of 709.)
    const int CG = p.Cout_pad / (128 * TM);
    const int within = blockIdx.x % (8 * CG);
    const int ptile = (blockIdx.x / (8 * CG)) * 8 + within % 8, cgrp = within / 8;
    if (ptile * NPX >= p.npix)
        return; // (the grid is padded to a multiple of eight pixel tiles)
    const int n0 = ptile * NPX, HW = p.OH * p.OW;
    const int KQ = p.Cin / 16, NCH = p.Cin / CK;

    if (wave >= 4) {
        // ---- PRODUCER wavefronts (4-7): B chunks global -> registers -> LDS.  The activations come from HBM (~2 us under load)
        // while a chunk's MFMAs last ~0.6 us, so FOUR chunks are kept in flight (ring slot = chunk % 4; the chunk loop is unrolled by
        // four so that the slots are register names).  They run in wavefronts of their own because vmcnt retires in order: in one
        // instruction stream with the weight loads - which are consumed one chunk after their request - every wait for a weight
        // fragment also waited for all older activation loads, i.e. the ring was one chunk deep whatever its size (measured: 22 us
        // for 512 -> 512 at 8 x 46 x 54 pixels with one stream, for 5 us of MFMAs).
        const int pt = tid - 256;
        long hoff[NTP];
#pragma unroll
        for (int k = 0; k < NTP; ++k) {
            const int n = min(n0 + (pt >> 3) + 32 * k, p.npix - 1);
            const int b = n / HW, r = n - b * HW, y = r / p.OW, x = r - y * p.OW;
            hoff[k] = tv_off(p.in, b, y * p.stride, x * p.stride) + (pt & 7) * 8;
        }
        constexpr int PF = 4;
        u32x4 hv[PF][NTP];
        auto hload = [&](u32x4 (&slot)[NTP], int chunk) {
#pragma unroll
            for (int k = 0; k < NTP; ++k)
                slot[k] = *reinterpret_cast<const u32x4*>(p.in.p + hoff[k] + chunk * CK);
        };
        auto to_lds = [&](int buf, const u32x4 (&slot)[NTP]) {
#pragma unroll
            for (int k = 0; k < NTP; ++k)
                *reinterpret_cast<u32x4*>(lds + buf * BUF + lds_off<CK>((pt >> 3) + 32 * k, pt & 7)) = slot[k];
        };
#pragma unroll
        for (int u = 0; u < PF; ++u)
            hload(hv[u], min(u, NCH - 1));
        int pdbg = 0;
#define HP_PSTAMP()                                                                                               \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 256 && pdbg < 30)                                    \
        p.dbg[32 + pdbg++] = __builtin_amdgcn_s_memtime();
        HP_PSTAMP();
        to_lds(0, hv[0]);
        hload(hv[0], min(PF, NCH - 1));
        HP_PSTAMP();
        lds_barrier();
        HP_PSTAMP();
#pragma unroll 1
        for (int c0 = 0; c0 < NCH; c0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                // chunk c + 1 (requested four iterations ago) -> the other buffer (its last readers passed the previous barrier);
                // the slot then takes chunk c + 5
                to_lds((u + 1) & 1, hv[(u + 1) & 3]);
                hload(hv[(u + 1) & 3], min(c0 + u + 1 + PF, NCH - 1));
                HP_PSTAMP();
                lds_barrier();
                HP_PSTAMP();
            }
        }
#undef HP_PSTAMP
        return;
    }

    // ---- CONSUMER wavefronts (0-3): wavefront w owns TM 32-row tiles over the full K
    const int frow = lane & 31, fk = lane >> 5;
    const int m_wave = (cgrp * 4 + wave) * TM * 32;
    const __half* const wbase = p.w + ((size_t)(m_wave / 32) * KQ * 64 + lane) * 8;
    const size_t row_stride = (size_t)KQ * 512;
    u32x4 a[KS][TM];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[ks][i] = *reinterpret_cast<const u32x4*>(wbase + i * row_stride + ks * 512);
    floatx16 acc[TM][NTP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    int dbg_i = 0; // HP_CONV_DBG: s_memtime stamps of block 0: consumer wave 0 at [0..], producer wave 4 at [32..]
#define HP_BSTAMP()                                                                                               \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && dbg_i < 30)                                    \
        p.dbg[(wave >= 4 ? 32 : 0) + dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_BSTAMP();
    lds_barrier();
    HP_BSTAMP();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const unsigned char* const bt = lds + (c & 1) * BUF;
        const __half* const wn = wbase + (size_t)min(c + 1, NCH - 1) * (KS * 512);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 fb[NTP];
#pragma unroll
            for (int j = 0; j < NTP; ++j)
                fb[j] = *reinterpret_cast<const half8*>(bt + lds_off<CK>(j * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                half8 fa;
                __builtin_memcpy(&fa, &a[ks][i], 16);
#pragma unroll
                for (int j = 0; j < NTP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[ks][i] = *reinterpret_cast<const u32x4*>(wn + i * row_stride + ks * 512);
        }
        HP_BSTAMP();
        lds_barrier(); // chunk c + 1 is complete in its buffer; everyone is done reading chunk c
        HP_BSTAMP();
    }

    int pb[NTP], py[NTP], px[NTP];
    bool pv[NTP];
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        const int n = n0 + j * 32 + frow, nc = min(n, p.npix - 1);
        pb[j] = nc / HW;
        const int r = nc - pb[j] * HW;
        py[j] = r / p.OW, px[j] = r - py[j] * p.OW;
        pv[j] = n < p.npix;
    }
    // (the producers are gone and the last barrier of the loop is behind every read of the B buffers: the wave-private slabs may
    // overlay them; the staged epilogue synchronises inside a wavefront only)
    // (a residual parked in LDS by the producers while the MFMAs run - no HBM round trip after the last MFMA - was built and measured:
    // 205 -> 223 us for ResNet's 256 -> 1024 expansion.  Round 4 then built the weights-stationary form of this kernel for that layer - one
    // persistent block per CU, its 256 x 256 weight fragments in registers, 1.2 GB of L2 weight traffic per launch gone - in three variants
    // (staged epilogue in the consumers; ALL global traffic incl. the output stores in the producer wavefronts; shortcut read in 512-byte
    // runs): 231 / 218 / 228 us against 217 - 220 us for this kernel in the same probe.  The layer moves its 709 MB at 3.3 - 3.5 TB/s
    // whatever the block structure: neither the weights, nor the matrix pipe (removing every MFMA: 227 us), nor who issues the stores
    // decide it.  DESIGN.md section 7, "weights-stationary 1 x 1"; the kernels were removed.)
    if (p.res.p) // uniform
        conv_epilogue_wide<TM, NTP>(p, acc, m_wave, lane, lds + wave * (NTP * stage_geom<TM>::SLAB), pb, py, px, pv);
    else
        conv_epilogue_packed<TM, NTP>(p, acc, m_wave, lane, lds + wave * (NTP * stage_geom<TM>::SLAB), pb, py, px, pv);
    HP_BSTAMP();
#undef HP_BSTAMP
}

// which (TM, NTP) the pixel-block GEMM runs a layer with: TM * 1000 + NTP, or 0 when the layer is not its kind
static int big1x1_variant(const conv_params& p)
{
    // (any stride: a strided 1x1 is the same GEMM over every stride-th pixel - the producers gather them; ResNet's projection shortcuts)
    if (p.KH != 1 || p.KW != 1 || p.stride < 1 || p.pad_t || p.pad_l || p.OH != (p.H + p.stride - 1) / p.stride
        || p.OW != (p.W + p.stride - 1) / p.stride || p.Cin % 256 /* four-chunk ring */ || p.Cout_pad % 128 || p.Cout % 8 || p.in.coff % 8
        || p.in.cs - p.in.coff < p.Cin)
        return 0;
    // (TM, NTP) by a small cost model: blocks are dealt to the 256 CUs in rounds (two blocks share a CU when each needs <= 256
    // registers); a round costs its MFMAs at ~80 % pipe efficiency plus ~6 k cycles of prologue / epilogue; 64-pixel blocks (NTP = 2)
    // pull twice the weights per MFMA through the texture path
    // (TM, NTP): measured over the ResNet-50 bottlenecks at 193^2 .. 12^2 pixels and LW-OpenPose's pointwise layers (sweep of all
    // instances, tools/profile_layers.py): 64 pixels x 256 output channels wins or ties almost everywhere - 118 registers and 72 KB
    // of LDS let TWO blocks share a CU, so one block's prologue (first chunk from HBM) and epilogue (stores) sit under the other's
    // MFMAs; wider or taller blocks run alone on their CU and pay both phases in full.  128-row blocks where the output has no
    // 256-row groups.
    // ... except where that grid is barely more than one block per CU (ResNet's reductions on 24 x 24 / 12 x 12 maps at batch 32: 288 / 144
    // blocks): 128-row blocks halve the last, nearly empty round (25.5 -> 21.4 us, 22.8 -> 19.2 us; a higher threshold loses with two streams)
    if (p.Cout_pad % 256 == 0 && (long)((p.npix + 63) / 64) * (p.Cout_pad / 256) < 320)
        return 1002;
    if (p.Cout_pad % 256 == 0)
        return 2002;
    const long blocks4 = (long)((p.npix + 127) / 128) * (p.Cout_pad / 128);
    return blocks4 >= 1024 ? 1004 : 1002;
}

static bool use_small1x1(const conv_params& p)
{
    return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.Cout_pad % 128 == 0 && p.Cout_pad <= 512
        && (p.Cout_pad == 128 || p.Cin <= 128) // (wider outputs only where the layer is HBM-bound: K <= 128)
        && (p.Cin == 64 || p.Cin == 128 || p.Cin == 192 || p.Cin == 256)
        && p.in.coff % 8 == 0 && p.in.cs - p.in.coff >= p.Cin && p.OH == p.H && p.OW == p.W;
}

int conv_mfma_tile(const conv_params& p)
{
    if (p.w_layout == 1 && p.KH == 1 && !use_small1x1(p))
        return 5200000 + big1x1_variant(p); // conv1x1_big_kernel<TM, NTP>
    if (p.w_layout == 1 && p.KH == 1)
        return 5100000 + p.Cin; // conv1x1_small_kernel
    if (p.w_layout == 1 && use_gdirect(p))
        return 6000000 + p.Cin * 1000 + p.KH * p.KW; // conv_direct_kernel
    if (p.w_layout == 1)
        return 5000000 + 64 * 1000 + 192;
    const int BM = (p.Cout_pad % 128 == 0) ? 128 : 64;
    // prefer the 128-pixel tile only when it still fills the 256 CUs at least once
    const long blocks128 = (long)((p.npix + 127) / 128) * (p.Cout_pad / BM);
    const int BN = blocks128 >= 256 ? 128 : 64;
    return BM * 1000 + BN;
}

// split-K for the chunk-pipelined 3x3 instance when its tiles leave CUs idle (configs[3]: 12 x 12 maps at batch 32 = 32 tiles x 4
// output-channel groups = 128 blocks; the 2048 -> 512 head convolution alone is 8 % of that network's conv time)
int conv_splitk(const conv_params& p, size_t* scratch_bytes)
{
    if (scratch_bytes)
        *scratch_bytes = 0;
    if (p.w_layout != 1 || p.KH != 3 || use_gdirect(p) != 64)
        return 1;
    const int nchunks = p.Cin / 64;
    const long blocks = (long)((p.OW + 11) / 12) * ((p.OH + 15) / 16) * p.B * (p.Cout_pad / 128);
    int ks = 1;
    if (blocks <= 64 && nchunks >= 8 && nchunks % 4 == 0)
        ks = 4;
    else if (blocks <= 160 && nchunks >= 4 && nchunks % 2 == 0)
        ks = 2;
    if (ks > 1 && scratch_bytes)
        *scratch_bytes = (size_t)ks * blocks * 8 * 3 * 4 * 64 * sizeof(float4); // [z][slot][wave][K0 = 3][4][64 lanes] float4
    return ks;
}

hipError_t launch_conv_mfma(const conv_params& p, hipStream_t s)
{
    if (p.w_layout == 1 && p.KH == 1 && !use_small1x1(p)) {
        const int v = big1x1_variant(p);
        if (!v || !fast_epilogue(p))
            return hipErrorInvalidValue;
        const int TM = v / 1000, NTP = v % 1000;
        const int ptiles = (p.npix + 32 * NTP - 1) / (32 * NTP);
        const dim3 grid((ptiles + 7) / 8 * 8 * (p.Cout_pad / (128 * TM)));
#define HP_BIG(TM_, NTP_) HP_LAUNCH((conv1x1_big_kernel<TM_, NTP_>), grid, dim3(512), 0, s, p)
        switch (v) { // (the instances big1x1_variant hands out; wider / taller ones were swept and lost: see there)
        case 2002: HP_BIG(2, 2); break;
        case 1004: HP_BIG(1, 4); break;
        default: HP_BIG(1, 2); break;
        }
#undef HP_BIG
        return hipGetLastError();
    }
    if (p.w_layout == 1 && p.KH == 1) {
        if (!(use_small1x1(p) && fast_epilogue(p)))
            return hipErrorInvalidValue;
        const dim3 grid((p.npix + 63) / 64);
        switch (p.Cin) {
        case 64: HP_LAUNCH((conv1x1_small_kernel<64>), grid, dim3(256), 0, s, p); break;
        case 128: HP_LAUNCH((conv1x1_small_kernel<128>), grid, dim3(256), 0, s, p); break;
        case 192: HP_LAUNCH((conv1x1_small_kernel<192>), grid, dim3(256), 0, s, p); break;
        default: HP_LAUNCH((conv1x1_small_kernel<256>), grid, dim3(256), 0, s, p); break;
        }
        return hipGetLastError();
    }
    if (p.w_layout == 1 && use_gdirect(p)) {
        const int ck = use_gdirect(p), nchunks = p.Cin / ck;
        const int tiles_x = (p.OW + 11) / 12, tiles_y = (p.OH + 15) / 16;
        const dim3 grid(tiles_x * tiles_y * p.B, p.Cout_pad / 128);
#define HP_GD(KS, CK, NBUF) HP_LAUNCH((conv_direct_kernel<KS, CK, NBUF>), grid, dim3(512), 0, s, p, tiles_x, tiles_y, nchunks)
        if (p.KH == 7 && ck == 128)
            HP_GD(7, 128, 1);
        else if (p.KH == 7)
            HP_GD(7, 64, 2);
        else if (p.KH == 5 && ck == 128)
            HP_GD(5, 128, 1);
        else if (p.KH == 5)
            HP_GD(5, 64, 2);
        else if (ck == 128)
            HP_GD(3, 128, 1);
        else if (nchunks == 1)
            HP_GD(3, 64, 1);
        else if (p.ksplit > 1 && p.splitk && nchunks % p.ksplit == 0) { // (the engine sized the scratch for its largest batch)
            const dim3 grid3(grid.x, grid.y, p.ksplit);
            hipEvent_t const ev0 = hp::prof_start, ev1 = hp::prof_stop; // (hp_engine_profile_sequence: begin of the first, end of the second launch)
            if (ev0) {
                hipExtLaunchKernelGGL((conv_direct_kernel<3, 64, 2>), grid3, dim3(512), 0, s, ev0, nullptr, 0, p, tiles_x, tiles_y, nchunks);
                hipExtLaunchKernelGGL((conv_direct_finish_kernel<3, 64>), grid, dim3(512), 0, s, nullptr, ev1, 0, p, tiles_x, tiles_y);
            } else {
                hipLaunchKernelGGL((conv_direct_kernel<3, 64, 2>), grid3, dim3(512), 0, s, p, tiles_x, tiles_y, nchunks);
                hipLaunchKernelGGL((conv_direct_finish_kernel<3, 64>), grid, dim3(512), 0, s, p, tiles_x, tiles_y);
            }
        } else
            HP_GD(3, 64, 2);
#undef HP_GD
        return hipGetLastError();
    }
    if (p.w_layout == 1) {
        if (!(use_halo(p) && fast_epilogue(p)))
            return hipErrorInvalidValue; // fragment-ordered weights only fit the direct kernel
        const int tiles_x = (p.OW + 11) / 12, tiles_y = (p.OH + 15) / 16;
        dim3 grid(tiles_x * tiles_y * p.B, p.Cout_pad / 64);
        if (p.Cin == 128)
            HP_LAUNCH((conv3x3_direct_kernel<128, 16>), grid, dim3(256), 0, s, p, tiles_x, tiles_y);
        else
            HP_LAUNCH((conv3x3_direct_kernel<64, 16>), grid, dim3(256), 0, s, p, tiles_x, tiles_y);
        return hipGetLastError();
    }
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
        __half* op = p.out.p + tv_off(p.out, b, oy, ox) + g * 8;
        if (g * 8 + 7 < p.Cout && ((p.out.coff & 7) == 0))
            *reinterpret_cast<half8*>(op) = h;
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (g * 8 + r < p.Cout)
                    reinterpret_cast<_Float16*>(op)[r] = h[r];
        }
    }
}

// The first layer on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulation like every other layer): 16 x the rate of the
// fp32 pipe per k.  u8 pixels are exact in fp16; the normalised value (x * factor - mean) / std and the weights are rounded to fp16 once
// (the rounding every other layer's operands already carry).  A block owns 16 rows x 32 columns of output pixels; its input patch is
// converted once into LDS as [row][x * 3 + c] halves, so the KS * 3 values of one kernel row of one pixel are contiguous: the im2col
// operand of a k16 step is four ds_read_b32.  K is laid out as KS kernel rows padded to ROWP = a multiple of 8 (24 for 7 x 7, 16 for 3 x 3):
// the pad positions read the next pixels' (finite) data against zero weights.  The weights sit in registers in fragment order
// (first_conv_params::w16, packed by the engine: [Cout / 32][step][64 lanes][8 halves]).  Stride 1 makes odd columns start on an odd
// half: a second copy of the patch shifted by one half keeps every read 4-byte aligned.
template <int KS, int MT, int S, bool CLAMP>
__global__ __launch_bounds__(256) void first_conv_f16_kernel(const first_conv_params p, int tiles_x, int tiles_y, float lo, float hi)
{
    constexpr int ROWP = (KS * 3 + 7) / 8 * 8, KP = KS * ROWP, STEPS = (KP + 15) / 16;
    constexpr int TH = 16, TW = 32, IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int PITCH = ((TW - 1) * S * 3 + ROWP + 2 + 1) / 2 * 2; // every read of a row stays inside it (+2: the shifted copy)
    constexpr int NCOPY = (S & 1) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 s_x[NCOPY][IH * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hh = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;

    // this lane's weights and bias (requested first: their latency hides behind the patch conversion)
    half8 wa[MT][STEPS];
    float bs[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
            wa[mt][st] = *reinterpret_cast<const half8*>(p.w16 + (((size_t)mt * STEPS + st) * 64 + lane) * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) { // accumulator registers 4g .. 4g+3 = channels mt*32 + 8g + 4hh + (0..3)
            const int ch = mt * 32 + 8 * g + 4 * hh;
            const float4 bv = ch + 3 < p.Cout ? *reinterpret_cast<const float4*>(p.bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            bs[mt][4 * g] = bv.x, bs[mt][4 * g + 1] = bv.y, bs[mt][4 * g + 2] = bv.z, bs[mt][4 * g + 3] = bv.w;
        }
    }
    // the patch: zero-fill (row pads are read against zero weights and must be finite), then all pixel loads, then the conversions
    for (int i = tid; i < NCOPY * IH * PITCH / 2; i += 256)
        reinterpret_cast<unsigned*>(&s_x[0][0])[i] = 0u;
    __syncthreads();
    {
        constexpr int NIT = (IH * IW + 255) / 256;
        const int c0 = p.flip_rb ? 2 : 0, c2 = p.flip_rb ? 0 : 2;
        auto put = [&](int py, int px, int c, float x) { // normalise, round to fp16, store (both copies)
            const _Float16 v = (_Float16)((x - p.mean[c]) * p.inv_std[c]);
            s_x[0][py * PITCH + px * 3 + c] = v;
            if (NCOPY == 2 && px * 3 + c >= 1)
                s_x[NCOPY - 1][py * PITCH + px * 3 + c - 1] = v; // copy 1 [j] = copy 0 [j + 1]
        };
        if (p.in_u8) {
            // all byte loads first (clamped addresses, three bytes of a pixel packed into one register): one memory round trip for the
            // patch instead of one per pass - the patch was half of the 7 x 7 block's time, and the 3 x 3 stride-2 stem's 33 x 65 patch is nine
            // passes (round 4: the ISA showed nine load -> wait -> convert rounds)
            unsigned raw[NIT];
            unsigned okm = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = min(tid + it * 256, IH * IW - 1);
                const int py = i / IW, px = i - py * IW;
                const int iy = iy0 + py, ix = ix0 + px;
                okm |= (tid + it * 256 < IH * IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (1u << it) : 0u;
                const uint8_t* q = p.in_u8 + (((size_t)b * p.H + min(max(iy, 0), p.H - 1)) * p.W + min(max(ix, 0), p.W - 1)) * 3;
                raw[it] = (unsigned)q[c0] | ((unsigned)q[1] << 8) | ((unsigned)q[c2] << 16);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (!((okm >> it) & 1u))
                    continue; // stays zero = the convolution's padding
                const int i = tid + it * 256;
                const int py = i / IW, px = i - py * IW;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    put(py, px, c, (float)((double)(float)((raw[it] >> (8 * c)) & 255u) * p.factor)); // src/data.cpp:48
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                if (i >= IH * IW)
                    break;
                const int py = i / IW, px = i - py * IW;
                const int iy = iy0 + py, ix = ix0 + px;
                if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W)
                    continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (p.in_u8) {
                        const int sc = c == 0 ? c0 : (c == 2 ? c2 : 1);
                        put(py, px, c, (float)((double)(float)p.in_u8[(((size_t)b * p.H + iy) * p.W + ix) * 3 + sc] * p.factor)); // src/data.cpp:48
                    } else
                        put(py, px, c, p.in_f32[(((size_t)b * 3 + c) * p.H + iy) * p.W + ix]);
                }
            }
        }
    }
    __syncthreads();

    const unsigned hmask = hh ? 0xffffffffu : 0u;
    const int odd = (NCOPY == 2) ? ((col * S * 3) & 1) : 0;
    const _Float16* const xcopy = &s_x[odd][0] + (col * S * 3 - odd);
#pragma unroll 1
    for (int rr = 0; rr < TH / 4; ++rr) {
        const int row = wave * (TH / 4) + rr, oy = oy0 + row, ox = ox0 + col;
        if (oy >= p.OH) // uniform per wavefront
            break;
        const _Float16* xb = xcopy + (row * S) * PITCH;
        floatx16 d[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                d[mt][r] = bs[mt][r];
        if constexpr (KS == 7) {
        // k' = st * 16 + hh * 8 + i  ->  kernel row k' / ROWP (clamped: the tail of the last step has zero weights), offset k' % ROWP;
        // the fragment of step st + 1 is read while step st multiplies (pinned: hipcc sinks a ds_read to just before its MFMA)
        auto frag = [&](int st) {
            const int ka = st * 16, kb = st * 16 + 8;
            const int oa = min(ka / ROWP, KS - 1) * PITCH + ka % ROWP, ob = min(kb / ROWP, KS - 1) * PITCH + kb % ROWP; // compile time
            const unsigned* q = reinterpret_cast<const unsigned*>(xb + (hh ? ob : oa));
            u32x4 raw;
            raw[0] = q[0], raw[1] = q[1], raw[2] = q[2], raw[3] = q[3];
            return raw;
        };
        u32x4 fr[2];
        fr[0] = frag(0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS)
                fr[(st + 1) & 1] = frag(st + 1);
            half8 xv;
            __builtin_memcpy(&xv, &fr[st & 1], 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                d[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[mt][st], xv, d[mt], 0, 0, 0);
            if (KS == 7) {
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
                if (st + 1 < STEPS)
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
        }
        } else {
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            // k' = st * 16 + hh * 8 + i  ->  kernel row k' / ROWP (clamped: the tail of the last step has zero weights), offset k' % ROWP
            constexpr int dummy = 0;
            (void)dummy;
            const int ka = st * 16, kb = st * 16 + 8;
            const int oa = min(ka / ROWP, KS - 1) * PITCH + ka % ROWP, ob = min(kb / ROWP, KS - 1) * PITCH + kb % ROWP; // compile time
            const unsigned* q = reinterpret_cast<const unsigned*>(xb + (hh ? ob : oa));
            u32x4 raw;
            raw[0] = q[0], raw[1] = q[1], raw[2] = q[2], raw[3] = q[3];
            half8 xv;
            __builtin_memcpy(&xv, &raw, 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                d[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[mt][st], xv, d[mt], 0, 0, 0);
        }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt * 32 >= p.Cout)
                break;
            // d[4g + e] = channel mt*32 + 8g + 4hh + e of pixel `col`.  Half 0 keeps groups 0, 1 and half 1 groups 2, 3: each sends the
            // other its two foreign groups and ends with 8 consecutive channels per group.
            unsigned mine[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v0 = d[mt][4 * g + 2 * e], v1 = d[mt][4 * g + 2 * e + 1];
                    const _Float16 h0 = (_Float16)(CLAMP ? __builtin_amdgcn_fmed3f(v0, lo, hi) : apply_act(v0, p.act, p.act_param, 0.f));
                    const _Float16 h1 = (_Float16)(CLAMP ? __builtin_amdgcn_fmed3f(v1, lo, hi) : apply_act(v1, p.act, p.act_param, 0.f));
                    unsigned short ul, uh;
                    __builtin_memcpy(&ul, &h0, 2), __builtin_memcpy(&uh, &h1, 2);
                    mine[g][e] = (unsigned)ul | ((unsigned)uh << 16);
                }
            unsigned got[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    got[q][e] = (unsigned)__shfl_xor((int)((mine[q][e] & hmask) | (mine[2 + q][e] & ~hmask)), 32);
            if (ox < p.OW) {
                __half* const op = p.out.p + tv_off(p.out, b, oy, ox) + mt * 32;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = 2 * hh + q;
                    u32x4 v;
                    v[0] = (got[q][0] & hmask) | (mine[q][0] & ~hmask), v[1] = (got[q][1] & hmask) | (mine[q][1] & ~hmask);
                    v[2] = (mine[2 + q][0] & hmask) | (got[q][0] & ~hmask), v[3] = (mine[2 + q][1] & hmask) | (got[q][1] & ~hmask);
                    if (mt * 32 + 8 * g < p.Cout)
                        *reinterpret_cast<u32x4*>(op + 8 * g) = v;
                }
            }
        }
    }
}

hipError_t launch_first_conv(const first_conv_params& p, hipStream_t s)
{
    if (p.w16 && p.KH == p.KW && (p.KH == 3 || p.KH == 7) && (p.stride == 1 || p.stride == 2) && p.Cout % 8 == 0 && p.Cout <= 64
        && p.out.coff % 8 == 0 && p.out.cs % 8 == 0) {
        const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 15) / 16;
        const dim3 grid(tiles_x * tiles_y * p.B);
        const bool clamp = p.act == ACT_NONE || p.act == ACT_RELU || p.act == ACT_RELU6;
        const float lo = p.act == ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = p.act == ACT_RELU6 ? 6.f : __builtin_huge_valf();
        const int mt = p.Cout <= 32 ? 1 : 2;
#define HP_F16(KS_, MT_, S_)                                                                                          \
    do {                                                                                                              \
        if (clamp)                                                                                                    \
            HP_LAUNCH((first_conv_f16_kernel<KS_, MT_, S_, true>), grid, dim3(256), 0, s, p, tiles_x, tiles_y, lo, hi);  \
        else                                                                                                          \
            HP_LAUNCH((first_conv_f16_kernel<KS_, MT_, S_, false>), grid, dim3(256), 0, s, p, tiles_x, tiles_y, lo, hi); \
    } while (0)
#define HP_F16S(KS_, MT_)      \
    do {                       \
        if (p.stride == 2)     \
            HP_F16(KS_, MT_, 2); \
        else                   \
            HP_F16(KS_, MT_, 1); \
    } while (0)
        if (p.KH == 7 && mt == 2)
            HP_F16S(7, 2);
        else if (p.KH == 7)
            HP_F16S(7, 1);
        else if (mt == 2)
            HP_F16S(3, 2);
        else
            HP_F16S(3, 1);
#undef HP_F16S
#undef HP_F16
        return hipGetLastError();
    }

    const int G = (p.Cout + 7) / 8;
    if (G > 256)
        return hipErrorInvalidValue;
    const int ppb = 256 / G;
    const long npix = (long)p.B * p.OH * p.OW;
    const int blocks = (int)std::min<long>((npix + ppb - 1) / ppb, 256 * 16);
    const size_t lds = (size_t)p.KH * p.KW * 3 * G * 8 * sizeof(float);
    HP_LAUNCH(first_conv_kernel, dim3(blocks), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Depthwise 3x3.  The vector-memory path of a CU delivers ~16 B/clk whether a request hits L1 or not, so re-reading
// every input pixel for each of its 9 taps (the naive gather) runs at 1/9 of that.  Here a block stages the
// (TH*stride + 2*dil) x (TW*stride + 2*dil) input tile of 64 channels in LDS with ONE coalesced 16-byte load per element
// (128 contiguous bytes per pixel), then every thread produces pixels x 8 channels from LDS: global traffic is
// one read + one write of the tensor.  The zero halo of the HBM layout makes every load unconditional except at the
// ragged right/bottom tile edges.
// acc[r] += (float)x.h[r] * (float)w.h[r] for 8 packed halves: one v_fma_mix_f32 per MAC (fp16 operands, fp32
// accumulate).  Written as inline asm because hipcc otherwise converts BOTH operands to fp32 first (2 v_cvt + half a
// v_pk_fma_f32 per MAC and ~70 extra VGPRs, which halves the occupancy of this latency-bound kernel).
__device__ __forceinline__ void mac8_f16(float (&acc)[8], const u32x4 x, const u32x4 w)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q]) : "v"(x[q]), "v"(w[q]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * q + 1]) : "v"(x[q]), "v"(w[q]));
    }
}

__device__ __forceinline__ void mac4_f16(float (&acc)[4], const uint2 x, const uint2 w)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[0]) : "v"(x.x), "v"(w.x));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[1]) : "v"(x.x), "v"(w.x));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2]) : "v"(x.y), "v"(w.y));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[3]) : "v"(x.y), "v"(w.y));
}

// the first tap of an output: acc = x * w + bias in one instruction (no copy of the bias into the accumulator first)
__device__ __forceinline__ void mac4_f16_init(float (&acc)[4], const uint2 x, const uint2 w, const float4 bias)
{
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(acc[0]) : "v"(x.x), "v"(w.x), "v"(bias.x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(acc[1]) : "v"(x.x), "v"(w.x), "v"(bias.y));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(acc[2]) : "v"(x.y), "v"(w.y), "v"(bias.z));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(acc[3]) : "v"(x.y), "v"(w.y), "v"(bias.w));
}

// depthwise activation y = v > 0 ? min(v, hi) : v * slope; the relu / relu6 family (slope == 0) is one v_med3_f32
template <bool CLAMP>
__device__ __forceinline__ float dw_act(float v, float slope, float hi)
{
    return CLAMP ? __builtin_amdgcn_fmed3f(v, 0.f, hi) : (v > 0.f ? fminf(v, hi) : v * slope);
}

constexpr int DW_TH = 8, DW_TW = 8, DW_CG = 8; // output tile 8x8 pixels, 8 chunks of 8 channels = 64 channels

template <int NLD> // 16-byte loads per thread per tile = ceil(IH*IW*8 / 256)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const dw_params p, int tiles_x, int tiles_y, int cgroups, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_lds[];
    const int tid = threadIdx.x;
    const int IH = (DW_TH - 1) * p.stride + 2 * p.dil + 1, IW = (DW_TW - 1) * p.stride + 2 * p.dil + 1;
    const int ymax = p.H + p.halo - 1, xmax = p.W + p.halo - 1; // extent of the zero halo in HBM
    const int chunk = tid & 7;
    const int nelem = IH * IW * DW_CG;
    const bool clamp_only = p.act_slope == 0.f; // uniform

    // persistent block: the loads of tile t+1 are in flight (in registers) while tile t is computed from LDS
    half8 nxt[NLD];
    auto decode = [&](int t, int& cgi, int& tx, int& ty, int& b) {
        cgi = t % cgroups;
        t /= cgroups;
        tx = t % tiles_x;
        t /= tiles_x;
        ty = t % tiles_y;
        b = t / tiles_y;
    };
#define HP_DW_LOAD(T)                                                                                             \
    {                                                                                                             \
        int cgi_, tx_, ty_, b_;                                                                                   \
        decode(T, cgi_, tx_, ty_, b_);                                                                            \
        const int c0_ = cgi_ * DW_CG * 8;                                                                         \
        const int iy0_ = ty_ * DW_TH * p.stride - p.pad_t, ix0_ = tx_ * DW_TW * p.stride - p.pad_l;               \
        const bool cv_ = c0_ + chunk * 8 < p.C;                                                                   \
        _Pragma("unroll") for (int k = 0; k < NLD; ++k)                                                           \
        {                                                                                                         \
            const int i = tid + k * 256;                                                                          \
            const int hp = min(i, nelem - 1) >> 3;                                                                \
            const int hy = hp / IW, hx = hp - hy * IW;                                                            \
            const int y = iy0_ + hy, x = ix0_ + hx;                                                               \
            const bool ok = cv_ && y <= ymax && x <= xmax;                                                        \
            half8 v = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b_, min(y, ymax), min(x, xmax)) + (cv_ ? c0_ + chunk * 8 : 0)); \
            if (!ok)                                                                                              \
                _Pragma("unroll") for (int r = 0; r < 8; ++r) v[r] = (_Float16)0.f;                               \
            nxt[k] = v;                                                                                           \
        }                                                                                                         \
    }
    int t = blockIdx.x;
    if (t < total_tiles)
        HP_DW_LOAD(t);
    for (; t < total_tiles; t += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * 256;
            if (i < nelem)
                *reinterpret_cast<half8*>(dw_lds + (size_t)i * 16) = nxt[k];
        }
        __syncthreads();
        if (t + (int)gridDim.x < total_tiles)
            HP_DW_LOAD(t + (int)gridDim.x);
        int cgi, tx, ty, b;
        decode(t, cgi, tx, ty, b);
        const int c0 = cgi * DW_CG * 8;
        // this tile's 9 x 64 weights go through LDS too: kept as fp16 and read right before use, otherwise hipcc
        // hoists 72 fp32 conversions into registers and the kernel drops to 2 waves/SIMD
        half8* s_w = reinterpret_cast<half8*>(dw_lds + (size_t)nelem * 16);
        if (tid < 72)
            s_w[tid] = (c0 + (tid & 7) * 8 < p.C) ? *reinterpret_cast<const half8*>(p.w + (size_t)(tid >> 3) * p.C + c0 + (tid & 7) * 8) : half8{};
        __syncthreads();
        if (c0 + chunk * 8 < p.C) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + c0 + chunk * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + c0 + chunk * 8 + 4);
            auto passes = [&](auto clamp_tag) {
            constexpr bool CLAMP = decltype(clamp_tag)::value;
#pragma unroll 1 // keep the live set small (occupancy hides the LDS / store latency here, not ILP)
            for (int pass = 0; pass < DW_TH * DW_TW * DW_CG / 256; ++pass) {
                const int pix = (tid >> 3) + pass * 32;
                const int py = pix / DW_TW, px = pix - py * DW_TW;
                const int oy = ty * DW_TH + py, ox = tx * DW_TW + px;
                float acc[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    u32x4 x[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int hp = (py * p.stride + ky * p.dil) * IW + px * p.stride + kx * p.dil;
                        x[kx] = *reinterpret_cast<const u32x4*>(dw_lds + ((size_t)hp * DW_CG + chunk) * 16);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        mac8_f16(acc, x[kx], *reinterpret_cast<const u32x4*>(&s_w[(ky * 3 + kx) * 8 + chunk]));
                }
                if (oy < p.OH && ox < p.OW) {
                    half8 h;
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        h[r] = (_Float16)dw_act<CLAMP>(acc[r], p.act_slope, p.act_hi);
                    *reinterpret_cast<half8*>(p.out.p + tv_off(p.out, b, oy, ox) + c0 + chunk * 8) = h;
                }
            }
            };
            if (clamp_only)
                passes(std::true_type{});
            else
                passes(std::false_type{});
        }
        __syncthreads(); // the tile in LDS is consumed before the next one overwrites it
    }
#undef HP_DW_LOAD
}

hipError_t launch_dwconv3x3(const dw_params& p_in, hipStream_t s)
{
    dw_params p = p_in;
    {   // piecewise-linear activations only (none / relu / relu6 / leaky), as y = v > 0 ? min(v, hi) : v * slope
        conv_params tmp{};
        tmp.act = p.act, tmp.act_param = p.act_param, tmp.alpha = nullptr;
        if (p.act == ACT_PRELU || !set_act(tmp))
            return hipErrorInvalidValue;
        p.act_slope = tmp.act_slope, p.act_hi = tmp.act_hi;
    }
    const int tiles_x = (p.OW + DW_TW - 1) / DW_TW, tiles_y = (p.OH + DW_TH - 1) / DW_TH;
    const int cgroups = (p.C + DW_CG * 8 - 1) / (DW_CG * 8);
    const int IH = (DW_TH - 1) * p.stride + 2 * p.dil + 1, IW = (DW_TW - 1) * p.stride + 2 * p.dil + 1;
    const size_t lds = (size_t)IH * IW * DW_CG * 16 + 72 * 16;
    const int total = tiles_x * tiles_y * cgroups * p.B;
    const int nld = (IH * IW * DW_CG + 255) / 256;
    const dim3 grid(std::min(total, 256 * 12));
    if (nld <= 4)
        HP_LAUNCH((dwconv3x3_kernel<4>), grid, dim3(256), lds, s, p, tiles_x, tiles_y, cgroups, total);
    else if (nld <= 5)
        HP_LAUNCH((dwconv3x3_kernel<5>), grid, dim3(256), lds, s, p, tiles_x, tiles_y, cgroups, total);
    else if (nld <= 10)
        HP_LAUNCH((dwconv3x3_kernel<10>), grid, dim3(256), lds, s, p, tiles_x, tiles_y, cgroups, total);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Depthwise 3x3 + pointwise 1x1 in ONE launch (MobileNet's separable block, backbones.py MobilenetDilated): the depthwise output
// never exists in HBM.  Every kernel boundary on this chip sends the whole activation through the memory-side fabric (the per-XCD
// L2s are not coherent with each other), so the unfused pair writes and re-reads B*H*W*C halves for nothing and pays the depthwise
// kernel's own launch.  Two forms below: sepconv_slot_kernel (64 .. 512 channels in 64-channel K chunks) and sepconv_small_kernel
// (32 .. 128 channels, all of them in LDS at once); the depthwise arithmetic is dwconv3x3_kernel's (bias first, fp32 v_fma_mix
// accumulation tap by tap, activation, RN to fp16) and the pointwise K order the un-fused convolution's, so the fused and the
// un-fused schedule produce the same bits (tests).
// The separable block cut to HALF a CU (<= 256 registers, < 80 KB of LDS) for 256 / 512 output channels at stride 1.
// With two kernels in flight a CU is two slots; a whole-CU form (one 8 x 12 tile x all output channels, depthwise taps interleaved
// between the MFMAs, 460 registers: rounds 1-2) held both, so its whole duration showed up end to end.
// Deliberately plain code (rolled loops, phases not interleaved) so that hipcc's register allocation stays small - the second
// block on the CU provides the overlap:
//   * 8 x 8 output pixels per block; the depthwise results of ALL K chunks stay in LDS (B_all: 64 px x C halves <= 64 KB);
//   * output channels in NP passes of 4 wavefronts x TP row tiles (128 or 256): pass 0 = per chunk depthwise -> B_all, then
//     its MFMAs; pass 1 = MFMAs only, straight out of B_all, no barrier.  <= 64 accumulator registers, nothing recomputed;
//   * pass 0 stores with the direct epilogue (B_all must survive), the last pass with the staged one.
// S = stride, D = dilation of the depthwise taps, CMAX = the largest channel count the instance holds (sizes B_all),
// CKH = channels per staged halo chunk (64 = one MFMA K chunk; 32 halves the halo buffer so that the dilated 512-channel
// block still fits 80 KB: two halo chunks then feed one K chunk).
template <int NP, int TP, int S, int D, int CMAX, int CKH = 64>
__global__ __launch_bounds__(256, 2) void sepconv_slot_kernel(const sep_params p, int tiles_x, int tiles_y)
{
    constexpr int NW = 4, NTHR = 64 * NW;
    constexpr int TH = 8, TW = 8, NPX = 64, NT = 2, CK = 64, CG = CKH / 8, KS = 4, HPK = CK / CKH;
    constexpr int IH = (TH - 1) * S + 2 * D + 1, IW = (TW - 1) * S + 2 * D + 1;
    constexpr int PIECES = IH * IW * CG, NLD = (PIECES + NTHR - 1) / NTHR, ITEMS = (NPX * CG + NTHR - 1) / NTHR;
    static_assert(NPX * CG % NTHR == 0, "whole items");
    constexpr int HALO_BYTES = PIECES * 16, BCH_BYTES = NPX * CK * 2, BALL_BYTES = (CMAX / CK) * BCH_BYTES;
    constexpr int DWW_BYTES = 9 * CKH * 2, DWB_BYTES = CKH * 4;
    constexpr int MAIN_BYTES = BALL_BYTES + HALO_BYTES + 2 * DWW_BYTES + 2 * DWB_BYTES;
    // epilogue slabs over B_all and the halo buffer: all pixel tiles of a wavefront (of both passes) at once (conv_epilogue_packed)
    constexpr int EPI_BYTES = NW * NP * packed_geom<TP, NT>::WAVE_BYTES;
    static_assert(EPI_BYTES <= 80 * 1024, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
    unsigned char* const s_ball = lds;
    unsigned char* const s_halo = lds + BALL_BYTES;
    unsigned char* const s_dww = s_halo + HALO_BYTES;   // [2][9][CKH] halves
    unsigned char* const s_dwb = s_dww + 2 * DWW_BYTES; // [2][CKH] floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ymax = p.H + p.halo - 1, xmax = p.W + p.halo - 1;
    const int C = p.C, KQ = C / 16, NCH = C / CK;

    // pointwise weights of (pass, K chunk): fragment (row tile pass*4*TP + wave*TP + i, k16 step chunk*4 + ks).  One pointer per
    // chunk, constant offsets per (i, ks): no per-load index arithmetic in the MFMA phase (it is issue-bound)
    const __half* const wbase = p.pw.w + (size_t)lane * 8 + (size_t)(wave * TP) * KQ * 512;
    const size_t pass_stride = (size_t)NW * TP * KQ * 512, row_stride = (size_t)KQ * 512;
    u32x4 a[KS][TP];
    auto a_load = [&](const __half* wp, int ks) {
#pragma unroll
        for (int i = 0; i < TP; ++i)
            a[ks][i] = *reinterpret_cast<const u32x4*>(wp + i * row_stride + ks * 512);
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        a_load(wbase, ks);
    // halo chunk + depthwise weights / bias: global -> registers (one chunk ahead) -> LDS
    u32x4 hv[NLD], wreg;
    int hoff[NLD];
    unsigned hmask = 0;
    {
        const int iy0 = y0 * S - p.pad_t, ix0 = x0 * S - p.pad_l;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = min(tid + k * NTHR, PIECES - 1);
            const int hp = i / CG, c = i - hp * CG;
            const int hy = hp / IW, hx = hp - hy * IW;
            const int y = iy0 + hy, x = ix0 + hx;
            hmask |= (y <= ymax && x <= xmax) ? (1u << k) : 0u;
            hoff[k] = (min(y, ymax) * p.in.wp + min(x, xmax)) * p.in.cs + c * 8;
        }
    }
    const __half* const hbase = p.in.p + (size_t)b * p.in.img * p.in.cs + p.in.coff;
    constexpr int NWT = 9 * CG, NBT = CKH / 4; // 16-B pieces of the [9][CKH] weights and of the CKH biases
    const bool w_thread = tid < NWT, b_thread = tid >= NWT && tid < NWT + NBT;
    auto hload = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            hv[k] = *reinterpret_cast<const u32x4*>(hbase + hoff[k] + chunk * CKH);
        const void* src = w_thread ? (const void*)(p.dw_w + (size_t)(tid / CG) * C + chunk * CKH + (tid % CG) * 8)
                                   : (const void*)(p.dw_bias + chunk * CKH + (b_thread ? (tid - NWT) * 4 : 0));
        wreg = *reinterpret_cast<const u32x4*>(src);
    };
    auto to_lds = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * NTHR < PIECES)
                *reinterpret_cast<u32x4*>(s_halo + (size_t)(tid + k * NTHR) * 16) = hv[k] & (((hmask >> k) & 1u) ? 0xffffffffu : 0u);
        if (w_thread)
            *reinterpret_cast<u32x4*>(s_dww + (chunk & 1) * DWW_BYTES + tid * 16) = wreg;
        else if (b_thread)
            *reinterpret_cast<u32x4*>(s_dwb + (chunk & 1) * DWB_BYTES + (tid - NWT) * 16) = wreg;
    };
    hload(0);

    floatx16 acc[TP][NT], acc1[NP == 2 ? TP : 1][NT]; // pass 0 / pass 1 results: both are stored after pass 1, through the then
                                                        // dead B_all (a direct epilogue for pass 0 cost 21 k of the block's 61 k cycles)
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (NP == 2)
                    acc1[i][j][r] = 0.f;
            }
    const int g = tid % CG, frow = lane & 31, fk = lane >> 5;
    const float dw_hi = p.dw_hi;

    // depthwise taps of halo chunk kd -> its CKH columns of B_all[kd / HPK] (same arithmetic as dwconv3x3_kernel / sepconv_kernel)
    auto dw_chunk = [&](int kd) {
        unsigned char* const bt = s_ball + (kd / HPK) * BCH_BYTES;
        const int gcol = (kd % HPK) * CG + g;
        const float* bsrc = reinterpret_cast<const float*>(s_dwb + (kd & 1) * DWB_BYTES) + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
        u32x4 wv[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
            wv[t9] = *reinterpret_cast<const u32x4*>(s_dww + (kd & 1) * DWW_BYTES + (t9 * CKH + g * 8) * 2);
#pragma unroll 1
        for (int r = 0; r < ITEMS; ++r) {
            const int pix = (tid + r * NTHR) / CG;
            const int py = pix / TW, px = pix - py * TW;
            const unsigned char* xs = s_halo + ((py * S * IW + px * S) * CG + g) * 16;
            u32x4 x[9];
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
                x[t9] = *reinterpret_cast<const u32x4*>(xs + (((t9 / 3) * D) * IW + (t9 % 3) * D) * CG * 16);
            float v[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
                mac8_f16(v, x[t9], wv[t9]);
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                h[e] = (_Float16)dw_act<true>(v[e], 0.f, dw_hi);
            *reinterpret_cast<half8*>(bt + lds_off<CK>(pix, gcol)) = h;
        }
    };
    // MFMAs of (pass ps, K chunk kc) out of B_all; the weights of the following step are requested as each k16 step's are consumed
    auto mm_chunk = [&](int ps, int kc, floatx16 (&acc)[TP][NT]) {
        const unsigned char* const bt = s_ball + kc * BCH_BYTES;
        const bool wrap = kc + 1 == NCH;
        const int ps2 = wrap ? min(ps + 1, NP - 1) : ps, kc2 = wrap ? (ps + 1 < NP ? 0 : kc) : kc + 1;
        const __half* const wn = wbase + ps2 * pass_stride + (size_t)kc2 * (KS * 512);
        // (the B fragments of step ks + 1 are read while step ks multiplies: read -> wait -> multiply per step left the matrix pipe idle
        // for an LDS latency four times per chunk)
        half8 fb[2][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[0][j] = *reinterpret_cast<const half8*>(bt + lds_off<CK>(j * 32 + frow, fk));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(bt + lds_off<CK>(j * 32 + frow, (ks + 1) * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                half8 fa;
                __builtin_memcpy(&fa, &a[ks][i], 16);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 1][j], acc[i][j], 0, 0, 0);
            }
            a_load(wn, ks);
        }
    };

    int pb[NT], py[NT], px[NT];
    bool pv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + (lane & 31);
        pb[j] = b;
        py[j] = y0 + n / TW;
        px[j] = x0 + n % TW;
        pv[j] = py[j] < p.OH && px[j] < p.OW;
    }

    // ---- pass 0
    int dbg_i = 0;
#define HP_STAMP()                                                   \
    if (p.pw.dbg && blockIdx.x == 0 && tid == 0 && dbg_i < 40)       \
        p.pw.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    const int NHC = NCH * HPK; // halo chunks
    HP_STAMP();
    to_lds(0);
    hload(min(1, NHC - 1));
    lds_barrier();
    HP_STAMP();
#pragma unroll 1
    for (int hc = 0; hc < NHC; ++hc) {
        dw_chunk(hc);
        HP_STAMP();
        lds_barrier(); // these columns of B_all complete; every thread is past its reads of halo chunk hc
        HP_STAMP();
        if (hc + 1 < NHC) {
            to_lds(hc + 1);
            hload(min(hc + 2, NHC - 1));
        }
        HP_STAMP();
        if (hc % HPK == HPK - 1)
            mm_chunk(0, hc / HPK, acc);
        HP_STAMP();
        lds_barrier(); // halo chunk hc+1 and its depthwise weights visible
        HP_STAMP();
    }
    if (NP == 1) {
        __syncthreads(); // every wave is done with B_all before the slabs overwrite it
        conv_epilogue_packed<TP, NT>(p.pw, acc, (wave * TP) * 32, lane, lds + wave * packed_geom<TP, NT>::WAVE_BYTES, pb, py, px, pv);
        HP_STAMP();
        return;
    }
    if (p.pw.dbg && blockIdx.x == 0 && tid == 0)
        p.pw.dbg[41] = __builtin_amdgcn_s_memtime();
    // ---- pass 1: MFMAs only
    if constexpr (NP == 2) {
#pragma unroll 1
        for (int kc = 0; kc < NCH; ++kc)
            mm_chunk(1, kc, acc1);
        if (p.pw.dbg && blockIdx.x == 0 && tid == 0)
            p.pw.dbg[42] = __builtin_amdgcn_s_memtime();
        __syncthreads(); // every wave is done with B_all before the slabs overwrite it
        // (both passes' tiles in their own slabs: nothing to wait for in between)
        conv_epilogue_packed<TP, NT>(p.pw, acc, (wave * TP) * 32, lane, lds + (2 * wave) * packed_geom<TP, NT>::WAVE_BYTES, pb, py, px, pv);
        conv_epilogue_packed<TP, NT>(p.pw, acc1, (NW * TP + wave * TP) * 32, lane, lds + (2 * wave + 1) * packed_geom<TP, NT>::WAVE_BYTES, pb, py, px, pv);
        if (p.pw.dbg && blockIdx.x == 0 && tid == 0)
            p.pw.dbg[43] = __builtin_amdgcn_s_memtime();
    }
#undef HP_STAMP
}

template <int NP, int TP, int S, int D, int CMAX, int CKH = 64>
static hipError_t launch_sep_slot(const sep_params& p, hipStream_t s)
{
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 7) / 8;
    HP_LAUNCH((sepconv_slot_kernel<NP, TP, S, D, CMAX, CKH>), dim3(tiles_x * tiles_y * p.B), dim3(256), 0, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The 512-output-channel separable blocks (five of them per frame in LightWeight-OpenPose: the network's dominant kernel) as a
// whole-CU block whose depthwise taps and MFMAs overlap INSIDE the block.  sepconv_slot_kernel's timeline (s_memtime, profiles/)
// is taps 1.75 k -> barrier -> MFMAs 1.3 k -> barrier per K chunk: the vector and the matrix pipe take turns, and the second
// block on the CU only partly fills the gaps.  Here eight wavefronts = two per SIMD work in ANTI-PHASE between two barriers:
// in interval k every wavefront owes the taps of chunk k+1 (-> B_all) and the MFMAs of chunk k (<- B_all); wavefronts 0-3 do the
// taps first, 4-7 the MFMAs first, so each SIMD has one wavefront on the vector pipe and one on the matrix pipe at any time.
// The halo chunks are double-buffered (chunk k+2 is written while chunk k+1 is read).  12 x 8 output pixels x all 512 output
// channels per block: 16 row tiles = 8 wavefronts x 2, three pixel tiles, ONE pass; 224 blocks for 8 x 46 x 54: one round on 256 CUs
// (the 64-pixel tile of an eight-wavefront block needed 311 blocks = two rounds, and lost: DESIGN.md section 7).
// Same arithmetic and K order as the two-launch form, so the results are bit-identical (tests).
template <int D, int CMAX>
__global__ __launch_bounds__(512, 1) void sepconv_pipe_kernel(const sep_params p, int tiles_x, int tiles_y)
{
    constexpr int NW = 8, NTHR = 512, TP = 2, TH = 12, TW = 8, NPX = TH * TW, NT = NPX / 32, CK = 64, CG = 8, KS = 4;
    constexpr int IH = TH + 2 * D, IW = TW + 2 * D, PIECES = IH * IW * CG, NLD = (PIECES + NTHR - 1) / NTHR;
    constexpr int HALO_BYTES = PIECES * 16, BCH_BYTES = NPX * CK * 2, BALL_BYTES = (CMAX / CK) * BCH_BYTES;
    constexpr int DWW_BYTES = 9 * CK * 2, DWB_BYTES = CK * 4;
    constexpr int MAIN_BYTES = BALL_BYTES + 2 * HALO_BYTES + 2 * DWW_BYTES + 2 * DWB_BYTES;
    constexpr int EPI_BYTES = NW * packed_geom<TP, NT>::WAVE_BYTES;
    static_assert(MAIN_BYTES <= 160 * 1024 && EPI_BYTES <= MAIN_BYTES, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds[MAIN_BYTES];
    unsigned char* const s_ball = lds;
    unsigned char* const s_halo = lds + BALL_BYTES;       // [2] halo chunks
    unsigned char* const s_dww = s_halo + 2 * HALO_BYTES; // [2][9][64] halves
    unsigned char* const s_dwb = s_dww + 2 * DWW_BYTES;   // [2][64] floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ymax = p.H + p.halo - 1, xmax = p.W + p.halo - 1;
    const int C = p.C, KQ = C / 16, NCH = C / CK;

    const __half* const wbase = p.pw.w + (size_t)lane * 8 + (size_t)(wave * TP) * KQ * 512;
    const size_t row_stride = (size_t)KQ * 512;
    u32x4 a[KS][TP];
    auto a_load = [&](const __half* wp, int ks) {
#pragma unroll
        for (int i = 0; i < TP; ++i)
            a[ks][i] = *reinterpret_cast<const u32x4*>(wp + i * row_stride + ks * 512);
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        a_load(wbase, ks);
    u32x4 hv[NLD], wreg;
    int hoff[NLD];
    unsigned hmask = 0;
    {
        const int iy0 = y0 - p.pad_t, ix0 = x0 - p.pad_l;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = min(tid + k * NTHR, PIECES - 1);
            const int hp = i / CG, c = i - hp * CG;
            const int hy = hp / IW, hx = hp - hy * IW;
            const int y = iy0 + hy, x = ix0 + hx;
            hmask |= (y <= ymax && x <= xmax) ? (1u << k) : 0u;
            hoff[k] = (min(y, ymax) * p.in.wp + min(x, xmax)) * p.in.cs + c * 8;
        }
    }
    const __half* const hbase = p.in.p + (size_t)b * p.in.img * p.in.cs + p.in.coff;
    constexpr int NWT = 9 * CG, NBT = CK / 4;
    const bool w_thread = tid < NWT, b_thread = tid >= NWT && tid < NWT + NBT;
    auto hload = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            hv[k] = *reinterpret_cast<const u32x4*>(hbase + hoff[k] + chunk * CK);
        const void* src = w_thread ? (const void*)(p.dw_w + (size_t)(tid / CG) * C + chunk * CK + (tid % CG) * 8)
                                   : (const void*)(p.dw_bias + chunk * CK + (b_thread ? (tid - NWT) * 4 : 0));
        wreg = *reinterpret_cast<const u32x4*>(src);
    };
    auto to_lds = [&](int chunk) {
        unsigned char* const hs = s_halo + (chunk & 1) * HALO_BYTES;
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * NTHR < PIECES)
                *reinterpret_cast<u32x4*>(hs + (size_t)(tid + k * NTHR) * 16) = hv[k] & (((hmask >> k) & 1u) ? 0xffffffffu : 0u);
        if (w_thread)
            *reinterpret_cast<u32x4*>(s_dww + (chunk & 1) * DWW_BYTES + tid * 16) = wreg;
        else if (b_thread)
            *reinterpret_cast<u32x4*>(s_dwb + (chunk & 1) * DWB_BYTES + (tid - NWT) * 16) = wreg;
    };

    floatx16 acc[TP][NT];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    const int g = tid % CG, frow = lane & 31, fk = lane >> 5;
    const float dw_hi = p.dw_hi;

    // 96 pixels x 8 channel groups = 768 items on 512 threads: one full item (pixels 0-63) + one HALF item (pixels 64-95, four
    // channels) per thread, so that the two wavefronts of a SIMD carry the same tap work (2 : 1 items left the one behind)
    auto dw_chunk = [&](int kd) {
        unsigned char* const bt = s_ball + kd * BCH_BYTES;
        const unsigned char* const hs = s_halo + (kd & 1) * HALO_BYTES;
        const unsigned char* const ws = s_dww + (kd & 1) * DWW_BYTES;
        const float* const bsrc = reinterpret_cast<const float*>(s_dwb + (kd & 1) * DWB_BYTES) + g * 8;
        {
            const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
            u32x4 wv[9];
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
                wv[t9] = *reinterpret_cast<const u32x4*>(ws + (t9 * CK + g * 8) * 2);
            const int pix = tid / CG;
            const int py = pix / TW, px = pix - py * TW;
            const unsigned char* xs = hs + ((py * IW + px) * CG + g) * 16;
            float v[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
            // one tap row at a time (the fence keeps hipcc from hoisting all nine reads: 24 registers the accumulators need)
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                u32x4 x[3];
#pragma unroll
                for (int tc = 0; tc < 3; ++tc)
                    x[tc] = *reinterpret_cast<const u32x4*>(xs + ((tr * D) * IW + tc * D) * CG * 16);
#pragma unroll
                for (int tc = 0; tc < 3; ++tc)
                    mac8_f16(v, x[tc], wv[tr * 3 + tc]);
                asm volatile("" ::: "memory");
            }
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                h[e] = (_Float16)dw_act<true>(v[e], 0.f, dw_hi);
            *reinterpret_cast<half8*>(bt + lds_off<CK>(pix, g)) = h;
        }
        {
            const int hf = (tid >> 3) & 1, pix = 64 + (tid >> 4);
            const float4 b0 = *reinterpret_cast<const float4*>(bsrc + hf * 4);
            uint2 wv[9];
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
                wv[t9] = *reinterpret_cast<const uint2*>(ws + (t9 * CK + g * 8 + hf * 4) * 2);
            const int py = pix / TW, px = pix - py * TW;
            const unsigned char* xs = hs + ((py * IW + px) * CG + g) * 16 + hf * 8;
            float v[4] = { b0.x, b0.y, b0.z, b0.w };
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                uint2 x[3];
#pragma unroll
                for (int tc = 0; tc < 3; ++tc)
                    x[tc] = *reinterpret_cast<const uint2*>(xs + ((tr * D) * IW + tc * D) * CG * 16);
#pragma unroll
                for (int tc = 0; tc < 3; ++tc)
                    mac4_f16(v, x[tc], wv[tr * 3 + tc]);
                asm volatile("" ::: "memory");
            }
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                h[e] = (_Float16)dw_act<true>(v[e], 0.f, dw_hi);
            *reinterpret_cast<half4*>(bt + lds_off<CK>(pix, g) + hf * 8) = h;
        }
    };
    auto mm_chunk = [&](int kc) {
        const unsigned char* const bt = s_ball + kc * BCH_BYTES;
        const __half* const wn = wbase + (size_t)min(kc + 1, NCH - 1) * (KS * 512);
        half8 fb[2][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[0][j] = *reinterpret_cast<const half8*>(bt + lds_off<CK>(j * 32 + frow, fk));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const half8*>(bt + lds_off<CK>(j * 32 + frow, (ks + 1) * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                half8 fa;
                __builtin_memcpy(&fa, &a[ks][i], 16);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 1][j], acc[i][j], 0, 0, 0);
            }
            a_load(wn, ks);
        }
    };

    int dbg_i = 0;
    // block 0: stamps of wavefront 0 (taps first) -> dbg[0..40); block 1: of wavefront 4 (MFMAs first) -> dbg[2112..2152)
#define HP_STAMP()                                                                                   \
    if (p.pw.dbg && blockIdx.x < 2 && tid == (int)blockIdx.x * 256 && dbg_i < 40)                    \
        p.pw.dbg[blockIdx.x * 2112 + dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    if (p.pw.dbg && tid == 0 && blockIdx.x < 1024) // every block's start / end on the shared 100 MHz clock
        p.pw.dbg[64 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    hload(0);
    to_lds(0);
    hload(min(1, NCH - 1));
    lds_barrier();
    HP_STAMP();
    dw_chunk(0);
    if (NCH > 1) {
        to_lds(1);
        hload(min(2, NCH - 1));
    }
    lds_barrier();
    HP_STAMP();
    const bool taps_first = wave < 4;
#pragma unroll 1
    for (int k = 0; k < NCH; ++k) {
        const bool more = k + 1 < NCH;
        if (k + 2 < NCH) { // halo chunk k+2 -> the buffer chunk k's taps were done with before the last barrier
            to_lds(k + 2);
            hload(min(k + 3, NCH - 1));
        }
        HP_STAMP();
        if (more && taps_first)
            dw_chunk(k + 1);
        HP_STAMP();
        mm_chunk(k);
        HP_STAMP();
        if (more && !taps_first)
            dw_chunk(k + 1);
        HP_STAMP();
        lds_barrier();
        HP_STAMP();
    }
    int pb[NT], py[NT], px[NT];
    bool pv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + (lane & 31);
        pb[j] = b;
        py[j] = y0 + n / TW;
        px[j] = x0 + n % TW;
        pv[j] = py[j] < p.OH && px[j] < p.OW;
    }
    // (the loop's last barrier put every wavefront past its reads of B_all: the slabs may overwrite it)
    conv_epilogue_packed<TP, NT>(p.pw, acc, (wave * TP) * 32, lane, lds + wave * packed_geom<TP, NT>::WAVE_BYTES, pb, py, px, pv);
    HP_STAMP();
    if (p.pw.dbg && tid == 0 && blockIdx.x < 1024)
        p.pw.dbg[65 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#undef HP_STAMP
}

// ---------------------------------------------------------------------------------------------------
// The 512-output separable blocks with the depthwise taps INSIDE the matrix-pipe stream of the same wavefront (round 4; the kernel is
// sepconv_pipe3_kernel below, this is its schedule).  The anti-phase form above lets the two wavefronts of a SIMD overlap each other, but
// every wavefront still runs its own taps (1.6-1.8 k cycles, latency-bound on its LDS reads) and its own 24 MFMAs (1.05-1.3 k) back to
// back: the interval between two barriers is their SUM (2.85 k).  Here interval k of a wavefront is ONE instruction
// stream of 24 slots - one MFMA of chunk k, then a few tap instructions of chunk k + 1, the order pinned by sched_barrier - so the tap
// reads' LDS latency sits under the MFMAs.  The tap work is re-cut so that it is the same in every thread and re-uses its operands:
//   thread = (channel quad q of the 64-channel chunk, column, row triple): three vertically adjacent output pixels x 4 channels.
//   The 3 + 2 D input rows of the triple are read once (15 / 21 ds_read_b64 instead of 27) and each feeds every output row it is a tap
//   of; the 9 x 4 depthwise weights of the quad are read once per chunk.  108 v_fma_mix_f32 per thread and chunk as before (bias folded
//   into the first tap, tap order (ky, kx) ascending per output: the same fp32 sums, bit for bit, as dwconv3x3_kernel), one v_med3 +
//   rounding, one ds_write_b64 per output pixel into the swizzled B tile.
template <int... I, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int D, int FIRST>
struct tap_sched {
    static constexpr int ROWS = 3 + 2 * D; // input rows of a triple of output rows
    // the nine (output row o, kernel row t) pairs in the order their input row o + t D arrives; within an output the kernel rows ascend
    static constexpr int pair_o(int i) { return D == 1 ? (int[9]){ 0, 0, 1, 0, 1, 2, 1, 2, 2 }[i] : (int[9]){ 0, 1, 2, 0, 1, 2, 0, 1, 2 }[i]; }
    static constexpr int pair_t(int i) { return D == 1 ? (int[9]){ 0, 1, 0, 2, 1, 0, 2, 1, 2 }[i] : (int[9]){ 0, 0, 0, 1, 1, 1, 2, 2, 2 }[i]; }
    static constexpr int pair_row(int i) { return pair_o(i) + pair_t(i) * D; }
    // program of 30 units: 27 column steps (pair i, column c) and, after the last step of an output row, its activation + store
    static constexpr int N_UNITS = 30;
    // unit u -> kind: >= 0: column step index 0 .. 26; -1 - o: finish output row o
    static constexpr int unit_kind(int u)
    {
        int n = 0;
        for (int i = 0; i < 9; ++i) {
            for (int c = 0; c < 3; ++c, ++n)
                if (n == u)
                    return i * 3 + c;
            if (pair_t(i) == 2) { // the last kernel row of output pair_o(i)
                if (n == u)
                    return -1 - pair_o(i);
                ++n;
            }
        }
        return 1 << 20;
    }
    static constexpr int FIRST_SLOT = FIRST, SLOTS = 24;
    static constexpr int unit_slot(int u) { return FIRST_SLOT + u * (SLOTS - FIRST_SLOT) / N_UNITS; }
    // the unit after which kernel row t's three weight registers are dead (its last pair's last column): they are re-loaded in place
    // with the NEXT chunk's weights right there, a good part of an interval before their first use
    static constexpr int last_unit_of_kernel_row(int t)
    {
        int last = -1;
        for (int u = 0; u < N_UNITS; ++u) {
            const int k = unit_kind(u);
            if (k >= 0 && pair_t(k / 3) == t)
                last = u;
        }
        return last;
    }
    static constexpr int last_bias_unit() // the first column step of the last output row to start: the bias registers are dead after it
    {
        int last = -1;
        for (int u = 0; u < N_UNITS; ++u) {
            const int k = unit_kind(u);
            if (k >= 0 && pair_t(k / 3) == 0 && k % 3 == 0)
                last = u;
        }
        return last;
    }
};

// ---------------------------------------------------------------------------------------------------
// sepconv_pipe3_kernel: the 24-slot stream above with its LDS traffic cut to what the stream can hide.  The timeline of its first form
// (tap reads as plain 8-byte loads of one base pointer, HP_SEP_DBG) put a slot at 90 - 95 cycles where the two
// wavefronts of a SIMD need 64 for their MFMAs:
// the LDS was busy ~1.7 k cycles per interval (ds_read2_b64, which hipcc makes of neighbouring 8-byte reads, costs 8 LDS cycles where
// two ds_read_b64 cost 2 + 2; nine 8-byte weight reads per thread; three ds_write_b128 of the halo in ONE slot by all eight wavefronts).
//   * every 8-byte read of the taps comes from its own base register (an integer laundered through an empty asm and cast back to an
//     LDS pointer) and halo rows are 17 pixels apart (2176 B > ds_read2's reach): no pairing, 2 cycles each;
//   * the depthwise weights lie [chunk][quad][9 taps x 4] (80 B per quad): four ds_read_b128 + one ds_read_b64 per thread and chunk;
//   * the halo pieces are stored one per slot in three different slots;
//   * the chunk loop is unrolled by two, so the buffer parity is a compile-time constant and every LDS address is a base register that
//     never changes plus an immediate (the v2 loop spent 31 v_add_u32 per interval on them).  Needs an even number of chunks.
template <int D, bool DBG = false, bool TAIL = true> // TAIL: the epilogue inside the last interval (relu / relu6, no residual)
__global__ __launch_bounds__(512, 1) void sepconv_pipe3_kernel(const sep_params p, int tiles_x, int tiles_y)
{
    using S = tap_sched<D, 1>;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const u32x2* lds_u2;
    typedef __attribute__((address_space(3))) const u32x4* lds_u4;
    constexpr int NW = 8, NTHR = 512, TP = 2, TH = 12, TW = 8, NPX = TH * TW, NT = NPX / 32, CK = 64, CG = 8, KS = 4, CMAX = 512;
    constexpr int IH = TH + 2 * D, IW = TW + 2 * D, PIECES = IH * IW * CG, NLD = (PIECES + NTHR - 1) / NTHR;
    constexpr int ROWPX = 17, ROWB = ROWPX * CG * 16; // halo rows 2176 B apart: two 8-byte reads of neighbouring rows cannot become one ds_read2_b64
    static_assert(IW <= ROWPX && ROWB / 8 > 255, "halo row stride");
    constexpr int HALO_BYTES = (IH * ROWPX + 1) * CG * 16; // (+ one pixel: the dummy slot of the pieces past the tile)
    constexpr int BCH_BYTES = NPX * CK * 2;
    constexpr int QW = 80, DWW_CHUNK = 16 * QW, DWW_BYTES = (CMAX / CK) * DWW_CHUNK, DWB_BYTES = CMAX * 4; // [chunk][quad][9 x 4 halves + pad]
    constexpr int OFF_HALO = 2 * BCH_BYTES, OFF_DWW = OFF_HALO + 2 * HALO_BYTES, OFF_DWB = OFF_DWW + DWW_BYTES;
    constexpr int MAIN_BYTES = OFF_DWB + DWB_BYTES;
    constexpr int EPI_BYTES = NW * packed_geom<TP, NT>::WAVE_BYTES;
    constexpr int TAIL_BYTES = OFF_HALO + NW * NT * packed_geom<TP, NT>::SLAB; // the fused tail's slabs lie behind the B tiles
    constexpr int LDS_BYTES = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) > TAIL_BYTES ? (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) : TAIL_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(256))) unsigned char lds[LDS_BYTES];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ymax = p.H + p.halo - 1, xmax = p.W + p.halo - 1;
    const int C = p.C, KQ = C / 16, NCH = C / CK;

    const __half* const wbase = p.pw.w + (size_t)lane * 8 + (size_t)(wave * TP) * KQ * 512;
    const size_t row_stride = (size_t)KQ * 512;
    u32x4 a[KS][TP];
    auto a_load = [&](const __half* wp, int ks) {
#pragma unroll
        for (int i = 0; i < TP; ++i)
            a[ks][i] = *reinterpret_cast<const u32x4*>(wp + i * row_stride + ks * 512);
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        a_load(wbase, ks);
    u32x4 hv[NLD];
    int hoff[NLD];
    unsigned hdst[NLD]; // LDS byte offset of the piece inside a halo buffer
    {   // halo pixels beyond the tensor's own zero halo (ragged bottom / right tiles) are read from its last halo row / column, which is
        // zero as well (sepconv_variant: halo >= dilation >= 1) - no mask needed
        const int iy0 = y0 - p.pad_t, ix0 = x0 - p.pad_l;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * NTHR, ic = min(i, PIECES - 1);
            const int hp = ic / CG, c = ic - hp * CG;
            const int hy = hp / IW, hx = hp - hy * IW;
            hoff[k] = (min(iy0 + hy, ymax) * p.in.wp + min(ix0 + hx, xmax)) * p.in.cs + c * 8;
            hdst[k] = lds0 + OFF_HALO + (i < PIECES ? ((hy * ROWPX + hx) * CG + c) * 16 : (IH * ROWPX * CG + c) * 16);
        }
    }
    const __half* const hbase = p.in.p + (size_t)b * p.in.img * p.in.cs + p.in.coff;
    typedef __attribute__((address_space(3))) u32x4* lds_w4;
    auto hload1 = [&](int k, int chunk) { hv[k] = *reinterpret_cast<const u32x4*>(hbase + hoff[k] + chunk * CK); };
    auto hstore1 = [&](int k, auto par_) { *(lds_w4)(hdst[k] + decltype(par_)::value * HALO_BYTES) = hv[k]; };

    floatx16 acc[TP][NT]; // (not cleared: the first k16 step of chunk 0 multiplies onto a literal zero)
    const int frow = lane & 31, fk = lane >> 5;
    const float dw_hi = p.dw_hi;
    // B-fragment address of k16 step 0; step ks is this XOR ks * 32 (the swizzled 16-byte slot is (2 ks + fk) ^ key = (fk ^ key) ^ 2 ks, the
    // array is 256-byte aligned), pixel tile j adds j * 4096
    const unsigned fbase0 = lds0 + lds_off<CK>(frow, fk);
    // tap role of this thread: channel quad q, column tcol, output rows 3 trow .. 3 trow + 2 of the tile
    const int q = tid & 15, tcol = (tid >> 4) & 7, trow = tid >> 7;
    unsigned xb[3]; // one base register per tap column, laundered so that hipcc cannot pair the reads
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        xb[c] = lds0 + OFF_HALO + ((3 * trow * ROWPX + tcol + c * D) * CG) * 16 + q * 8;
        asm volatile("" : "+v"(xb[c]));
    }
    // where the three outputs go in a B tile: rows of 8 pixels are 1024 B apart and alternate the swizzle key by 4 (output 2 = output 0 + 2048)
    const unsigned b_dst0 = lds0 + lds_off<CK>((3 * trow) * TW + tcol, q >> 1) + (q & 1) * 8;
    const unsigned b_dst1 = lds0 + lds_off<CK>((3 * trow + 1) * TW + tcol, q >> 1) + (q & 1) * 8;
    const unsigned wq = lds0 + OFF_DWW + q * QW, bq = lds0 + OFF_DWB + q * 16;
    // depthwise weights / bias of the chunk whose taps run next: taps (2g, 2g + 1) in wv[g], tap 8 in w8
    u32x4 wv[4];
    uint2 w8;
    float4 bs;
    auto tap_w = [&](auto t9_) -> uint2 {
        constexpr int t9 = decltype(t9_)::value;
        if constexpr (t9 == 8)
            return w8;
        else
            return (t9 & 1) ? make_uint2(wv[t9 / 2][2], wv[t9 / 2][3]) : make_uint2(wv[t9 / 2][0], wv[t9 / 2][1]);
    };
    auto w_load_lo = [&](int chunk) { // taps 0 .. 5 (kernel rows 0 and 1)
#pragma unroll
        for (int g = 0; g < 3; ++g)
            wv[g] = *(lds_u4)(wq + chunk * DWW_CHUNK + g * 16);
    };
    auto w_load_hi = [&](int chunk) { // taps 6 .. 8 (kernel row 2)
        wv[3] = *(lds_u4)(wq + chunk * DWW_CHUNK + 48);
        const u32x2 t2 = *(lds_u2)(wq + chunk * DWW_CHUNK + 64);
        w8 = make_uint2(t2.x, t2.y);
    };
    auto b_load = [&](int chunk) {
        const u32x4 t4 = *(lds_u4)(bq + chunk * (CK * 4));
        bs = make_float4(__uint_as_float(t4[0]), __uint_as_float(t4[1]), __uint_as_float(t4[2]), __uint_as_float(t4[3]));
    };

    int dbg_i = 0;
    // one chunk interval: MFMAs of chunk kc out of B buffer PAR, taps of chunk kd = kc + 1 out of halo buffer
    // 1 - PAR into B buffer 1 - PAR, halo chunk kd + 1 -> halo buffer PAR, weights of chunk knext in place
    auto interval = [&](auto mm_, auto taps_, auto par_, int kc, int kd, int knext, auto first_) {
        constexpr bool MM = decltype(mm_)::value, TAPS = decltype(taps_)::value, FIRST = decltype(first_)::value;
        constexpr int PAR = decltype(par_)::value, TPAR = MM ? 1 - PAR : PAR; // (taps only: the prologue's chunk 0 in buffer PAR)
        constexpr bool STAGE = MM && TAPS;
        const __half* const wn = wbase + (size_t)min(kc + 1, NCH - 1) * (KS * 512);
        half8 fb[NT]; // ONE set: fragment j of step ks + 1 is requested right after the last MFMA of step ks that reads fragment j
        uint2 xr[S::ROWS][3];
        float v[3][4];
        auto read_row = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
#pragma unroll
            for (int c = 0; c < 3; ++c)
            {
                const u32x2 t2 = *(lds_u2)(xb[c] + TPAR * HALO_BYTES + r * ROWB);
                xr[r][c] = make_uint2(t2.x, t2.y);
            }
        };
        auto read_fb = [&](auto ks_, auto j_) {
            constexpr int ks = decltype(ks_)::value, j = decltype(j_)::value;
            const u32x4 t4 = *(lds_u4)((fbase0 ^ (ks * 32)) + PAR * BCH_BYTES + j * 32 * CK * 2);
            __builtin_memcpy(&fb[j], &t4, 16);
        };
        if (MM) { // the first MFMA's operands first: its s_waitcnt then only covers these three reads
            static_for<NT>([&](auto j_) { read_fb(std::integral_constant<int, 0>{}, j_); });
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAPS) {
            read_row(std::integral_constant<int, 0>{});
            read_row(std::integral_constant<int, 1>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<S::SLOTS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int ks = s / (TP * NT), i = (s % (TP * NT)) / NT, j = s % NT;
            if constexpr (MM) {
                half8 fa;
                __builtin_memcpy(&fa, &a[ks][i], 16);
                if constexpr (FIRST && ks == 0)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], floatx16{}, 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], acc[i][j], 0, 0, 0);
                if constexpr (i == TP - 1 && ks + 1 < KS) // fragment j is free: the next step's, three slots ahead of its first MFMA
                    read_fb(std::integral_constant<int, (ks + 1) % KS>{}, std::integral_constant<int, j>{});
                if constexpr (s % (TP * NT) == TP * NT - 1)
                    a_load(wn, ks);
            }
            if constexpr (TAPS) {
                static_for<S::N_UNITS>([&](auto u_) {
                    constexpr int u = decltype(u_)::value;
                    if constexpr (S::unit_slot(u) == s) {
                        constexpr int kind = S::unit_kind(u);
                        if constexpr (kind >= 0) {
                            constexpr int pi = kind / 3, c = kind % 3, o = S::pair_o(pi), tr = S::pair_t(pi), r = S::pair_row(pi);
                            // input rows are requested two rows ahead of their first use
                            if constexpr (c == 0 && r + 2 < S::ROWS && (pi == 0 || S::pair_row(pi - 1) != r))
                                read_row(std::integral_constant<int, r + 2>{});
                            if constexpr (tr == 0 && c == 0)
                                mac4_f16_init(v[o], xr[r][c], tap_w(std::integral_constant<int, 0>{}), bs);
                            else
                                mac4_f16(v[o], xr[r][c], tap_w(std::integral_constant<int, tr * 3 + c>{}));
                            // the next chunk's weights / bias into the registers that just died
                            if constexpr (u == S::last_unit_of_kernel_row(1))
                                w_load_lo(knext);
                            if constexpr (u == S::last_unit_of_kernel_row(2))
                                w_load_hi(knext);
                            if constexpr (u == S::last_bias_unit())
                                b_load(knext);
                        } else {
                            constexpr int o = -1 - kind;
                            half4 h;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                h[e] = (_Float16)dw_act<true>(v[o][e], 0.f, dw_hi);
                            typedef __attribute__((address_space(3))) half4* lds_h4;
                            *(lds_h4)((o == 1 ? b_dst1 : b_dst0) + (o == 2 ? 2 * TW * CK * 2 : 0) + TPAR * BCH_BYTES) = h;
                        }
                    }
                });
            }
            // halo chunk kd + 1: one piece per slot, registers -> LDS, then the piece of chunk kd + 2 is requested.  MID-interval: at the
            // top of the loop the store's s_waitcnt would have to be vmcnt(0) - the waitcnt pass merges the loop entry, where the halo
            // loads are the youngest requests, with the back edge - and would drain the weight-fragment prefetch every iteration; here
            // the merged count only covers fragments requested half an interval earlier.  Every staging store is unconditional for the
            // same reason (a branch around one drains vmcnt at the join): the pieces past the tile go to a dummy slot.
            if constexpr (STAGE && s >= 10 && s < 10 + 2 * NLD && (s - 10) % 2 == 0) {
                hstore1((s - 10) / 2, std::integral_constant<int, PAR>{});
                hload1((s - 10) / 2, min(kd + 2, NCH - 1));
            }
            if constexpr (DBG && MM && TAPS && (s == 0 || s == 3 || s == 7 || s == 11 || s == 12 || s == 15 || s == 19 || s == 23)) {
                // (timeline build only: stamps of intervals 2 and 3 of wavefront 0 of block 0 / wavefront 4 of block 1)
                if ((kc == 2 || kc == 3) && blockIdx.x < 2 && tid == (int)blockIdx.x * 256 && dbg_i < 40)
                    p.pw.dbg[blockIdx.x * 2112 + dbg_i++] = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};

#define HP_STAMP()                                                                                   \
    if (p.pw.dbg && blockIdx.x < 2 && tid == (int)blockIdx.x * 256 && dbg_i < 40)                    \
        p.pw.dbg[blockIdx.x * 2112 + dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    if (p.pw.dbg && tid == 0 && blockIdx.x < 1024) // every block's start / end on the shared 100 MHz clock
        p.pw.dbg[64 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < NLD; ++k)
        hload1(k, 0);
    {   // all depthwise weights ([9][C] halves in HBM -> [chunk][quad][9 x 4]) and biases of the block, once: thread cq < C / 4 takes the
        // nine taps of channel quad cq (no index arithmetic), and every request goes out before the first store - a load -> store loop
        // pays one memory latency per trip
        if (tid < C / 4) { // (C / 4 is a multiple of 32 and at most 128: the first one or two wavefronts)
            const int cq = tid;
            uint2 dv[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                dv[tp] = *reinterpret_cast<const uint2*>(p.dw_w + (size_t)tp * C + cq * 4);
            const u32x4 bv = *reinterpret_cast<const u32x4*>(p.dw_bias + cq * 4);
            unsigned char* const dq = lds + OFF_DWW + (cq >> 4) * DWW_CHUNK + (cq & 15) * QW;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                *reinterpret_cast<uint2*>(dq + tp * 8) = dv[tp];
            *reinterpret_cast<u32x4*>(lds + OFF_DWB + cq * 16) = bv;
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        hstore1(k, P0);
        hload1(k, min(1, NCH - 1));
    }
    lds_barrier();
    HP_STAMP();
    w_load_lo(0);
    w_load_hi(0);
    b_load(0);
    interval(NO, YES, P0, 0, 0, min(1, NCH - 1), NO); // taps of chunk 0: halo buffer 0 -> B buffer 0
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        hstore1(k, P1);
        hload1(k, min(2, NCH - 1));
    }
    lds_barrier();
    HP_STAMP();
    // NCH - 1 combined intervals (an odd number: NCH is even), buffer parity = k & 1
    interval(YES, YES, P0, 0, 1, min(2, NCH - 1), YES);
    lds_barrier();
#pragma unroll 1
    for (int k = 1; k + 1 < NCH; k += 2) {
        interval(YES, YES, P1, k, k + 1, min(k + 2, NCH - 1), NO);
        lds_barrier();
        interval(YES, YES, P0, k + 1, k + 2, min(k + 3, NCH - 1), NO);
        lds_barrier();
    }
    float4 ebias[TP][4]; // the epilogue's bias vectors: requested before the last interval's MFMAs (the taps' registers are free there)
    packed_bias<TP>(p.pw, (wave * TP) * 32, lane, ebias);
    int pb[NT], py[NT], px[NT];
    bool pv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + (lane & 31);
        pb[j] = b;
        py[j] = y0 + n / TW;
        px[j] = x0 + n % TW;
        pv[j] = py[j] < p.OH && px[j] < p.OW;
    }
    // The last interval has no taps to hide: with relu / relu6 and no residual (every block of the network) its vector slots take the
    // epilogue instead.  The MFMAs run pixel tile by pixel tile (j, then k16 step, then row tile), so tile j's accumulators are final
    // after slot 8 j + 7; bias + clamp + fp16 + the slab write of tile j - 1 (conv_epilogue_packed's arithmetic, unit = one (row tile,
    // channel group)) sit behind the MFMAs of tile j, the global stores of tile j - 2 behind those.  The slabs lie over the halo buffers
    // and the depthwise weights, which nobody reads any more; the B tile of the last chunk stays where it is.
    using EG = packed_geom<TP, NT>;
    static_assert(OFF_HALO + NW * NT * EG::SLAB <= LDS_BYTES, "the fused tail's slabs");
    if constexpr (TAIL) {
        unsigned char* const slab = lds + OFF_HALO + wave * (NT * EG::SLAB);
        const float hi = p.pw.act_hi;
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                reinterpret_cast<long*>(slab + j * EG::SLAB + 32 * EG::ROW)[lane] = pv[j] ? tv_off(p.pw.out, pb[j], py[j], px[j]) : -1;
        }
        const int echunk = lane % EG::CPP, eprow = lane / EG::CPP;
        const int emc = (wave * TP) * 32 + echunk * 8;
        const bool emvalid = emc < p.pw.Cout;
        half8 fbq[2]; // B fragments in MFMA order: n = j * KS + ks, two in flight
        auto read_fbq = [&](auto n_) {
            constexpr int n = decltype(n_)::value, j = n / KS, ks = n % KS;
            const u32x4 t4 = *(lds_u4)((fbase0 ^ (ks * 32)) + BCH_BYTES + j * 32 * CK * 2);
            __builtin_memcpy(&fbq[n & 1], &t4, 16);
        };
        auto unit = [&](auto j_, auto u_) { // bias + clamp + fp16 of accumulator registers 4 g .. 4 g + 3 of tile (i, j) -> slab j
            constexpr int j = decltype(j_)::value, i = decltype(u_)::value / 4, g = decltype(u_)::value % 4;
            half4 h;
            h[0] = (_Float16)__builtin_amdgcn_fmed3f(acc[i][j][4 * g + 0] + ebias[i][g].x, 0.f, hi);
            h[1] = (_Float16)__builtin_amdgcn_fmed3f(acc[i][j][4 * g + 1] + ebias[i][g].y, 0.f, hi);
            h[2] = (_Float16)__builtin_amdgcn_fmed3f(acc[i][j][4 * g + 2] + ebias[i][g].z, 0.f, hi);
            h[3] = (_Float16)__builtin_amdgcn_fmed3f(acc[i][j][4 * g + 3] + ebias[i][g].w, 0.f, hi);
            *reinterpret_cast<half4*>(slab + j * EG::SLAB + (lane & 31) * EG::ROW + (i * 32 + 8 * g + 4 * (lane >> 5)) * 2) = h;
        };
        auto pass = [&](auto j_, auto ps_) { // one store pass of slab j: 8 pixels x this wavefront's 64 channels
            constexpr int j = decltype(j_)::value, ps = decltype(ps_)::value;
            if constexpr (ps == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's slab writes have landed (DS ops retire in order)
                __builtin_amdgcn_wave_barrier();
            }
            const unsigned char* const sj = slab + j * EG::SLAB;
            const int pix = ps * EG::PPP + eprow;
            const half8 h = *reinterpret_cast<const half8*>(sj + pix * EG::ROW + echunk * 16);
            const long oo = reinterpret_cast<const long*>(sj + 32 * EG::ROW)[pix];
            if (oo >= 0 && emvalid)
                *reinterpret_cast<half8*>(p.pw.out.p + oo + emc) = h;
        };
        read_fbq(std::integral_constant<int, 0>{});
        read_fbq(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<TP * NT * KS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int j = s / (TP * KS), ks = (s % (TP * KS)) / TP, i = s % TP, n = j * KS + ks;
            half8 fa;
            __builtin_memcpy(&fa, &a[ks][i], 16);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fbq[n & 1], acc[i][j], 0, 0, 0);
            if constexpr (i == TP - 1 && n + 2 < NT * KS) // this fragment's last reader has been issued: its register takes fragment n + 2
                read_fbq(std::integral_constant<int, (n + 2 < NT * KS ? n + 2 : 0)>{});
            if constexpr (j >= 1)
                unit(std::integral_constant<int, (j >= 1 ? j - 1 : 0)>{}, std::integral_constant<int, s % (TP * KS)>{});
            if constexpr (j >= 2 && (s % (TP * KS)) % 2 == 0)
                pass(std::integral_constant<int, (j >= 2 ? j - 2 : 0)>{}, std::integral_constant<int, (s % (TP * KS)) / 2>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        HP_STAMP();
        static_for<TP * 4>([&](auto u_) { unit(std::integral_constant<int, NT - 1>{}, u_); });
        static_for<EG::PASSES>([&](auto ps_) { pass(std::integral_constant<int, NT - 2>{}, ps_); });
        static_for<EG::PASSES>([&](auto ps_) { pass(std::integral_constant<int, NT - 1>{}, ps_); });
    } else {
        interval(YES, NO, P1, NCH - 1, 0, 0, NO);
        HP_STAMP();
        lds_barrier(); // (every wavefront past its reads of the B tiles: the slabs may overwrite them)
        conv_epilogue_packed<TP, NT>(p.pw, acc, (wave * TP) * 32, lane, lds + wave * packed_geom<TP, NT>::WAVE_BYTES, pb, py, px, pv, ebias);
    }
    HP_STAMP();
    if (p.pw.dbg && tid == 0 && blockIdx.x < 1024)
        p.pw.dbg[65 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#undef HP_STAMP
}

template <int D, int CMAX>
static hipError_t launch_sep_pipe(const sep_params& p, hipStream_t s)
{
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 11) / 12;
    // the round-3 anti-phase form: A/B switch (HP_SEP_PIPE1=1) and the blocks with an odd number of 64-channel chunks
    static const bool anti_phase = getenv("HP_SEP_PIPE1") != nullptr;
    if (anti_phase || p.C % 128)
        HP_LAUNCH((sepconv_pipe_kernel<D, CMAX>), dim3(tiles_x * tiles_y * p.B), dim3(512), 0, s, p, tiles_x, tiles_y);
    else if (p.pw.res.p || p.pw.alpha || p.pw.act_slope != 0.f) // (no block of the built-in networks: the general epilogue after the last interval)
        HP_LAUNCH((sepconv_pipe3_kernel<D, false, false>), dim3(tiles_x * tiles_y * p.B), dim3(512), 0, s, p, tiles_x, tiles_y);
    else if (p.pw.dbg) // HP_SEP_DBG: the build with s_memtime stamps inside the interval (tools/sep_timeline.py)
        HP_LAUNCH((sepconv_pipe3_kernel<D, true>), dim3(tiles_x * tiles_y * p.B), dim3(512), 0, s, p, tiles_x, tiles_y);
    else
        HP_LAUNCH((sepconv_pipe3_kernel<D>), dim3(tiles_x * tiles_y * p.B), dim3(512), 0, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The separable blocks of the high-resolution end of the MobileNet backbones (32 -> 64 at 184x216, 64 -> 128 stride 2,
// 128 -> 128 at 92x108: the largest activations of the network, little arithmetic per byte) as ONE small block per 8 x 8
// output pixels with ALL channels of the tile in LDS at once: halo tile -> LDS, depthwise taps (one thread = one pixel x 8
// channels per item, same arithmetic as dwconv3x3_kernel) -> swizzled B tile, pointwise GEMM with K = C (2 / 4 / 8 k16 steps,
// fragment-ordered weights straight from L2), staged epilogue.  No K chunks, two barriers, < 128 registers and < 50 KB of LDS:
// three or more blocks per CU, so loads, taps and stores of different blocks overlap.
//   C = input channels (32 / 64 / 128), S = stride, RT = 32-row tiles of output channels (2: <= 64 outputs, wavefronts as
//   2 row tiles x 2 pixel halves; 4: <= 128 outputs, one row tile x both pixel halves per wavefront).
template <int C, int S, int RT>
__global__ __launch_bounds__(256) void sepconv_small_kernel(const sep_params p, int tiles_x, int tiles_y)
{
    constexpr int TH = 8, TW = 8, CG = C / 8, KQ = C / 16, NT = RT == 4 ? 2 : 1;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, PIECES = IH * IW * CG, NLD = (PIECES + 255) / 256;
    constexpr int ITEMS = TH * TW * CG / 256;
    constexpr int HALO_BYTES = PIECES * 16, B_BYTES = TH * TW * C * 2, DWW_BYTES = 9 * C * 2, DWB_BYTES = C * 4;
    constexpr int SLAB_BYTES = 4 * packed_geom<1, NT>::WAVE_BYTES;
    constexpr bool OVERLAY = HALO_BYTES >= SLAB_BYTES; // the epilogue slabs reuse the halo tile (dead after the taps) when it is large enough
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_BYTES + B_BYTES + DWW_BYTES + DWB_BYTES + (OVERLAY ? 0 : SLAB_BYTES)];
    unsigned char* const s_halo = lds;
    unsigned char* const s_b = s_halo + HALO_BYTES;
    unsigned char* const s_dww = s_b + B_BYTES;
    unsigned char* const s_dwb = s_dww + DWW_BYTES;
    unsigned char* const s_slab = OVERLAY ? s_halo : s_dwb + DWB_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = RT == 4 ? wave : (wave & 1), wn = RT == 4 ? 0 : (wave >> 1); // row tile, first pixel tile
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ymax = p.H + p.halo - 1, xmax = p.W + p.halo - 1;

    // pointwise weights: row tile wm, all k16 steps (fragment order)
    u32x4 a[KQ];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks)
        a[ks] = *reinterpret_cast<const u32x4*>(p.pw.w + ((size_t)(wm * KQ + ks) * 64 + lane) * 8);

    // halo tile + depthwise weights / bias -> LDS
    {
        const __half* const hbase = p.in.p + (size_t)b * p.in.img * p.in.cs + p.in.coff;
        const int iy0 = y0 * S - p.pad_t, ix0 = x0 * S - p.pad_l;
        u32x4 hv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = min(tid + k * 256, PIECES - 1);
            const int hp = i / CG, c = i - hp * CG;
            const int hy = hp / IW, hx = hp - hy * IW;
            const int y = iy0 + hy, x = ix0 + hx;
            const bool ok = y <= ymax && x <= xmax; // y, x >= -halo: inside the zero halo of the HBM tensor
            const u32x4 v = *reinterpret_cast<const u32x4*>(hbase + (size_t)(min(y, ymax) * p.in.wp + min(x, xmax)) * p.in.cs + c * 8);
            hv[k] = v & (ok ? 0xffffffffu : 0u);
        }
        constexpr int NWT = 9 * CG, NBT = C / 4; // 16-byte pieces of the [9][C] weights and of the C biases
        u32x4 wreg[(NWT + NBT + 255) / 256];
#pragma unroll
        for (int k = 0; k < (NWT + NBT + 255) / 256; ++k) {
            const int i = tid + k * 256;
            wreg[k] = u32x4{ 0, 0, 0, 0 };
            if (i < NWT)
                wreg[k] = *reinterpret_cast<const u32x4*>(p.dw_w + (size_t)(i / CG) * C + (i % CG) * 8);
            else if (i < NWT + NBT)
                wreg[k] = *reinterpret_cast<const u32x4*>(p.dw_bias + (i - NWT) * 4);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * 256 < PIECES)
                *reinterpret_cast<u32x4*>(s_halo + (size_t)(tid + k * 256) * 16) = hv[k];
#pragma unroll
        for (int k = 0; k < (NWT + NBT + 255) / 256; ++k) {
            const int i = tid + k * 256;
            if (i < NWT)
                *reinterpret_cast<u32x4*>(s_dww + i * 16) = wreg[k];
            else if (i < NWT + NBT)
                *reinterpret_cast<u32x4*>(s_dwb + (i - NWT) * 16) = wreg[k];
        }
    }
    lds_barrier();
    // depthwise taps
    {
        const int g = tid % CG;
        const float* bsrc = reinterpret_cast<const float*>(s_dwb) + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
        u32x4 wv[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
            wv[t9] = *reinterpret_cast<const u32x4*>(s_dww + (t9 * C + g * 8) * 2);
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int pix = (tid + r * 256) / CG;
            const int py = pix / TW, px = pix - py * TW;
            const unsigned char* xs = s_halo + ((py * S * IW + px * S) * CG + g) * 16;
            float v[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(xs + ((t9 / 3) * IW + (t9 % 3)) * CG * 16);
                mac8_f16(v, x, wv[t9]);
            }
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                h[e] = (_Float16)dw_act<true>(v[e], 0.f, p.dw_hi);
            *reinterpret_cast<half8*>(s_b + lds_off<C>(pix, g)) = h;
        }
    }
    lds_barrier();
    // pointwise: D[32 rows of tile wm][NT x 32 pixels], K = C
    floatx16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[0][j][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
        half8 fa;
        __builtin_memcpy(&fa, &a[ks], 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const half8 fb = *reinterpret_cast<const half8*>(s_b + lds_off<C>((wn + j) * 32 + frow, ks * 2 + fk));
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0][j], 0, 0, 0);
        }
    }
    int pb[NT], py1[NT], px1[NT];
    bool pv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn + j) * 32 + frow;
        pb[j] = b, py1[j] = y0 + n / TW, px1[j] = x0 + n % TW;
        pv[j] = py1[j] < p.OH && px1[j] < p.OW;
    }
    // (where the slabs overlay the halo tile: the barrier after the taps already put every wavefront past its reads of it)
    conv_epilogue_packed<1, NT>(p.pw, acc, wm * 32, lane, s_slab + wave * packed_geom<1, NT>::WAVE_BYTES, pb, py1, px1, pv);
}

template <int C, int S, int RT>
static hipError_t launch_sep_small(const sep_params& p, hipStream_t s)
{
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 7) / 8;
    HP_LAUNCH((sepconv_small_kernel<C, S, RT>), dim3(tiles_x * tiles_y * p.B), dim3(256), 0, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Two consecutive separable blocks in ONE launch: block a (C0 -> C1, stride 1) and block b (C1 -> C2, stride 2) of the MobileNet
// stem (32 -> 64 at 184 x 216, 64 -> 128 down to 92 x 108 in LW-OpenPose: block a's output is the largest tensor of the network,
// 40.7 MB per batch of 8, written once and read once).  A block of 256 threads owns 4 x 8 output pixels of block b: the 9 x 17 pixels
// of block a's output under them are computed in LDS and never reach HBM (1.2 x the work of block a for 61 % of the two blocks' bytes).
//   S1  11 x 19 x C0 input halo (zero outside the image) + both depthwise weight sets -> LDS; pointwise A fragments -> registers
//   S2  depthwise 3x3 of block a on the 153 mid pixels -> swizzled B tile [160][C0]
//   S3  pointwise of block a: C1 / 32 row tiles x 5 pixel tiles, K = C0; bias + clamp + fp16 on the accumulators; mid pixels outside
//       the image are block b's zero padding -> [153][C1] (+ 16 B per pixel against bank conflicts of the stride-2 reads)
//   S4  depthwise 3x3 stride 2 of block b: one (pixel, 8 channels) per thread -> swizzled B tile [32][C1]
//   S5  pointwise of block b: one 32-row tile per wavefront, K = C1, packed epilogue
// Every value is computed by the same operations in the same order as sepconv_small_kernel computes it for the two blocks one after
// the other (fp32 tap chain from the bias, one rounding to fp16 per tensor, MFMA k16 steps ascending from zero): bit-identical.
template <int C0, int C1, int C2>
__global__ __launch_bounds__(256, 4) void sepconv_pair_kernel(const seppair_params p, int tiles_x, int tiles_y)
{
    constexpr int TH = 4, TW = 8, NO = TH * TW;
    constexpr int MH = (TH - 1) * 2 + 3, MW = (TW - 1) * 2 + 3, NM = MH * MW; // 9 x 17 mid pixels
    constexpr int IH = MH + 2, IW = MW + 2, NI = IH * IW;                      // 11 x 19 input pixels
    constexpr int G0 = C0 / 8, G1 = C1 / 8, KQ1 = C0 / 16, KQ2 = C1 / 16, NPT1 = (NM + 31) / 32, RT1 = C1 / 32;
    static_assert(NO == 32 && C2 == 128 && RT1 == 2 && 256 % G0 == 0 && NO * G1 == 256, "thread roles");
    constexpr int NLD = (NI * G0 + 255) / 256, ITEMS1 = (NM * G0 + 255) / 256;
    constexpr int H1_BYTES = NI * C0 * 2, B1_BYTES = NPT1 * 32 * C0 * 2;
    constexpr int PXB2 = C1 * 2 + 16, H2_BYTES = (NM * PXB2 + 15) / 16 * 16, B2_BYTES = NO * C1 * 2;
    constexpr int W1_BYTES = 9 * C0 * 2, BI1_BYTES = C0 * 4, W2_BYTES = 9 * C1 * 2, BI2_BYTES = C1 * 4;
    constexpr int NWP = (W1_BYTES + BI1_BYTES + W2_BYTES + BI2_BYTES) / 16;
    // LDS: region X = the input halo, then (S3 on) block a's output over it; region Y = block a's B tile, then (S4 on) block b's; each
    // is dead when its successor is written (a barrier in between).  34 KB: four blocks per CU.
    constexpr int X_BYTES = H1_BYTES > H2_BYTES ? H1_BYTES : H2_BYTES, Y_BYTES = B1_BYTES > B2_BYTES ? B1_BYTES : B2_BYTES;
    static_assert(NWP <= 256 && 4 * packed_geom<1, 1>::WAVE_BYTES <= X_BYTES, "weight pieces / epilogue slabs");
    __shared__ __attribute__((aligned(16))) unsigned char lds[X_BYTES + Y_BYTES + NWP * 16];
    unsigned char* const s_h1 = lds;
    unsigned char* const s_h2 = lds;
    unsigned char* const s_b1 = lds + X_BYTES;
    unsigned char* const s_b2 = lds + X_BYTES;
    unsigned char* const s_w1 = s_b1 + Y_BYTES;
    unsigned char* const s_bi1 = s_w1 + W1_BYTES;
    unsigned char* const s_w2 = s_bi1 + BI1_BYTES;
    unsigned char* const s_bi2 = s_w2 + W2_BYTES;

    const sep_params& A = p.a;
    const sep_params& Bk = p.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 31, fk = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int my0 = oy0 * 2 - Bk.pad_t, mx0 = ox0 * 2 - Bk.pad_l; // first mid pixel (may be -1: block b's padding)
    const int iy0 = my0 - A.pad_t, ix0 = mx0 - A.pad_l;

    // ---- S1: every request first
    const int rt1 = wave & 1;
    u32x4 a1[KQ1], a2[KQ2];
#pragma unroll
    for (int ks = 0; ks < KQ1; ++ks)
        a1[ks] = *reinterpret_cast<const u32x4*>(A.pw.w + ((size_t)(rt1 * KQ1 + ks) * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks)
        a2[ks] = *reinterpret_cast<const u32x4*>(Bk.pw.w + ((size_t)(wave * KQ2 + ks) * 64 + lane) * 8);
    float4 bs1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bs1[g] = *reinterpret_cast<const float4*>(A.pw.bias + rt1 * 32 + 8 * g + 4 * fk);
    {
        const __half* const hbase = A.in.p + (size_t)b * A.in.img * A.in.cs + A.in.coff;
        u32x4 hv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = min(tid + k * 256, NI * G0 - 1);
            const int hp = i / G0, c = i - hp * G0;
            const int hy = hp / IW, hx = hp - hy * IW;
            const int y = iy0 + hy, x = ix0 + hx;
            const bool ok = y >= 0 && y < A.H && x >= 0 && x < A.W;
            const u32x4 v = *reinterpret_cast<const u32x4*>(hbase + (size_t)(min(max(y, 0), A.H - 1) * A.in.wp + min(max(x, 0), A.W - 1)) * A.in.cs + c * 8);
            hv[k] = v & (ok ? 0xffffffffu : 0u);
        }
        u32x4 wreg = u32x4{ 0, 0, 0, 0 };
        {
            constexpr int E1 = W1_BYTES / 16, E2 = E1 + BI1_BYTES / 16, E3 = E2 + W2_BYTES / 16;
            const int i = min(tid, NWP - 1);
            const unsigned char* src = i < E1 ? reinterpret_cast<const unsigned char*>(A.dw_w) + i * 16
                : i < E2                      ? reinterpret_cast<const unsigned char*>(A.dw_bias) + (i - E1) * 16
                : i < E3                      ? reinterpret_cast<const unsigned char*>(Bk.dw_w) + (i - E2) * 16
                                              : reinterpret_cast<const unsigned char*>(Bk.dw_bias) + (i - E3) * 16;
            wreg = *reinterpret_cast<const u32x4*>(src);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (tid + k * 256 < NI * G0)
                *reinterpret_cast<u32x4*>(s_h1 + (size_t)(tid + k * 256) * 16) = hv[k];
        if (tid < NWP)
            *reinterpret_cast<u32x4*>(s_w1 + tid * 16) = wreg;
    }
    lds_barrier();

    // ---- S2: depthwise taps of block a (thread = channel group tid % G0 of the pixels (tid + 256 r) / G0)
    {
        const int g = tid % G0;
        const float* bsrc = reinterpret_cast<const float*>(s_bi1) + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
        u32x4 wv[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
            wv[t9] = *reinterpret_cast<const u32x4*>(s_w1 + (t9 * C0 + g * 8) * 2);
        const float dw_hi = A.dw_hi;
#pragma unroll
        for (int r = 0; r < ITEMS1; ++r) {
            const int pix = min((tid + r * 256) / G0, NM - 1); // (threads past the end redo the last pixel: same value to the same place)
            const int py = pix / MW, px = pix - py * MW;
            const unsigned char* xs = s_h1 + ((py * IW + px) * G0 + g) * 16;
            float v[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(xs + ((t9 / 3) * IW + (t9 % 3)) * G0 * 16);
                mac8_f16(v, x, wv[t9]);
            }
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                h[e] = (_Float16)dw_act<true>(v[e], 0.f, dw_hi);
            *reinterpret_cast<half8*>(s_b1 + lds_off<C0>(pix, g)) = h;
        }
    }
    lds_barrier();

    // ---- S3: pointwise of block a; wavefront = row tile wave & 1, pixel tiles (wave >> 1) + 2 q
    {
        const float hi = A.pw.act_hi;
#pragma unroll
        for (int q = 0; q < (NPT1 + 1) / 2; ++q) {
            const int pt = (wave >> 1) + 2 * q;
            if (pt >= NPT1) // uniform per wavefront
                break;
            floatx16 acc;
#pragma unroll
            for (int ks = 0; ks < KQ1; ++ks) {
                half8 fa;
                __builtin_memcpy(&fa, &a1[ks], 16);
                const half8 fb = *reinterpret_cast<const half8*>(s_b1 + lds_off<C0>(pt * 32 + frow, ks * 2 + fk));
                acc = ks == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, floatx16{}, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
            }
            const int n = pt * 32 + frow;
            const int my = n / MW, mx = n - my * MW;
            const int y = my0 + my, x = mx0 + mx;
            const unsigned okm = (y >= 0 && y < A.OH && x >= 0 && x < A.OW) ? 0xffffffffu : 0u;
            if (n < NM) {
                unsigned char* const row = s_h2 + n * PXB2 + (rt1 * 32 + 4 * fk) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h;
                    h[0] = (_Float16)__builtin_amdgcn_fmed3f(acc[4 * g + 0] + bs1[g].x, 0.f, hi);
                    h[1] = (_Float16)__builtin_amdgcn_fmed3f(acc[4 * g + 1] + bs1[g].y, 0.f, hi);
                    h[2] = (_Float16)__builtin_amdgcn_fmed3f(acc[4 * g + 2] + bs1[g].z, 0.f, hi);
                    h[3] = (_Float16)__builtin_amdgcn_fmed3f(acc[4 * g + 3] + bs1[g].w, 0.f, hi);
                    uint2 u;
                    __builtin_memcpy(&u, &h, 8);
                    u.x &= okm, u.y &= okm; // (outside the image: + 0.0, block b's padding)
                    *reinterpret_cast<uint2*>(row + 16 * g) = u;
                }
            }
        }
    }
    lds_barrier();

    // ---- S4: depthwise taps of block b, stride 2: thread = (output pixel tid / G1, channel group tid % G1)
    {
        const int g = tid % G1, pix = tid / G1;
        const int py = pix / TW, px = pix - py * TW;
        const float* bsrc = reinterpret_cast<const float*>(s_bi2) + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
        const unsigned char* xs = s_h2 + ((py * 2) * MW + px * 2) * PXB2 + g * 16;
        float v[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(xs + ((t9 / 3) * MW + (t9 % 3)) * PXB2);
            const u32x4 w = *reinterpret_cast<const u32x4*>(s_w2 + (t9 * C1 + g * 8) * 2);
            mac8_f16(v, x, w);
        }
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            h[e] = (_Float16)dw_act<true>(v[e], 0.f, Bk.dw_hi);
        *reinterpret_cast<half8*>(s_b2 + lds_off<C1>(pix, g)) = h;
    }
    lds_barrier();

    // ---- S5: pointwise of block b: row tile `wave`, the 32 output pixels, K = C1
    floatx16 acc[1][1];
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) {
        half8 fa;
        __builtin_memcpy(&fa, &a2[ks], 16);
        const half8 fb = *reinterpret_cast<const half8*>(s_b2 + lds_off<C1>(frow, ks * 2 + fk));
        acc[0][0] = ks == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, floatx16{}, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0][0], 0, 0, 0);
    }
    int pb[1] = { b }, py1[1] = { oy0 + frow / TW }, px1[1] = { ox0 + frow % TW };
    bool pv[1] = { py1[0] < Bk.OH && px1[0] < Bk.OW };
    // (the slabs lie over region X: every wavefront passed its last reads of block a's output at the barrier above)
    conv_epilogue_packed<1, 1>(Bk.pw, acc, wave * 32, lane, s_h1 + wave * packed_geom<1, 1>::WAVE_BYTES, pb, py1, px1, pv);
}

// which pair kernel serves two consecutive separable blocks (0 = none: one launch per block)
int seppair_variant(const seppair_params& p)
{
    const sep_params &a = p.a, &b = p.b;
    const conv_params &qa = a.pw, &qb = b.pw;
    if (a.C != 32 || qa.Cout != 64 || qa.Cout_pad < 64 || a.stride != 1 || a.dil != 1 || a.pad_t != 1 || a.pad_l != 1 || a.OH != a.H || a.OW != a.W)
        return 0;
    if (b.C != 64 || qb.Cout_pad != 128 || b.stride != 2 || b.dil != 1 || b.pad_t < 0 || b.pad_t > 1 || b.pad_l < 0 || b.pad_l > 1 || b.H != a.OH || b.W != a.OW)
        return 0;
    if (a.dw_slope != 0.f || b.dw_slope != 0.f || qa.alpha || qa.act_slope != 0.f || qa.res.p || qa.out_f32 || qb.out_f32 || !fast_epilogue(qb))
        return 0;
    if (a.in.coff % 8 || a.in.cs % 8)
        return 0;
    // block b reads exactly what block a writes
    if (b.in.p != qa.out.p || b.in.coff != qa.out.coff || b.in.cs != qa.out.cs || b.in.wp != qa.out.wp || b.in.img != qa.out.img)
        return 0;
    return 1;
}

hipError_t launch_seppair(const seppair_params& p, hipStream_t s)
{
    if (seppair_variant(p) != 1)
        return hipErrorInvalidValue;
    const int tiles_x = (p.b.OW + 7) / 8, tiles_y = (p.b.OH + 3) / 4;
    HP_LAUNCH((sepconv_pair_kernel<32, 64, 128>), dim3(tiles_x * tiles_y * p.b.B), dim3(256), 0, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

// which instantiation serves (Cout_pad, stride, dilation, C); 0 = none (the engine then keeps the two launches)
// which fused instance serves a block (0 = none: the caller keeps dwconv3x3 + a 1x1 convolution)
int sepconv_variant_for(int C, int cout_pad, int stride, int dil, int cout)
{
    if (C > SEP_CMAX || C % 32 || cout_pad % 128)
        return 0;
    if (C == 32 && cout > 0 && cout <= 64 && stride == 1 && dil == 1)
        return 7; // sepconv_small_kernel<32, 1, 2>
    if (C % 64)
        return 0;
    const int tm = cout_pad / 128;
    if (tm == 1 && C <= 128 && dil == 1 && (stride == 1 || stride == 2))
        return stride; // 1 / 2: 128 output channels
    if (tm == 2 && stride == 2 && dil == 1 && C <= 128)
        return 3;
    if (tm == 2 && stride == 1 && dil == 1 && C <= 256)
        return 4;
    if (tm == 4 && stride == 1 && dil == 1)
        return 5;
    if (tm == 4 && stride == 1 && dil == 2)
        return 6;
    return 0;
}

int sepconv_variant(const sep_params& p)
{
    const conv_params& q = p.pw;
    if (q.res.p || q.out_f32 || !fast_epilogue(q) || p.in.coff % 8 || p.halo < p.dil || p.dw_slope != 0.f)
        return 0;
    return sepconv_variant_for(p.C, q.Cout_pad, p.stride, p.dil, q.Cout);
}

hipError_t launch_sepconv(const sep_params& p, hipStream_t s)
{
    int v = sepconv_variant(p);
    static const bool half_cu = getenv("HP_SEP_SLOT") != nullptr; // A/B switch: the half-CU forms of the 512-channel blocks
    if (half_cu && (v == 5 || v == 6))
        v += 10;
    // <= 128 output channels at 64 / 128 input channels: ONE block of all channels (sepconv_small_kernel)
    if (p.pw.Cout <= 128) {
        if (v == 1 && p.C == 128)
            return launch_sep_small<128, 1, 4>(p, s);
        if (v == 1 && p.C == 64)
            return launch_sep_small<64, 1, 4>(p, s);
        if (v == 2 && p.C == 64)
            return launch_sep_small<64, 2, 4>(p, s);
    }
    switch (v) { // <passes, row tiles per wavefront, stride, dilation, max channels, halo chunk>
    case 7:
        return launch_sep_small<32, 1, 2>(p, s);
    case 1:
        return launch_sep_slot<1, 1, 1, 1, 128>(p, s);
    case 2:
        return launch_sep_slot<1, 1, 2, 1, 128>(p, s);
    case 3:
        return launch_sep_slot<2, 1, 2, 1, 128>(p, s); // (one pass of two row tiles spills next to the 17 x 17 halo prefetch)
    case 4:
        return launch_sep_slot<1, 2, 1, 1, 256>(p, s);
    case 5:
        return launch_sep_pipe<1, 512>(p, s);
    case 6:
        return launch_sep_pipe<2, 512>(p, s);
    case 15: // (the half-CU forms of 5 / 6: HP_SEP_SLOT=1, kept for the A/B in DESIGN.md section 7)
        return launch_sep_slot<2, 2, 1, 1, 512>(p, s);
    case 16:
        return launch_sep_slot<2, 2, 1, 2, 512, 32>(p, s);
    default:
        return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------
// Two chained 1x1 convolutions in ONE launch: K1 -> 512 (relu) -> Cout2 <= 64 (the OpenPose-style heads,
// lw_openpose.py:123-191: "1x1 128->512 relu, 1x1 512->19|38").  The 512-channel hidden tensor (20 MB per head and batch
// at 46x54x8) never exists in HBM - and not even in LDS: the first GEMM's accumulator layout IS a B-fragment layout of
// the second GEMM once the second layer's weights are packed in the matching K order.
//   block = one (image, 8 x 12 pixel tile); the input tile [96 px][K1] goes to LDS once (swizzled B tiles of 64 ch).
//   GEMM1: wavefront w owns hidden rows 128w .. 128w+127 (4 x 32-row MFMA tiles) x all 96 pixels; its weight fragments
//          come straight from L2 in fragment order (as in conv3x3_direct_kernel), two k16 steps ahead.
//   hidden = fp16(clamp(acc + bias1)) in registers.  In the 32x32 MFMA result, lane (n, h) holds rows
//          (r & 3) + 8 (r >> 2) + 4h of column n: registers 8s .. 8s+7 are 8 of the 16 rows of k16 step s, and the h = 0 /
//          h = 1 lanes together hold all 16 - exactly what a B fragment is, for a permuted row order.
//   GEMM2: out[Cout2 x 96] += W2[:, rows of this wavefront] x hidden: W2 is packed host-side with that row order
//          (head_params), 8 k16 steps per wavefront; the four partial sums meet through LDS.
//   store: bias2 / activation, fp16 NHWC slice and / or fp32 NCHW network output.
template <int NCH>
__device__ __forceinline__ void mlp_head_body(const head_params& p, int tiles_x, int tiles_y)
{
    // 8 x 8 pixel tile and the hidden rows in two passes of 2 x 32 per wavefront: 64 + 64 accumulator registers instead
    // of 192 + 96, so that the kernel fits half a CU (<= 256 registers, 64 KB of LDS) and shares it with whatever the
    // other hardware queue is running - alone on its CU it cost its full duration end to end.
    constexpr int TH = 8, TW = 8, NPX = 64, NT = 2, TP = 2; // TP = row tiles per pass (2 passes x 2 = 4 per wavefront)
    constexpr int K1 = NCH * 64, KQ1 = K1 / 16; // k16 steps of GEMM1
    constexpr int X_BYTES = NCH * NPX * 128;    // NCH tiles [64 px][64 ch]
    constexpr int NSLOT = 2 * NT * 4;           // float4 slots of the second GEMM's result per lane
    constexpr int RED_BYTES = 4 * NSLOT * 64 * 16; // [wave][slot][lane]
    constexpr int LDS_BYTES = X_BYTES > RED_BYTES ? X_BYTES : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES + 512 * 4]; // + the first layer's 512 biases
    float* const s_b1 = reinterpret_cast<float*>(lds + LDS_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const conv_params& q = p.pw;
    // <= 32 outputs (the 19 heat-maps): the second GEMM's rows 32 .. 63 are all padding - their MFMAs and weight fragments are skipped
    const bool two_tiles = q.Cout > 32; // uniform
    int dbg_i = 0;
#define HP_STAMP()                                                                                                \
    if (q.dbg && blockIdx.x == 0 && tid == 0)                                                                     \
        q.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();

    // ---- GEMM1 weights of pass 0: fragments (row tile wave*4 + i, k16 step 0 and 1) in flight first
    const __half* w1 = p.w1 + ((size_t)(wave * 4) * KQ1 * 64 + lane) * 8;
    const __half* w2 = p.w2 + ((size_t)(wave * 8) * 64 + lane) * 8;
    u32x4 a[2][TP];
#pragma unroll
    for (int i = 0; i < TP; ++i) {
        a[0][i] = *reinterpret_cast<const u32x4*>(w1 + ((size_t)i * KQ1 + 0) * 512);
        a[1][i] = *reinterpret_cast<const u32x4*>(w1 + ((size_t)i * KQ1 + 1) * 512);
    }
    // ---- input tile -> LDS (pixels outside the map alias pixel (H-1, W-1): finite values, masked at the store)
    {
        constexpr int PIECES = NPX * NCH * 8, NLD = PIECES / 256;
        static_assert(PIECES % 256 == 0, "whole pieces per thread");
        u32x4 xv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * 256;
            const int pix = i / (NCH * 8), c = i % (NCH * 8);
            const int y = min(y0 + pix / TW, p.H - 1), x = min(x0 + pix % TW, p.W - 1);
            xv[k] = *reinterpret_cast<const u32x4*>(p.in.p + tv_off(p.in, b, y, x) + c * 8);
        }
        // (the first layer's biases ride along: read from LDS after each pass instead of from L2 - a memory latency per row tile)
        const float4 b1v = *reinterpret_cast<const float4*>(p.b1 + (tid & 127) * 4);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * 256;
            const int pix = i / (NCH * 8), c = i % (NCH * 8);
            *reinterpret_cast<u32x4*>(lds + (c >> 3) * (NPX * 128) + lds_off<64>(pix, c & 7)) = xv[k];
        }
        if (tid < 128)
            *reinterpret_cast<float4*>(s_b1 + tid * 4) = b1v;
    }
    floatx16 acc2[2][NT];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc2[i2][j][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;
    const float hi1 = p.hi1;
    HP_STAMP();
    lds_barrier(); // the input tile is complete
    HP_STAMP();

#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // GEMM2 weights of this pass (k16 steps 4 pass .. 4 pass + 3 of this wavefront, 2 row tiles): needed after GEMM1
        u32x4 a2[4][2];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
                if (i2 == 0 || two_tiles)
                    a2[s4][i2] = *reinterpret_cast<const u32x4*>(w2 + ((size_t)i2 * 32 + pass * 4 + s4) * 512);
        // ---- GEMM1 for hidden row tiles 2 pass, 2 pass + 1 of this wavefront
        floatx16 acc[TP][NT];
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] = 0.f;
#pragma unroll
        for (int qs = 0; qs < KQ1; ++qs) {
            half8 fb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
                fb[j] = *reinterpret_cast<const half8*>(lds + (qs / 4) * (NPX * 128) + lds_off<64>(j * 32 + frow, (qs % 4) * 2 + fk));
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                half8 fa;
                __builtin_memcpy(&fa, &a[qs & 1][i], 16);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], acc[i][j], 0, 0, 0);
                // two k16 steps ahead; the last two steps of pass 0 already fetch the first two of pass 1
                const int qn = qs + 2 < KQ1 ? qs + 2 : qs + 2 - KQ1;
                const int pn = qs + 2 < KQ1 ? pass : pass + 1;
                if (pn < 2)
                    a[qs & 1][i] = *reinterpret_cast<const u32x4*>(w1 + ((size_t)(pn * TP + i) * KQ1 + qn) * 512);
            }
        }
        // ---- hidden activations -> fp16 B fragments, GEMM2 on these 64 hidden rows
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            float bs[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(s_b1 + wave * 128 + (pass * TP + i) * 32 + 8 * g + 4 * fk);
                bs[4 * g] = bv.x, bs[4 * g + 1] = bv.y, bs[4 * g + 2] = bv.z, bs[4 * g + 3] = bv.w;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 hb[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        hb[j][e] = (_Float16)__builtin_amdgcn_fmed3f(acc[i][j][8 * s2 + e] + bs[8 * s2 + e], 0.f, hi1);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    if (i2 == 1 && !two_tiles)
                        continue;
                    half8 fa;
                    __builtin_memcpy(&fa, &a2[i * 2 + s2][i2], 16);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc2[i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, hb[j], acc2[i2][j], 0, 0, 0);
                }
            }
        }
    }
    HP_STAMP();

    // second-layer bias / slopes of the float4 slots this wave will finish (host arrays are padded to 64 rows)
    constexpr int PERW = NSLOT / 4;
    float4 bias4[PERW], slope4[PERW];
#pragma unroll
    for (int f6 = 0; f6 < PERW; ++f6) {
        const int f = wave * PERW + f6, m = ((f >> 2) / NT) * 32 + 8 * (f & 3) + 4 * fk;
        bias4[f6] = *reinterpret_cast<const float4*>(q.bias + m);
        slope4[f6] = q.alpha ? *reinterpret_cast<const float4*>(q.alpha + m) : make_float4(q.act_slope, q.act_slope, q.act_slope, q.act_slope);
    }
    // ---- the four partial sums meet: wave w finishes float4 slots PERW w .. PERW w + PERW - 1 of the NSLOT per lane
    __syncthreads(); // every wave is done with the input tile
    float4* const red = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[((size_t)wave * NSLOT + (i2 * NT + j) * 4 + g) * 64 + lane]
                    = make_float4(acc2[i2][j][4 * g], acc2[i2][j][4 * g + 1], acc2[i2][j][4 * g + 2], acc2[i2][j][4 * g + 3]);
    __syncthreads();
    HP_STAMP();
    const long plane = (long)q.OH * q.OW;
#pragma unroll
    for (int f6 = 0; f6 < PERW; ++f6) {
        const int f = wave * PERW + f6, tile = f >> 2, g = f & 3, i2 = tile / NT, j = tile % NT;
        float4 v = red[((size_t)0 * NSLOT + f) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = red[((size_t)w * NSLOT + f) * 64 + lane];
            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
        }
        const int m = i2 * 32 + 8 * g + 4 * fk;
        const int n = j * 32 + (lane & 31);
        const int oy = y0 + n / TW, ox = x0 + n % TW;
        if (m < q.Cout && oy < q.OH && ox < q.OW) {
            const float vv[4] = { v.x + bias4[f6].x, v.y + bias4[f6].y, v.z + bias4[f6].z, v.w + bias4[f6].w };
            const float sl[4] = { slope4[f6].x, slope4[f6].y, slope4[f6].z, slope4[f6].w };
            __half* const o16 = q.out.p ? q.out.p + tv_off(q.out, b, oy, ox) + m : nullptr;
            float* const o32 = q.out_f32 ? q.out_f32 + ((long)b * q.Cout + m) * plane + (long)oy * q.OW + ox : nullptr;
            float xs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xs[e] = vv[e] > 0.f ? fminf(vv[e], q.act_hi) : vv[e] * sl[e];
            if (o16) {
                if (m + 3 < q.Cout && ((q.out.coff | q.out.cs) & 3) == 0) { // whole, 8-byte aligned group
                    half4 h4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        h4[e] = (_Float16)xs[e];
                    *reinterpret_cast<half4*>(o16) = h4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < q.Cout)
                            o16[e] = __float2half(xs[e]);
                }
            }
            if (o32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m + e < q.Cout)
                        o32[e * plane] = xs[e];
            }
        }
    }
    HP_STAMP();
#undef HP_STAMP
}

// 0: no fused head kernel for this pair
int mlp_head_variant(int k1, int hidden, int cout2)
{
    const int off = getenv("HP_NO_FUSE_HEAD") ? atoi(getenv("HP_NO_FUSE_HEAD")) : 0; // read per engine build (tests toggle it)
    if (off || hidden != 512 || cout2 > 64 || cout2 < 1)
        return 0;
    return k1 == 64 ? 1 : k1 == 128 ? 2 : k1 == 256 ? 4 : 0;
}

template <int NCH>
__global__ __launch_bounds__(256, 2) void mlp_head_kernel(const head_params p, int tiles_x, int tiles_y)
{
    mlp_head_body<NCH>(p, tiles_x, tiles_y);
}
// Two heads that read the same feature map (the conf / paf branches of one stage) in ONE launch: blockIdx.y picks the head.
// Each is a grid of ~340 latency-bound blocks on 512 half-CU slots, so together they take about as long as one alone.
template <int NCH>
__global__ __launch_bounds__(256, 2) void mlp_head_pair_kernel(const head_params p0, const head_params p1, int tiles_x, int tiles_y)
{
    mlp_head_body<NCH>(blockIdx.y ? p1 : p0, tiles_x, tiles_y);
}

hipError_t launch_mlp_head_pair(const head_params& p0, const head_params& p1, hipStream_t s)
{
    const int tiles_x = (p0.W + 7) / 8, tiles_y = (p0.H + 7) / 8;
    const dim3 grid(tiles_x * tiles_y * p0.B, 2);
    const int v = mlp_head_variant(p0.K1, 512, p0.pw.Cout);
    if (v == 0 || v != mlp_head_variant(p1.K1, 512, p1.pw.Cout) || p0.H != p1.H || p0.W != p1.W || p0.B != p1.B)
        return hipErrorInvalidValue;
    switch (v) {
    case 1:
        HP_LAUNCH((mlp_head_pair_kernel<1>), grid, dim3(256), 0, s, p0, p1, tiles_x, tiles_y);
        break;
    case 2:
        HP_LAUNCH((mlp_head_pair_kernel<2>), grid, dim3(256), 0, s, p0, p1, tiles_x, tiles_y);
        break;
    default:
        HP_LAUNCH((mlp_head_pair_kernel<4>), grid, dim3(256), 0, s, p0, p1, tiles_x, tiles_y);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_mlp_head(const head_params& p, hipStream_t s)
{
    const int tiles_x = (p.W + 7) / 8, tiles_y = (p.H + 7) / 8;
    const dim3 grid(tiles_x * tiles_y * p.B);
    switch (mlp_head_variant(p.K1, 512, p.pw.Cout)) {
    case 1:
        HP_LAUNCH((mlp_head_kernel<1>), grid, dim3(256), 0, s, p, tiles_x, tiles_y);
        break;
    case 2:
        HP_LAUNCH((mlp_head_kernel<2>), grid, dim3(256), 0, s, p, tiles_x, tiles_y);
        break;
    case 4:
        HP_LAUNCH((mlp_head_kernel<4>), grid, dim3(256), 0, s, p, tiles_x, tiles_y);
        break;
    default:
        return hipErrorInvalidValue;
    }
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
            if (iy < 0 || iy >= p.H) // SAME max-pool pads with -inf, not with the zero halo
                continue;
            for (int kx = 0; kx < p.k; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx;
                if (ix < 0 || ix >= p.W)
                    continue;
                const half8 x = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, iy, ix) + cg * 8);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    m[r] = fmaxf(m[r], (float)x[r]);
            }
        }
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            h[r] = (_Float16)m[r];
        *reinterpret_cast<half8*>(p.out.p + tv_off(p.out, b, oy, ox) + cg * 8) = h;
    }
}

hipError_t launch_maxpool(const pool_params& p, hipStream_t s)
{
    const long total = (long)p.B * p.OH * p.OW * (p.C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    HP_LAUNCH(maxpool_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Integer up-scaling (HP_OP_UPSAMPLE): one thread = one output pixel x 8 channels.  Bilinear = half-pixel centres, source
// coordinate clamped at 0, the far neighbour clamped to the last row / column (tf.image.resize, ONNX Resize half_pixel,
// torch align_corners = False all agree for integer scales); interpolation in fp32, one rounding to fp16.
__global__ __launch_bounds__(256) void upsample_kernel(const pool_params p)
{
    const int CG = p.C / 8, sc = p.stride;
    const float inv = 1.f / (float)sc;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        half8 h;
        if (p.k == 0) {
            h = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, oy / sc, ox / sc) + cg * 8);
        } else {
            const float sy = fmaxf(((float)oy + 0.5f) * inv - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * inv - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = min(y0 + 1, p.H - 1), x1 = min(x0 + 1, p.W - 1);
            const float fy = sy - (float)y0, fx = sx - (float)x0;
            const half8 a = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, y0, x0) + cg * 8);
            const half8 bq = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, y0, x1) + cg * 8);
            const half8 c = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, y1, x0) + cg * 8);
            const half8 d = *reinterpret_cast<const half8*>(p.in.p + tv_off(p.in, b, y1, x1) + cg * 8);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float top = (float)a[r] * (1.f - fx) + (float)bq[r] * fx;
                const float bot = (float)c[r] * (1.f - fx) + (float)d[r] * fx;
                h[r] = (_Float16)(top * (1.f - fy) + bot * fy);
            }
        }
        *reinterpret_cast<half8*>(p.out.p + tv_off(p.out, b, oy, ox) + cg * 8) = h;
    }
}

hipError_t launch_upsample(const pool_params& p, hipStream_t s)
{
    const long total = (long)p.B * p.OH * p.OW * (p.C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    HP_LAUNCH(upsample_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void output_transform_kernel(tview in, int B, int H, int W, out_xform x, float* __restrict__ out)
{
    const int sc = x.shuffle, CO = x.C / (sc * sc), OH = x.out_h, OW = x.out_w;
    const long total = (long)B * CO * OH * OW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        long t = i / OW;
        const int oy = (int)(t % OH);
        t /= OH;
        const int c = (int)(t % CO), b = (int)(t / CO);
        // pixel_shuffle: [b, c, sy, sx, h, w] -> [b, c, h, sy, w, sx]  (hyperpose/Model/pifpaf/utils.py:371-379)
        const int y = oy / sc, sy = oy - y * sc, xx = ox / sc, sx = ox - xx * sc;
        const int cin = c * sc * sc + sy * sc + sx;
        float v = __half2float(in.p[tv_off(in, b, y, xx) + cin]);
        int act = x.act;
        if (x.group > 0) {
            const int comp = c % x.group;
            act = ((x.sigmoid_mask >> comp) & 1u) ? ACT_SIGMOID : (((x.softplus_mask >> comp) & 1u) ? ACT_SOFTPLUS : ACT_NONE);
        }
        v = apply_act(v, act, 0.f, 0.f);
        if (x.grid == 1)
            v += (float)ox;
        else if (x.grid == 2)
            v += (float)oy;
        out[i] = v * x.scale;
    }
}

hipError_t launch_output_transform(tview in, int B, int H, int W, const out_xform& x, float* out, hipStream_t s)
{
    const long total = (long)B * (x.C / (x.shuffle * x.shuffle)) * x.out_h * x.out_w;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    HP_LAUNCH(output_transform_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W, x, out);
    return hipGetLastError();
}

} // namespace hp
