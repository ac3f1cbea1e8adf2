// conv32_epilogue.hpp — the row-major epilogue of the fp32 convolution kernels (conv_fp32.hip, conv32_direct.hip).
//
// An MFMA accumulator tile gives lane (n, h) four consecutive channels of pixel n per register quad: stored from there, one
// global_store_dwordx4 of a wavefront touches 64 different cache lines with 16 bytes each, and the CU's texture path takes ~6 cycles per
// LINE whatever the bytes: block timelines (tools/direct_timeline.py) showed 13 - 18 k cycles of epilogue per block, 16 % of a 3 x 3 layer on
// the fp32 pipe, a third to two thirds of a 1 x 1 layer on the split path.  Here a wavefront transposes its 32-pixel column tile through
// a private LDS slab ([32 pixels][32 TM channels + 4 floats of padding]: ds_write_b128 / ds_read_b128 conflict-free, no block barrier) and
// then owns whole pixel rows: 8 (TM = 1) or 16 (TM = 2) lanes cover the 128 / 256 contiguous bytes of one pixel, so bias / slopes are one
// coalesced load, the residual and the output full-line accesses.
#pragma once
#include "conv_fp32.hpp"

#include "conv_device.hpp"

namespace hp {

typedef float f32x4e __attribute__((ext_vector_type(4)));

template <int TM>
struct rows_geom {
    static constexpr int CH = TM * 32, PITCH = CH + 4, LPR = CH / 4, RPP = 64 / LPR, NP = 32 / RPP;
    static constexpr int SLAB_BYTES = 32 * PITCH * 4; // per wavefront
};

// Second half of the row-major epilogue: the wavefront's slab holds NROWS finished pixel rows of TM * 32 channels ([NROWS][PITCH] floats, written by
// this wavefront only); m_base = first channel; pixel(r, ok, ooff, roff): validity and element offsets (tview32 units, without the channel) of
// slab row r in the output / residual tensors.  Only the NHWC output (p.out) is written here; the NCHW network output keeps the lane = pixel
// form (its runs lie along x).
template <int TM, int NROWS, class F>
__device__ __forceinline__ void conv32_drain_rows(const conv32_params& p, const float* slab, int lane, int m_base, F&& pixel)
{
    using R = rows_geom<TM>;
    const int c4 = lane % R::LPR, r0 = lane / R::LPR;
    const int m = m_base + c4 * 4;
    const bool any = m < p.Cout, full = m + 3 < p.Cout;
    const bool out_vec = ((p.out.coff | p.out.cs) & 3) == 0, res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0;
    // (m + 3 < Cout_pad: m is a multiple of 4 below the padded channel count, which is a multiple of 32)
    const f32x4e bs = *reinterpret_cast<const f32x4e*>(p.bias + m);
    f32x4e sl = { p.act_slope, p.act_slope, p.act_slope, p.act_slope };
    if (p.alpha)
        sl = *reinterpret_cast<const f32x4e*>(p.alpha + m);
    // rows in groups of at most four: the slab reads and residual requests of a group first, then its arithmetic and stores (all eight passes
    // of a two-tile wavefront at once cost conv32_kernel<64, 128> 56 more registers)
    constexpr int NPASS = NROWS / R::RPP, GRP = NPASS < 4 ? NPASS : NPASS % 4 == 0 ? 4 : NPASS % 3 == 0 ? 3 : 1;
    static_assert(NROWS % R::RPP == 0 && NPASS % GRP == 0, "slab rows per pass");
#pragma unroll
    for (int g = 0; g < NPASS; g += GRP) {
        f32x4e v[GRP], rr[GRP];
        bool ok[GRP];
        long ooff[GRP];
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
            const int r = (g + k) * R::RPP + r0;
            long roff = 0;
            pixel(r, ok[k], ooff[k], roff);
            ok[k] = ok[k] && any;
            v[k] = *reinterpret_cast<const f32x4e*>(slab + r * R::PITCH + c4 * 4);
            rr[k] = f32x4e{ 0.f, 0.f, 0.f, 0.f };
            if (p.res.p && ok[k]) {
                if (full && res_vec)
                    rr[k] = *reinterpret_cast<const f32x4e*>(p.res.p + roff + m);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.Cout)
                            rr[k][e] = p.res.p[roff + m + e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
            f32x4e o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[k][e] + bs[e];
                if (p.res.p && p.res_before_act)
                    x += rr[k][e];
                x = x > 0.f ? fminf(x, p.act_hi) : x * sl[e];
                if (p.res.p && !p.res_before_act)
                    x += rr[k][e];
                o[e] = x;
            }
            if (ok[k]) {
                if (full && out_vec)
                    *reinterpret_cast<f32x4e*>(p.out.p + ooff[k] + m) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.Cout)
                            p.out.p[ooff[k] + m + e] = o[e];
                }
            }
        }
    }
}

// acc[i] = the finished sums of the wavefront's i-th 32-channel tile (v_mfma_f32_32x32x*: lane (n, fk) holds rows (r & 3) + 8 (r >> 2) + 4 fk of
// column n) for ONE 32-pixel column tile; m_base = first channel of acc[0]; pixel(): see conv32_drain_rows.
template <int TM, class F>
__device__ __forceinline__ void conv32_store_rows(const conv32_params& p, const floatx16 (&acc)[TM], float* slab, int lane, int m_base, F&& pixel)
{
    using R = rows_geom<TM>;
    const int n = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4e*>(slab + n * R::PITCH + 32 * i + 8 * q + 4 * fk) = f32x4e{ acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3] };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slab is private to the wavefront: its own writes are all it waits for
    conv32_drain_rows<TM, 32>(p, slab, lane, m_base, pixel);
}

} // namespace hp
