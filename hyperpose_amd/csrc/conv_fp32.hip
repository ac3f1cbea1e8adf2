// conv_fp32.hip — the fp32-faithful kernel family for gfx950 (interface and rationale: conv_fp32.hpp).
//
// The reference's engine computes in fp32 unless the caller asks for kHALF (include/hyperpose/operator/dnn/tensorrt.hpp:14-21,48;
// src/tensorrt.cpp:327,353 hand the data type to the TensorRT builder).  An engine created with HP_DTYPE_F32 runs every layer of the
// exported graphs through the kernels below, one launch per layer (except where conv32_winograd.hip / conv32_head.hip / conv32_direct.hip take a
// layer or a pair: conv_fp32.hpp):
//   conv32_kernel          dense KH x KW convolution, implicit GEMM D[cout][pixel] = sum_{tap,cin} W[tap][cout][cin] * X[pixel@tap][cin]
//                          on v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157 TFLOP/s: MI355X_MICROARCH.md) - exact fp32 products and sums;
//   first_conv32_kernel    the 3-channel input layer with the u8 -> f32 pre-processing of src/data.cpp:21-51 folded into its load;
//   dwconv32_kernel        depthwise 3 x 3;   maxpool32_kernel / upsample32_kernel;   output_transform32_kernel (NHWC fp32 -> NCHW fp32).
// Activations keep the zero-halo NHWC layout of the fp16 path with 4-byte elements.
#include "conv_fp32.hpp"

#include "conv32_epilogue.hpp"
#include "conv_device.hpp"

#include <algorithm>
#include <cstdlib>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act32(float v, int act, float param, float alpha)
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
        return 1.f / (1.f + expf(-v));
    case ACT_SOFTPLUS:
        return v > 20.f ? v : log1pf(expf(v));
    default:
        return v;
    }
}

__device__ __forceinline__ long tv32_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

} // namespace

// ---------------------------------------------------------------------------------------------------
// Block = 256 threads = WM x WN wavefronts; block tile BM output channels x BN pixels, K-step of 16 input channels of one tap.
// A (weights) and B (activations) tiles go global -> registers -> LDS, double-buffered, ONE LDS-only barrier per K-step; the global
// loads of step s + 1 are in flight while step s multiplies (32 MFMAs of 64 cycles per wavefront at 128 x 128: the loads hide).
// K order inside a group of 8 channels: lane (row, h) reads channels [4h, 4h + 4) of its row with one ds_read_b128 and feeds element e
// to MFMA e, i.e. MFMA e multiplies channels {e, 4 + e} - A and B use the same permutation, so the sum over the 8 channels is complete.
// LDS rows are 20 floats (80 B): the 16 lanes of a ds_read_b128 service group then touch 64 distinct banks.
// ROWS = the row-major epilogue (NHWC output only); !ROWS = the lane = pixel epilogue (the network's heads; HP_LANE_EPILOGUE=1: every layer).
// A template parameter, not a branch: with both epilogues in one body the kernel carries the register peak of the larger one into every
// launch (conv32_kernel<64, 128>: 92 registers with the lane form alone, 140 - 184 with both) and the store-bound 1 x 1 layers lose a block per CU.
// Where step s + 1's global loads are requested inside step s was measured in round 6 ("software-pipeline the loads";
// 512 -> 512 at 8 x 46 x 54, us alone | in sequence | with a second stream): hipcc's own placement - it sinks them below 12 of the step's 16
// MFMAs, the s_waitcnt two instructions later - 107 | 102 | 95; requested before the step's fragment reads behind a sched_barrier 127 | 119 | 100;
// after the first half of the MFMAs 133 | 129 | 97 (profiles/r06_ab_layers_f32_conv32_prefetch.txt).  The block residency trace
// (HP_DIRECT_DBG, engine.cpp) says why: with the loads early the four blocks of a CU run in step - all of them at their barrier at once, the
// pipe idle - while the late form lets the oldest block run ahead (blocks of one CU end 58 / 68 / 78 / 88 us after they started together)
// and the phases interleave.  The form below is therefore round 5's, unchanged.
template <int BM, int BN, int WM, int WN, bool ROWS>
__global__ __launch_bounds__(256) void conv32_kernel(const conv32_params p)
{
    constexpr int BK = 16, LDR = 20;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NA = BM / 64, NB = BN / 64;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && NA >= 1 && NB >= 1, "tile");
    // [2][BM * LDR] A tiles, then [2][BN * LDR] B tiles; the epilogue's transposition slabs (conv32_epilogue.hpp) lie over them afterwards
    constexpr int AB_BYTES = 2 * (BM + BN) * LDR * 4, SLABS = 4 * rows_geom<TM>::SLAB_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[AB_BYTES > SLABS ? AB_BYTES : SLABS];
    float (*const sA)[BM * LDR] = reinterpret_cast<float (*)[BM * LDR]>(lds_raw);
    float (*const sB)[BN * LDR] = reinterpret_cast<float (*)[BN * LDR]>(lds_raw + 2 * BM * LDR * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware block order (1-D grid).  The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch) and every XCD
    // has its own L2: the G = Cout_pad / BM blocks that read the SAME pixel tile are b, b + 8, .., b + 8 (G - 1) - dispatched back to back to
    // one XCD, so the tile comes from HBM once and from that L2 G - 1 times.  (With blockIdx.y = channel group the whole input streamed past
    // once per group: conv32_kernel<64, 128> on 512 -> 512 at 8 x 46 x 54 moved 213 MB per launch against 76 MB algorithmic, rocprofv3
    // FETCH_SIZE / WRITE_SIZE.)
    const int G = p.Cout_pad / BM, nx = (p.npix + BN - 1) / BN;
    const int bj = blockIdx.x >> 3, ptile = (bj / G) * 8 + (blockIdx.x & 7);
    if (ptile >= nx)
        return; // (the grid is padded to whole groups of eight pixel tiles)
    const int n0 = ptile * BN, m0 = (bj % G) * BM;
    const int lrow = tid >> 2, lchunk = (tid & 3) * 4;
    const int OHW = p.OH * p.OW, kc = p.Cin / BK, steps = p.KH * p.KW * kc;

    long bbase[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = min(n0 + lrow + 64 * j, p.npix - 1);
        const int b = n / OHW, rem = n - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        bbase[j] = tv32_off(p.in, b, oy * p.stride - p.pad_t, ox * p.stride - p.pad_l) + lchunk;
    }
    const float* const wrow = p.w + (long)(m0 + lrow) * p.Cin + lchunk;

    f32x4 ra[NA], rb[NB];
    auto gload = [&](int s) {
        const int tap = s / kc, k0 = (s - tap * kc) * BK;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const long toff = ((long)(ky * p.dil) * p.in.wp + kx * p.dil) * p.in.cs + k0;
        const float* const wt = wrow + (long)tap * p.Cout_pad * p.Cin + k0;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            ra[j] = *reinterpret_cast<const f32x4*>(wt + (long)(64 * j) * p.Cin);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            rb[j] = *reinterpret_cast<const f32x4*>(p.in.p + bbase[j] + toff);
    };
    auto to_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            *reinterpret_cast<f32x4*>(&sA[buf][(lrow + 64 * j) * LDR + lchunk]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<f32x4*>(&sB[buf][(lrow + 64 * j) * LDR + lchunk]) = rb[j];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = (lane >> 5) * 4;
    int dbg_i = 0;
#define HP_STAMP()                                       \
    if (p.dbg && blockIdx.x == 9 && tid == 0)            \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    // block residency trace: [128 + 3 b] = start, [.. + 1] = end (100 MHz clock), [.. + 2] = XCC_ID << 32 | HW_ID of block b < 4096
    if (p.dbg && tid == 0 && blockIdx.x < 4096) {
        p.dbg[128 + 3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        p.dbg[128 + 3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    gload(0);
    to_lds(0);
    lds_barrier();
    HP_STAMP();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        gload(min(s + 1, steps - 1));
        const float* const a_s = &sA[s & 1][(wm * TM * 32 + frow) * LDR + fh];
        const float* const b_s = &sB[s & 1][(wn * TN * 32 + frow) * LDR + fh];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(a_s + i * 32 * LDR + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const f32x4*>(b_s + j * 32 * LDR + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        to_lds((s + 1) & 1);
        lds_barrier();
        if ((s & 7) == 7)
            HP_STAMP();
    }

    // epilogue
    if constexpr (ROWS) {
        // NHWC output only (every layer but the network's heads): row-major through a private LDS slab (conv32_epilogue.hpp)
        lds_barrier(); // every wavefront is done with the last K-step's tiles the slabs lie over
        float* const slab = reinterpret_cast<float*>(lds_raw) + wave * (rows_geom<TM>::SLAB_BYTES / 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            floatx16 fin[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fin[i] = acc[i][j];
            const int nb = n0 + (wn * TN + j) * 32;
            conv32_store_rows<TM>(p, fin, slab, lane, m0 + wm * TM * 32, [&](int r, bool& ok, long& ooff, long& roff) {
                const int n = nb + r;
                ok = n < p.npix;
                const int nc = min(n, p.npix - 1);
                const int b = nc / OHW, rem = nc - b * OHW;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                ooff = tv32_off(p.out, b, oy, ox);
                roff = p.res.p ? tv32_off(p.res, b, oy, ox) : 0;
            });
        }
    } else {
    // the network's heads (fp32 NCHW for the parsers, runs along x): lane (n, h) of a 32 x 32 tile holds rows (r & 3) + 8 (r >> 2) + 4 h of
    // column n, four consecutive channels per r >> 2
    const bool out_vec = p.out.p && ((p.out.coff | p.out.cs) & 3) == 0;
    const bool res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0;
    const bool res_pre = p.res.p && p.res_before_act, res_post = p.res.p && !p.res_before_act; // (selected with branches: 0 * Inf would be NaN)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + frow;
        const bool pix_ok = n < p.npix;
        const int nc = min(n, p.npix - 1);
        const int b = nc / OHW, rem = nc - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const long ooff = p.out.p ? tv32_off(p.out, b, oy, ox) : 0;
        const long roff = p.res.p ? tv32_off(p.res, b, oy, ox) : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + (wm * TM + i) * 32 + 8 * q + fh;
                if (pix_ok && m < p.Cout) {
                    const bool full = m + 3 < p.Cout;
                    float v[4], rr[4] = { 0.f, 0.f, 0.f, 0.f };
                    if (p.res.p) {
                        if (full && res_vec) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(p.res.p + roff + m);
                            rr[0] = t[0], rr[1] = t[1], rr[2] = t[2], rr[3] = t[3];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (m + e < p.Cout)
                                    rr[e] = p.res.p[roff + m + e];
                        }
                    }
                    // (m + 3 < Cout_pad: m is a multiple of 4 below Cout <= Cout_pad, a multiple of 64)
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(p.bias + m);
                    f32x4 sl = { p.act_slope, p.act_slope, p.act_slope, p.act_slope };
                    if (p.alpha)
                        sl = *reinterpret_cast<const f32x4*>(p.alpha + m);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][4 * q + e] + bs[e];
                        if (res_pre)
                            x += rr[e];
                        x = x > 0.f ? fminf(x, p.act_hi) : x * sl[e];
                        if (res_post)
                            x += rr[e];
                        v[e] = x;
                    }
                    if (p.out.p) {
                        if (full && out_vec) {
                            f32x4 t = { v[0], v[1], v[2], v[3] };
                            *reinterpret_cast<f32x4*>(p.out.p + ooff + m) = t;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (m + e < p.Cout)
                                    p.out.p[ooff + m + e] = v[e];
                        }
                    }
                    if (p.out_f32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (m + e < p.Cout)
                                p.out_f32[((long)b * p.Cout + m + e) * OHW + rem] = v[e];
                    }
                }
            }
        }
    }
    }
    HP_STAMP();
    if (p.dbg && tid == 0 && blockIdx.x < 4096)
        p.dbg[128 + 3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#undef HP_STAMP
}

// ---------------------------------------------------------------------------------------------------
// conv32_t16_kernel: 64 output channels x 160 pixels per block on v_mfma_f32_16x16x4_f32 (round 6).
// Why 160: conv32_kernel<64, 128> runs four blocks per CU = 1024 slots; a 512-output layer at 8 x 46 x 54 is 156 pixel tiles x 8 channel groups
// = 1248 blocks - one full round and a second round that is 22 % full, while the K loop itself runs at the pipe's rate.  160 pixels
// make it 125 x 8 = 1000 blocks: ONE round, 250 of 256 CUs busy to the end.  A tile of 160 pixels does not divide into 32-pixel MFMA tiles over four
// wavefronts (five wavefronts per block were tried first: the hardware places a block's wavefronts on SIMD 0, 1, 2, 3, 0 - SIMD 0 then holds
// eight of the CU's twenty and every block waits for it: 168 us against 105, profiles/r06_ab_layers_f32_conv32_64x160_five_wavefronts.txt), so the
// tile is 2 x 2 wavefronts of 32 channels x 80 pixels = 2 x 5 MFMA tiles of 16 x 16 (same 64 FLOP/clk/SIMD as the 32 x 32 x 2 form).
// LDS: rows of 16 floats (one K-step) WITHOUT padding, the 16-byte quad index XOR-ed with 2 * ((row >> 3) & 1): the fragment read
// (lane = (row & 15, quad)) and the staging write (lane = (row, quad) in thread order) are both conflict-free; (64 + 192) rows x 64 B x 2 buffers
// = 32 KB per block.  K order as in conv32_kernel: lane (row, kq) reads channels [4 kq, 4 kq + 4) with one ds_read_b128 and feeds element e to
// MFMA e - A and B use the same permutation.  Epilogue: lane (pixel n, q) of a 16 x 16 tile holds channels 4 q .. 4 q + 3 of pixel n: one
// global_store_dwordx4 of a wavefront covers 16 pixels x 64 contiguous bytes.
// WN = wavefronts side by side over the pixels (2: 2 x 2 wavefronts of 32 channels x BN / 2 pixels; 1: four wavefronts of 16 channels x all BN pixels -
// any multiple of 16 pixels per block, e.g. 176: OpenPose-VGG19's 7 x 7 layers at 16 x 54 x 96 are 1 296 tiles of 64 x 128, 1 038 of 64 x 160 - 14 more
// than the chip's 1 024 slots - and 944 of 64 x 176: one round).
template <int BN, int WN = 2>
__global__ __launch_bounds__(256) void conv32_t16_kernel(const conv32_params p)
{
    constexpr int BM = 64, BK = 16, WM = 4 / WN, TMW = BM / 16 / WM, TNW = BN / WN / 16; // a wavefront: TMW x TNW tiles of 16 x 16
    constexpr int NB = (BN + 63) / 64, BR = NB * 64;             // staging passes of the B tile (64 rows each); rows allocated
    static_assert(BN % (16 * WN) == 0 && (WN == 1 || WN == 2), "wavefront columns of whole 16-pixel tiles");
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BR) * BK];
    float (*const sA)[BM * BK] = reinterpret_cast<float (*)[BM * BK]>(lds);
    float (*const sB)[BR * BK] = reinterpret_cast<float (*)[BR * BK]>(lds + 2 * BM * BK);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int G = p.Cout_pad / BM, nx = (p.npix + BN - 1) / BN; // XCD-aware 1-D block order: see conv32_kernel
    const int bj = blockIdx.x >> 3, ptile = (bj / G) * 8 + (blockIdx.x & 7);
    if (ptile >= nx)
        return;
    const int n0 = ptile * BN, m0 = (bj % G) * BM;
    const int lrow = tid >> 2, lq = tid & 3;
    const int OHW = p.OH * p.OW, kc = p.Cin / BK, steps = p.KH * p.KW * kc;
    auto swz = [](int row, int quad) { return row * BK + ((quad ^ (((row >> 3) & 1) << 1)) << 2); };

    long bbase[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = min(n0 + min(lrow + 64 * j, BN - 1), p.npix - 1); // (rows past the tile / the tensor: a clamped pixel nobody reads)
        const int b = n / OHW, rem = n - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        bbase[j] = tv32_off(p.in, b, oy * p.stride - p.pad_t, ox * p.stride - p.pad_l) + lq * 4;
    }
    const float* const wrow = p.w + (long)(m0 + lrow) * p.Cin + lq * 4;
    f32x4 ra, rb[NB];
    auto gload = [&](int s) { // (requested where hipcc puts them: see conv32_kernel)
        const int tap = s / kc, k0 = (s - tap * kc) * BK;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const long toff = ((long)(ky * p.dil) * p.in.wp + kx * p.dil) * p.in.cs + k0;
        ra = *reinterpret_cast<const f32x4*>(wrow + (long)tap * p.Cout_pad * p.Cin + k0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            rb[j] = *reinterpret_cast<const f32x4*>(p.in.p + bbase[j] + toff);
    };
    auto to_lds = [&](int buf) {
        *reinterpret_cast<f32x4*>(&sA[buf][swz(lrow, lq)]) = ra;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<f32x4*>(&sB[buf][swz(lrow + 64 * j, lq)]) = rb[j];
    };

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
            acc[i][j] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    const int fr = lane & 15, kq = lane >> 4;
    int dbg_i = 0;
#define HP_STAMP()                                       \
    if (p.dbg && blockIdx.x == 9 && tid == 0)            \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    // block residency trace: [128 + 3 b] = start, [.. + 1] = end (100 MHz clock), [.. + 2] = XCC_ID << 32 | HW_ID of block b < 4096
    if (p.dbg && tid == 0 && blockIdx.x < 4096) {
        p.dbg[128 + 3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        p.dbg[128 + 3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    gload(0);
    to_lds(0);
    lds_barrier();
    HP_STAMP();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        gload(min(s + 1, steps - 1));
        f32x4 fa[TMW], fb[TNW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(&sA[s & 1][swz(wm * (16 * TMW) + i * 16 + fr, kq)]);
#pragma unroll
        for (int j = 0; j < TNW; ++j)
            fb[j] = *reinterpret_cast<const f32x4*>(&sB[s & 1][swz(wn * (BN / WN) + j * 16 + fr, kq)]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        to_lds((s + 1) & 1);
        lds_barrier();
        if ((s & 7) == 7)
            HP_STAMP();
    }

    // epilogue: lane (n, q) of tile (i, j) holds channels m0 + wm * 16 TMW + i * 16 + 4 q + {0..3} of pixel n0 + wn * BN / WN + j * 16 + n
    const int q = lane >> 4;
    const bool out_vec = p.out.p && ((p.out.coff | p.out.cs) & 3) == 0;
    const bool res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0;
    const bool res_pre = p.res.p && p.res_before_act, res_post = p.res.p && !p.res_before_act;
    f32x4 bs[TMW], sl[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int m = m0 + wm * (16 * TMW) + i * 16 + 4 * q; // (m + 3 < Cout_pad)
        bs[i] = *reinterpret_cast<const f32x4*>(p.bias + m);
        sl[i] = f32x4{ p.act_slope, p.act_slope, p.act_slope, p.act_slope };
        if (p.alpha)
            sl[i] = *reinterpret_cast<const f32x4*>(p.alpha + m);
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 16 + fr;
        const bool pix_ok = n < p.npix;
        const int nc = min(n, p.npix - 1);
        const int b = nc / OHW, rem = nc - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const long ooff = p.out.p ? tv32_off(p.out, b, oy, ox) : 0;
        const long roff = p.res.p ? tv32_off(p.res, b, oy, ox) : 0;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int m = m0 + wm * (16 * TMW) + i * 16 + 4 * q;
            if (!pix_ok || m >= p.Cout)
                continue;
            const bool full = m + 3 < p.Cout;
            float v[4], rr[4] = { 0.f, 0.f, 0.f, 0.f };
            if (p.res.p) {
                if (full && res_vec) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.res.p + roff + m);
                    rr[0] = t[0], rr[1] = t[1], rr[2] = t[2], rr[3] = t[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.Cout)
                            rr[e] = p.res.p[roff + m + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[i][j][e] + bs[i][e];
                if (res_pre)
                    x += rr[e];
                x = x > 0.f ? fminf(x, p.act_hi) : x * sl[i][e];
                if (res_post)
                    x += rr[e];
                v[e] = x;
            }
            if (p.out.p) {
                if (full && out_vec)
                    *reinterpret_cast<f32x4*>(p.out.p + ooff + m) = f32x4{ v[0], v[1], v[2], v[3] };
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.Cout)
                            p.out.p[ooff + m + e] = v[e];
                }
            }
            if (p.out_f32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m + e < p.Cout)
                        p.out_f32[((long)b * p.Cout + m + e) * OHW + rem] = v[e];
            }
        }
    }
    HP_STAMP();
    if (p.dbg && tid == 0 && blockIdx.x < 4096)
        p.dbg[128 + 3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#undef HP_STAMP
}

// ---------------------------------------------------------------------------------------------------
// conv32_wk_kernel ("whole K"; round 6): 1 x 1 layers with 32 / 64 input channels whose time is their memory traffic - ResNet's expansions
// (64 -> 256 at 32 x 96 x 96 with a residual: 75 MB in, 302 MB residual, 302 MB out for 9.7 GFLOP), MobileNet's first pointwise layer (32 -> 64 at
// 8 x 184 x 216).  On conv32_t16_kernel such a layer is two to four K-steps, each behind a barrier with 14 KB of loads in flight per block, then bias,
// then the residual, then the stores, one wait after the other: 205 us for 680 MB (3.3 TB/s; a copy reaches 6.3), 172 of them with the MFMAs
// REMOVED.  What was tried on it, in measured order (profiles/r06_ab_layers_f32_whole_k.txt): bias and residual requested before the
// last K-step (185), persistent blocks (192), staggered starts (197), a barrier-free per-wavefront stream with the weights in registers (203) - and
// this form, which gains with the NUMBER of blocks a CU holds (64 x 160 pixels, two per CU: 188; 64 x 96, three: 172; 64 x 64, four: 165):
//   * a block requests EVERYTHING it will read at once - the weights' 64 rows and its 64 pixels for all K channels (whole pixel rows of K * 4
//     contiguous bytes), the residual and the bias - waits once, and has no block-wide barrier left;
//   * a wavefront owns ALL 64 output channels of 16 pixels: it multiplies the K / 16 slices out of LDS, turns the tile through a private LDS slab
//     (32 channels at a time) into pixel rows - lane = (pixel, 16-byte chunk) - and applies bias / residual / activation there: a store (and a
//     residual load) instruction covers 8 pixels x 128 contiguous bytes.  (Half-line stores - 16 pixels x 64 bytes, what the accumulator layout
//     gives - to a tensor beyond the 256 MB Infinity Cache run at 3.3 TB/s against 5.4 when nobody writes the other half, tools/storebench.hip /
//     profiles/r06_storebench.txt; here the other half follows at once and L2 merges them: whole-line stores measured 168 against 165 us);
//   * pixel -> address arithmetic (two integer divisions per pixel and tensor) is done once per pixel by 64 threads into an LDS table.
// Still far from the traffic's 110 us: with parts removed the layer takes 166 (no activation loads), 133 (no stores), 128 us (no MFMAs) - no single
// resource is the bound, a block's own chain of latencies is, three blocks per CU deep.  What would cover it is a loader that runs tiles ahead
// (LDS-DMA ring per wavefront); not built.
//   tile   64 output channels x 64 pixels, four wavefronts side by side over the pixels, MFMA tiles of 16 x 16 x 4
//   LDS    [K / 16 slices][rows][16 floats] per operand, conv32_t16_kernel's XOR swizzle inside a slice, slices one row apart in bank space
//   K order, accumulation order and epilogue arithmetic (per element) are conv32_t16_kernel's: the same bits.
template <int KT, int BN>
__global__ __launch_bounds__(256) void conv32_wk_kernel(const conv32_params p)
{
    constexpr int BM = 64, KS = KT / 16, CPR = KT / 4, TMW = 4, TNW = BN / 64, WPX = BN / 4;
    constexpr int SA = BM * 16 + 16, SB = BN * 16 + 16;   // slice strides in floats (+ 64 B: the slices of one pixel row land in different bank quads)
    constexpr int SLAB = 16 * 36;                         // a wavefront's slab: 16 pixels x (32 channels + 4) floats
    constexpr int NA = BM * CPR / 256, NBC = (BN * CPR + 255) / 256; // 16-byte chunks per thread: weights, pixels
    static_assert(BN % 64 == 0 && KT % 16 == 0 && (BM * CPR) % 256 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[]; // wk_lds_bytes: the two operands, the slabs, then the address table
    float* const sA = lds;
    float* const sB = lds + KS * SA;
    long* const s_in = reinterpret_cast<long*>(lds + KS * (SA + SB) + 4 * SLAB);
    long* const s_out = s_in + BN;
    long* const s_res = s_out + BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const slab = lds + KS * (SA + SB) + wave * SLAB;
    const int G = p.Cout_pad / BM, nx = (p.npix + BN - 1) / BN; // XCD-aware 1-D block order: see conv32_kernel
    const int bj = blockIdx.x >> 3, ptile = (bj / G) * 8 + (blockIdx.x & 7);
    if (ptile >= nx)
        return;
    const int n0 = ptile * BN, m0 = (bj % G) * BM;
    const int OHW = p.OH * p.OW;
    auto swz = [](int row, int quad) { return row * 16 + ((quad ^ (((row >> 3) & 1) << 1)) << 2); };

    if (tid < BN) { // (pixels past the tensor: the last pixel again - read, multiplied, never stored)
        const int n = min(n0 + tid, p.npix - 1);
        const int b = n / OHW, rem = n - b * OHW;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        s_in[tid] = tv32_off(p.in, b, oy * p.stride - p.pad_t, ox * p.stride - p.pad_l);
        s_out[tid] = tv32_off(p.out, b, oy, ox);
        s_res[tid] = p.res.p ? tv32_off(p.res, b, oy, ox) : 0;
    }
    const int fr = lane & 15, kq = lane >> 4;
    const int rrow = lane >> 3, rc = lane & 7; // row form: lane = (pixel rrow of 8, 16-byte chunk rc of 8) of a half tile
    const bool out_vec = ((p.out.coff | p.out.cs) & 3) == 0;
    const bool res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0 && p.res.cs >= p.Cout_pad; // (whole channel quads readable for every group of the padded matrix)
    const bool res_pre = p.res.p && p.res_before_act, res_post = p.res.p && !p.res_before_act;
    f32x4 bsr[2], slr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m0 + h * 32 + 4 * rc; // (m + 3 < Cout_pad)
        bsr[h] = *reinterpret_cast<const f32x4*>(p.bias + m);
        slr[h] = f32x4{ p.act_slope, p.act_slope, p.act_slope, p.act_slope };
        if (p.alpha)
            slr[h] = *reinterpret_cast<const f32x4*>(p.alpha + m);
    }
    f32x4 va[NA], vb[NBC];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + 256 * i, row = idx / CPR, c = idx % CPR;
        va[i] = *reinterpret_cast<const f32x4*>(p.w + (long)(m0 + row) * p.Cin + c * 4);
    }
    __syncthreads(); // the address table
#pragma unroll
    for (int i = 0; i < NBC; ++i) {
        const int idx = min(tid + 256 * i, BN * CPR - 1), row = idx / CPR, c = idx % CPR;
        vb[i] = *reinterpret_cast<const f32x4*>(p.in.p + s_in[row] + c * 4);
    }
    f32x4 rr[TNW][2][2]; // [tile][half][8-pixel group] in row form
    if (res_vec) {
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long roff = s_res[wave * WPX + j * 16 + 8 * r + rrow] + m0 + 4 * rc;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    rr[j][h][r] = *reinterpret_cast<const f32x4*>(p.res.p + roff + h * 32);
            }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + 256 * i, row = idx / CPR, c = idx % CPR;
        *reinterpret_cast<f32x4*>(&sA[(c >> 2) * SA + swz(row, c & 3)]) = va[i];
    }
#pragma unroll
    for (int i = 0; i < NBC; ++i) {
        const int idx = tid + 256 * i, row = idx / CPR, c = idx % CPR;
        if (idx < BN * CPR)
            *reinterpret_cast<f32x4*>(&sB[(c >> 2) * SB + swz(row, c & 3)]) = vb[i];
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        f32x4 acc[TMW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
            acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
        const int prow0 = wave * WPX + j * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 fa[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(&sA[s * SA + swz(i * 16 + fr, kq)]);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(&sB[s * SB + swz(prow0 + fr, kq)]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[e], acc[i], 0, 0, 0);
        }
        // accumulator form: lane (pixel fr, kq) holds channels i * 16 + 4 kq + {0..3}.  Through the slab, 32 channels at a time, into row form.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<f32x4*>(&slab[fr * 36 + i * 16 + 4 * kq]) = acc[2 * h + i];
            __builtin_amdgcn_wave_barrier(); // (a wavefront's LDS operations complete in order: the barrier only pins the compiler's order)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&slab[(8 * r + rrow) * 36 + 4 * rc]);
                const int prow = prow0 + 8 * r + rrow, m = m0 + h * 32 + 4 * rc;
                if (n0 + prow >= p.npix || m >= p.Cout)
                    continue;
                const bool full = m + 3 < p.Cout;
                const long ooff = s_out[prow] + m;
                float v[4], rv[4] = { 0.f, 0.f, 0.f, 0.f };
                if (p.res.p) {
                    if (res_vec) {
                        rv[0] = rr[j][h][r][0], rv[1] = rr[j][h][r][1], rv[2] = rr[j][h][r][2], rv[3] = rr[j][h][r][3];
                    } else {
                        const long roff = s_res[prow] + m;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (m + e < p.Cout)
                                rv[e] = p.res.p[roff + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = a[e] + bsr[h][e];
                    if (res_pre)
                        x += rv[e];
                    x = x > 0.f ? fminf(x, p.act_hi) : x * slr[h][e];
                    if (res_post)
                        x += rv[e];
                    v[e] = x;
                }
                if (full && out_vec)
                    *reinterpret_cast<f32x4*>(p.out.p + ooff) = f32x4{ v[0], v[1], v[2], v[3] };
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.Cout)
                            p.out.p[ooff + e] = v[e];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int KT, int BN>
static hipError_t launch_wk(const conv32_params& p, dim3 grid, hipStream_t s)
{
    constexpr int lds = KT / 16 * ((64 * 16 + 16) + (BN * 16 + 16)) * 4 + 4 * 16 * 36 * 4 + 3 * BN * 8;
    static bool granted = false;
    if (lds > 64 * 1024 && !granted) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv32_wk_kernel<KT, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess)
            return e;
        granted = true;
    }
    HP_LAUNCH((conv32_wk_kernel<KT, BN>), grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// Which layers take conv32_wk_kernel: 1 x 1 (any stride), exactly 32 / 64 / 128 input channels, an NHWC output only, and enough pixel tiles that
// two blocks per CU stay busy (HP_C32_WK=0: none, the A/B switch; =1: every layer of that shape).  Looks at pick_npix like conv32_pick.
static int conv32_wk_bn(const conv32_params& p)
{
    const int wk = getenv("HP_C32_WK") ? atoi(getenv("HP_C32_WK")) : -1;
    if (wk == 0 || p.KH != 1 || p.KW != 1 || p.dil != 1 || p.out_f32 || !p.out.p || (p.Cin != 32 && p.Cin != 64 && p.Cin != 128))
        return 0;
    // K = 128 (two blocks per CU) wins nothing: ResNet's 128 -> 512 at 32 x 48 x 48 140 -> 135 us, MobileNet's 128 -> 128 at 8 x 92 x 108 38.6 -> 41.6
    // (profiles/r06_ab_layers_f32_whole_k.txt): forced only (the tests)
    if (p.Cin == 128 && wk != 1)
        return 0;
    const int bn = 64; // (LDS: 27 / 44 / 77 KB per block at K = 32 / 64 / 128: five / three / two blocks per CU)
    const long np = p.pick_npix > 0 ? p.pick_npix : p.npix, blocks = (np + bn - 1) / bn * (p.Cout_pad / 64);
    return wk == 1 || blocks >= 768 ? bn : 0;
}

// Block tile (BM output channels x BN pixels) of a layer.  The fp32 matrix pipe is slow enough (64 cycles per MFMA) that small tiles cost
// little per MFMA, and a layer that is one wave of 128 x 128 tiles leaves CUs idle: LW-OpenPose's 3 x 3 128 -> 128 layers at 8 x 46 x 54 pixels
// are 156 such tiles on 256 CUs (0.38 of the fp32 MFMA peak) - as 64 x 64 tiles they are 622 blocks, several per CU.
static void conv32_pick(const conv32_params& p, int& BM, int& BN)
{
    BM = p.Cout_pad % 128 == 0 ? 128 : 64, BN = 128;
    static const int force = getenv("HP_C32_TILE") ? atoi(getenv("HP_C32_TILE")) : 0; // A/B switch: 1 = the largest tile, 2 = 64 x 128 at most
    if (force == 1)
        return;
    if (force == 2) {
        BM = 64;
        return;
    }
    const long blocks = (long)(((p.pick_npix > 0 ? p.pick_npix : p.npix) + 127) / 128) * (p.Cout_pad / BM);
    if (blocks < 448) // fewer than ~1.75 blocks per CU: quarter the tile
        BM = 64, BN = 64;
    else if (blocks < 1024 && BM == 128) // up to four blocks per CU: halve the tile (512 -> 512 at 8 x 46 x 54: 115 -> 100 us alone, same machine time)
        BM = 64;
    // Round quantisation: 64 x 128 tiles have 1024 slots (four blocks per CU); a layer that is "one round and a bit" of them but ONE round of
    // 64 x 160 tiles takes conv32_t16_kernel<160> (see there).  HP_C32_BN160=0: the A/B switch back; =1: every layer (tests).  Read per launch.
    const int bn160 = getenv("HP_C32_BN160") ? atoi(getenv("HP_C32_BN160")) : -1;
    if (bn160 == 1 || bn160 == 176) { // (tests: every layer on the 64 x 160 | 64 x 176 tile)
        BM = 64, BN = bn160 == 1 ? 160 : 176;
        return;
    }
    if (BM == 64 && BN == 128 && bn160 != 0) {
        const long np = p.pick_npix > 0 ? p.pick_npix : p.npix; // (the engine's max_batch geometry: see conv32_params::pick_npix)
        const long g = p.Cout_pad / 64, b128 = (long)((np + 127) / 128) * g, b160 = (long)((np + 159) / 160) * g;
        if (b128 > 1024 && b160 <= 1024)
            BN = 160;
        // (b160 a few blocks over the 1 024 slots - OpenPose-VGG19's 7 x 7 layers at 16 x 54 x 96: 1 296 tiles of 64 x 128, 1 038 of 64 x 160, 944 of
        // 64 x 176 = conv32_t16_kernel<176, 1>, one round: 1 272 -> 1 172 us alone, but 1 039 -> 1 053 with a second stream, whose blocks fill the
        // second round's empty slots anyway: not taken; HP_C32_BN160=176 forces it, tests/test_engine_fp32_gpu.py::test_one_round_tile_64x160)
    }
    // ResNet's 1 x 1 expansions with few input channels (64 -> 256 at 96 x 96, 128 -> 512 at 48 x 48, batch 32): four to eight K-steps, then 300 MB
    // of output + residual - the layer is its epilogue.  conv32_t16_kernel's stores cover 16 pixels x 64 contiguous bytes per instruction where
    // the 32 x 32 accumulator layout covers 64 pixels x 16 bytes: 259 -> 205 us alone, 218 -> 188 with a second stream for 64 -> 256, 146 -> 142 | 130 -> 126
    // for 128 -> 512; every wider-K layer is slower on it (profiles/r06_ab_layers_f32_config3_t16.txt)
    if (bn160 != 0 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.Cin <= 128 && p.Cout_pad >= 256 && !p.out_f32 && (BN == 128))
        BM = 64, BN = 160;
}

static bool conv32_rows(const conv32_params& p, int BN)
{
    if (BN == 160 || BN == 176)
        return false; // conv32_t16_kernel has the one epilogue (16 pixels x 64 contiguous bytes per store instruction)
    // The row-major epilogue pays on the 64-pixel tile only.  Measured per layer of LW-OpenPose @ 8 x 46 x 54 (us alone | with a second
    // stream; rows -> lane = pixel): <64, 64> 256 -> 256 33.2 | 26.9 -> 33.8 | 29.2, 128 -> 256 21.3 | 15.8 -> 21.7 | 19.0;  <64, 128> (two
    // channel tiles per wavefront: eight pixel rows of 256 B in flight per lane group) 128 -> 512 61.5 | 40.3 -> 35.1 | 27.4, 32 -> 64 at
    // 184 x 216 59.2 | 46.6 -> 36.0 | 29.6, 512 -> 512 129 | 93 -> 106 | 94 (profiles/r05_ab_layers_f32_*_epilogue.txt).
    // HP_LANE_EPILOGUE=1: lane form everywhere, HP_LANE_EPILOGUE=-1: row form everywhere.
    return !p.out_f32 && (p.lane_epilogue < 0 || (p.lane_epilogue == 0 && BN == 64));
}

int conv32_tile(const conv32_params& p)
{
    if (const int bn = conv32_wk_bn(p))
        return 39000000 + p.Cin * 1000 + bn;
    int BM, BN;
    conv32_pick(p, BM, BN);
    return 32000000 + (conv32_rows(p, BN) ? 400000 : 0) + BM * 1000 + BN;
}

bool set_act32(conv32_params& p)
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

hipError_t launch_conv32(const conv32_params& p, hipStream_t s)
{
    if (p.Cin % 16 || p.Cout_pad % 64 || p.npix <= 0)
        return hipErrorInvalidValue;
    if (const int bn = conv32_wk_bn(p)) {
        const dim3 g(((p.npix + bn - 1) / bn + 7) / 8 * 8 * (p.Cout_pad / 64));
        return p.Cin == 32 ? launch_wk<32, 64>(p, g, s) : p.Cin == 64 ? launch_wk<64, 64>(p, g, s) : launch_wk<128, 64>(p, g, s);
    }
    int BM, BN;
    conv32_pick(p, BM, BN);
    const dim3 grid(((p.npix + BN - 1) / BN + 7) / 8 * 8 * (p.Cout_pad / BM)); // XCD-aware 1-D order: see conv32_kernel
    const bool rows = conv32_rows(p, BN);
#define HP_C32_CASE(BM_, BN_, WM_, WN_)                                                    \
    if (rows)                                                                              \
        HP_LAUNCH((conv32_kernel<BM_, BN_, WM_, WN_, true>), grid, dim3(256), 0, s, p);    \
    else                                                                                   \
        HP_LAUNCH((conv32_kernel<BM_, BN_, WM_, WN_, false>), grid, dim3(256), 0, s, p);
    if (BN == 176) {
        HP_LAUNCH((conv32_t16_kernel<176, 1>), grid, dim3(256), 0, s, p);
    } else if (BN == 160) {
        // (three blocks per CU instead of four - 12 KB of unused dynamic LDS, to leave registers and LDS to the other pipes' depthwise kernels - cost four
        // pipes 1.3 %: 1.466 -> 1.485 ms of conv stack per batch, two runs each)
        HP_LAUNCH((conv32_t16_kernel<160>), grid, dim3(256), 0, s, p);
    } else if (BM == 128) {
        HP_C32_CASE(128, 128, 2, 2)
    } else if (BN == 128) {
        HP_C32_CASE(64, 128, 1, 4)
    } else {
        HP_C32_CASE(64, 64, 2, 2)
    }
#undef HP_C32_CASE
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// First layer: Cin = 3.  Block = 256 threads = TH x 8 output pixels; thread = 4 horizontally adjacent pixels x 8 output channels (a
// weight read from LDS feeds four FMAs), GP = min(G, 32) channel groups side by side, a thread loops over the groups beyond.  The
// weights and the block's input patch - pre-processed ONCE per input pixel (src/data.cpp:21-51: x factor in double, BGR -> RGB,
// (x - mean) / std; zero outside the image = the convolution's padding of the normalised tensor) - live in LDS; the taps run
// branch-free in (ky, kx, c) order.  (The first form loaded and converted three bytes per tap and thread behind two bounds checks:
// 44 us for the 184 x 216 x 32 stem of LW-OpenPose.)
// HP_FIRST_CONV_VERIFY=1 (diagnostic): after its outputs are computed a block re-reads its staged weights and input patch
// from LDS and compares them with what global memory holds; [0] = patch words that differ, [1] = weight words, [2] = blocks checked
__device__ unsigned g_first_conv_verify[4];
__global__ __launch_bounds__(256) void first_conv32_kernel(const first_conv32_params p, int tiles_x, int tiles_y, int TH, int GP, int verify)
{
    extern __shared__ __attribute__((aligned(16))) float s_w32[]; // [KH*KW*3][Cout_pad8], then the patch [IH][IW][3]
    constexpr int TW = 8, PX = 4;
    const int G = (p.Cout + 7) / 8, CP = G * 8, taps = p.KH * p.KW;
    const int IH = (TH - 1) * p.stride + p.KH, IW = (TW - 1) * p.stride + p.KW;
    float* const s_x = s_w32 + taps * 3 * CP;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * p.stride - p.pad_t, ix0 = ox0 * p.stride - p.pad_l;
    for (int i = threadIdx.x; i < taps * 3 * CP; i += 256) {
        const int co = i % CP, tc = i / CP; // tc = tap * 3 + c
        s_w32[i] = co < p.Cout ? p.w[(size_t)co * taps * 3 + tc] : 0.f;
    }
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        const int py = i / IW, px = i - py * IW;
        const int iy = iy0 + py, ix = ix0 + px;
        float v[3] = { 0.f, 0.f, 0.f };
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            if (p.in_u8) {
                const uint8_t* q = p.in_u8 + (((size_t)b * p.H + iy) * p.W + ix) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[c] = ((float)((double)q[p.flip_rb ? 2 - c : c] * p.factor) - p.mean[c]) * p.inv_std[c]; // src/data.cpp:48
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[c] = (p.in_f32[(((size_t)b * 3 + c) * p.H + iy) * p.W + ix] - p.mean[c]) * p.inv_std[c];
            }
        }
        s_x[i * 3] = v[0], s_x[i * 3 + 1] = v[1], s_x[i * 3 + 2] = v[2];
    }
    __syncthreads();
    const int g0 = threadIdx.x % GP, slot = threadIdx.x / GP; // slot = (row of the tile, left / right half of its eight pixels)
    auto verify_lds = [&]() {
        if (!verify)
            return;
        unsigned bad_x = 0, bad_w = 0;
        for (int i = threadIdx.x; i < IH * IW; i += 256) {
            const int py = i / IW, px = i - py * IW;
            const int iy = iy0 + py, ix = ix0 + px;
            float v[3] = { 0.f, 0.f, 0.f };
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && p.in_u8) {
                const uint8_t* q = p.in_u8 + (((size_t)b * p.H + iy) * p.W + ix) * 3;
                for (int c = 0; c < 3; ++c)
                    v[c] = ((float)((double)q[p.flip_rb ? 2 - c : c] * p.factor) - p.mean[c]) * p.inv_std[c];
            }
            for (int c = 0; c < 3; ++c)
                bad_x += __float_as_uint(s_x[i * 3 + c]) != __float_as_uint(v[c]);
        }
        for (int i = threadIdx.x; i < taps * 3 * CP; i += 256) {
            const int co = i % CP, tc = i / CP;
            const float wv = co < p.Cout ? p.w[(size_t)co * taps * 3 + tc] : 0.f;
            bad_w += __float_as_uint(s_w32[i]) != __float_as_uint(wv);
        }
        if (bad_x)
            atomicAdd(&g_first_conv_verify[0], bad_x);
        if (bad_w)
            atomicAdd(&g_first_conv_verify[1], bad_w);
        if (threadIdx.x == 0)
            atomicAdd(&g_first_conv_verify[2], 1u);
    };
    if (slot >= TH * (TW / PX)) {
        verify_lds();
        return;
    }
    const int ly = slot / (TW / PX), lx0 = (slot % (TW / PX)) * PX;
    const int oy = oy0 + ly;
    if (oy >= p.OH) {
        verify_lds();
        return;
    }
    const bool vec_ok = ((p.out.coff | p.out.cs) & 3) == 0;
    for (int g = g0; g < G; g += GP) {
        float acc[PX][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float bv = (g * 8 + r < p.Cout) ? p.bias[g * 8 + r] : 0.f;
#pragma unroll
            for (int j = 0; j < PX; ++j)
                acc[j][r] = bv;
        }
        for (int ky = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx) {
                const float* xp = s_x + ((ly * p.stride + ky) * IW + lx0 * p.stride + kx) * 3;
                const float* wt = s_w32 + (size_t)((ky * p.KW + kx) * 3) * CP + g * 8;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + c * CP);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + c * CP + 4);
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const float xv = xp[j * p.stride * 3 + c]; // (a tap in the padding is 0: fma(0, w, acc) = acc, the same value as skipping it)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[j][e] = fmaf(xv, w0[e], acc[j][e]), acc[j][4 + e] = fmaf(xv, w1[e], acc[j][4 + e]);
                    }
                }
            }
        if (verify) { // the same taps a second time: a difference is arithmetic / register state that changed under the kernel ([3])
            float acc2[PX][8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float bv = (g * 8 + r < p.Cout) ? p.bias[g * 8 + r] : 0.f;
#pragma unroll
                for (int j = 0; j < PX; ++j)
                    acc2[j][r] = bv;
            }
            for (int ky = 0; ky < p.KH; ++ky)
                for (int kx = 0; kx < p.KW; ++kx) {
                    const float* xp = s_x + ((ly * p.stride + ky) * IW + lx0 * p.stride + kx) * 3;
                    const float* wt = s_w32 + (size_t)((ky * p.KW + kx) * 3) * CP + g * 8;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + c * CP);
                        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + c * CP + 4);
#pragma unroll
                        for (int j = 0; j < PX; ++j) {
                            const float xv = xp[j * p.stride * 3 + c];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc2[j][e] = fmaf(xv, w0[e], acc2[j][e]), acc2[j][4 + e] = fmaf(xv, w1[e], acc2[j][4 + e]);
                        }
                    }
                }
            unsigned bad = 0;
#pragma unroll
            for (int j = 0; j < PX; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    bad += __float_as_uint(acc[j][r]) != __float_as_uint(acc2[j][r]);
            if (bad)
                atomicAdd(&g_first_conv_verify[3], bad);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int ox = ox0 + lx0 + j;
            if (ox >= p.OW)
                break;
            float* op = p.out.p + tv32_off(p.out, b, oy, ox) + g * 8;
            if (g * 8 + 7 < p.Cout && vec_ok) {
                f32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o0[e] = act32(acc[j][e], p.act, p.act_param, 0.f), o1[e] = act32(acc[j][4 + e], p.act, p.act_param, 0.f);
                *reinterpret_cast<f32x4*>(op) = o0, *reinterpret_cast<f32x4*>(op + 4) = o1;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (g * 8 + r < p.Cout)
                        op[r] = act32(acc[j][r], p.act, p.act_param, 0.f);
            }
        }
    }
    verify_lds();
}

void first_conv32_verify_counts(unsigned out[4], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_first_conv_verify), 16);
    if (reset) {
        const unsigned z[4] = { 0, 0, 0, 0 };
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_first_conv_verify), z, 16);
    }
}

hipError_t launch_first_conv32(const first_conv32_params& p, hipStream_t s)
{
    const int G = (p.Cout + 7) / 8, GP = std::min(G, 32);
    const int TH = 256 / GP / 2; // two threads (of four pixels) per row of eight
    const int IH = (TH - 1) * p.stride + p.KH, IW = 7 * p.stride + p.KW;
    const size_t lds = ((size_t)p.KH * p.KW * 3 * G * 8 + (size_t)IH * IW * 3) * sizeof(float);
    if (lds > 64 * 1024)
        return hipErrorInvalidValue;
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + TH - 1) / TH;
    HP_LAUNCH(first_conv32_kernel, dim3(tiles_x * tiles_y * p.B), dim3(256), lds, s, p, tiles_x, tiles_y, TH, GP, getenv("HP_FIRST_CONV_VERIFY") ? 1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Depthwise 3 x 3: one thread = 4 channels of one output COLUMN segment, marching down RUN output rows with the 2 D + 1 input rows it
// needs in registers: every input value is loaded once per thread instead of once per tap row (3 loads per output at stride 1 instead
// of 9 - the per-pixel form ran at 2.1 TB/s of its compulsory bytes, bound by the vector memory path, not by HBM).  Taps in the padding
// read the zero halo; the tap order per output is (ky, kx) ascending as in the fp16 kernels.
// PX = 2 (stride 1): a thread owns TWO output columns D apart - their windows share two of three tap columns, so a row of the window is
// four loads for two outputs instead of six (1.5 + 1 requests of the vector memory path per output instead of 3 + 1: that path, not HBM,
// bounds the kernel - 512 channels at 8 x 46 x 54 ran at 3.3 TB/s of its compulsory bytes).  Each output is the same chain of fmaf as before.
template <int S, int D, int PX>
__global__ __launch_bounds__(256) void dwconv32_kernel(const dw32_params p, int run)
{
    static_assert(PX == 1 || (PX == 2 && S == 1), "column pairs share taps at stride 1 only");
    constexpr int NR = 2 * D + 1; // input rows one output needs, from its first to its last tap row
    constexpr int NC = 2 + PX;    // tap columns of a window row: offsets 0, D, 2 D (, 3 D)
    const int CG = p.C / 4, segs = (p.OH + run - 1) / run;
    const int ncol = PX == 1 ? p.OW : (p.OW + 2 * D - 1) / (2 * D) * D; // column slots: PX = 2 pairs columns (j, j + D) inside groups of 2 D
    const long total = (long)p.B * segs * ncol * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        long n = i / CG;
        const int q = (int)(n % ncol);
        n /= ncol;
        const int ox = PX == 1 ? q : (q / D) * 2 * D + q % D;
        if (ox >= p.OW)
            continue; // (a ragged last group of 2 D columns)
        const int seg = (int)(n % segs), b = (int)(n / segs);
        const int oy0 = seg * run, oy1 = min(oy0 + run, p.OH);
        f32x4 w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w[t] = *reinterpret_cast<const f32x4*>(p.w + t * p.C + cg * 4);
        const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
        // input row r of the window = tensor row oy * S - pad_t + r; tap columns at ox * S - pad_l + {0, D, 2 D (, 3 D)}.  A column past the
        // tensor's halo (the second output of a pair beyond the map) is clamped to it: it only feeds an output that is not stored
        const int cx0 = ox * S - p.pad_l, cmax = p.W - 1 + (2 * D - p.pad_l);
        long coff[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
            coff[c] = (long)(min(cx0 + c * D, cmax) - cx0) * p.in.cs;
        const float* const col0 = p.in.p + tv32_off(p.in, b, oy0 * S - p.pad_t, cx0) + cg * 4;
        const long rstride = (long)p.in.wp * p.in.cs;
        f32x4 x[NR][NC];
        auto load_row = [&](int r, long row) { // window slot r <- input row `row` (relative to col0)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                x[r][c] = *reinterpret_cast<const f32x4*>(col0 + row * rstride + coff[c]);
        };
#pragma unroll
        for (int r = 0; r < NR; ++r)
            load_row(r, r);
        float* op = p.out.p + tv32_off(p.out, b, oy0, ox) + cg * 4;
        const long ostride = (long)p.out.wp * p.out.cs, o2 = (long)D * p.out.cs;
        const bool second = PX == 2 && ox + D < p.OW;
        for (int oy = oy0; oy < oy1; ++oy) {
#pragma unroll
            for (int px = 0; px < PX; ++px) {
                f32x4 acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[e] = fmaf(x[ky * D][kx + px][e], w[ky * 3 + kx][e], acc[e]);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = act32(acc[e], p.act, p.act_param, 0.f);
                if (px == 0 || second)
                    *reinterpret_cast<f32x4*>(op + px * o2) = o;
            }
            op += ostride;
            // slide the window down by S rows: keep NR - S rows, load S new ones (rows past the tensor's halo are never used: the last
            // outputs of a segment load rows that only the NEXT output would read - clamp them to the window's last valid row)
#pragma unroll
            for (int r = 0; r + S < NR; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    x[r][c] = x[r + S][c];
            if (oy + 1 < oy1) {
#pragma unroll
                for (int r = NR - S; r < NR; ++r)
                    load_row(r, (long)(oy + 1 - oy0) * S + r);
            }
        }
    }
}

// any other stride / dilation: one thread = one output pixel x 4 channels, nine loads per output
__global__ __launch_bounds__(256) void dwconv32_any_kernel(const dw32_params p)
{
    const int CG = p.C / 4;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        f32x4 acc = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
        const float* const x0 = p.in.p + tv32_off(p.in, b, oy * p.stride - p.pad_t, ox * p.stride - p.pad_l) + cg * 4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(x0 + ((long)(ky * p.dil) * p.in.wp + kx * p.dil) * p.in.cs);
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (ky * 3 + kx) * p.C + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[e] = fmaf(x[e], w[e], acc[e]);
            }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = act32(acc[e], p.act, p.act_param, 0.f);
        *reinterpret_cast<f32x4*>(p.out.p + tv32_off(p.out, b, oy, ox) + cg * 4) = o;
    }
}

hipError_t launch_dwconv32(const dw32_params& p, hipStream_t s)
{
    if (p.C % 4)
        return hipErrorInvalidValue;
    // rows per thread: long runs re-use more, short runs give more threads; 8 keeps > 100 k threads on the 46 x 54 maps of LW-OpenPose
    const int run = p.OH >= 32 ? 8 : 4;
    const int px_env = getenv("HP_DW32_PX") ? atoi(getenv("HP_DW32_PX")) : 0; // A/B switch and test hook: 1 = one column per thread always, 2 = pairs always (read per launch: tests toggle it)
    const bool pairs = px_env != 1;
    // column pairs where they still leave > 120 k threads.  LW-OpenPose @ 8 x 46 x 54, us alone | with a second stream, one column -> pairs: 512 channels
    // 24.4 | 19.5 -> 20.0 | 16.6, dilation 2 34.0 | 29.4 -> 27.1 | 22.3, 128 channels at 92 x 108 20.5 | 15.4 -> 17.6 | 14.8; 256 channels (83 k pair
    // threads) 12.0 | 9.0 -> 13.9 | 9.5: those keep one column per thread
    const long pair_threads = (long)p.B * ((p.OH + run - 1) / run) * ((p.OW + 2 * p.dil - 1) / (2 * p.dil) * p.dil) * (p.C / 4);
    const bool two = pairs && p.stride == 1 && (p.dil == 1 || p.dil == 2) && (pair_threads >= 120000 || px_env == 2);
    const int ncol = two ? (p.OW + 2 * p.dil - 1) / (2 * p.dil) * p.dil : p.OW;
    const long total = (long)p.B * ((p.OH + run - 1) / run) * ncol * (p.C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 32);
    if (p.stride == 1 && p.dil == 1) {
        if (two)
            HP_LAUNCH((dwconv32_kernel<1, 1, 2>), dim3(blocks), dim3(256), 0, s, p, run);
        else
            HP_LAUNCH((dwconv32_kernel<1, 1, 1>), dim3(blocks), dim3(256), 0, s, p, run);
    } else if (p.stride == 2 && p.dil == 1)
        HP_LAUNCH((dwconv32_kernel<2, 1, 1>), dim3(blocks), dim3(256), 0, s, p, run);
    else if (p.stride == 1 && p.dil == 2) {
        if (two)
            HP_LAUNCH((dwconv32_kernel<1, 2, 2>), dim3(blocks), dim3(256), 0, s, p, run);
        else
            HP_LAUNCH((dwconv32_kernel<1, 2, 1>), dim3(blocks), dim3(256), 0, s, p, run);
    }
    else {
        const long px = (long)p.B * p.OH * p.OW * (p.C / 4);
        HP_LAUNCH(dwconv32_any_kernel, dim3((int)std::min<long>((px + 255) / 256, 256 * 32)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool32_kernel(const pool32_params p)
{
    const int CG = p.C / 4;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        f32x4 m = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
        for (int ky = 0; ky < p.k; ++ky) {
            const int iy = oy * p.stride - p.pad_t + ky;
            if (iy < 0 || iy >= p.H) // SAME max-pool pads with -inf, not with the zero halo
                continue;
            for (int kx = 0; kx < p.k; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx;
                if (ix < 0 || ix >= p.W)
                    continue;
                const f32x4 x = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, iy, ix) + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    m[e] = fmaxf(m[e], x[e]);
            }
        }
        *reinterpret_cast<f32x4*>(p.out.p + tv32_off(p.out, b, oy, ox) + cg * 4) = m;
    }
}

hipError_t launch_maxpool32(const pool32_params& p, hipStream_t s)
{
    if (p.C % 4)
        return hipErrorInvalidValue;
    const long total = (long)p.B * p.OH * p.OW * (p.C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 32);
    HP_LAUNCH(maxpool32_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// Integer up-scaling: nearest, or bilinear with half-pixel centres (the fp16 kernel's definition, conv_kernels.hip upsample_kernel).
__global__ __launch_bounds__(256) void upsample32_kernel(const pool32_params p)
{
    const int CG = p.C / 4, sc = p.stride;
    const float inv = 1.f / (float)sc;
    const long total = (long)p.B * p.OH * p.OW * CG;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        const long n = i / CG;
        const int ox = (int)(n % p.OW);
        const long t = n / p.OW;
        const int oy = (int)(t % p.OH), b = (int)(t / p.OH);
        f32x4 o;
        if (p.k == 0) {
            o = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, oy / sc, ox / sc) + cg * 4);
        } else {
            const float sy = fmaxf(((float)oy + 0.5f) * inv - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * inv - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = min(y0 + 1, p.H - 1), x1 = min(x0 + 1, p.W - 1);
            const float fy = sy - (float)y0, fx = sx - (float)x0;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, y0, x0) + cg * 4);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, y0, x1) + cg * 4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, y1, x0) + cg * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.in.p + tv32_off(p.in, b, y1, x1) + cg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float top = a[e] * (1.f - fx) + bq[e] * fx;
                const float bot = c[e] * (1.f - fx) + d[e] * fx;
                o[e] = top * (1.f - fy) + bot * fy;
            }
        }
        *reinterpret_cast<f32x4*>(p.out.p + tv32_off(p.out, b, oy, ox) + cg * 4) = o;
    }
}

hipError_t launch_upsample32(const pool32_params& p, hipStream_t s)
{
    if (p.C % 4)
        return hipErrorInvalidValue;
    const long total = (long)p.B * p.OH * p.OW * (p.C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 32);
    HP_LAUNCH(upsample32_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void output_transform32_kernel(tview32 in, int B, int H, int W, out_xform x, float* __restrict__ out)
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
        float v = in.p[tv32_off(in, b, y, xx) + cin];
        int act = x.act;
        if (x.group > 0) {
            const int comp = c % x.group;
            act = ((x.sigmoid_mask >> comp) & 1u) ? ACT_SIGMOID : (((x.softplus_mask >> comp) & 1u) ? ACT_SOFTPLUS : ACT_NONE);
        }
        v = act32(v, act, 0.f, 0.f);
        if (x.grid == 1)
            v += (float)ox;
        else if (x.grid == 2)
            v += (float)oy;
        out[i] = v * x.scale;
    }
}

hipError_t launch_output_transform32(tview32 in, int B, int H, int W, const out_xform& x, float* out, hipStream_t s)
{
    const long total = (long)B * (x.C / (x.shuffle * x.shuffle)) * x.out_h * x.out_w;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 32);
    HP_LAUNCH(output_transform32_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W, x, out);
    return hipGetLastError();
}

} // namespace hp
