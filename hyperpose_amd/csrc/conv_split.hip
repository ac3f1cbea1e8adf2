// conv_split.hip — fp32-accurate dense convolution on the fp16 matrix pipe (interface: conv_fp32.hpp, launch_conv32_split).
//
// The fp32 matrix pipe (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 pipe.  An fp32 value x splits EXACTLY into two fp16 numbers
// and a remainder below fp32's own rounding:   x = hi + 2^-11 lo + r,   hi = fp16(x),  lo = fp16((x - hi) * 2^11),  |r| <= 2^-24 |x|
// (hi carries 11 significant bits, x - hi is exact in fp32 and has at most 13, lo keeps 11 of them).  Then
//      a b = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + O(2^-22 a b)
// and every product on the right is a product of two fp16 numbers - EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_f16.  Three
// MFMAs of the 2.5 PFLOP/s pipe replace sixteen-rate-units of the 157 TFLOP/s pipe: 16 / 3 = 5.3 x the ceiling of conv32_kernel, with
// products good to ~2^-22 relative (fp32: exact products) and the same fp32 accumulation.  Two accumulators per tile (the hi-hi sum
// and the cross sum, combined once in the epilogue as acc0 + 2^-11 acc1) keep lo in fp16's normal range for every |x| >= 2^-12; smaller
// values degrade gracefully to an ABSOLUTE error of 2^-36.  |x| > 65504 does not fit fp16: the staging code raises a sticky flag
// (conv32_params::ovf) and the engine re-runs on the fp32 pipe (engine.cpp).  VERDICT r4 item 2, step 2.
//
// Kernel shape (stride 1, dilation 1, KS = 1 or 3 - every dense layer of LW-OpenPose, VGG and the ResNet 3x3 / 1x1 layers at stride 1):
//   block  = WM wavefronts; wavefront w owns 64 output channels (two 32-row MFMA tiles) x the block's 8 x 8 output pixels (two 32-pixel
//            tiles of 4 rows x 8 columns): 2 x 2 tiles x 2 accumulators = 128 accumulator registers;
//   K loop = chunks of 32 input channels.  A chunk's halo tile ((8 + KS - 1)^2 pixels x 32 channels) is read from HBM as fp32 ONCE per
//            block, split, and stored in LDS as [pixel][32 hi | 32 lo] halves (144-byte pixel rows, pixel-row pitch = 8 mod 16 sixteen-byte
//            units: every ds_read_b128 service group of MI355X_MICROARCH.md's LDS table touches 16 distinct bank quads); it then serves
//            all KS*KS taps x 2 k16 steps.  The next chunk's fp32 values are requested into registers before the chunk's MFMAs start.
//   A      = weights pre-split on the host, in MFMA-fragment order [chunk][tap][k16][32-row tile][hi | lo][lane][8 halves]: a wavefront's
//            four fragments of one k16 step are 4 KB contiguous, straight from L2 into registers, two steps ahead.
//   per k16 step and wavefront: 4 A fragments + 4 B fragments (ds_read_b128) feed 12 MFMAs (384 matrix-pipe cycles).
#include "conv_fp32.hpp"

#include "conv_device.hpp"

#include <algorithm>
#include <cstdlib>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int SPLIT_CK = 32; // input channels per chunk at 3 x 3 (1 x 1: 64), and the granularity a layer's channel slice must have

__device__ __forceinline__ long tv32s_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

template <int KS, int CK>
struct split_geom {
    static constexpr int HP = 8 + KS - 1;                          // halo tile is HP x HP pixels
    static constexpr int PBU = CK / 4 + 1;                         // 16-byte units per halo pixel: CK hi + CK lo halves + one unit of padding (9 / 17: odd)
    static constexpr int PB = PBU * 16;
    static constexpr int RP = ((HP * PBU + 7) / 16 * 16 + 8) * 16; // pixel-row pitch in bytes: >= HP pixels, = 8 mod 16 units
    static constexpr int LDS_BYTES = HP * RP;
    static constexpr int QPP = CK / 4;                             // float4 quads per pixel
    static constexpr int QUADS = HP * HP * QPP;                    // ... per chunk
    static constexpr int SPC = KS * KS * (CK / 16);                // k16 steps per chunk
    static constexpr int RING = SPC % 3 == 0 ? 3 : 4;              // A-fragment ring: the step in use + two in flight (SPC % RING == 0)
    static_assert(SPC % RING == 0 && SPC % 2 == 0, "ring / double buffer periods");
};

} // namespace

template <int KS, int CK, int WM>
__global__ __launch_bounds__(64 * WM) void conv32_split_kernel(const conv32_params p, int tiles_x, int tiles_y)
{
    using G = split_geom<KS, CK>;
    constexpr int NT = 64 * WM, HP = G::HP, RP = G::RP, PB = G::PB, KQ = CK / 16, SPC = G::SPC, RING = G::RING;
    constexpr int NQ = (G::QUADS + NT - 1) / NT; // float4 per thread and chunk
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * 8, x0 = tx * 8;
    const int MT = p.Cout_pad / 32, mt0 = (blockIdx.y * WM + wave) * 2;
    const int nch = p.Cin / CK;

    // ---- staging geometry: quad q of a chunk = (halo pixel q / 8, channels 4 (q % 8) ..)
    long goff[NQ];
    int soff[NQ];
    bool qok[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * NT;
        const int hp = min(q, G::QUADS - 1) / G::QPP, c4 = q % G::QPP;
        const int hy = hp / HP, hx = hp - hy * HP;
        // (the tensor's zero halo covers the padding rows / columns below and right of the image too; pixels of a ragged last tile
        // beyond it are clamped to it and zeroed - they only feed output pixels that are never stored)
        const int y = y0 + hy - p.pad_t, x = x0 + hx - p.pad_l;
        const int ymax = p.H - 1 + (KS - 1 - p.pad_t), xmax = p.W - 1 + (KS - 1 - p.pad_l);
        qok[i] = y <= ymax && x <= xmax && q < G::QUADS;
        goff[i] = tv32s_off(p.in, b, min(y, ymax), min(x, xmax)) + c4 * 4;
        soff[i] = hy * RP + hx * PB + c4 * 8;
    }
    f32x4 stage[NQ];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            stage[i] = *reinterpret_cast<const f32x4*>(p.in.p + goff[i] + c * CK);
    };
    unsigned ovf = 0;
    auto to_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (i * NT + tid < G::QUADS) {
                _Float16 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = qok[i] ? stage[i][e] : 0.f;
                    h[e] = (_Float16)x;
                    l[e] = (_Float16)((x - (float)h[e]) * 2048.f);
                    ovf |= __builtin_fabsf(x) > 65504.f;
                }
                typedef _Float16 half4v __attribute__((ext_vector_type(4)));
                *reinterpret_cast<half4v*>(lds + soff[i]) = half4v{ h[0], h[1], h[2], h[3] };
                *reinterpret_cast<half4v*>(lds + soff[i] + CK * 2) = half4v{ l[0], l[1], l[2], l[3] };
            }
        }
    };

    // ---- A fragments: steps run (chunk, tap, k16) in packing order; a step's four fragments of this wavefront are 4 KB contiguous
    const long step_stride = (long)MT * 1024; // halves per k16 step: MT tiles x (hi, lo) x 512
    const _Float16* wp = p.w_split + (long)mt0 * 1024 + lane * 8;
    const int nsteps = nch * SPC;
    half8 fa[RING][4]; // ring: the step in use and two in flight
    auto aload = [&](int slot, int s) {
        const _Float16* q = wp + (long)min(s, nsteps - 1) * step_stride;
#pragma unroll
        for (int f = 0; f < 4; ++f)
            fa[slot][f] = *reinterpret_cast<const half8*>(q + f * 512);
    };

    floatx16 acc[2][2][2]; // [m tile][n tile][hi-hi | cross]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][0][r] = 0.f, acc[i][j][1][r] = 0.f;

    // B fragment base of n-tile j: pixel (4 j + (n >> 3), n & 7) of the tile = halo pixel (.. + ky, .. + kx) at tap (ky, kx)
    const int n = lane & 31, fk = lane >> 5;
    int bbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        bbase[j] = (4 * j + (n >> 3)) * RP + (n & 7) * PB + fk * 16;

    gload(0);
    aload(0, 0);
    aload(1, 1);
    int s = 0; // global step index
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        if (c)
            lds_barrier(); // every wavefront is done reading the previous chunk's tile
        to_lds();
        if (c + 1 < nch)
            gload(c + 1);
        lds_barrier();
        half8 fb[2][2][2]; // [buffer][n tile][hi | lo]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[0][j][0] = *reinterpret_cast<const half8*>(lds + bbase[j]);
            fb[0][j][1] = *reinterpret_cast<const half8*>(lds + bbase[j] + CK * 2);
        }
#pragma unroll
        for (int st = 0; st < SPC; ++st) { // st = tap * KQ + k16 step of the chunk (the packing order of the weights)
            constexpr int dummy = 0;
            (void)dummy;
            const int cur = st & 1;
            // B fragments of the next step of this chunk (the last step re-reads its own: harmless, keeps the loop uniform)
            {
                const int nst = st + 1 < SPC ? st + 1 : st, ntap = nst / KQ, nks = nst % KQ;
                const int toff = (ntap / KS) * RP + (ntap % KS) * PB + nks * 32;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fb[cur ^ 1][j][0] = *reinterpret_cast<const half8*>(lds + bbase[j] + toff);
                    fb[cur ^ 1][j][1] = *reinterpret_cast<const half8*>(lds + bbase[j] + toff + CK * 2);
                }
            }
            const int slot = st % RING; // (SPC % RING == 0: the ring position is a compile-time function of st in every chunk)
            aload((st + 2) % RING, s + 2);
            // hi-hi first (four independent accumulators), then the two cross products of every tile
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][2 * i], fb[cur][j][0], acc[i][j][0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][2 * i], fb[cur][j][1], acc[i][j][1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][2 * i + 1], fb[cur][j][0], acc[i][j][1], 0, 0, 0);
            // pin the issue order of the step (hipcc otherwise sinks every load to just before its first use and the wavefront - alone on
            // its SIMD - eats the latency): MFMA, LDS read, ... (the next step's B), MFMA, L2 read, ... (A two steps ahead), 4 MFMAs
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0); // nothing crosses a step boundary: the reads above belong to LATER steps and must stay here
            ++s;
        }
    }
    if (ovf && p.ovf)
        atomicOr(p.ovf, 1u);

    // ---- epilogue: lane (n, fk) of a 32 x 32 tile holds rows (r & 3) + 8 (r >> 2) + 4 fk of column n: four consecutive channels per r >> 2
    const bool out_vec = p.out.p && ((p.out.coff | p.out.cs) & 3) == 0;
    const bool res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0;
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int oy = y0 + 4 * j + (n >> 3), ox = x0 + (n & 7);
        const bool pix_ok = oy < p.OH && ox < p.OW;
        const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 1);
        const long ooff = p.out.p ? tv32s_off(p.out, b, oyc, oxc) : 0;
        const long roff = p.res.p ? tv32s_off(p.res, b, oyc, oxc) : 0;
        const int rem = oyc * p.OW + oxc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = (mt0 + i) * 32 + 8 * q + 4 * fk;
                if (pix_ok && m < p.Cout) {
                    const bool full = m + 3 < p.Cout;
                    float v[4], rr[4] = { 0.f, 0.f, 0.f, 0.f };
                    if (p.res.p) {
                        if (full && res_vec) {
                            const f32x4 tt = *reinterpret_cast<const f32x4*>(p.res.p + roff + m);
                            rr[0] = tt[0], rr[1] = tt[1], rr[2] = tt[2], rr[3] = tt[3];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (m + e < p.Cout)
                                    rr[e] = p.res.p[roff + m + e];
                        }
                    }
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(p.bias + m);
                    f32x4 sl = { p.act_slope, p.act_slope, p.act_slope, p.act_slope };
                    if (p.alpha)
                        sl = *reinterpret_cast<const f32x4*>(p.alpha + m);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = __builtin_fmaf(acc[i][j][1][4 * q + e], 1.f / 2048.f, acc[i][j][0][4 * q + e]) + bs[e];
                        if (p.res.p && p.res_before_act)
                            x += rr[e];
                        x = x > 0.f ? fminf(x, p.act_hi) : x * sl[e];
                        if (p.res.p && !p.res_before_act)
                            x += rr[e];
                        v[e] = x;
                    }
                    if (p.out.p) {
                        if (full && out_vec) {
                            f32x4 tt = { v[0], v[1], v[2], v[3] };
                            *reinterpret_cast<f32x4*>(p.out.p + ooff + m) = tt;
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

// Which layers take the split kernel: square 1 x 1 / 3 x 3, stride 1, dilation 1, channel slice readable in whole 32-channel chunks.
bool conv32_split_ok(const conv32_params& p)
{
    return p.KH == p.KW && ((p.KH == 1 && p.Cin % 64 == 0) || (p.KH == 3 && p.Cin % SPLIT_CK == 0)) && p.stride == 1 && p.dil == 1 && p.Cout_pad % 64 == 0
        && p.OH == p.H && p.OW == p.W && p.pad_t == (p.KH - 1) / 2 && p.pad_l == (p.KW - 1) / 2;
}

// Wavefronts per block: all of them share one 8 x 8 pixel tile, so more wavefronts amortise the tile's staging over more output channels -
// but a layer needs enough blocks for 256 CUs.
static int split_wm(const conv32_params& p)
{
    const int groups = p.Cout_pad / 64;
    const long tiles = (long)p.B * ((p.OH + 7) / 8) * ((p.OW + 7) / 8);
    static const int force = getenv("HP_SPLIT_WM") ? atoi(getenv("HP_SPLIT_WM")) : 0;
    for (int wm : { 4, 2, 1 }) {
        if (force && wm != force)
            continue;
        if (groups % wm == 0 && (force || wm == 1 || tiles * (groups / wm) >= 512))
            return wm;
    }
    return 1;
}

int conv32_split_tile(const conv32_params& p) { return 33000000 + p.KH * 1000 + split_wm(p); }

// Host side of the split: [chunk][tap][k16][32-row tile][hi | lo][lane][8] halves from the packed fp32 matrix [tap][Cout_pad][Cin].
void conv32_split_pack(const float* packed, int taps, int cout_pad, int cin, _Float16* out)
{
    const int ck = taps == 1 ? 64 : SPLIT_CK, kq = ck / 16, nch = cin / ck, MT = cout_pad / 32;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < taps; ++t)
            for (int ks = 0; ks < kq; ++ks)
                for (int mt = 0; mt < MT; ++mt) {
                    _Float16* dst = out + ((((size_t)(c * taps + t) * kq + ks) * MT + mt) * 2) * 512;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int m = mt * 32 + (lane & 31), k = c * ck + ks * 16 + (lane >> 5) * 8 + e;
                            const float x = packed[((size_t)t * cout_pad + m) * cin + k];
                            const _Float16 h = (_Float16)x;
                            dst[lane * 8 + e] = h;
                            dst[512 + lane * 8 + e] = (_Float16)((x - (float)h) * 2048.f);
                        }
                }
}

hipError_t launch_conv32_split(const conv32_params& p, hipStream_t s)
{
    if (!conv32_split_ok(p) || !p.w_split || p.npix <= 0)
        return hipErrorInvalidValue;
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 7) / 8, wm = split_wm(p);
    const dim3 grid(tiles_x * tiles_y * p.B, p.Cout_pad / (64 * wm));
#define HP_SPLIT_CASE(KS_, CK_, WM_)                                                                              \
    if (p.KH == KS_ && wm == WM_) {                                                                               \
        HP_LAUNCH((conv32_split_kernel<KS_, CK_, WM_>), grid, dim3(64 * WM_), 0, s, p, tiles_x, tiles_y);         \
        return hipGetLastError();                                                                                 \
    }
    HP_SPLIT_CASE(1, 64, 1)
    HP_SPLIT_CASE(1, 64, 2)
    HP_SPLIT_CASE(1, 64, 4)
    HP_SPLIT_CASE(3, 32, 1)
    HP_SPLIT_CASE(3, 32, 2)
    HP_SPLIT_CASE(3, 32, 4)
#undef HP_SPLIT_CASE
    return hipErrorInvalidValue;
}

} // namespace hp
