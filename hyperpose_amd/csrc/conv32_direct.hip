// conv32_direct.hip — the barrier-free direct convolution of the fp32 engines (interface: conv_fp32.hpp), in two arithmetic modes:
//
//   SPLIT = false  (HP_DTYPE_F32)   exact fp32 products and sums on v_mfma_f32_32x32x2_f32 - conv32_kernel's arithmetic in a form without a
//                                   barrier per K-step: see "Kernel shape" below;
//   SPLIT = true   (HP_DTYPE_F32S)  fp32-accurate products on the fp16 matrix pipe.  The fp32 pipe runs at 1/16 of the fp16 pipe.  An fp32
//       value x splits into two fp16 numbers and a remainder below fp32's own rounding:
//                 x = hi + 2^-11 lo + r,     hi = fp16(x),   lo = fp16((x - hi) * 2^11),   |r| <= 2^-23 |x|
//       (hi carries 11 significant bits; x - hi is exact in fp32 and at most 12 bits wide; lo keeps 11 of them).  Then
//                 a b = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + O(2^-22 a b)
//       and every product on the right is a product of two fp16 numbers - EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_f16.  Three
//       MFMAs of the 2.5 PFLOP/s pipe replace 16 rate-units of the 157 TFLOP/s pipe: 833 TFLOP/s of fp32-equivalent work, 5.3 x conv32_kernel's
//       ceiling, with products good to ~2^-22 relative (fp32: exact) and the same fp32 accumulation.  Two accumulators per tile (the hi-hi
//       sum and the cross sum, combined once in the epilogue as acc0 + 2^-11 acc1) keep lo in fp16's normal range for every |x| >= 2^-12;
//       smaller values degrade gracefully to an ABSOLUTE error of 2^-36.  |x| > 65504 does not fit fp16: the staging code raises a sticky
//       flag (conv32_params::ovf) and the engine re-runs the batch on the fp32 pipe (engine.cpp).  Measured against the pure fp32 oracle on
//       LW-OpenPose @ 368x432x8: 2.4e-6 of the output scale (the fp32 pipe: 2.65e-6), 9755 / 9755 peaks and 1533 / 1533 key-points identical
//       (tests/test_pipeline_gpu.py::test_split_engine_vs_fp32_oracle_keypoint_drift).  VERDICT r4 item 2, step 2.
//
// Kernel shape (stride 1, dilation 1, SAME padding, KS = 1 or 3 - every dense layer of LW-OpenPose and VGG, the stride-1 layers of ResNet):
//   block  = 8 x 8 output pixels (two 32-pixel MFMA column tiles of 4 rows x 8 columns) x a group of output channels;
//            SPLIT: MW wavefronts, each 64 output channels x all 64 pixels (2 x 2 tiles x 2 accumulators = 128 accumulator registers);
//            fp32:  MW x 2 wavefronts, each ONE 32 x 32 tile - the smallest unit, because a layer of LW-OpenPose is only ~2500 such tiles
//                   for 1024 SIMDs and larger units leave SIMDs idle (64 x 64 units: 621 for 1024 SIMDs);
//   K loop = chunks of CK input channels.  A chunk's halo tile ((8 + KS - 1)^2 pixels x CK channels) is read from HBM as fp32 ONCE per
//            block and stored in LDS - as it is (fp32) or split into [CK hi | CK lo] halves (the same 4 bytes per element) - in pixel rows
//            of CK * 4 + 16 bytes with a pixel-row pitch of 8 mod 16 sixteen-byte units: every ds_read_b128 service group of
//            MI355X_MICROARCH.md's LDS table touches 16 distinct bank quads.  The tile then serves all KS * KS taps; the next chunk's
//            values are requested into registers before the chunk's MFMAs start.  TWO barriers per chunk, none per K-step.
//   A      = weights packed on the host in MFMA-fragment order [chunk][tap][step][32-row tile][fragment][lane][16 bytes] (SPLIT: a step
//            is 16 channels, fragments hi | lo of 8 halves; fp32: a step is 8 channels, lane (row, h) holds channels 4h .. 4h + 3 and
//            feeds element e to MFMA e, exactly like the B side - conv32_kernel's permutation): coalesced 1 KB loads straight from L2
//            into registers, two steps ahead, never through LDS.
#include "conv_fp32.hpp"

#include "conv32_epilogue.hpp"
#include "conv_device.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long tvd_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

template <bool SPLIT, int KS, int CK, int MW, int DWD = 0>
struct direct32_geom {
    static constexpr int HP = 8 + KS - 1;                          // halo tile is HP x HP pixels
    static constexpr int PBU = CK / 4 + 1;                         // 16-byte units per halo pixel: CK elements of 4 bytes + one unit of padding (odd)
    static constexpr int PB = PBU * 16;
    static constexpr int RP = ((HP * PBU + 7) / 16 * 16 + 8) * 16; // pixel-row pitch in bytes: >= HP pixels, = 8 mod 16 units
    static constexpr int LDS_BYTES = HP * RP;
    static constexpr int QPP = CK / 4;                             // float4 quads per pixel
    // DWD > 0 (a depthwise 3 x 3 of dilation DWD fused in front of the 1 x 1, KS = 1): the chunk's DEPTHWISE input - (8 + 2 DWD)^2 pixels x CK
    // fp32 channels in pixel rows of PB bytes - is what arrives from HBM, into a second LDS tile behind the first; the depthwise outputs are
    // computed from it into the first tile, which the MFMAs read as before.  The depthwise weights and biases of ALL chunks ([9][C] + [C]
    // floats) sit behind that, loaded once per block.
    static constexpr int XP = DWD ? 8 + 2 * DWD : HP;              // staged tile is XP x XP pixels
    static constexpr int XT_BYTES = DWD ? XP * XP * PB : 0;
    static constexpr int QUADS = XP * XP * QPP;                    // float4 quads staged per chunk
    static constexpr int TM = SPLIT ? 2 : 1, TN = SPLIT ? 2 : 1, NWN = 2 / TN; // MFMA tiles per wavefront; wavefronts side by side over the pixels
    static constexpr int SLABS = MW * NWN * (32 * (TM * 32 + 4) * 4); // the epilogue's transposition slabs (rows_geom<TM>::SLAB_BYTES per wavefront) lie over the dead tiles
    static constexpr int TILES_BYTES = LDS_BYTES + XT_BYTES;
    static constexpr int DWW_OFF = TILES_BYTES > SLABS ? TILES_BYTES : SLABS; // dynamic LDS: this + 40 bytes per depthwise channel (DWD)
    static constexpr int KQ = SPLIT ? CK / 16 : CK / 8;            // steps per tap: 32 bytes of a pixel row each
    static constexpr int SPC = KS * KS * KQ;                       // steps per chunk
    static constexpr int RING = SPC % 3 == 0 ? 3 : (SPLIT ? 2 : 4); // A-fragment ring: the step in use + one or two in flight
    static constexpr int AHEAD = RING - 1;                         // steps between an A fragment's request and its use
    // chunks in flight between HBM and LDS (registers): a 1 x 1 layer's chunk is only KQ steps of MFMAs - far shorter than an HBM round trip
    // under load - and Little's law asks for ~40 KB in flight per CU to stream at the rate the layer needs; a 3 x 3 chunk covers its successor
    static constexpr int DEPTH = KS != 1 ? 1 : DWD ? (SPLIT && MW == 8 ? 1 : 2) : (SPLIT && (MW < 4 || MW == 8) ? 2 : 3); // (fewer threads share a tile's staging when MW < 4: 16 quads per thread and chunk at MW = 1)
    static_assert(SPC % RING == 0 && SPC % 2 == 0, "ring / double buffer periods");
};

} // namespace

template <bool SPLIT, int KS, int CK, int MW, int DWD = 0>
__global__ __launch_bounds__(SPLIT ? 64 * MW : 128 * MW, (SPLIT && (MW < 4 || (DWD && MW == 4))) ? 1 : 2) void conv32_direct_kernel(const conv32_params p, int tiles_x, int tiles_y)
{
    static_assert(DWD == 0 || KS == 1, "a fused depthwise layer feeds a 1 x 1 convolution");
    using G = direct32_geom<SPLIT, KS, CK, MW, DWD>;
    constexpr int TM = G::TM, TN = G::TN, NWN = G::NWN, NT = 64 * MW * NWN, NACC = SPLIT ? 2 : 1;
    static_assert(G::SLABS == MW * NWN * rows_geom<TM>::SLAB_BYTES, "slab size");
    constexpr int NFA = SPLIT ? 2 * TM : TM; // A fragments (16 bytes per lane each) of one step
    constexpr int RP = G::RP, PB = G::PB, KQ = G::KQ, SPC = G::SPC, RING = G::RING, AHEAD = G::AHEAD, DEPTH = G::DEPTH;
    constexpr int NQ = (G::QUADS + NT - 1) / NT; // float4 per thread and chunk
    constexpr int XT_OFF = G::LDS_BYTES, XP = G::XP, DWW_OFF = G::DWW_OFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[]; // G::DWW_OFF bytes (+ 40 bytes per depthwise channel when DWD)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * 8, x0 = tx * 8;
    const int MT = p.Cout_pad / 32, mt0 = (blockIdx.y * MW + wm) * TM;
    const int nch = p.Cin / CK;
    int dbg_i = 0;
#define HP_STAMP()                                                     \
    if (p.dbg && blockIdx.x == 1 && blockIdx.y == 0 && tid == 0)       \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();

    // ---- staging geometry: quad q of a chunk = (halo pixel q / QPP, channels 4 (q % QPP) ..)
    long goff[NQ];
    int soff[NQ];
    bool qok[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * NT;
        const int hp = min(q, G::QUADS - 1) / G::QPP, c4 = q % G::QPP;
        const int hy = hp / XP, hx = hp - hy * XP;
        // (the tensor's zero halo covers the padding rows / columns below and right of the image too; pixels of a ragged last tile
        // beyond it are clamped to it and zeroed - they only feed output pixels that are never stored)
        const int pt = DWD ? DWD : p.pad_t, pl = DWD ? DWD : p.pad_l, reach = DWD ? 2 * DWD : KS - 1;
        const int y = y0 + hy - pt, x = x0 + hx - pl;
        const int ymax = p.H - 1 + (reach - pt), xmax = p.W - 1 + (reach - pl);
        qok[i] = y <= ymax && x <= xmax && q < G::QUADS;
        goff[i] = tvd_off(p.in, b, min(y, ymax), min(x, xmax)) + c4 * 4;
        soff[i] = DWD ? XT_OFF + hp * PB + c4 * 16 : hy * RP + hx * PB + (SPLIT ? c4 * 8 : c4 * 16);
    }
    f32x4 stage[DEPTH][NQ]; // stage[0] = the chunk about to be written to LDS, stage[d] = d chunks later
    auto gload = [&](int d, int c) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            stage[d][i] = *reinterpret_cast<const f32x4*>(p.in.p + goff[i] + min(c, nch - 1) * CK);
    };
    unsigned ovf = 0;
    auto to_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (i * NT + tid < G::QUADS) {
                f32x4 x = stage[0][i];
                if (!qok[i])
                    x = f32x4{ 0.f, 0.f, 0.f, 0.f };
                if constexpr (DWD != 0) {
                    *reinterpret_cast<f32x4*>(lds + soff[i]) = x; // the depthwise input, as it is
                } else if constexpr (SPLIT) {
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = (_Float16)x[e];
                        l[e] = (_Float16)((x[e] - (float)h[e]) * 2048.f);
                        ovf |= __builtin_fabsf(x[e]) > 65504.f;
                    }
                    *reinterpret_cast<half4v*>(lds + soff[i]) = half4v{ h[0], h[1], h[2], h[3] };
                    *reinterpret_cast<half4v*>(lds + soff[i] + CK * 2) = half4v{ l[0], l[1], l[2], l[3] };
                } else
                    *reinterpret_cast<f32x4*>(lds + soff[i]) = x;
            }
        }
    };

    // ---- fused depthwise 3 x 3 (DWD > 0): the chunk's 64 pixels x CK channels from the staged input tile into the MFMAs' tile.  Quad q =
    // (pixel q / QPP, channels 4 (q % QPP) ..): a thread's channel quad is the same for all its pixels; taps in (ky, kx) order, fmaf, bias
    // first - dwconv32_kernel's arithmetic, so fusing does not change a bit of the depthwise result
    auto dw_compute = [&](int c) {
        if constexpr (DWD != 0) {
            constexpr int OQ = 64 * G::QPP;
            const float* const dww = reinterpret_cast<const float*>(lds + DWW_OFF);
#pragma unroll 1
            for (int i = 0; i < (OQ + NT - 1) / NT; ++i) { // (one quad at a time: nine tile reads + nine weight reads live, not 4 x that)
                const int q = tid + i * NT;
                if (q < OQ) {
                    const int px = q / G::QPP, c4 = q % G::QPP, oy = px >> 3, ox = px & 7;
                    const int ch = c * CK + c4 * 4;
                    f32x4 a = *reinterpret_cast<const f32x4*>(dww + 9 * p.Cin + ch);
                    const unsigned char* const xt = lds + XT_OFF + (oy * XP + ox) * PB + c4 * 16;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(xt + (ky * DWD * XP + kx * DWD) * PB);
                            const f32x4 w = *reinterpret_cast<const f32x4*>(dww + (ky * 3 + kx) * p.Cin + ch);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                a[e] = __builtin_fmaf(x[e], w[e], a[e]);
                        }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        a[e] = a[e] > 0.f ? fminf(a[e], p.dw_hi) : a[e] * p.dw_slope;
                    unsigned char* const dst = lds + oy * RP + ox * PB;
                    if constexpr (SPLIT) {
                        _Float16 h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h[e] = (_Float16)a[e];
                            l[e] = (_Float16)((a[e] - (float)h[e]) * 2048.f);
                            ovf |= __builtin_fabsf(a[e]) > 65504.f;
                        }
                        *reinterpret_cast<half4v*>(dst + c4 * 8) = half4v{ h[0], h[1], h[2], h[3] };
                        *reinterpret_cast<half4v*>(dst + c4 * 8 + CK * 2) = half4v{ l[0], l[1], l[2], l[3] };
                    } else
                        *reinterpret_cast<f32x4*>(dst + c4 * 16) = a;
                }
            }
        }
    };
    if constexpr (DWD != 0) { // the depthwise weights and biases of every chunk: [9][C] + [C] floats, once per block (the first barrier covers them)
        float* const dww = reinterpret_cast<float*>(lds + DWW_OFF);
        for (int i = tid; i < 10 * p.Cin / 4; i += NT)
            *reinterpret_cast<f32x4*>(dww + i * 4) = *reinterpret_cast<const f32x4*>(p.dw_w + i * 4);
    }

    // ---- A fragments: steps run (chunk, tap, step of the tap) in packing order; a step's fragments of this wavefront are contiguous
    constexpr int FRAG = SPLIT ? 512 : 256;             // elements (halves / floats) of one 1 KB fragment
    using wt = typename std::conditional<SPLIT, _Float16, float>::type;
    const long step_stride = (long)MT * (SPLIT ? 2 : 1) * FRAG; // elements per step
    const wt* wp = reinterpret_cast<const wt*>(SPLIT ? (const void*)p.w_split : (const void*)p.w_frag) + (long)mt0 * (SPLIT ? 2 : 1) * FRAG + lane * (SPLIT ? 8 : 4);
    const int nsteps = nch * SPC;
    u32x4 fa[RING][NFA]; // ring: the step in use and two in flight
    auto aload = [&](int slot, int s) {
        const wt* q = wp + (long)min(s, nsteps - 1) * step_stride;
#pragma unroll
        for (int f = 0; f < NFA; ++f)
            fa[slot][f] = *reinterpret_cast<const u32x4*>(q + f * FRAG);
    };

    floatx16 acc[TM][TN][NACC]; // SPLIT: [hi-hi | cross]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][a][r] = 0.f;

    // B fragment base of n-tile j: pixel (4 j + (n >> 3), n & 7) of the tile = halo pixel (.. + ky, .. + kx) at tap (ky, kx)
    const int n = lane & 31, fk = lane >> 5;
    int bbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bbase[j] = (4 * (wn * TN + j) + (n >> 3)) * RP + (n & 7) * PB + fk * 16;

#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        gload(d, d);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        aload(a, a);
    int s = 0; // global step index
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        if (c && DWD == 0)
            lds_barrier(); // every wavefront is done reading the previous chunk's tile
        to_lds();
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                stage[d][i] = stage[d + 1][i];
        gload(DEPTH - 1, c + DEPTH); // (past the last chunk: a harmless re-read of it)
        lds_barrier();
        if constexpr (DWD != 0) { // (that barrier: the depthwise input tile is complete AND every wavefront has left the previous chunk's MFMAs)
            dw_compute(c);
            lds_barrier();
        }
        HP_STAMP();
        u32x4 fb[2][TN][NACC]; // [buffer][n tile][SPLIT: hi | lo]
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                fb[0][j][a] = *reinterpret_cast<const u32x4*>(lds + bbase[j] + a * CK * 2);
#pragma unroll
        for (int st = 0; st < SPC; ++st) { // st = tap * KQ + step of the tap (the packing order of the weights)
            const int cur = st & 1;
            // B fragments of the next step of this chunk (the last step re-reads its own: harmless, keeps the loop uniform)
            {
                const int nst = st + 1 < SPC ? st + 1 : st, ntap = nst / KQ, nks = nst % KQ;
                const int toff = (ntap / KS) * RP + (ntap % KS) * PB + nks * 32;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int a = 0; a < NACC; ++a)
                        fb[cur ^ 1][j][a] = *reinterpret_cast<const u32x4*>(lds + bbase[j] + toff + a * CK * 2);
            }
            const int slot = st % RING; // (SPC % RING == 0: the ring position is a compile-time function of st in every chunk)
            aload((st + AHEAD) % RING, s + AHEAD);
            if constexpr (SPLIT) {
                // hi-hi first (independent accumulators), then the two cross products of every tile
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[slot][2 * i]), __builtin_bit_cast(half8, fb[cur][j][0]), acc[i][j][0], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[slot][2 * i]), __builtin_bit_cast(half8, fb[cur][j][NACC - 1]), acc[i][j][NACC - 1], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[slot][2 * i + 1]), __builtin_bit_cast(half8, fb[cur][j][0]), acc[i][j][NACC - 1], 0, 0, 0);
                // pin the issue order of the step (hipcc otherwise sinks every load to just before its first use and the wavefront - alone
                // on its SIMD - eats the latency): MFMA, LDS read, ... (the next step's B), MFMA, L2 read, ... (A two steps ahead), 4 MFMAs
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
            } else {
                // 8 channels: lane (row / pixel, h) holds channels 4h .. 4h + 3, MFMA e multiplies channels {e, 4 + e} (A and B alike)
                const f32x4 a4 = __builtin_bit_cast(f32x4, fa[slot][0]), b4 = __builtin_bit_cast(f32x4, fb[cur][0][0]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[0][0][0], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0); // nothing crosses a step boundary: the reads above belong to LATER steps and must stay here
            ++s;
        }
        HP_STAMP();
    }
    if (SPLIT && ovf && p.ovf)
        atomicOr(p.ovf, 1u);

    // ---- epilogue
    if (!p.out_f32 && p.lane_epilogue <= 0) {
        // NHWC output only (every layer but the network's heads): row-major through a private LDS slab (conv32_epilogue.hpp)
        lds_barrier(); // every wavefront is done with the halo tile the slabs lie over
        float* const slab = reinterpret_cast<float*>(lds) + wave * (rows_geom<TM>::SLAB_BYTES / 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            floatx16 fin[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    fin[i][r] = SPLIT ? __builtin_fmaf(acc[i][j][NACC - 1][r], 1.f / 2048.f, acc[i][j][0][r]) : acc[i][j][0][r];
            const int ty0 = y0 + 4 * (wn * TN + j);
            conv32_store_rows<TM>(p, fin, slab, lane, mt0 * 32, [&](int r, bool& ok, long& ooff, long& roff) {
                const int oy = ty0 + (r >> 3), ox = x0 + (r & 7);
                ok = oy < p.OH && ox < p.OW;
                const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 1);
                ooff = tvd_off(p.out, b, oyc, oxc);
                roff = p.res.p ? tvd_off(p.res, b, oyc, oxc) : 0;
            });
        }
    } else {
    // the network's heads (fp32 NCHW for the parsers, runs along x): lane (n, fk) of a 32 x 32 tile holds rows (r & 3) + 8 (r >> 2) + 4 fk of
    // column n, four consecutive channels per r >> 2
    const bool out_vec = p.out.p && ((p.out.coff | p.out.cs) & 3) == 0;
    const bool res_vec = p.res.p && ((p.res.coff | p.res.cs) & 3) == 0;
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oy = y0 + 4 * (wn * TN + j) + (n >> 3), ox = x0 + (n & 7);
        const bool pix_ok = oy < p.OH && ox < p.OW;
        const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 1);
        const long ooff = p.out.p ? tvd_off(p.out, b, oyc, oxc) : 0;
        const long roff = p.res.p ? tvd_off(p.res, b, oyc, oxc) : 0;
        const int rem = oyc * p.OW + oxc;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
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
                        float x = SPLIT ? __builtin_fmaf(acc[i][j][NACC - 1][4 * q + e], 1.f / 2048.f, acc[i][j][0][4 * q + e]) : acc[i][j][0][4 * q + e];
                        x += bs[e];
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
    HP_STAMP();
#undef HP_STAMP
}

// Which layers the direct kernels take: square 1 x 1 / 3 x 3, stride 1, dilation 1, SAME padding, channel slice readable in whole chunks
// (32 channels at 3 x 3, 64 at 1 x 1).
bool conv32_direct_ok(const conv32_params& p)
{
    return p.KH == p.KW && ((p.KH == 1 && p.Cin % 64 == 0) || (p.KH == 3 && p.Cin % 32 == 0)) && p.stride == 1 && p.dil == 1 && p.Cout_pad % 64 == 0
        && p.OH == p.H && p.OW == p.W && p.pad_t == (p.KH - 1) / 2 && p.pad_l == (p.KW - 1) / 2;
}

// Wavefront groups per block.  All wavefronts of a block share one 8 x 8 pixel tile, so more of them amortise the tile's staging over more
// output channels - but a layer needs enough blocks for 256 CUs.  SPLIT: MW wavefronts of 64 channels; fp32: MW x 2 wavefronts, 32 MW channels.
// dwd = dilation of a depthwise 3 x 3 fused in front (0: none).  The fused forms exist for 2 / 4 (and, split at dilation 1, 8) groups only: a
// fused layer takes the largest of those that still leaves enough blocks, 2 at least; 0 = no fused form fits (odd group count).
static int direct_mw(const conv32_params& p, bool split, int dwd)
{
    const int groups = p.Cout_pad / (split ? 64 : 32);
    const long tiles = (long)p.B * ((p.OH + 7) / 8) * ((p.OW + 7) / 8);
    static const int force = getenv("HP_DIRECT_MW") ? atoi(getenv("HP_DIRECT_MW")) : 0;
    static const int mw_max = getenv("HP_DIRECT_MW_MAX") ? atoi(getenv("HP_DIRECT_MW_MAX")) : 8;
    for (int mw : { 8, 4, 2, 1 }) {
        if (mw == 8 && (!split || p.KH != 1 || dwd == 2))
            continue; // (8 wavefronts of 64 channels: the split 1 x 1 layers with 512 outputs read their input tile once; behind a depthwise layer of dilation 2: 68 spilled registers - not compiled)
        if (dwd && mw == 1)
            break;
        if ((force && mw != force) || mw > mw_max)
            continue;
        if (groups % mw == 0 && (force || mw == 1 || (dwd && mw == 2) || tiles * (groups / mw) >= (split ? 256 : 640)))
            return mw;
    }
    return dwd ? 0 : 1;
}
static int direct_mw(const conv32_params& p, bool split) { return direct_mw(p, split, p.dw_w ? p.dw_dil : 0); }

int conv32_direct_tile(const conv32_params& p, bool split) { return (split ? 33000000 : 34000000) + (p.dw_w ? 100000 * p.dw_dil : 0) + p.KH * 1000 + direct_mw(p, split); }

// Host side: the packed fp32 matrix [tap][Cout_pad][Cin] (conv32_params::w's layout) in the kernels' fragment order.
//   split: [chunk][tap][k16][32-row tile][hi | lo][lane][8 halves]      (2 * taps * cout_pad * cin halves)
//   fp32:  [chunk][tap][k8][32-row tile][lane][4 floats], lane (row, h) = channels 8 k8 + 4 h + {0..3}   (taps * cout_pad * cin floats)
void conv32_split_pack(const float* packed, int taps, int cout_pad, int cin, _Float16* out)
{
    const int ck = taps == 1 ? 64 : 32, kq = ck / 16, nch = cin / ck, MT = cout_pad / 32;
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

void conv32_frag_pack(const float* packed, int taps, int cout_pad, int cin, float* out)
{
    const int ck = taps == 1 ? 64 : 32, kq = ck / 8, nch = cin / ck, MT = cout_pad / 32;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < taps; ++t)
            for (int ks = 0; ks < kq; ++ks)
                for (int mt = 0; mt < MT; ++mt) {
                    float* dst = out + (((size_t)(c * taps + t) * kq + ks) * MT + mt) * 256;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int m = mt * 32 + (lane & 31), k = c * ck + ks * 8 + (lane >> 5) * 4 + e;
                            dst[lane * 4 + e] = packed[((size_t)t * cout_pad + m) * cin + k];
                        }
                }
}

// The depthwise-fused forms that are compiled: dilation 1 | 2; split: 2 / 4 / 8 wavefronts of 64 channels, fp32: 2 / 4 wavefront pairs of 32.
bool conv32_dw_fusable(const conv32_params& p, bool split, int dil)
{
    return conv32_direct_ok(p) && p.KH == 1 && (dil == 1 || dil == 2) && direct_mw(p, split, dil) != 0;
}

template <bool SPLIT, int KS, int CK, int MW, int DWD>
static hipError_t launch_direct_case(const conv32_params& p, dim3 grid, int tiles_x, int tiles_y, hipStream_t s)
{
    using G = direct32_geom<SPLIT, KS, CK, MW, DWD>;
    const size_t lds = (size_t)G::DWW_OFF + (DWD ? (size_t)40 * p.Cin : 0);
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    static size_t granted = 0; // (per instantiation: the largest dynamic LDS size the runtime has been told about)
    if (lds > 64 * 1024 && lds > granted) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv32_direct_kernel<SPLIT, KS, CK, MW, DWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        granted = lds;
    }
    HP_LAUNCH((conv32_direct_kernel<SPLIT, KS, CK, MW, DWD>), grid, dim3((SPLIT ? 64 : 128) * MW), lds, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

hipError_t launch_conv32_direct(const conv32_params& p, bool split, hipStream_t s)
{
    if (!conv32_direct_ok(p) || !(split ? (const void*)p.w_split : (const void*)p.w_frag) || p.npix <= 0)
        return hipErrorInvalidValue;
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 7) / 8, mw = direct_mw(p, split);
    const dim3 grid(tiles_x * tiles_y * p.B, p.Cout_pad / ((split ? 64 : 32) * mw));
    const int dwd = p.dw_w ? p.dw_dil : 0;
#define HP_DIRECT_CASE(SPLIT_, KS_, CK_, MW_, DWD_)                                                               \
    if (split == SPLIT_ && p.KH == KS_ && mw == MW_ && dwd == DWD_)                                               \
        return launch_direct_case<SPLIT_, KS_, CK_, MW_, DWD_>(p, grid, tiles_x, tiles_y, s);
    HP_DIRECT_CASE(true, 1, 64, 1, 0)
    HP_DIRECT_CASE(true, 1, 64, 2, 0)
    HP_DIRECT_CASE(true, 1, 64, 4, 0)
    HP_DIRECT_CASE(true, 1, 64, 8, 0)
    HP_DIRECT_CASE(true, 3, 32, 1, 0)
    HP_DIRECT_CASE(true, 3, 32, 2, 0)
    HP_DIRECT_CASE(true, 3, 32, 4, 0)
    HP_DIRECT_CASE(false, 1, 64, 1, 0)
    HP_DIRECT_CASE(false, 1, 64, 2, 0)
    HP_DIRECT_CASE(false, 1, 64, 4, 0)
    HP_DIRECT_CASE(false, 3, 32, 1, 0)
    HP_DIRECT_CASE(false, 3, 32, 2, 0)
    HP_DIRECT_CASE(false, 3, 32, 4, 0)
    HP_DIRECT_CASE(true, 1, 64, 2, 1)
    HP_DIRECT_CASE(true, 1, 64, 4, 1)
    HP_DIRECT_CASE(true, 1, 64, 8, 1)
    HP_DIRECT_CASE(true, 1, 64, 2, 2)
    HP_DIRECT_CASE(true, 1, 64, 4, 2)
    HP_DIRECT_CASE(false, 1, 64, 2, 1)
    HP_DIRECT_CASE(false, 1, 64, 4, 1)
    HP_DIRECT_CASE(false, 1, 64, 2, 2)
    HP_DIRECT_CASE(false, 1, 64, 4, 2)
#undef HP_DIRECT_CASE
    return hipErrorInvalidValue;
}

} // namespace hp
