// conv_chain.hip — a chain of 128-channel convolutions in ONE launch: [1x1 ->] 3x3 -> 3x3 [+ residual], intermediates in LDS.
//
// LW-OpenPose's head (hyperpose/Model/openpose/model/lw_openpose.py:106-191) is seventeen 3x3 128 -> 128 convolutions and six 1x1s
// on a 46 x 54 map.  At batch 8 each of them is 23 MFLOP per CU: as one launch per layer the matrix pipe idles through a halo
// prologue, an epilogue and a kernel boundary for every 7 k cycles of MFMAs (conv3x3_direct_kernel: 13 us per layer, 0.16-0.18 of
// the fp16 MFMA peak).  Here one block owns a 10 x 6 pixel output tile and ALL 128 channels and runs the whole chain on it:
//
//     S0 (refinement blocks, lw_openpose.py:176-191):  X0 = input tile + 2-pixel halo (14 x 10 px) -> 1x1 + relu -> T1 (14 x 10 px)
//        (otherwise T1 = the input tile + 2-pixel halo, straight from HBM)
//     S1:  T1 -> 3x3 + relu [+ external residual] -> T2 (12 x 8 px: the output tile + 1-pixel halo)
//     S2:  T2 -> 3x3 + relu [+ residual: external, or T1's interior = the 1x1's output] -> 10 x 6 px -> HBM
//
// The halo pixels of the intermediates are recomputed by the neighbouring blocks (x2 MFMAs in S1 incl. tile padding, x2.5 in the small S0) - MFMA
// time is what this network has to spare - and in exchange two of three launches, their prologues / epilogues and the HBM round
// trips of both intermediates (5 MB each way per layer) disappear.  Intermediate pixels outside the image are ZERO (the next
// convolution's padding), not convolution outputs.
//
// Four wavefronts, one per SIMD, wavefront w = output channels 32w .. 32w+31 over the FULL K of every stage: no split-K exchange;
// A fragments straight from L2 in MFMA-fragment order (conv_kernels.hpp, w_layout 1: one coalesced 1 KB load per fragment that only
// this wavefront needs), re-requested one tap ahead as they are consumed; B fragments from the swizzled LDS tiles, read one k16
// step ahead of their MFMAs.  Activations are stored fp16 exactly where the per-layer schedule stores them (every intermediate is
// rounded once), so the chain's results differ from the unfused schedule only by fp32 summation order.
#include "conv_device.hpp"

#include <cstdlib>

namespace hp {

namespace {

constexpr int CH = 128;              // channels of every tensor in the chain
constexpr int PXB = CH * 2;          // bytes per pixel in LDS
constexpr int KQ = CH / 16;          // k16 steps per tap
// Output tile <= 64 pixels: with it a block needs <= 72 KB of LDS and <= 256 registers, i.e. HALF a CU - two blocks, or one block and a kernel
// of the other hardware queue, share a CU.  The first form (8 x 12 pixels, 85 / 134 KB, 270 registers: 8 % fewer MFMAs per pixel)
// was 14 % faster alone (conv stack 0.59 -> 0.53 ms) and gained NOTHING end to end: a whole-CU kernel serialises the two queues
// the four pipes run on ("a CU is two slots").
// 10 x 6 (round 3; 8 x 8 before): the middle stage - the bulk of the MFMAs - covers 12 x 8 = 96 pixels = exactly three column tiles
// (10 x 10 = 100 needed four: 22 % of that stage multiplied padding), and 46 x 54 is 5 x 9 tiles with 4 spare rows instead of 6 x 7 with
// 2 rows + 2 columns: 576 k instead of 634 k MFMAs per batch of 8.  71.7 KB of LDS with the 1x1 in front.
constexpr int TH = 10, TW = 6;       // output tile
constexpr int H2 = TH + 4, W2 = TW + 4, N2 = H2 * W2; // S0 / T1 region (halo 2): 14 x 10 = 140 px -> 5 column tiles (20 pad lanes)
constexpr int H1 = TH + 2, W1 = TW + 2, N1 = H1 * W1; // T2 region (halo 1): 12 x 8 = 96 px -> 3 column tiles
constexpr int N0 = TH * TW;                            // 60 px -> 2 column tiles (4 pad lanes)
constexpr int NT2 = (N2 + 31) / 32, NT1 = (N1 + 31) / 32, NT0 = (N0 + 31) / 32;

// swizzle key of a pixel of a tile that is CONSUMED in pixel order of width CW: 16 consecutive consumer pixels, shifted by any tap,
// read 16 distinct 16-byte slots of the 256-byte bank row (a pixel is exactly one bank row: 128 channels x 2 B)
__device__ __forceinline__ int t1_key(int hy, int hx) { return (hy * W1 + hx) & 15; } // T1 [H2][W2], consumed by S1 over W1
__device__ __forceinline__ int t2_key(int hy, int hx) { return (hy * TW + hx) & 15; } // T2 [H1][W1], consumed by S2 over TW

// One stage's MFMAs for this wavefront: TAPS taps x 8 k16 steps x NT column tiles.
//   src        LDS tile the B fragments come from, pixel stride 256 B, row width WIN pixels
//   pix0[j]    byte offset of column tile j's pixel (this lane's) at tap (0, 0)
//   nkey[j]    that pixel's index in consumer order (its swizzle key at tap (ky, kx) is (nkey + ky * KW + kx) & 15)
//   a[]        A fragments of the first tap (already requested); wcur = this stage's weights (this wave's rows), wnext = the NEXT
//              stage's (its first tap is requested while this stage's last tap is consumed), or nullptr
template <int NT, int TAPS, int WIN, int KW, int D>
__device__ __forceinline__ void chain_stage(floatx16 (&acc)[NT], u32x4 (&a)[KQ], const __half* wcur, long tap_stride, const __half* wnext,
    const unsigned char* src, const int (&pix0)[NT], const int (&nkey)[NT], int fk)
{
    // D = how many k16 steps ahead of their MFMAs the B fragments are read (ring of fragment sets).  Measured on the 8 x 12 form: D = 1, 2,
    // 3 run within 1 % of each other, and a timing build without ANY memory operation in the loop takes 83 % of the loop's time (14.9 k
    // of 17.9 k ticks for S1): the loop is at the matrix pipe's rate at the clock the chip sustains; the reads are not what it waits
    // for.  D = 1 (two sets) keeps the block under 256 registers.
    static_assert(D >= 1 && D <= 3 && KQ == 8, "prefetch ring");
    constexpr int KS = TAPS == 9 ? 3 : 1;
    auto tap_addr = [&](int tap, int (&ad)[NT]) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const int toff = (ky * WIN + kx) * PXB, tkey = ky * KW + kx; // uniform
#pragma unroll
        for (int j = 0; j < NT; ++j)
            ad[j] = pix0[j] + toff + ((((nkey[j] + tkey) & 15) ^ fk) << 4);
    };
    half8 fb[4][NT];
    int ad[NT];
    tap_addr(0, ad);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[d][j] = *reinterpret_cast<const half8*>(src + (ad[j] ^ (d << 5)));
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        int adn[NT];
        tap_addr(min(tap + 1, TAPS - 1), adn);
        const bool last = tap + 1 == TAPS; // uniform
        // (unconditional: a conditional prefetch makes hipcc drain the load queue at the join; the last stage re-requests its own tap 0)
        const __half* wn = last ? (wnext ? wnext : wcur) : wcur + (long)(tap + 1) * tap_stride;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            // the B fragments of k16 step ks + D (of the next tap past the end of this one) are read while this step multiplies
            if (ks + D < KQ) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[(ks + D) & 3][j] = *reinterpret_cast<const half8*>(src + (ad[j] ^ ((ks + D) << 5)));
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[(ks + D) & 3][j] = *reinterpret_cast<const half8*>(src + (adn[j] ^ ((ks + D - KQ) << 5)));
            }
            half8 fa;
            __builtin_memcpy(&fa, &a[ks], 16);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[ks & 3][j], acc[j], 0, 0, 0);
            a[ks] = *reinterpret_cast<const u32x4*>(wn + (size_t)ks * 512);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
            ad[j] = adn[j];
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(floatx16 (&acc)[NT])
{
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[j][r] = 0.f;
}

} // namespace

// S0: the chain starts with a 1x1; RES: 0 none, 1 external tensor added to S1's output, 2 external tensor added to S2's output,
// 3 the 1x1's output (T1) added to S2's output
template <bool S0, int RES, int D = 1>
__global__ __launch_bounds__(256, 2) void conv_chain_kernel(const chain_params p, int tiles_x, int tiles_y)
{
    // LDS: [X0 | T1] with T2 over X0 (dead once S0 is done), or [T2 | T1] without S0: 72 / 61 KB
    constexpr int T1_BYTES = N2 * PXB, T2_BYTES = N1 * PXB, X0_BYTES = S0 ? N2 * PXB : T2_BYTES;
    static_assert(T2_BYTES <= X0_BYTES && X0_BYTES + T1_BYTES <= 80 * 1024, "half a CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[X0_BYTES + T1_BYTES];
    unsigned char* const s_x0 = lds;
    unsigned char* const s_t2 = lds;
    unsigned char* const s_t1 = lds + X0_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = p.c2.OH, W = p.c2.OW; // every tensor of the chain has this size
    int dbg_i = 0; // HP_CHAIN_DBG: s_memtime stamps of block 0 / thread 0 (engine.cpp prints the deltas)
#define HP_CSTAMP()                                                                                               \
    if (p.c2.dbg && blockIdx.x == 0 && tid == 0)                                                                  \
        p.c2.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_CSTAMP();

    // this wavefront's rows of the three weight matrices (fragment order: [tap][32-row tile][k16][lane][8])
    const long tap_stride = (long)(CH / 32) * KQ * 512;
    const __half* const w0 = S0 ? p.c0.w + ((size_t)(wave * KQ) * 64 + lane) * 8 : nullptr;
    const __half* const w1 = p.c1.w + ((size_t)(wave * KQ) * 64 + lane) * 8;
    const __half* const w2 = p.c2.w + ((size_t)(wave * KQ) * 64 + lane) * 8;
    u32x4 a[KQ];
    {
        const __half* wf = S0 ? w0 : w1;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks)
            a[ks] = *reinterpret_cast<const u32x4*>(wf + (size_t)ks * 512);
    }

    // An external residual (RES 1: added to S1's output, RES 2: to S2's) is requested right BEHIND the halo requests and waits in registers:
    // requests complete in order, so one issued next to a stage's MFMA loop is older than the loop's weight-fragment ring and the loop's
    // first s_waitcnt vmcnt exposes its whole latency (HP_CHAIN_DBG: 2.6 k of a block's 30 k ticks for RES 1, 1.7 k for RES 2), and one
    // issued AHEAD of the halo is waited for with it (the tensor was written several launches ago: it comes from HBM, the halo from L2)
    half4 rs1[RES == 1 ? NT1 : 1][4], rs2[RES == 2 ? NT0 : 1][4];
    // ---- the input tile + 2-pixel halo: all loads first (one round trip), then the LDS stores; pixels outside the image are zero
    {
        const tview& in = S0 ? p.c0.in : p.c1.in;
        unsigned char* const dst = S0 ? s_x0 : s_t1;
        constexpr int NIT = (N2 * 16 + 255) / 256;
        u32x4 hv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + it * 256, N2 * 16 - 1), px = i >> 4, c = i & 15;
            const int hy = px / W2, hx = px - hy * W2;
            const int y = y0 - 2 + hy, x = x0 - 2 + hx;
            const bool ok = y >= 0 && y < H && x >= 0 && x < W;
            const u32x4 v = *reinterpret_cast<const u32x4*>(in.p + tv_off(in, b, min(max(y, 0), H - 1), min(max(x, 0), W - 1)) + c * 8);
            hv[it] = v & (ok ? 0xffffffffu : 0u);
        }
        if (RES == 1) {
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
                const int nc = min(j * 32 + fr, N1 - 1), br = nc / W1, bc = nc - br * W1;
                const int y = y0 - 1 + br, x = x0 - 1 + bc;
                const __half* rp = p.c1.res.p + tv_off(p.c1.res, b, min(max(y, 0), H - 1), min(max(x, 0), W - 1)) + wave * 32 + 4 * fk;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rs1[j][g] = *reinterpret_cast<const half4*>(rp + 8 * g);
            }
        }
        if (RES == 2) {
#pragma unroll
            for (int j = 0; j < NT0; ++j) {
                const int n = min(j * 32 + fr, N0 - 1), br = n / TW, bc = n - br * TW;
                const __half* rp = p.c2.res.p + tv_off(p.c2.res, b, min(y0 + br, H - 1), min(x0 + bc, W - 1)) + wave * 32 + 4 * fk;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rs2[j][g] = *reinterpret_cast<const half4*>(rp + 8 * g);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256, px = i >> 4, c = i & 15;
            const int hy = px / W2, hx = px - hy * W2;
            const int key = S0 ? (px & 15) : t1_key(hy, hx); // X0 is consumed in its own pixel order (S0 is a 1x1)
            if (i < N2 * 16)
                *reinterpret_cast<u32x4*>(dst + px * PXB + ((c ^ key) << 4)) = hv[it];
        }
    }
    HP_CSTAMP();
    lds_barrier();
    HP_CSTAMP();

    // epilogue constants of this lane: its 16 accumulator rows are channels 32 wave + 8 g + 4 fk + {0..3}, g = 0..3
    auto load_bias = [&](const float* bias, float (&bs)[16]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(bias + wave * 32 + 8 * g + 4 * fk);
            bs[4 * g] = v.x, bs[4 * g + 1] = v.y, bs[4 * g + 2] = v.z, bs[4 * g + 3] = v.w;
        }
    };

    // ---- S0: 1x1 on the 14 x 10 region -> T1
    if (S0) {
        floatx16 acc[NT2];
        zero_acc(acc);
        int pix0[NT2], nkey[NT2];
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int n = min(j * 32 + fr, N2 - 1);
            pix0[j] = n * PXB, nkey[j] = n;
        }
        chain_stage<NT2, 1, W2, W2, D>(acc, a, w0, 0, w1, s_x0, pix0, nkey, fk);
        HP_CSTAMP();
        float bs[16];
        load_bias(p.c0.bias, bs);
        const float hi = p.c0.act_hi;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int n = j * 32 + fr, hy = n / W2, hx = n - hy * W2;
            if (n >= N2)
                continue; // pad lanes of the last column tile
            const int y = y0 - 2 + hy, x = x0 - 2 + hx;
            const bool ok = y >= 0 && y < H && x >= 0 && x < W;
            unsigned char* const row = s_t1 + n * PXB + fk * 8;
            const int key = t1_key(hy, hx);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h[r] = (_Float16)(ok ? __builtin_amdgcn_fmed3f(acc[j][4 * g + r] + bs[4 * g + r], 0.f, hi) : 0.f);
                *reinterpret_cast<half4*>(row + (((wave * 4 + g) ^ key) << 4)) = h;
            }
        }
        lds_barrier();
        HP_CSTAMP();
    }

    // ---- S1: 3x3 on the 12 x 8 region -> T2
    {
        floatx16 acc[NT1];
        zero_acc(acc);
        int pix0[NT1], nkey[NT1];
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            const int n = min(j * 32 + fr, N1 - 1), br = n / W1, bc = n - br * W1;
            pix0[j] = (br * W2 + bc) * PXB, nkey[j] = n;
        }
        chain_stage<NT1, 9, W2, W1, D>(acc, a, w1, tap_stride, w2, s_t1, pix0, nkey, fk);
        HP_CSTAMP();
        float bs[16];
        load_bias(p.c1.bias, bs);
        const float hi = p.c1.act_hi;
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            const int n = j * 32 + fr;
            const int nc = min(n, N1 - 1), br = nc / W1, bc = nc - br * W1;
            const int y = y0 - 1 + br, x = x0 - 1 + bc;
            const bool ok = y >= 0 && y < H && x >= 0 && x < W;
            if (n < N1) {
                unsigned char* const row = s_t2 + n * PXB + fk * 8;
                const int key = t2_key(br, bc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = __builtin_amdgcn_fmed3f(acc[j][4 * g + r] + bs[4 * g + r], 0.f, hi);
                        if (RES == 1)
                            v += (float)rs1[j][g][r];
                        h[r] = (_Float16)(ok ? v : 0.f);
                    }
                    *reinterpret_cast<half4*>(row + (((wave * 4 + g) ^ key) << 4)) = h;
                }
            }
        }
        lds_barrier();
        HP_CSTAMP();
    }

    // ---- S2: 3x3 on the 10 x 6 tile; the result goes (fp16) into T1's interior, in place of the residual it may have read there
    {
        floatx16 acc[NT0];
        zero_acc(acc);
        int pix0[NT0], nkey[NT0];
#pragma unroll
        for (int j = 0; j < NT0; ++j) {
            const int n = min(j * 32 + fr, N0 - 1), br = n / TW, bc = n - br * TW; // (pad lanes of the last column tile redo its last pixel)
            pix0[j] = (br * W1 + bc) * PXB, nkey[j] = n;
        }
        chain_stage<NT0, 9, W1, TW, D>(acc, a, w2, tap_stride, nullptr, s_t2, pix0, nkey, fk);
        HP_CSTAMP();
        float bs[16];
        load_bias(p.c2.bias, bs);
        const float hi = p.c2.act_hi;
#pragma unroll
        for (int j = 0; j < NT0; ++j) {
            const int n = j * 32 + fr, br = n / TW, bc = n - br * TW;
            if (n >= N0)
                continue; // pad lanes
            unsigned char* const row = s_t1 + ((br + 2) * W2 + bc + 2) * PXB + fk * 8;
            const int key = t1_key(br + 2, bc + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* const at = row + (((wave * 4 + g) ^ key) << 4);
                half4 h;
                if (RES == 3)
                    h = *reinterpret_cast<const half4*>(at);
                else if (RES == 2)
                    h = rs2[j][g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = __builtin_amdgcn_fmed3f(acc[j][4 * g + r] + bs[4 * g + r], 0.f, hi);
                    if (RES >= 2)
                        v += (float)h[r];
                    h[r] = (_Float16)v;
                }
                *reinterpret_cast<half4*>(at) = h;
            }
        }
        lds_barrier();
        HP_CSTAMP();
    }

    // ---- the finished tile: 16 lanes per pixel, 256 contiguous bytes of HBM each
    {
        constexpr int NIT = (N0 * 16 + 255) / 256;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256, px = min(i >> 4, N0 - 1), c = i & 15;
            const int br = px / TW, bc = px - br * TW;
            const int y = y0 + br, x = x0 + bc;
            const u32x4 v = *reinterpret_cast<const u32x4*>(s_t1 + ((br + 2) * W2 + bc + 2) * PXB + ((c ^ t1_key(br + 2, bc + 2)) << 4));
            if (i < N0 * 16 && y < H && x < W)
                *reinterpret_cast<u32x4*>(p.c2.out.p + tv_off(p.c2.out, b, y, x) + c * 8) = v;
        }
    }
    HP_CSTAMP();
#undef HP_CSTAMP
}

// what the chain kernel takes: 128 -> 128 channels everywhere, fragment-ordered weights, relu / relu6 (one clamp), the residual
// (if any) added after the activation, fp16 NHWC in and out with whole 16-byte channel groups
static bool chain_conv_ok(const conv_params& c, int k)
{
    return c.KH == k && c.KW == k && c.stride == 1 && c.dil == 1 && c.pad_t == k / 2 && c.pad_l == k / 2 && c.Cin == CH && c.Cout == CH
        && c.Cout_pad == CH && c.w_layout == 1 && c.OH == c.H && c.OW == c.W && !c.alpha && c.act_slope == 0.f && !c.out_f32
        && c.in.coff % 8 == 0 && c.in.cs % 8 == 0 && c.res_before_act == 0;
}

int conv_chain_variant(const chain_params& p)
{
    if (!chain_conv_ok(p.c1, 3) || !chain_conv_ok(p.c2, 3) || (p.has_c0 && !chain_conv_ok(p.c0, 1)))
        return 0;
    if (p.c1.H != p.c2.H || p.c1.W != p.c2.W || (p.has_c0 && (p.c0.H != p.c1.H || p.c0.W != p.c1.W)))
        return 0;
    if (!p.c2.out.p || p.c2.out.coff % 8 || p.c2.out.cs % 8)
        return 0;
    const tview* res = p.res_mode == 1 ? &p.c1.res : p.res_mode == 2 ? &p.c2.res : nullptr;
    if (res && (!res->p || res->coff % 4 || res->cs % 4))
        return 0;
    if (p.has_c0)
        return p.res_mode == 3 ? 13 : p.res_mode == 0 ? 10 : 0;
    return p.res_mode >= 0 && p.res_mode <= 2 ? 1 + p.res_mode : 0;
}

hipError_t launch_conv_chain(const chain_params& p, hipStream_t s)
{
    const int v = conv_chain_variant(p);
    if (!v)
        return hipErrorInvalidValue;
    const int tiles_x = (p.c2.OW + TW - 1) / TW, tiles_y = (p.c2.OH + TH - 1) / TH;
    const dim3 grid(tiles_x * tiles_y * p.c2.B);
    switch (v) {
    case 1: HP_LAUNCH((conv_chain_kernel<false, 0>), grid, dim3(256), 0, s, p, tiles_x, tiles_y); break;
    case 2: HP_LAUNCH((conv_chain_kernel<false, 1>), grid, dim3(256), 0, s, p, tiles_x, tiles_y); break;
    case 3: HP_LAUNCH((conv_chain_kernel<false, 2>), grid, dim3(256), 0, s, p, tiles_x, tiles_y); break;
    case 10: HP_LAUNCH((conv_chain_kernel<true, 0>), grid, dim3(256), 0, s, p, tiles_x, tiles_y); break;
    default: HP_LAUNCH((conv_chain_kernel<true, 3>), grid, dim3(256), 0, s, p, tiles_x, tiles_y); break;
    }
    return hipGetLastError();
}

} // namespace hp
