// conv32_head.hip — the two-layer heads of the fp32 engine (HP_DTYPE_F32, data_type::kFLOAT) in one launch (interface: conv_fp32.hpp):
//        1 x 1  K1 = 128 -> HID (a multiple of 128; ReLU family)  ->  1 x 1  HID -> C2 <= 64 (any fused activation)
// LW-OpenPose ends its initial and every refinement stage with two such pairs (128 -> 512 -> 19 heat-maps | 38 PAFs:
// hyperpose/Model/openpose/model/lw_openpose.py:106-200).  As two launches the 512-channel hidden tensor (40 MB at 8 x 46 x 54) is written
// and read back for 0.4 GFLOP of second-layer work: 35 + 27 us alone, 27 + 17.5 with a second stream, per pair.  Here it never leaves the
// registers:
//   block   = 32 pixels (one column tile of v_mfma_f32_32x32x2_f32), four wavefronts; the pixels' 128 input channels go through LDS once
//             and then sit in every wavefront's registers as B fragments (16 x 4 registers);
//   hidden  = wavefront w computes hidden 32-row tiles w, w + 4, ..: 64 MFMAs over K1 (A = W1 in conv32_frag_pack's fragment order, 1 KB
//             loads from L2), bias + activation on the accumulator tile;
//   second  = the accumulator layout of that tile IS a B operand: lane (pixel n, fk) holds hidden rows (r & 3) + 8 (r >> 2) + 4 fk in register
//             r, and one MFMA step multiplies two K indices supplied by fk = 0 | 1 - so register r feeds the step over hidden channels
//             {h_r, h_r + 4} directly, with W2 packed to match (conv32_head_pack): 16 MFMAs per 32 outputs, no LDS round trip;
//   output  = the four wavefronts' partial sums meet in LDS (summed in wavefront order: deterministic), then the lane = pixel epilogue
//             (NHWC slice of any alignment + the fp32 NCHW network output the parsers read).
#include "conv_fp32.hpp"

#include "conv_device.hpp"

#include <cstdlib>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HK1 = 128;            // input channels
constexpr int HXP = (HK1 + 4) * 4;  // LDS row (one pixel) pitch in bytes
constexpr int HX_BYTES = 32 * HXP;  // 16896

__device__ __forceinline__ long tvh_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

} // namespace

// q = the SECOND layer's parameters (bias, activation, out, out_f32, Cout = C2, Cout_pad = 32 TM2, OH / OW / npix) with q.in = the FIRST layer's input;
// h = the first layer's
// (the body, with its LDS handed in: xs = HX_BYTES, red_ = 4 x TM2 x 4 x 64 x 4 floats; blk = the block's 32-pixel tile)
template <int TM2>
__device__ __forceinline__ void conv32_head_body(const conv32_params& q, const head32_hidden& h, unsigned char* const xs, float* const red_, const int blk)
{
    float (*const red)[TM2][4][64][4] = reinterpret_cast<float (*)[TM2][4][64][4]>(red_); // [wavefront][output tile][register quad][lane][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blk * 32;
    const int OHW = q.OH * q.OW;

    // ---- the 32 pixels' input channels -> LDS: thread = (pixel, quads q0, q0 + 8, q0 + 16, q0 + 24)
    {
        const int px = tid >> 3, q0 = tid & 7;
        const int n = min(n0 + px, q.npix - 1);
        const int b = n / OHW, rem = n - b * OHW;
        const int oy = rem / q.OW, ox = rem - oy * q.OW;
        const float* const src = q.in.p + tvh_off(q.in, b, oy, ox);
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = *reinterpret_cast<const f32x4*>(src + (q0 + 8 * i) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(xs + px * HXP + (q0 + 8 * i) * 16) = v[i];
    }
    lds_barrier();
    const int n = lane & 31, fk = lane >> 5;
    f32x4 xb[HK1 / 8]; // B fragments of the first layer: lane (pixel, fk) holds channels 8 k + 4 fk .. + 3 of step k
#pragma unroll
    for (int k = 0; k < HK1 / 8; ++k)
        xb[k] = *reinterpret_cast<const f32x4*>(xs + n * HXP + (k * 8 + fk * 4) * 4);

    floatx16 acc2[TM2];
#pragma unroll
    for (int m = 0; m < TM2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc2[m][r] = 0.f;

    const int MT1 = h.HID / 32, nt = MT1 / 4; // hidden tiles of this wavefront: wave, wave + 4, ..
    const float* const w1 = h.w1_frag + lane * 4;
    const float* const w2 = h.w2_frag + lane * 4;
    // first-layer A fragments: steps run (hidden tile of this wavefront, 8-channel step) flat; a ring of four, three steps ahead - across
    // the tile boundaries too, so that a tile's first MFMAs do not wait for L2
    constexpr int KS = HK1 / 8, RING = 4, AHEAD = 3;
    const int nsteps = nt * KS;
    f32x4 fa[RING];
    auto aload = [&](int slot, int st) {
        const int sc = min(st, nsteps - 1), ht = wave + 4 * (sc / KS), k = sc % KS;
        fa[slot] = *reinterpret_cast<const f32x4*>(w1 + (long)(k * MT1 + ht) * 256);
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        aload(a, a);
#pragma unroll 1
    for (int it = 0; it < nt; ++it) {
        const int ht = wave + 4 * it;
        floatx16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc1[r] = 0.f;
        f32x4 a2[TM2][4]; // the second layer's A fragments of this hidden tile: requested now, used after the first layer's 64 MFMAs
#pragma unroll
        for (int m = 0; m < TM2; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                a2[m][qd] = *reinterpret_cast<const f32x4*>(w2 + (long)(((ht * TM2 + m) * 4 + qd) * 256));
        f32x4 bs[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            bs[qd] = *reinterpret_cast<const f32x4*>(h.bias1 + ht * 32 + 8 * qd + 4 * fk);
#pragma unroll
        for (int k = 0; k < KS; ++k) { // (KS % RING == 0: the ring position is a compile-time function of k)
            aload((k + AHEAD) % RING, it * KS + k + AHEAD);
            const f32x4 a = fa[k % RING];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xb[k][e], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = acc1[r] + bs[r >> 2][r & 3];
            acc1[r] = x > 0.f ? fminf(x, h.hi1) : x * h.slope1;
        }
#pragma unroll
        for (int m = 0; m < TM2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m][r >> 2][r & 3], acc1[r], acc2[m], 0, 0, 0);
    }

    // ---- the four partial sums -> LDS; wavefront m adds them in wavefront order and stores output tile m
#pragma unroll
    for (int m = 0; m < TM2; ++m)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<f32x4*>(&red[wave][m][qd][lane][0]) = f32x4{ acc2[m][4 * qd], acc2[m][4 * qd + 1], acc2[m][4 * qd + 2], acc2[m][4 * qd + 3] };
    lds_barrier();
    if (wave >= TM2)
        return;
    const int np = n0 + n;
    const bool pix_ok = np < q.npix;
    const int nc = min(np, q.npix - 1);
    const int b = nc / OHW, rem = nc - b * OHW;
    const int oy = rem / q.OW, ox = rem - oy * q.OW;
    const long ooff = q.out.p ? tvh_off(q.out, b, oy, ox) : 0;
    const bool out_vec = q.out.p && ((q.out.coff | q.out.cs) & 3) == 0;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(&red[0][wave][qd][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w)
            sum += *reinterpret_cast<const f32x4*>(&red[w][wave][qd][lane][0]);
        const int m = wave * 32 + 8 * qd + 4 * fk;
        if (pix_ok && m < q.Cout) {
            const bool full = m + 3 < q.Cout;
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(q.bias + m); // (m + 3 < the padded bias length, a multiple of 64)
            f32x4 sl = { q.act_slope, q.act_slope, q.act_slope, q.act_slope };
            if (q.alpha)
                sl = *reinterpret_cast<const f32x4*>(q.alpha + m);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = sum[e] + b2[e];
                v[e] = x > 0.f ? fminf(x, q.act_hi) : x * sl[e];
            }
            if (q.out.p) {
                if (full && out_vec)
                    *reinterpret_cast<f32x4*>(q.out.p + ooff + m) = f32x4{ v[0], v[1], v[2], v[3] };
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < q.Cout)
                            q.out.p[ooff + m + e] = v[e];
                }
            }
            if (q.out_f32) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m + e < q.Cout)
                        q.out_f32[((long)b * q.Cout + m + e) * OHW + rem] = v[e];
            }
        }
    }
}

template <int TM2>
__global__ __launch_bounds__(256, 2) void conv32_head_kernel(const conv32_params q, const head32_hidden h)
{
    __shared__ __attribute__((aligned(16))) unsigned char xs[HX_BYTES];
    __shared__ __attribute__((aligned(16))) float red[4 * TM2 * 4 * 64 * 4];
    conv32_head_body<TM2>(q, h, xs, red, blockIdx.x);
}

// Two heads that read the SAME tensor (LW-OpenPose's heat-map and PAF heads of a stage) in ONE launch (round 6): alone each is 621 blocks of 32
// pixels on 512 slots at 8 x 46 x 54 - 45 + 48 us one after the other, 31 + 33 side by side on two streams; a captured graph cannot run two branches
// side by side, one grid can.  Block ids b and b + 8 - the same XCD, back to back - are the two heads of one pixel tile: the tile's
// 16 KB of input come from HBM once.
template <int TMA, int TMB>
__global__ __launch_bounds__(256, 2) void conv32_head_pair_kernel(const conv32_params qa, const head32_hidden ha, const conv32_params qb, const head32_hidden hb)
{
    __shared__ __attribute__((aligned(16))) unsigned char xs[HX_BYTES];
    __shared__ __attribute__((aligned(16))) float red[4 * (TMA > TMB ? TMA : TMB) * 4 * 64 * 4];
    const int blk = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    if (blk * 32 >= qa.npix)
        return; // (the grid is padded to whole groups of eight tiles)
    if ((blockIdx.x >> 3) & 1)
        conv32_head_body<TMB>(qb, hb, xs, red, blk);
    else
        conv32_head_body<TMA>(qa, ha, xs, red, blk);
}

// The pairs this kernel takes: 1 x 1 / stride 1 / no padding on both layers, 128 input channels (a 4-aligned slice), a hidden width that is a
// multiple of 128 (32 rows x four wavefronts), at most 64 outputs, no residual on either layer
bool conv32_head_ok(int k1, int hid, int c2) { return k1 == HK1 && hid >= 128 && hid % 128 == 0 && c2 >= 1 && c2 <= 64; }

int conv32_head_tile(int hid, int c2) { return 37000000 + hid * 100 + c2; }

// W2 = [c2_pad = 32 tm2][hid] row-major (zero rows in the padding) -> [hidden tile][output tile][register quad][lane][4]: lane (m, fk), register
// r = 4 quad + e  <->  W2[32 output tile + m][32 hidden tile + (r & 3) + 8 (r >> 2) + 4 fk]  (hid * 32 * tm2 floats)
void conv32_head_pack(const float* w2, int tm2, int hid, float* out)
{
    for (int ht = 0; ht < hid / 32; ++ht)
        for (int m2 = 0; m2 < tm2; ++m2)
            for (int qd = 0; qd < 4; ++qd)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e)
                        out[((((size_t)ht * tm2 + m2) * 4 + qd) * 64 + lane) * 4 + e] =
                            w2[(size_t)(m2 * 32 + (lane & 31)) * hid + ht * 32 + e + 8 * qd + 4 * (lane >> 5)];
}

static bool head_args_ok(const conv32_params& q, const head32_hidden& h)
{
    return conv32_head_ok(HK1, h.HID, q.Cout) && h.w1_frag && h.w2_frag && h.bias1 && q.npix > 0 && (q.out.p || q.out_f32);
}

// whether two head launches may share one grid: the same input view and map
bool conv32_head_pair_ok(const conv32_params& a, const conv32_params& b)
{
    const bool off = getenv("HP_HEAD_PAIR") && atoi(getenv("HP_HEAD_PAIR")) == 0; // A/B switch (read per launch / capture: the tests compare both in one process)
    return !off && a.in.p == b.in.p && a.in.cs == b.in.cs && a.in.coff == b.in.coff && a.in.wp == b.in.wp && a.in.img == b.in.img && a.OH == b.OH && a.OW == b.OW
        && a.npix == b.npix;
}

hipError_t launch_conv32_head_pair(const conv32_params& qa, const head32_hidden& ha, const conv32_params& qb, const head32_hidden& hb, hipStream_t s)
{
    if (!head_args_ok(qa, ha) || !head_args_ok(qb, hb) || !conv32_head_pair_ok(qa, qb))
        return hipErrorInvalidValue;
    const dim3 grid(((qa.npix + 31) / 32 + 7) / 8 * 8 * 2);
    const bool a1 = qa.Cout <= 32, b1 = qb.Cout <= 32;
    if (a1 && b1)
        HP_LAUNCH((conv32_head_pair_kernel<1, 1>), grid, dim3(256), 0, s, qa, ha, qb, hb);
    else if (a1)
        HP_LAUNCH((conv32_head_pair_kernel<1, 2>), grid, dim3(256), 0, s, qa, ha, qb, hb);
    else if (b1)
        HP_LAUNCH((conv32_head_pair_kernel<2, 1>), grid, dim3(256), 0, s, qa, ha, qb, hb);
    else
        HP_LAUNCH((conv32_head_pair_kernel<2, 2>), grid, dim3(256), 0, s, qa, ha, qb, hb);
    return hipGetLastError();
}

hipError_t launch_conv32_head(const conv32_params& q, const head32_hidden& h, hipStream_t s)
{
    if (!conv32_head_ok(HK1, h.HID, q.Cout) || !h.w1_frag || !h.w2_frag || !h.bias1 || q.npix <= 0 || (!q.out.p && !q.out_f32))
        return hipErrorInvalidValue;
    const dim3 grid((q.npix + 31) / 32);
    if (q.Cout <= 32)
        HP_LAUNCH((conv32_head_kernel<1>), grid, dim3(256), 0, s, q, h);
    else
        HP_LAUNCH((conv32_head_kernel<2>), grid, dim3(256), 0, s, q, h);
    return hipGetLastError();
}

} // namespace hp
