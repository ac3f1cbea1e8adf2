// conv32_winograd3.hip — the 3 x 3, stride-1 convolutions of the fp32 engine in Winograd's F(3 x 3, 3 x 3) form (round 6; interface: conv_fp32.hpp).
//
// F(2 x 2, 3 x 3) (conv32_winograd.hip) spends 16 multiplications on a 2 x 2 output tile: 4 per pixel.  F(3 x 3, 3 x 3) - Cook-Toom at the points
// 0, 1, -1, 2, inf - spends 25 on a 3 x 3 tile: 2.78 per pixel, 1.44 x fewer MFMA cycles on the pipe that bounds these layers:
//        Y = At [ (G g Gt) . (Bt d B) ] A            d: the 5 x 5 input patch, g: the filter, ".": element-wise, summed over input channels
//        Bt = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
//        G  = [1/2 0 0; -1/2 -1/2 -1/2; -1/6 1/6 -1/6; 1/6 1/3 2/3; 0 0 1]          At = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
// (tools/winograd_accuracy.py checks the identity in exact rationals).  Accuracy, measured before this kernel was written (CPU, fp32 arithmetic, the drift
// test's frames and weights, all seventeen 3 x 3 layers of LW-OpenPose in this form; profiles/r06_winograd_accuracy_f22_f33_f43.txt): 1.9e-6 of the
// heat-map scale against fp64 - the same as F(2 x 2) and the direct form - and every peak and human identical.  U = G g Gt is formed in double and
// rounded to fp32 once on the host; the transforms' small integers (2, 3, 4) multiply exactly or round once like any fp32 operation.
//
// Kernel shape (the pipelined one-column form of conv32_winograd_kernel<4, true, 1>, which showed that one 16-tile MFMA column per wavefront is enough):
//   block   = 8 x 2 tiles of 3 x 3 = 24 x 6 output pixels x 64 output channels; four wavefronts, each ONE 16-row MFMA tile of channels x 16 tiles x
//             25 positions = 100 accumulator registers; two blocks per CU (65 KB of LDS)
//   K loop  = chunks of 16 input channels.  The chunk's 26 x 8 halo patch arrives from HBM as fp32 (requested two chunks ahead, into registers), goes
//             to LDS (pixel rows of 528 bytes), is transformed into V[pos][tile][16 channels] of the OTHER V buffer while the current chunk multiplies:
//             wavefront w forms row w of Bt d B for all 16 tiles (lane = (tile, 4-channel quad)), wavefront 0 row 4 as well
//   MFMA    = per position one step of 16 channels: A = U in fragment order straight from L2 (1 KB per step, four steps ahead), B = one ds_read_b128
//             from V; four v_mfma_f32_16x16x4_f32 per step, 25 steps per chunk
//   output  = At M A per lane from its own registers, into a slab the block shares ([144 pixels][64 channels]), then conv32_epilogue.hpp's whole pixel rows
// A frame's tiles start at its own first row and column: the same bits in every batch and at every frame offset.
#include "conv_fp32.hpp"

#include "conv32_epilogue.hpp"
#include "conv_device.hpp"

#include <cstdlib>
#include <vector>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W3_CK = 16;                         // input channels per chunk
constexpr int W3_TY = 8, W3_TX = 2;               // tiles of a block
constexpr int W3_BH = 3 * W3_TY, W3_BW = 3 * W3_TX; // 24 x 6 output pixels
constexpr int W3_HH = W3_BH + 2, W3_HW = W3_BW + 2; // 26 x 8 halo patch
constexpr int W3_RP = W3_HW * 64 + 16;            // patch row pitch in bytes (+ 16: the eight tile rows of a 16-lane group spread over the banks)
constexpr int W3_QUADS = W3_HH * W3_HW * 4;       // 832 float4 quads per chunk
constexpr int W3_NQ = (W3_QUADS + 255) / 256;     // 4 per thread
constexpr int W3_RAW = (W3_HH * W3_RP + 16 + 255) / 256 * 256; // + a slot the surplus threads write to
constexpr int W3_VPOS = 16 * 64, W3_VBUF = 25 * W3_VPOS; // one position (16 tiles x 64 B), one buffer (25 600 B)
constexpr int W3_SLAB_PITCH = rows_geom<2>::PITCH;      // 68 floats
constexpr int W3_SLAB = W3_BH * W3_BW * W3_SLAB_PITCH * 4; // 39 168 B, over the V buffers
constexpr int W3_LDS = W3_RAW + 2 * W3_VBUF;      // 65 024 B

__device__ __forceinline__ long tv3_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

} // namespace

__global__ __launch_bounds__(256, 2) void conv32_winograd3_kernel(const conv32_params p, int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[]; // W3_LDS
    unsigned char* const raw = lds;
    unsigned char* const vb = lds + W3_RAW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order (1-D grid): see conv32_winograd_kernel
    const int NG = p.Cout_pad / 64, ntiles = tiles_x * tiles_y * p.B;
    const int bj = blockIdx.x >> 3, by = bj % NG;
    int t = (bj / NG) * 8 + (blockIdx.x & 7);
    if (t >= ntiles)
        return;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * W3_BH, x0 = tx * W3_BW;
    const int MT = p.Cout_pad / 16, mt = by * 4 + wave;
    const int nch = p.Cin / W3_CK;
    int dbg_i = 0;
#define HP_STAMP()                                                          \
    if (p.dbg && blockIdx.x == 9 && tid == 0 && dbg_i < 60)                \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();

    // ---- staging geometry: quad q of a chunk = (halo pixel q / 4, channels 4 (q % 4) ..); halo pixel (hy, hx) = image pixel (y0 - 1 + hy, x0 - 1 + hx).
    // The tensor's zero halo is the convolution's padding; pixels further out (ragged last tiles) are clamped to it and zeroed
    long goff[W3_NQ];
    int soff[W3_NQ];
    bool qok[W3_NQ];
#pragma unroll
    for (int i = 0; i < W3_NQ; ++i) {
        const int q = tid + i * 256, qc = min(q, W3_QUADS - 1);
        const int hp = qc >> 2, c4 = qc & 3;
        const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        qok[i] = y <= p.H && x <= p.W;
        goff[i] = tv3_off(p.in, b, min(y, p.H), min(x, p.W)) + c4 * 4;
        soff[i] = q < W3_QUADS ? hy * W3_RP + hx * 64 + c4 * 16 : W3_HH * W3_RP; // (surplus threads: the spare slot)
    }
    f32x4 stage[W3_NQ];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < W3_NQ; ++i)
            stage[i] = *reinterpret_cast<const f32x4*>(p.in.p + goff[i] + min(c, nch - 1) * W3_CK);
    };
    auto to_lds = [&]() { // (no branch: see conv32_winograd_kernel)
#pragma unroll
        for (int i = 0; i < W3_NQ; ++i)
            *reinterpret_cast<f32x4*>(raw + soff[i]) = qok[i] ? stage[i] : f32x4{ 0.f, 0.f, 0.f, 0.f };
    };

    // ---- input transform: item i = row i of Bt d B for the block's 16 tiles; lane (tile, quad) like the MFMA's B read.  Wavefront w forms row w;
    // row 4 is formed by wavefront (chunk & 3).
    const int btile = lane & 15, kq = lane >> 4;
    const int vdst = btile * 64 + ((kq ^ ((4 - (btile >> 2)) & 3)) * 16); // (quad index XOR-ed with a function of the tile: ds_write_b128 / ds_read_b128 conflict-free)
    const int tsrc = (3 * (btile >> 1)) * W3_RP + (3 * (btile & 1)) * 64 + kq * 16;
    // The transform in pieces, one per MFMA step, and never a wait inside a step: step k requests column k of the patch (five ds_read_b128), step k + 1
    // applies the wavefront's row of Bt down it (coefficients in registers: no branch on the row inside the pinned steps - a switch splits the step into
    // basic blocks and the loads then sit right in front of their use: 48.6 | 28.6 us with it), five more steps apply Bt along the row and store
    // position (i, l).  Own row: steps 1 - 11; row 4 (one wavefront per chunk): steps 12 - 22.
    float cw[4]; // row `wave` of Bt (rows 0 - 3 of Bt have no fifth entry: patch rows 0 - 3 are all they read)
    {
        const float bt[4][4] = { { 2.f, -1.f, -2.f, 1.f }, { 0.f, -2.f, -1.f, 1.f }, { 0.f, 2.f, -3.f, 1.f }, { 0.f, -1.f, 0.f, 1.f } };
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cw[k] = wave == 0 ? bt[0][k] : wave == 1 ? bt[1][k] : wave == 2 ? bt[2][k] : bt[3][k];
    }
    f32x4 T[5], dc[4];
    auto col_load = [&](int row0, int j) { // patch rows row0 .. row0 + 3 of column j
#pragma unroll
        for (int r = 0; r < 4; ++r)
            dc[r] = *reinterpret_cast<const f32x4*>(raw + tsrc + j * 64 + (row0 + r) * W3_RP);
    };
    auto col_own = [&](int j) { T[j] = ((cw[0] * dc[0] + cw[1] * dc[1]) + cw[2] * dc[2]) + cw[3] * dc[3]; };
    auto col_row4 = [&](int j) { T[j] = ((2.f * dc[0] - dc[1]) - 2.f * dc[2]) + dc[3]; }; // (dc = patch rows 1 .. 4)
    auto v_store = [&](int i, int l, int dstoff, int buf) {
        const f32x4 v = l == 0 ? ((2.f * T[0] - T[1]) - 2.f * T[2]) + T[3]
            : l == 1       ? (T[3] - T[2]) - 2.f * T[1]
            : l == 2       ? (2.f * T[1] - 3.f * T[2]) + T[3]
            : l == 3       ? T[3] - T[1]
                           : ((2.f * T[1] - T[2]) - 2.f * T[3]) + T[4];
        *reinterpret_cast<f32x4*>(vb + buf * W3_VBUF + (i * 5 + l) * W3_VPOS + dstoff) = v;
    };
    // piece k = 0 .. 10 of an item: k <= 4 requests column k, 1 <= k <= 5 combines column k - 1, k >= 6 stores position l = k - 6
    auto piece = [&](bool own, int k, int buf) {
        if (k >= 1 && k <= 5) {
            if (own)
                col_own(k - 1);
            else
                col_row4(k - 1);
        }
        if (k <= 4)
            col_load(own ? 0 : 1, k);
        if (k >= 6)
            v_store(own ? wave : 4, k - 6, vdst, buf);
    };
    // row 4 of a chunk is formed by ONE wavefront, in turn: wavefront (chunk & 3) (a branch on the wavefront's index, around whole pieces only)
    auto transform_all = [&](int buf) { // chunk 0: nothing to hide it under
#pragma unroll
        for (int k = 0; k < 11; ++k)
            piece(true, k, buf);
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 11; ++k)
                piece(false, k, buf);
        }
    };

    // ---- A fragments: [chunk][pos][16-row tile][lane][4 floats]; step s = chunk * 25 + pos
    const long step_stride = (long)MT * 256;
    const float* const wp = p.w_wino3 + (long)mt * 256 + lane * 4;
    const int nsteps = nch * 25;
    constexpr int RING = 5, AHEAD = 4;
    f32x4 fa[RING];
    auto aload = [&](int slot, int s) { fa[slot] = *reinterpret_cast<const f32x4*>(wp + (long)min(s, nsteps - 1) * step_stride); };

    f32x4 acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i)
        acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    const unsigned char* const vsrc = vb + vdst;

    gload(0);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        aload(a, a);
    to_lds();
    gload(1);
    lds_barrier();
    transform_all(0);
    HP_STAMP();
    int s = 0;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        lds_barrier(); // V[c & 1] is complete; every wavefront has left the patch (transform c) and V[(c + 1) & 1] (chunk c - 1's MFMAs)
        to_lds();      // chunk c + 1's patch
        gload(c + 2);  // (past the last chunk: a harmless re-read of it)
        lds_barrier(); // the patch is complete
        HP_STAMP();
        const unsigned char* const vcur = vsrc + (c & 1) * W3_VBUF;
        f32x4 fb[2];
        fb[0] = *reinterpret_cast<const f32x4*>(vcur);
#pragma unroll
        for (int pos = 0; pos < 25; ++pos) {
            const int cur = pos & 1, npos = pos + 1 < 25 ? pos + 1 : pos;
            fb[cur ^ 1] = *reinterpret_cast<const f32x4*>(vcur + npos * W3_VPOS);
            aload((pos + AHEAD) % RING, s + AHEAD); // (25 % RING == 0: the ring position is a compile-time function of pos)
            const int slot = pos % RING;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[slot][e], fb[cur][e], acc[pos], 0, 0, 0);
            // chunk c + 1's transform under this chunk's MFMAs, a piece per step: the wavefront's own row in steps 1 - 11, its quarter of row 4 in 12 - 22
            if (pos >= 1 && pos <= 11)
                piece(true, pos - 1, (c + 1) & 1);
            if (pos >= 12 && pos <= 22 && wave == ((c + 1) & 3))
                piece(false, pos - 12, (c + 1) & 1);
            // issue order of a step (hipcc otherwise sinks every load to just before its use): MFMA, LDS read (the next position's B), MFMA, L2 read (A four
            // steps ahead), two MFMAs; the transform's piece goes wherever hipcc finds room between them
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            ++s;
        }
        HP_STAMP();
    }

    // ---- output transform Y = At M A, per lane: tile btile, channels 16 wave + 4 kq + r; then whole pixel rows through the block's slab.
    // Slab row (3 tile_y + a) * 6 + 3 tile_x + bb = output pixel (a, bb) of the tile
    lds_barrier(); // every wavefront is done with V, which the slab lies over
    float* const slab = reinterpret_cast<float*>(vb);
    {
        f32x4 S[3][5]; // At M
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            S[0][j] = ((acc[j] + acc[5 + j]) + acc[10 + j]) + acc[15 + j];
            S[1][j] = (acc[5 + j] - acc[10 + j]) + 2.f * acc[15 + j];
            S[2][j] = ((acc[5 + j] + acc[10 + j]) + 4.f * acc[15 + j]) + acc[20 + j];
        }
        const int prow = 3 * (btile >> 1), pcol = 3 * (btile & 1);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const f32x4 y0v = ((S[a][0] + S[a][1]) + S[a][2]) + S[a][3];
            const f32x4 y1v = (S[a][1] - S[a][2]) + 2.f * S[a][3];
            const f32x4 y2v = ((S[a][1] + S[a][2]) + 4.f * S[a][3]) + S[a][4];
            float* const row = slab + ((prow + a) * W3_BW + pcol) * W3_SLAB_PITCH + wave * 16 + kq * 4;
            *reinterpret_cast<f32x4*>(row) = y0v;
            *reinterpret_cast<f32x4*>(row + W3_SLAB_PITCH) = y1v;
            *reinterpret_cast<f32x4*>(row + 2 * W3_SLAB_PITCH) = y2v;
        }
    }
    HP_STAMP();
    lds_barrier(); // the slab holds all 64 channels of the block's 144 pixels; wavefront w stores rows 36 w ..
    constexpr int RPW = W3_BH * W3_BW / 4;
    conv32_drain_rows<2, RPW>(p, slab + wave * RPW * W3_SLAB_PITCH, lane, by * 64, [&](int r, bool& ok, long& ooff, long& roff) {
        const int rr = wave * RPW + r;
        const int oy = y0 + rr / W3_BW, ox = x0 + rr % W3_BW;
        ok = oy < p.OH && ox < p.OW;
        const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 1);
        ooff = tv3_off(p.out, b, oyc, oxc);
        roff = p.res.p ? tv3_off(p.res, b, oyc, oxc) : 0;
    });
    HP_STAMP();
#undef HP_STAMP
}

// 3 x 3, stride 1, dilation 1, SAME padding, input slice readable in whole 16-channel chunks, NHWC output only
bool conv32_winograd3_ok(const conv32_params& p)
{
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.dil == 1 && p.Cin % W3_CK == 0 && p.Cout_pad % 64 == 0 && p.OH == p.H && p.OW == p.W && p.pad_t == 1
        && p.pad_l == 1 && !p.out_f32 && p.out.p;
}

int conv32_winograd3_tile(const conv32_params&) { return 35005004; }

// packed = [9 taps][cout_pad][cin] fp32 (conv32_params::w's layout) -> U = G g Gt (5 x 5) in fragment order [chunk][pos][16-row tile][lane][4 floats]
// (25 * cout_pad * cin floats); lane (row, kq) = channels chunk * 16 + 4 kq + {0..3}.  U is formed in double and rounded to fp32 once.
void conv32_winograd3_pack(const float* packed, int cout_pad, int cin, float* out)
{
    static const double G[5][3] = { { .5, 0, 0 }, { -.5, -.5, -.5 }, { -1. / 6, 1. / 6, -1. / 6 }, { 1. / 6, 1. / 3, 2. / 3 }, { 0, 0, 1 } };
    const int nch = cin / W3_CK, MT = cout_pad / 16;
    std::vector<double> U((size_t)25 * cout_pad * cin);
    for (int m = 0; m < cout_pad; ++m)
        for (int k = 0; k < cin; ++k) {
            double g[3][3], Gg[5][3];
            for (int t = 0; t < 9; ++t)
                g[t / 3][t % 3] = packed[((size_t)t * cout_pad + m) * cin + k];
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 3; ++j)
                    Gg[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 5; ++j)
                    U[((size_t)(i * 5 + j) * cout_pad + m) * cin + k] = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
        }
    for (int c = 0; c < nch; ++c)
        for (int pos = 0; pos < 25; ++pos)
            for (int mt = 0; mt < MT; ++mt) {
                float* dst = out + (((size_t)c * 25 + pos) * MT + mt) * 256;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int m = mt * 16 + (lane & 15), k = c * W3_CK + (lane >> 4) * 4 + e;
                        dst[lane * 4 + e] = (float)U[((size_t)pos * cout_pad + m) * cin + k];
                    }
            }
}

hipError_t launch_conv32_winograd3(const conv32_params& p, hipStream_t s)
{
    if (!conv32_winograd3_ok(p) || !p.w_wino3 || p.npix <= 0)
        return hipErrorInvalidValue;
    static bool granted = false;
    if (!granted) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv32_winograd3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
        if (e != hipSuccess)
            return e;
        granted = true;
    }
    const int tiles_x = (p.OW + W3_BW - 1) / W3_BW, tiles_y = (p.OH + W3_BH - 1) / W3_BH;
    const dim3 grid((tiles_x * tiles_y * p.B + 7) / 8 * 8 * (p.Cout_pad / 64)); // XCD-aware 1-D order: see the kernel
    HP_LAUNCH(conv32_winograd3_kernel, grid, dim3(256), W3_LDS, s, p, tiles_x, tiles_y);
    return hipGetLastError();
}

} // namespace hp
