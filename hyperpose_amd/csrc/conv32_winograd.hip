// conv32_winograd.hip — the 3 x 3, stride-1 convolutions of the fp32 engine (HP_DTYPE_F32, data_type::kFLOAT) in Winograd's minimal form
// F(2 x 2, 3 x 3) on the fp32 matrix pipe (interface: conv_fp32.hpp).
//
// The fp32 pipe (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD, 157 TFLOP/s) is the bound of every dense fp32 layer here, so the lever left is the
// number of multiplications.  For a 2 x 2 output tile and a 3 x 3 filter (Lavin & Gray; what cuDNN / TensorRT pick for fp32 3 x 3 layers - the
// reference's engine, src/tensorrt.cpp:327-353, leaves the choice to the TensorRT builder):
//        Y = At [ (G g Gt) . (Bt d B) ] A            d: the 4 x 4 input patch, g: the filter, ".": element-wise, summed over input channels
//        Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    At = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 multiplications per tile and channel pair instead of 36: the sum over input channels is 16 independent matrix products
//        M[pos][cout][tile] = sum_c U[pos][cout][c] V[pos][c][tile]          U = G g Gt (host, once),  V = Bt d B (per launch, in LDS)
// - 2.25 x fewer MFMA cycles than the direct form.  Input and output transforms are additions only; U's factors 1/2 and 1/4 are exact in binary
// and U is rounded to fp32 once on the host (computed in double).  The products and sums of the 16 matrix products are exact fp32 FMAs on the
// matrix pipe like everywhere else in this engine; what changes against the direct form is the order of summation and the cancellation inside the
// transforms - measured against the pure fp32 oracle in tests/test_engine_fp32_gpu.py (same 1e-4 bound as every fp32 kernel; DESIGN.md 7 has the
// measured figures).
//
// Kernel shape (MW = wavefronts per block, 4 by default):
//   block   = 16 x 8 output pixels = 8 x 4 Winograd tiles (two N = 16 column tiles of the MFMA) x 16 MW output channels; a wavefront = ONE
//             16-row MFMA tile of channels x 32 tiles x 16 positions = 128 accumulator registers.  MW = 4: two blocks per CU - one block's
//             transform phases and epilogue run under the other's MFMAs; MW = 8 (HP_WINO_MW=8): one block of two wavefronts per SIMD per
//             CU, the staging / transform of a pixel tile shared by 128 channels - 44.6 us alone against 46.3 for a 128 -> 128 layer at
//             8 x 46 x 54, but 31.5 against 28.8 with a second stream (its phases are in step inside the one block).
//             (The first form - 32 channels x 16 tiles per wavefront - read 2 KB of A per eight MFMAs: 68 us alone.)
//   K loop  = chunks of 16 input channels.  The chunk's 18 x 10 halo patch arrives from HBM as fp32 (requested two chunks ahead, into
//             registers), goes to LDS, is transformed cooperatively (a wavefront = ONE row of Bt d B for 16 tiles, lane = (tile, 4-channel
//             quad): 8 LDS reads, 8 vector additions, 4 LDS writes) into V[pos][tile][16 channels].  MW = 4 (PIPE): two V buffers, the
//             transform of chunk c + 1 cut into 16 pieces that sit inside the 16 MFMA steps of chunk c (46.6 -> 44.3 us alone, 28.9 -> 27.0
//             with a second stream; HP_WINO_PIPE=0 is the A/B switch back); MW = 8: one buffer, the transform between two barriers;
//   MFMA    = per position one step of 16 channels: lane (row / tile, kq) holds channels 4 kq .. 4 kq + 3 of its row (A, 1 KB from L2 in
//             fragment order, six steps ahead) / tile (B, two ds_read_b128 from V) and feeds element e to MFMA e - 8 MFMAs of 32 cycles per step;
//   output  = At M A per lane from its own registers (lane (tile, kq) holds channels 4 kq + r at all 16 positions), written into a slab the
//             block shares ([128 pixels][16 MW channels]), then the row-major epilogue of conv32_epilogue.hpp: whole pixel rows.
// Block timeline (tools/direct_timeline.py f32; shader cycles at 2.2 GHz, MW = 4, PIPE, 128 -> 128 at 8 x 46 x 54, two blocks per CU): start +
// chunk 0's transform 5.7 k | per chunk: patch stored 0.6 - 1.2 k, multiplied 7.4 - 8.0 k (8.2 k = the pipe's time for the two blocks' 2 x 128
// MFMAs per SIMD: the loop is at the pipe's rate) | output transform + stores 17 k.  89 k cycles per block of which 2 x 32.8 k are MFMA issue.
#include "conv_fp32.hpp"

#include "conv32_epilogue.hpp"
#include "conv_device.hpp"

#include <cstdlib>
#include <vector>

namespace hp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WCK = 16;                        // input channels per chunk
constexpr int HALO_W = 10;                     // halo patch of the 16 NC x 8 pixel tile: (8 NC + 2) x 10 pixels
constexpr int RAW_PB = WCK * 4;
constexpr int VP = 24 * 4;                     // V row (one tile, 16 channels) pitch in bytes: 24 floats - the transform's ds_write_b128 and the MFMA's ds_read_b128 are conflict-free

// NC = 16-tile MFMA column tiles per block and wavefront: 2 = a block of 16 x 8 pixels (8 x 4 tiles), 1 = 8 x 8 pixels (4 x 4 tiles)
template <int MW, int NC = 2>
struct wino_geom {
    static constexpr int NT = 64 * MW;
    static constexpr int HALO_H = 8 * NC + 2;
    static constexpr int RAW_Q = HALO_H * HALO_W * 4; // float4 quads of the halo patch (64 bytes per pixel): 720 | 400
    static constexpr int NQ = (RAW_Q + NT - 1) / NT;  // quads per thread and chunk
    static constexpr int RAW_BYTES = NQ * NT * 16;    // + room for the surplus threads' (discarded) quads: the store to LDS has no branch
    static constexpr int NI = 4 * NC / MW;            // transform items (row of Bt d B, 16 tiles) per wavefront and chunk
    static_assert(NI >= 1, "a transform item per wavefront");
    static constexpr int TMS = MW / 2;                // the block's output slab: [64 NC pixel rows][16 MW channels + 4] = rows_geom<TMS>
    static constexpr int SLAB_PITCH = rows_geom<TMS>::PITCH, SLAB_BYTES = 64 * NC * SLAB_PITCH * 4;
    static constexpr int VPOS = 16 * NC * VP, VBUF = 16 * VPOS; // one position (16 NC tiles), all 16 positions
    static constexpr int LDS_BYTES = RAW_BYTES + (VBUF > SLAB_BYTES ? VBUF : SLAB_BYTES); // MW = 4: 61440 (two blocks per CU), MW = 8: 83968 (one)
    // the pipelined form (MW = 4): two V buffers without padding - 64 bytes per tile, the 16-byte quad index XOR-ed with a function of the
    // tile so that the transform's ds_write_b128 and the MFMA's ds_read_b128 (same lane -> (tile, quad) map) stay conflict-free
    static constexpr int PVPOS = 16 * NC * 64, PVBUF = 16 * PVPOS;                  // 32768 | 16384 bytes per buffer
    static constexpr int PLDS_BYTES = RAW_BYTES + (2 * PVBUF > SLAB_BYTES ? 2 * PVBUF : SLAB_BYTES); // 12288 + 65536 = 77824: two blocks per CU
};

__device__ __forceinline__ long tvw_off(const tview32& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

} // namespace

// PIPE: the input transform of chunk c + 1 runs inside the MFMA steps of chunk c (two V buffers): see "K loop" above
template <int MW, bool PIPE, int NC = 2>
__global__ __launch_bounds__(64 * MW, MW == 8 ? 1 : NC == 1 ? 3 : 2) void conv32_winograd_kernel(const conv32_params p, int tiles_x, int tiles_y, int vh)
{
    static_assert(!PIPE || MW == 4, "the pipelined form is the four-wavefront one");
    using G = wino_geom<MW, NC>;
    constexpr int NT = G::NT, NQ = G::NQ, SLAB_PITCH = G::SLAB_PITCH, RAW_Q = G::RAW_Q, VPOS = G::VPOS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[]; // G::LDS_BYTES
    unsigned char* const raw = lds;
    unsigned char* const vb = lds + G::RAW_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order (1-D grid): block id -> XCD id % 8; the NG channel groups of one pixel tile are ids b, b + 8, .. - back to back on one
    // XCD, whose L2 then serves the tile's halo patch NG - 1 times (conv_fp32.hip has the measurement behind this)
    // vh > 0, the "tall" form: the batch as ONE image of (B - 1) vh + H rows - an image's rows plus its halo rows are vh (even) rows of the buffer,
    // the halo rows between two images are zeros, i.e. the padding both of them need.  Tiles then run down the whole batch: a 12 x 12 map is
    // 14 rows of a 446-row image (28 blocks of 16 rows) instead of one three-quarters-empty block per image (32).  vh is even, so every 2 x 2
    // Winograd tile covers the same rows of every image: the same bits as the per-image form, whatever the batch.  Output rows that fall on
    // separator rows are computed and not stored.
    const int NG = p.Cout_pad / (16 * MW), ntiles = tiles_x * tiles_y * (vh ? 1 : p.B);
    const int Hin = vh ? (p.B - 1) * vh + p.H : p.H; // rows 0 .. Hin - 1; row Hin = the (last) image's bottom halo row
    const int bj = blockIdx.x >> 3, by = bj % NG;
    int t = (bj / NG) * 8 + (blockIdx.x & 7);
    if (t >= ntiles)
        return; // (the grid is padded to whole groups of eight pixel tiles)
    const int bx = t;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * (8 * NC), x0 = tx * 8;
    const int MT = p.Cout_pad / 16, mt = by * MW + wave;
    const int nch = p.Cin / WCK;
    int dbg_i = 0;
#define HP_STAMP()                                                     \
    if (p.dbg && bx == 1 && by == 0 && tid == 0)       \
        p.dbg[dbg_i++] = __builtin_amdgcn_s_memtime();
    HP_STAMP();
    if (p.dbg && bx == 1 && by == 0 && tid == 0)
        p.dbg[119] = __builtin_readcyclecounter(), p.dbg[120] = __builtin_amdgcn_s_memrealtime();

    // ---- staging geometry: quad q of a chunk = (halo pixel q / 4, channels 4 (q % 4) ..); halo pixel (hy, hx) = image pixel (y0 - 1 + hy, x0 - 1 + hx).
    // The tensor's zero halo is the convolution's padding; pixels further out (ragged last tiles, the fourth patch row / column of an odd-sized
    // map) are clamped to it and zeroed - they only feed outputs that are never stored
    long goff[NQ];
    bool qok[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = min(tid + i * NT, RAW_Q - 1);
        const int hp = q >> 2, c4 = q & 3;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        qok[i] = y <= Hin && x <= p.W;
        goff[i] = tvw_off(p.in, b, min(y, Hin), min(x, p.W)) + c4 * 4;
    }
    f32x4 stage[NQ];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            stage[i] = *reinterpret_cast<const f32x4*>(p.in.p + goff[i] + min(c, nch - 1) * WCK);
    };
    auto to_lds = [&]() { // (no branch: with one, hipcc drains EVERY load in flight - the A fragments of the next steps too - before the first store)
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            *reinterpret_cast<f32x4*>(raw + (tid + i * NT) * 16) = qok[i] ? stage[i] : f32x4{ 0.f, 0.f, 0.f, 0.f };
    };
    // ---- input transform: item of a wavefront = (row i of Bt d B, 16 of the 32 tiles); lane (tile, quad) like the MFMA's B read.  Row i of
    // T = Bt d combines two patch rows; V[i][.] = T[i][.] B
    auto transform = [&]() {
#pragma unroll
        for (int k = 0; k < G::NI; ++k) {
            const int item = wave + k * MW, i = NC == 2 ? item >> 1 : item;
            const int quad = lane >> 4, tile = (NC == 2 ? (item & 1) * 16 : 0) + (lane & 15);
            const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
            const float sg = i == 1 ? 1.f : -1.f;
            const unsigned char* const src = raw + ((2 * (tile >> 2)) * HALO_W + 2 * (tile & 3)) * RAW_PB + quad * 16;
            f32x4 T[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + (ra * HALO_W + j) * RAW_PB);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(src + (rb * HALO_W + j) * RAW_PB);
                T[j] = a + sg * bq; // (an exact sign change, then one rounded addition)
            }
            constexpr int PS = PIPE ? G::PVPOS : VPOS;
            unsigned char* const dst = PIPE ? vb + (i * 4) * PS + tile * 64 + ((quad ^ ((4 - (tile >> 2)) & 3)) * 16) : vb + (i * 4) * PS + tile * VP + quad * 16;
            *reinterpret_cast<f32x4*>(dst) = T[0] - T[2];
            *reinterpret_cast<f32x4*>(dst + PS) = T[1] + T[2];
            *reinterpret_cast<f32x4*>(dst + 2 * PS) = T[2] - T[1];
            *reinterpret_cast<f32x4*>(dst + 3 * PS) = T[1] - T[3];
        }
    };
    // the same transform in pieces, one per MFMA step (PIPE): item k = 0 in steps 0 - 7, k = 1 in steps 8 - 15 of the PREVIOUS chunk's
    // multiplication: steps 0 - 3 read a column pair each, step 4 forms T, steps 4 - 7 form and store one position each
    f32x4 ta[4], tb[4];
    auto transform_piece = [&](int pos, int buf) {
        const int k = pos >> 3, ps = pos & 7;
        if (k >= G::NI)
            return; // (NC = 1: one item per wavefront, in steps 0 - 7)
        const int item = wave + k * MW, i = NC == 2 ? item >> 1 : item;
        const int quad = lane >> 4, tile = (NC == 2 ? (item & 1) * 16 : 0) + (lane & 15);
        const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
        const unsigned char* const src = raw + ((2 * (tile >> 2)) * HALO_W + 2 * (tile & 3)) * RAW_PB + quad * 16;
        if (ps < 4) {
            ta[ps] = *reinterpret_cast<const f32x4*>(src + (ra * HALO_W + ps) * RAW_PB);
            tb[ps] = *reinterpret_cast<const f32x4*>(src + (rb * HALO_W + ps) * RAW_PB);
        } else {
            if (ps == 4) {
                const float sg = i == 1 ? 1.f : -1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ta[j] = ta[j] + sg * tb[j];
            }
            unsigned char* const dst = vb + buf * G::PVBUF + (i * 4 + (ps - 4)) * G::PVPOS + tile * 64 + ((quad ^ ((4 - (tile >> 2)) & 3)) * 16);
            *reinterpret_cast<f32x4*>(dst) = ps == 4 ? ta[0] - ta[2] : ps == 5 ? ta[1] + ta[2] : ps == 6 ? ta[2] - ta[1] : ta[1] - ta[3];
        }
    };

    // ---- A fragments: [chunk][pos][16-row tile][lane][4 floats]; step s = chunk * 16 + pos
    const long step_stride = (long)MT * 256;
    const float* const wp = p.w_wino + (long)mt * 256 + lane * 4;
    const int nsteps = nch * 16;
    constexpr int RING = PIPE ? 4 : 8, AHEAD = PIPE ? 3 : 6; // (PIPE: the transform's eight patch quads want the registers)
    f32x4 fa[RING];
    auto aload = [&](int slot, int s) { fa[slot] = *reinterpret_cast<const f32x4*>(wp + (long)min(s, nsteps - 1) * step_stride); };

    f32x4 acc[16][NC]; // [position][column tile: tile rows 0 - 3 | 4 - 7 of the block]
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
            acc[i][nc] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    const int btile = lane & 15, kq = lane >> 4;
    const unsigned char* const vsrc = PIPE ? vb + btile * 64 + ((kq ^ ((4 - (btile >> 2)) & 3)) * 16) : vb + btile * VP + kq * 16;
    constexpr int BPOS = PIPE ? G::PVPOS : VPOS, BNT = PIPE ? 16 * 64 : 16 * VP; // B fragment strides: position, second column tile

    gload(0);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        aload(a, a);
    if constexpr (PIPE) { // chunk 0's transform is the one that nothing hides
        to_lds();
        gload(1);
        lds_barrier();
        transform();
    }
    int s = 0;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        if constexpr (PIPE)
            lds_barrier(); // V[c & 1] is complete and every wavefront has left the patch (transform c) and V[(c + 1) & 1] (chunk c - 1's MFMAs)
        to_lds();     // PIPE: chunk c + 1's patch; else chunk c's (the previous chunk's transform is behind every wavefront: its second barrier)
        gload(PIPE ? c + 2 : c + 1); // (past the last chunk: a harmless re-read of it)
        lds_barrier(); // the patch is complete (!PIPE: AND every wavefront has left the previous chunk's MFMAs: V may be overwritten)
        HP_STAMP();
        if constexpr (!PIPE) {
            transform();
            lds_barrier();
            HP_STAMP();
        }
        const unsigned char* const vcur = vsrc + (PIPE ? (c & 1) * G::PVBUF : 0);
        f32x4 fb[2][NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
            fb[0][nc] = *reinterpret_cast<const f32x4*>(vcur + nc * BNT);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            const int cur = pos & 1, npos = pos + 1 < 16 ? pos + 1 : pos;
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
                fb[cur ^ 1][nc] = *reinterpret_cast<const f32x4*>(vcur + npos * BPOS + nc * BNT);
            aload((pos + AHEAD) % RING, s + AHEAD); // (16 % RING == 0: the ring position is a compile-time function of pos)
            const int slot = pos % RING;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
                    acc[pos][nc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[slot][e], fb[cur][nc][e], acc[pos][nc], 0, 0, 0);
            if constexpr (PIPE)
                transform_piece(pos, (c + 1) & 1); // chunk c + 1's transform, one piece per step, under this step's MFMAs
            // issue order of a step: MFMA, LDS read, MFMA, LDS read (the next position's B), MFMA, L2 read (A some steps ahead), 5 MFMAs
            // (the transform's piece - two LDS reads, or a few additions and one LDS write - goes wherever hipcc finds room between them)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (NC == 2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NC == 2 ? 5 : 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            ++s;
        }
        HP_STAMP();
    }

    // ---- output transform Y = At M A, per lane: tile 16 nt + (lane & 15), channels 16 wave + 4 kq + r; then whole pixel rows through the
    // block's slab.  Slab row (a * 2 + bb) * 32 + tile = output pixel (2 tile_y + a, 2 tile_x + bb) of the block's 16 x 8
    lds_barrier(); // every wavefront is done with V, which the slab lies over (MW = 8: and beyond)
    HP_STAMP();
    float* const slab = reinterpret_cast<float*>(vb);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        f32x4 S[2][4]; // At M: S[0][j] = M0j + M1j + M2j, S[1][j] = M1j - M2j - M3j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            S[0][j] = acc[j][nt] + acc[4 + j][nt] + acc[8 + j][nt];
            S[1][j] = acc[4 + j][nt] - acc[8 + j][nt] - acc[12 + j][nt];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const f32x4 y0v = S[a][0] + S[a][1] + S[a][2], y1v = S[a][1] - S[a][2] - S[a][3];
            *reinterpret_cast<f32x4*>(slab + ((a * 2 + 0) * (16 * NC) + nt * 16 + btile) * SLAB_PITCH + wave * 16 + kq * 4) = y0v;
            *reinterpret_cast<f32x4*>(slab + ((a * 2 + 1) * (16 * NC) + nt * 16 + btile) * SLAB_PITCH + wave * 16 + kq * 4) = y1v;
        }
    }
    HP_STAMP();
    lds_barrier(); // the slab holds all 16 MW channels of the block's 128 pixels; wavefront w stores rows (128 / MW) w ..
    HP_STAMP();
    constexpr int RPW = 64 * NC / MW;
    conv32_drain_rows<G::TMS, RPW>(p, slab + wave * RPW * SLAB_PITCH, lane, by * 16 * MW, [&](int r, bool& ok, long& ooff, long& roff) {
        const int rr = wave * RPW + r, ab = rr / (16 * NC), tile = rr % (16 * NC);
        int oy = y0 + 2 * (tile >> 2) + (ab >> 1), ob = b;
        const int ox = x0 + 2 * (tile & 3) + (ab & 1);
        if (vh) // tall form: row oy of the batch = row oy % vh of image oy / vh
            ob = oy / vh, oy -= ob * vh;
        ok = oy < p.OH && ox < p.OW && ob < p.B;
        const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 1), obc = min(ob, p.B - 1);
        ooff = tvw_off(p.out, obc, oyc, oxc);
        roff = p.res.p ? tvw_off(p.res, obc, oyc, oxc) : 0;
    });
    HP_STAMP();
#undef HP_STAMP
    if (p.dbg && bx == 1 && by == 0 && tid == 0) // slots 119 - 122: the shader clock and the constant 100 MHz clock at the start / end
        p.dbg[121] = __builtin_readcyclecounter(), p.dbg[122] = __builtin_amdgcn_s_memrealtime();
}

// 3 x 3, stride 1, dilation 1, SAME padding, input slice readable in whole 16-channel chunks, NHWC output only (a network head keeps the
// direct kernel's lane = pixel epilogue for its NCHW copy)
bool conv32_winograd_ok(const conv32_params& p)
{
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.dil == 1 && p.Cin % WCK == 0 && p.Cout_pad % 64 == 0 && p.OH == p.H && p.OW == p.W && p.pad_t == 1
        && p.pad_l == 1 && !p.out_f32 && p.out.p;
}

// Wavefronts (16-channel MFMA tiles) per block: 4 (two blocks per CU); HP_WINO_MW=8 where the channel count allows: see "Kernel shape"
static int winograd_mw(const conv32_params& p)
{
    static const int force = getenv("HP_WINO_MW") ? atoi(getenv("HP_WINO_MW")) : 0;
    return (force == 8 && p.Cout_pad % 128 == 0) ? 8 : 4;
}

int conv32_winograd_tile(const conv32_params& p) { return p.w_wino3 ? conv32_winograd3_tile(p) : 35000000 + 3000 + winograd_mw(p); }

// MFMA work of one launch (what the roofline fraction of this kernel is computed from): 16 products per tile and channel pair
double conv32_winograd_flops(const conv32_params& p)
{
    return 2.0 * 16 * (double)p.B * ((p.OH + 1) / 2) * ((p.OW + 1) / 2) * p.Cout * p.Cin;
}

// packed = [9 taps][cout_pad][cin] fp32 (conv32_params::w's layout) -> U = G g Gt in fragment order [chunk][pos][16-row tile][lane][4 floats]
// (16 * cout_pad * cin floats); lane (row, kq) = channels chunk * 16 + 4 kq + {0..3}.  U is formed in double and rounded to fp32 once.
void conv32_winograd_pack(const float* packed, int cout_pad, int cin, float* out)
{
    static const double G[4][3] = { { 1, 0, 0 }, { .5, .5, .5 }, { .5, -.5, .5 }, { 0, 0, 1 } };
    const int nch = cin / WCK, MT = cout_pad / 16;
    std::vector<double> U((size_t)16 * cout_pad * cin);
    for (int m = 0; m < cout_pad; ++m)
        for (int k = 0; k < cin; ++k) {
            double g[3][3], Gg[4][3];
            for (int t = 0; t < 9; ++t)
                g[t / 3][t % 3] = packed[((size_t)t * cout_pad + m) * cin + k];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j)
                    Gg[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[((size_t)(i * 4 + j) * cout_pad + m) * cin + k] = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
        }
    for (int c = 0; c < nch; ++c)
        for (int pos = 0; pos < 16; ++pos)
            for (int mt = 0; mt < MT; ++mt) {
                float* dst = out + (((size_t)c * 16 + pos) * MT + mt) * 256;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int m = mt * 16 + (lane & 15), k = c * WCK + (lane >> 4) * 4 + e;
                        dst[lane * 4 + e] = (float)U[((size_t)pos * cout_pad + m) * cin + k];
                    }
            }
}

template <int MW, bool PIPE, int NC = 2>
static hipError_t launch_wino_case(const conv32_params& q, dim3 grid, int tiles_x, int tiles_y, int vh, hipStream_t s)
{
    constexpr int lds = PIPE ? wino_geom<MW, NC>::PLDS_BYTES : wino_geom<MW, NC>::LDS_BYTES;
    static bool granted = false;
    if (lds > 64 * 1024 && !granted) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv32_winograd_kernel<MW, PIPE, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess)
            return e;
        granted = true;
    }
    HP_LAUNCH((conv32_winograd_kernel<MW, PIPE, NC>), grid, dim3(64 * MW), lds, s, q, tiles_x, tiles_y, vh);
    return hipGetLastError();
}

// Column tiles per block: 2 (16 x 8 pixels) or 1 (8 x 8 pixels: twice the blocks, half the accumulators, three blocks per CU - and twice the U
// bytes per MFMA).  A 128 -> 128 layer at 8 x 46 x 54 is 336 blocks of the first form for 256 CUs x 2: 176 CUs run one block, 80 run two and
// set the launch's time (45.0 us); as 672 blocks of the second it takes 35.9 us alone and the same 27 us next to a second stream's launch -
// but four pipes lose 1.4 % (5 113 -> 5 042 frames/s, three runs each: the U stream).  So: the small form for a caller with ONE batch in flight
// (conv32_params::latency, i.e. hp_engine_set_concurrency(e, 2)) when the large form would not fill the chip's 512 slots twice; the same
// tiles, the same arithmetic: the same bits (tests/test_engine_fp32_gpu.py).  HP_WINO_NC=1 | 2 forces one (read per launch).
static int winograd_nc(const conv32_params& p, int blocks_nc2)
{
    const int force = getenv("HP_WINO_NC") ? atoi(getenv("HP_WINO_NC")) : 0;
    if (force == 1 || force == 2)
        return force;
    return p.latency && blocks_nc2 <= 512 ? 1 : 2;
}

// Rows per image of the tall form, or 0 where it does not apply: the input's images must lie vh = even rows apart with at least one (zero) halo
// row above and below each.  HP_WINO_TALL=0: the A/B switch (read per launch).
static int winograd_tall(const conv32_params& p)
{
    if (getenv("HP_WINO_TALL") && atoi(getenv("HP_WINO_TALL")) == 0)
        return 0;
    if (p.B < 2 || p.in.wp <= 0 || p.in.img % p.in.wp)
        return 0;
    const int vh = p.in.img / p.in.wp;
    return (vh & 1) == 0 && vh >= p.H + 2 ? vh : 0;
}

static bool winograd_pipe() // HP_WINO_PIPE=0: the A/B switch back to the form with the transform between two barriers
{
    static const bool off = getenv("HP_WINO_PIPE") && atoi(getenv("HP_WINO_PIPE")) == 0;
    return !off;
}

hipError_t launch_conv32_winograd(const conv32_params& p, hipStream_t s)
{
    if (!conv32_winograd_ok(p) || !p.w_wino || p.npix <= 0)
        return hipErrorInvalidValue;
    const int tiles_x = (p.OW + 7) / 8, mw = winograd_mw(p);
    const int nc = mw == 4 && winograd_pipe() ? winograd_nc(p, tiles_x * ((p.OH + 15) / 16) * p.B * (p.Cout_pad / 64)) : 2, bh = 8 * nc;
    int tiles_y = (p.OH + bh - 1) / bh, vh = winograd_tall(p), images = p.B;
    if (vh) {
        const int tall_y = ((p.B - 1) * vh + p.H + bh - 1) / bh;
        if (tall_y < tiles_y * p.B)
            tiles_y = tall_y, images = 1;
        else
            vh = 0; // (a map that is whole 16-row blocks already: 48 rows + 2 halo rows would only add separator rows)
    }
    const dim3 grid((tiles_x * tiles_y * images + 7) / 8 * 8 * (p.Cout_pad / (16 * mw))); // XCD-aware 1-D order: see the kernel
    if (mw == 8)
        return launch_wino_case<8, false>(p, grid, tiles_x, tiles_y, vh, s);
    if (nc == 1)
        return launch_wino_case<4, true, 1>(p, grid, tiles_x, tiles_y, vh, s);
    return winograd_pipe() ? launch_wino_case<4, true>(p, grid, tiles_x, tiles_y, vh, s) : launch_wino_case<4, false>(p, grid, tiles_x, tiles_y, vh, s);
}

hipError_t conv32_winograd_occupancy(const conv32_params& p, int* blocks_per_cu)
{
    if (winograd_mw(p) == 8)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, conv32_winograd_kernel<8, false>, 512, wino_geom<8>::LDS_BYTES);
    if (winograd_pipe())
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, conv32_winograd_kernel<4, true>, 256, wino_geom<4>::PLDS_BYTES);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, conv32_winograd_kernel<4, false>, 256, wino_geom<4>::LDS_BYTES);
}

} // namespace hp
