// paf_parser.hip — hyperpose::parser::paf on gfx950 (replaces reference src/paf.cpp + src/post_process.hpp).
//
// The reference runs, per frame and on ONE CPU thread: 57 cv::resize calls (post_process.hpp:47-51),
// 19 GaussianBlur calls (:61-68), a scalar 3x3 max-pool (:71-102), a full scan for peaks (:184-192), the
// line-integral scoring of every peak pair of every limb (paf.cpp:93-144), a sort + greedy assignment
// per limb (:234-272) and the sequential skeleton assembly (:146-232), materialising ~15 MB of
// up-sampled / smoothed / pooled temporaries on the way.
//
// Here a whole batch of frames is parsed by four launches that never materialise the up-sampled maps:
//   1. paf_peaks_kernel    grid (bands x strips, 18 parts, frames): one thread per column marches down the
//                          rows: up-sample (INTER_AREA semantics) -> 17-tap row filter through LDS ->
//                          symmetric column filter out of a register window -> 3x3 max / threshold on a
//                          4-row LDS ring; append peaks with an atomic.
//   2. paf_sort_kernel     grid (18 parts, frames): rank-sort each part's peaks into the reference's scan
//                          order so that peak ids equal the reference's running index.
//   3. paf_limbs_kernel    grid (19 limbs, frames): the limb's two PAF source channels live in LDS; one
//                          thread per (peak a, peak b) pair evaluates the 10-sample line integral by
//                          interpolating the un-up-sampled PAF on the fly; rank-sort of the surviving
//                          candidates and a wave-ballot greedy bipartite assignment.
//   4. paf_assemble_kernel grid (frames): one wavefront walks the connections in reference order; the
//                          "which humans touch this connection" scan is a ballot over up to 256 humans.
// Only `hp_human`s (<= 292 B each) come back over PCIe.
//
// Every floating-point expression keeps the operand order and the (non-fused) rounding of the CPU
// code it replaces; this translation unit is compiled with -ffp-contract=off.  See DESIGN.md.
#include "hp_common.hpp"

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

namespace {

constexpr int KSIZE = 17; // paf.cpp:331 (peak_finder ksize)
constexpr int KR = KSIZE / 2;
constexpr int PEAK_THREADS = 256;
constexpr int PEAK_WCOLS = 46;                 // columns a wavefront of the peaks kernel owns (lanes 9 .. 54)
constexpr int PEAK_BCOLS = 4 * PEAK_WCOLS;       // ... and a block
constexpr int MAXH = 1024;         // humans (skeleton fragments, alive or merged) in flight per frame in the assembly kernel: 80 KB of its 149 KB of LDS (512 until round 5)
constexpr int THRESH_VECTOR_CNT1 = 8; // paf.cpp:57
constexpr int THRESH_PART_CNT = 4;    // paf.cpp:58
constexpr float THRESH_HUMAN_SCORE = 0.4; // paf.cpp:59
constexpr int STEP_PAF = 10;          // paf.cpp:60

// src/coco.hpp:10-30 / :32-51
__constant__ int c_pairs_net[19][2] = {
    { 12, 13 }, { 20, 21 }, { 14, 15 }, { 16, 17 }, { 22, 23 }, { 24, 25 }, { 0, 1 }, { 2, 3 },
    { 4, 5 }, { 6, 7 }, { 8, 9 }, { 10, 11 }, { 28, 29 }, { 30, 31 }, { 34, 35 }, { 32, 33 },
    { 36, 37 }, { 18, 19 }, { 26, 27 },
};
constexpr int K_PAIRS[19][2] = { // (the same table for compile-time part indices)
    { 1, 2 }, { 1, 5 }, { 2, 3 }, { 3, 4 }, { 5, 6 }, { 6, 7 }, { 1, 8 }, { 8, 9 }, { 9, 10 },
    { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 0 }, { 0, 14 }, { 14, 16 }, { 0, 15 }, { 15, 17 },
    { 2, 16 }, { 5, 17 },
};
__constant__ int c_pairs[19][2] = {
    { 1, 2 }, { 1, 5 }, { 2, 3 }, { 3, 4 }, { 5, 6 }, { 6, 7 }, { 1, 8 }, { 8, 9 }, { 9, 10 },
    { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 0 }, { 0, 14 }, { 14, 16 }, { 0, 15 }, { 15, 17 },
    { 2, 16 }, { 5, 17 },
};

struct gauss_t {
    float k[KSIZE];
};

// Geometry + interpolation tables of one parser instance (device pointers into one table buffer).
struct geom_t {
    int J, L2;      // channels of conf / paf
    int R, Cc;      // source rows / cols (memory order [ch][R][Cc])
    int UH, UW;     // up-sampled rows / cols  (m_resolution_size.height / .width)
    int vmax_x;     // first dx that uses the 1-tap copy (OpenCV's xmax)
    int feat_height; // m_feature_size.height == Cc (paf.cpp:329)
    const int* ofs_x;  // [UW]
    const float* c0_x; // [UW]
    const float* c1_x; // [UW]
    const int* ofs_y0; // [UH]
    const int* ofs_y1; // [UH]  min(sy+1, R-1)
    const float* c0_y; // [UH]
    const float* c1_y; // [UH]
};

struct dpeak {
    int x, y;
    float score;
    int lin; // y * UW + x: the reference's scan position inside the channel
};

struct dconn {
    int cid1, cid2;
    float score;
};

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// One up-sampled sample = HResizeLinear on the two source rows, then VResizeLinear (see oracle/paf_oracle.cpp
// for the OpenCV derivation).  `src` points at a [rows][Cc] plane whose row 0 is source row `row_base`.
__device__ __forceinline__ float up_at(const float* __restrict__ src, int row_base, const geom_t& g, int y, int x)
{
    const int sx = g.ofs_x[x];
    const int r0 = g.ofs_y0[y] - row_base, r1 = g.ofs_y1[y] - row_base;
    float h0, h1;
    if (x < g.vmax_x) {
        const float a0 = g.c0_x[x], a1 = g.c1_x[x];
        h0 = src[r0 * g.Cc + sx] * a0 + src[r0 * g.Cc + sx + 1] * a1;
        h1 = src[r1 * g.Cc + sx] * a0 + src[r1 * g.Cc + sx + 1] * a1;
    } else {
        h0 = src[r0 * g.Cc + sx] * 1.f;
        h1 = src[r1 * g.Cc + sx] * 1.f;
    }
    return h0 * g.c0_y[y] + h1 * g.c1_y[y];
}

// ---------------------------------------------------------------------------------------------------
// 1. peaks: resize_area + smooth + same_max_pool_3x3 + find_peak_coords (post_process.hpp:26-195), fused.
// One block = one (frame, part, band of BH up-sampled rows, strip of CW <= 184 columns).  One thread = one column; the four wavefronts of
// a block are INDEPENDENT: wavefront w owns the 46 columns x0 + 46 w .. + 45 (lanes 9 .. 54) plus 9 halo columns on either side, marches
// down the rows TWO at a time, and keeps everything but the source rows in registers:
//   * the 17-tap row filter takes its neighbours' samples with v_mov_b32_dpp wave_shr:1 / wave_shl:1 (sixteen shifts of the lane's
//     (row m, row m+1) pair) instead of an LDS exchange - the kernel used to move 224 B of LDS per thread and step and was bound by the
//     128 B/clk of the LDS, now 64 B (the up-sampling reads of the staged source rows), and there is no barrier inside the march;
//   * the column filter runs out of the 18-row register window of row-filtered values as before;
//   * the last four smoothed rows stay in registers, the 3x3 maximum test reads the left / right neighbours with two more shifts.
// The two rows of a step sit in the two halves of packed-fp32 registers (v_pk_mul_f32 / v_pk_add_f32: the same IEEE roundings as the
// scalar forms, no fusion); every value is computed by the same expression in the same order as before - only the data movement changed.
// BORDER_REFLECT_101 is handled by evaluating the reflected row / column itself (the same value the CPU code reads).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool DUMP>
__global__ __launch_bounds__(PEAK_THREADS) void paf_peaks_kernel(const float* __restrict__ conf, geom_t g,
    gauss_t gk, float thresh, int BH, int CW, int strips, int src_rows_cap,
    dpeak* __restrict__ plist, int* __restrict__ pcount, int peak_cap,
    float* __restrict__ dump_up, float* __restrict__ dump_smooth)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int band = blockIdx.x / strips, strip = blockIdx.x % strips;
    const int k = blockIdx.y, f = blockIdx.z;

    const int y0 = band * BH, y1 = min(g.UH, y0 + BH);
    const int x0 = strip * CW, x1 = min(g.UW, x0 + CW);
    // rows really touched (reflections of the rows above / below the map fall inside this range)
    const int ry_lo = max(0, y0 - KR - 1), ry_hi = min(g.UH, y1 + KR + 1);
    const int row_base = g.ofs_y0[ry_lo];
    const int nrows = g.ofs_y1[ry_hi - 1] - row_base + 1;

    // LDS: the source rows of the band and, per logical row, its two source rows and vertical coefficients (read-only after staging)
    float* s_src = smem; // [src_rows_cap][Cc]
    const int tab_n = BH + 2 * (KR + 1) + 6;
    int* s_r0 = reinterpret_cast<int*>(s_src + ((src_rows_cap * g.Cc + 1) & ~1));
    int* s_r1 = s_r0 + tab_n;
    float* s_cy0 = reinterpret_cast<float*>(s_r1 + tab_n);
    float* s_cy1 = s_cy0 + tab_n;
    for (int i = tid; i < tab_n; i += PEAK_THREADS) {
        const int ry = reflect101(y0 - KR - 1 + i, g.UH);
        s_r0[i] = min(max(g.ofs_y0[ry] - row_base, 0), nrows - 1) * g.Cc; // clamped: rows past the band's needs are never used
        s_r1[i] = min(max(g.ofs_y1[ry] - row_base, 0), nrows - 1) * g.Cc;
        s_cy0[i] = g.c0_y[ry];
        s_cy1[i] = g.c1_y[ry];
    }
    const float* plane = conf + ((size_t)f * g.J + k) * g.R * g.Cc;
    float vmax = 0.f;
    for (int base = 0; base < nrows * g.Cc; base += 4 * PEAK_THREADS) { // four requests per thread in flight: one round trip per 1024 floats
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            v[q] = plane[row_base * g.Cc + min(base + q * PEAK_THREADS + tid, nrows * g.Cc - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = base + q * PEAK_THREADS + tid;
            if (i < nrows * g.Cc) {
                s_src[i] = v[q];
                vmax = fmaxf(vmax, v[q]); // NaN-safe for the purpose: a NaN source keeps the band alive below
                if (!(v[q] == v[q]))
                    vmax = __builtin_huge_valf();
            }
        }
    }

    // lane l of wavefront w owns column x0 + 46 w - 9 + l; columns outside the map read their reflection (far-out halo lanes of a
    // narrow map are clamped first: their values are never used)
    const int lane = tid & 63, wv = tid >> 6;
    const int ux = x0 + PEAK_WCOLS * wv - (KR + 1) + lane;
    const bool interior = lane >= KR + 1 && lane < KR + 1 + PEAK_WCOLS && ux < x1;
    const int rc = reflect101(min(max(ux, -(g.UW - 1)), 2 * g.UW - 2), g.UW);
    const int sx = g.ofs_x[rc];
    const bool two_tap = rc < g.vmax_x;
    // single-tap columns (x >= OpenCV's xmax) multiply by 1.f and add nothing: up_row keeps that form
    const float a0 = two_tap ? g.c0_x[rc] : 1.f, a1 = two_tap ? g.c1_x[rc] : 0.f;
    const int sx1 = two_tap ? sx + 1 : sx;
    // Every smoothed value of this band is a chain of convex combinations of the source rows staged above (interpolation
    // weights c0 + c1 = 1 and the Gaussian taps sum to 1, each within a few ulp; ~40 roundings of 2^-24 relative), so it
    // cannot exceed max(0, max source) * (1 + 1e-5).  A band whose sources all stay below thresh / 1.001 therefore has no
    // peak (the test is `smoothed > thresh`): skip it.  Real heat-maps are empty almost everywhere.
    const int alive = __syncthreads_or(vmax * 1.001f >= thresh); // (also the barrier that makes the staged rows visible to every wavefront)
    if (!DUMP && alive == 0)
        return;
    // The same bound per wavefront: its 46 columns' smoothed values and their 3 x 3 neighbourhoods are made of the samples of its own 64
    // lanes (9 halo columns on either side = the 8-column filter radius + 1), i.e. of the source columns sx / sx1 of those lanes over the
    // staged rows.  A strip of four wavefronts usually has people under one or two of them only.  (No barrier follows: a wavefront may leave.)
    if (!DUMP) {
        float wmax = 0.f;
        for (int r = 0; r < nrows; ++r) {
            const float v0 = s_src[r * g.Cc + sx], v1 = s_src[r * g.Cc + sx1];
            wmax = fmaxf(wmax, fmaxf(v0, v1));
            if (!(v0 == v0) || !(v1 == v1))
                wmax = __builtin_huge_valf(); // a NaN source keeps the wavefront alive
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
            wmax = fmaxf(wmax, __shfl_xor(wmax, d));
        if (!(wmax * 1.001f >= thresh))
            return;
    }

    // w[j] = (R[m-16+j], R[m-15+j]): overlapping pairs of the row-filtered column, logical rows m-16 .. m+1
    f32x2 w[KSIZE];
#pragma unroll
    for (int i = 0; i < KSIZE; ++i)
        w[i] = f32x2{ 0.f, 0.f };
    float sm0 = 0.f, sm1 = 0.f; // smoothed rows ys - 2, ys - 1 of this column (the two rows below them are this step's)

    auto up_row = [&](int i) { // row m_begin + i: HResizeLinear on the two source rows, then VResizeLinear
        const int r0 = s_r0[i], r1 = s_r1[i];
        // branch-free, same values: single-tap columns have a0 = 1, so t0 = S*1.f is already their result; the sum is
        // only selected for two-tap columns
        const float t0 = s_src[r0 + sx] * a0, t1 = s_src[r1 + sx] * a0;
        const float u0 = s_src[r0 + sx1] * a1, u1 = s_src[r1 + sx1] * a1;
        const float h0 = two_tap ? t0 + u0 : t0, h1 = two_tap ? t1 + u1 : t1;
        return h0 * s_cy0[i] + h1 * s_cy1[i];
    };
    auto shr1 = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)); }; // lane l <- l - 1
    auto shl1 = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false)); }; // lane l <- l + 1

    const int m_begin = y0 - KR - 1;
    const int m_end = y1 + KR; // the step holding m = m_end (or m_end - 1) tests row y1 - 1 at the latest
    const bool emit = interior && k < HP_COCO_N_PARTS;
    int it = 0;
    for (int m = m_begin; m <= m_end; m += 2, ++it) {
        // ---- a. up-sample this thread's samples of (reflected) rows m, m+1
        f32x2 u;
        u[0] = up_row(2 * it);
        u[1] = up_row(2 * it + 1);
        if (DUMP && dump_up && interior) {
            if (m >= y0 && m < y1)
                dump_up[(((size_t)f * g.J + k) * g.UH + m) * g.UW + ux] = u[0];
            if (m + 1 >= y0 && m + 1 < y1)
                dump_up[(((size_t)f * g.J + k) * g.UH + m + 1) * g.UW + ux] = u[1];
        }

        // ---- b. RowFilter<float,float> on both rows: s = k0*S[0]; s += kk*S[kk], S[t] = the sample of column ux - 8 + t
#pragma unroll
        for (int i = 0; i + 2 < KSIZE; ++i)
            w[i] = w[i + 2];
        {
            f32x2 left[KR]; // left[d - 1] = the pair of lane - d
            f32x2 x = u;
#pragma unroll
            for (int d = 0; d < KR; ++d) {
                x = f32x2{ shr1(x[0]), shr1(x[1]) };
                left[d] = x;
            }
            f32x2 s = gk.k[0] * left[KR - 1];
#pragma unroll
            for (int t = 1; t < KR; ++t)
                s += gk.k[t] * left[KR - 1 - t];
            s += gk.k[KR] * u;
            x = u;
#pragma unroll
            for (int d = 1; d <= KR; ++d) {
                x = f32x2{ shl1(x[0]), shl1(x[1]) };
                s += gk.k[KR + d] * x;
            }
            w[KSIZE - 2] = f32x2{ w[KSIZE - 3][1], s[0] };
            w[KSIZE - 1] = s;
        }

        // ---- c. SymmColumnFilter<float> for rows ys = m - KR, ys + 1: s = k8*C + 0; s += kk*(S[+kk] + S[-kk])
        const int ys = m - KR;
        f32x2 sm;
        {
            f32x2 s = gk.k[KR] * w[KR] + 0.f;
#pragma unroll
            for (int t = 1; t <= KR; ++t)
                s += gk.k[KR + t] * (w[KR + t] + w[KR - t]);
            sm = s;
            if (DUMP && dump_smooth && interior) {
                if (ys >= y0 && ys < y1)
                    dump_smooth[(((size_t)f * g.J + k) * g.UH + ys) * g.UW + ux] = s[0];
                if (ys + 1 >= y0 && ys + 1 < y1)
                    dump_smooth[(((size_t)f * g.J + k) * g.UH + ys + 1) * g.UW + ux] = s[1];
            }
        }

        // ---- d. peaks of rows ys - 1 and ys: their 3x3 neighbourhoods are the smoothed rows ys - 2 .. ys + 1 of this column and of the
        // two neighbouring lanes
        {
            const float r[4] = { sm0, sm1, sm[0], sm[1] }; // rows ys - 2 .. ys + 1
            float rl[4], rr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                rl[q] = shr1(r[q]), rr[q] = shl1(r[q]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int yn = ys - 1 + q;
                const float v = r[1 + q];
                if (emit && v > thresh && yn >= y0 && yn < y1) {
                    // same_max_pool_3x3_2d: max over in-range taps == v  <=>  no in-range tap exceeds v
                    bool is_max = true;
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int ny = yn + dy;
                        if (ny < 0 || ny >= g.UH)
                            continue;
                        if (ux - 1 >= 0)
                            is_max = is_max && !(rl[1 + q + dy] > v);
                        is_max = is_max && !(r[1 + q + dy] > v);
                        if (ux + 1 < g.UW)
                            is_max = is_max && !(rr[1 + q + dy] > v);
                    }
                    if (is_max) {
                        const int pos = atomicAdd(&pcount[f * HP_COCO_N_PARTS + k], 1);
                        if (pos < peak_cap) {
                            dpeak p;
                            p.x = ux;
                            p.y = yn;
                            p.score = up_at(s_src, row_base, g, yn, ux); // raw up-sampled value (post_process.hpp:180)
                            p.lin = yn * g.UW + ux;
                            plist[((size_t)f * HP_COCO_N_PARTS + k) * peak_cap + pos] = p;
                        }
                    }
                }
            }
            sm0 = sm[0], sm1 = sm[1];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2. order every part's peaks by scan position (row-major), i.e. the reference's push_back order.
__global__ __launch_bounds__(256) void paf_sort_kernel(const dpeak* __restrict__ plist, const int* __restrict__ pcount,
    int peak_cap, dpeak* __restrict__ sorted)
{
    extern __shared__ __attribute__((aligned(16))) int s_lin[];
    const int k = blockIdx.x, f = blockIdx.y;
    const int n = min(pcount[f * HP_COCO_N_PARTS + k], peak_cap);
    const dpeak* in = plist + ((size_t)f * HP_COCO_N_PARTS + k) * peak_cap;
    dpeak* out = sorted + ((size_t)f * HP_COCO_N_PARTS + k) * peak_cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        s_lin[i] = in[i].lin;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int me = s_lin[i];
        int rank = 0;
        for (int j = 0; j < n; ++j)
            rank += (s_lin[j] < me);
        out[rank] = in[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// 3. get_connection_candidates + get_connections (paf.cpp:93-144, :234-272) for one limb of one frame.
struct cand_t {
    float score;
    int ab;  // (a << 16) | b, indices inside the two part lists
    int seq; // a * n2 + b: generation order (tie-break for equal scores)
};

// libstdc++'s std::sort (bits/stl_algo.h: __introsort_loop with median-of-three __unguarded_partition_pivot down to 16 elements, then
// __final_insertion_sort) restated on an index array, comparator std::greater<connection_candidate> = by score (src/paf.cpp:47-50).  The
// standard leaves the order of equal elements open; the reference's results depend on this implementation's choice, so it is
// reproduced step by step (the sequence of comparisons and swaps is a pure function of the scores).  When the depth limit
// (2 * floor(log2 n)) is exhausted libstdc++ heap-sorts the range it is looking at (`std::__partial_sort(first, last, last)` =
// __heap_select + __sort_heap: __make_heap, then __pop_heap down to one element, both through __adjust_heap / __push_heap); that is
// restated too (`used_heap` reports that it ran).  Returns false only if the explicit stack overflowed (impossible: depth <= 2 log2 n).
__device__ void libstdcxx_adjust_heap(int* v, const cand_t* c, int first, int hole, int len, int value)
{
    // bits/stl_heap.h __adjust_heap(first, holeIndex, len, value, comp) followed by __push_heap; comp(a, b) = a.score > b.score
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (c[v[first + child]].score > c[v[first + child - 1]].score)
            --child;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && c[v[first + parent]].score > c[value].score) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

__device__ void libstdcxx_heap_sort_greater(int* v, const cand_t* c, int first, int last)
{
    const int len = last - first;
    if (len >= 2) // __make_heap
        for (int parent = (len - 2) / 2;; --parent) {
            libstdcxx_adjust_heap(v, c, first, parent, len, v[first + parent]);
            if (parent == 0)
                break;
        }
    // __heap_select's scan over [middle, last) is empty (middle == last); __sort_heap:
    for (int l = last; l - first > 1;) {
        --l;
        const int value = v[l]; // __pop_heap(first, l, l)
        v[l] = v[first];
        libstdcxx_adjust_heap(v, c, first, 0, l - first, value);
    }
}

__device__ bool libstdcxx_sort_greater(int* v, int n, const cand_t* c, bool* used_heap = nullptr)
{
#define HP_GT(i, j) (c[v[i]].score > c[v[j]].score)
#define HP_SWAP(i, j)                                                                                             \
    {                                                                                                             \
        const int t_ = v[i];                                                                                      \
        v[i] = v[j];                                                                                              \
        v[j] = t_;                                                                                                \
    }
    if (n <= 1)
        return true;
    bool ok = true;
    int lg = 0;
    while ((2 << lg) <= n)
        ++lg;
    // __introsort_loop(first, last, depth): `while (last - first > 16) { ...; __introsort_loop(cut, last, depth); last = cut; }` with
    // the recursion on the RIGHT part first: an explicit stack of (first, last, depth) reproduces the same sequence of partitions
    int stk_f[64], stk_l[64], stk_d[64], sp = 0;
    stk_f[0] = 0, stk_l[0] = n, stk_d[0] = 2 * lg, sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_f[sp], last = stk_l[sp], depth = stk_d[sp];
        // iterative form of the loop: every partition pushes the LEFT remainder to be continued after the right recursion returns
        while (last - first > 16) {
            if (depth == 0) {
                libstdcxx_heap_sort_greater(v, c, first, last);
                if (used_heap)
                    *used_heap = true;
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            { // __move_median_to_first(result = first, a = first + 1, b = mid, c = last - 1)
                const int a = first + 1, b = mid, cc = last - 1;
                if (HP_GT(a, b)) {
                    if (HP_GT(b, cc))
                        HP_SWAP(first, b)
                    else if (HP_GT(a, cc))
                        HP_SWAP(first, cc)
                    else
                        HP_SWAP(first, a)
                } else if (HP_GT(a, cc))
                    HP_SWAP(first, a)
                else if (HP_GT(b, cc))
                    HP_SWAP(first, cc)
                else
                    HP_SWAP(first, b)
            }
            int lo = first + 1, hi = last; // __unguarded_partition(first + 1, last, pivot = first)
            for (;;) {
                while (HP_GT(lo, first))
                    ++lo;
                --hi;
                while (HP_GT(first, hi))
                    --hi;
                if (!(lo < hi))
                    break;
                HP_SWAP(lo, hi)
                ++lo;
            }
            const int cut = lo;
            // recursion on [cut, last) happens NOW in libstdc++, the loop then continues with [first, cut): push the continuation
            // first (it is popped after the right part and everything below it is done), then descend into the right part
            if (sp < 63) {
                stk_f[sp] = first, stk_l[sp] = cut, stk_d[sp] = depth, ++sp;
            } else
                ok = false;
            first = cut;
        }
    }
    // __final_insertion_sort: guarded insertion sort of the first 16, unguarded linear inserts for the rest
    const int head = n > 16 ? 16 : n;
    for (int i = 1; i < head; ++i) {
        const int val = v[i];
        if (c[val].score > c[v[0]].score) {
            for (int k = i; k > 0; --k)
                v[k] = v[k - 1];
            v[0] = val;
        } else {
            int k = i;
            while (c[val].score > c[v[k - 1]].score) {
                v[k] = v[k - 1];
                --k;
            }
            v[k] = val;
        }
    }
    for (int i = head; i < n; ++i) {
        const int val = v[i];
        int k = i;
        while (k > 0 && c[val].score > c[v[k - 1]].score) { // (k > 0 never decides after a completed introsort loop; kept as a guard)
            v[k] = v[k - 1];
            --k;
        }
        v[k] = val;
    }
#undef HP_GT
#undef HP_SWAP
    return ok;
}

// hp_paf_debug_sort: the restated std::sort alone, on an arbitrary score sequence in generation order (tests: median-of-three
// killers that drive libstdc++ into its heap-sort fallback, mass ties) - compared with the host's real std::sort.
__global__ void paf_debug_sort_kernel(const float* __restrict__ scores, int n, cand_t* __restrict__ cand, int* __restrict__ order,
    int* __restrict__ flag)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        cand[i].score = scores[i];
        cand[i].ab = i;
        cand[i].seq = i;
        order[i] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool heap = false;
        const bool ok = libstdcxx_sort_greater(order, n, cand, &heap);
        *flag = (heap ? 1 : 0) | (ok ? 0 : 2);
    }
}

// LIMB_THREADS threads per (limb, frame): the pair loop is the kernel's time, a block is alone on its CU (19 x frames blocks), and what a frame of
// dense maps costs is its LARGEST limb (a part with 300 maxima on both ends is 90 000 pairs): sixteen wavefronts - four per SIMD - walk the pairs
// four times as wide and cover each other's LDS / division latencies (round 6; four wavefronts before)
constexpr int LIMB_THREADS = 1024;
__global__ __launch_bounds__(LIMB_THREADS) void paf_limbs_kernel(const float* __restrict__ paf, geom_t g, float paf_thresh,
    const dpeak* __restrict__ sorted, const int* __restrict__ pcount, int peak_cap, int cand_cap,
    dconn* __restrict__ conns, int* __restrict__ conn_count, int* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_ncand;
    const int tid = threadIdx.x;
    const int pair_id = blockIdx.x, f = blockIdx.y;
    const int p1 = c_pairs[pair_id][0], p2 = c_pairs[pair_id][1];
    const int ch1 = c_pairs_net[pair_id][0], ch2 = c_pairs_net[pair_id][1];

    int start1 = 0, start2 = 0;
    for (int c = 0; c < HP_COCO_N_PARTS; ++c) {
        const int nc = min(pcount[f * HP_COCO_N_PARTS + c], peak_cap);
        if (c < p1)
            start1 += nc;
        if (c < p2)
            start2 += nc;
    }
    const int n1 = min(pcount[f * HP_COCO_N_PARTS + p1], peak_cap);
    const int n2 = min(pcount[f * HP_COCO_N_PARTS + p2], peak_cap);

    const int plane = g.R * g.Cc;
    float* s_px = smem;            // PAF x-channel of this limb, [R][Cc]
    float* s_py = smem + plane;    // PAF y-channel
    cand_t* s_cand = reinterpret_cast<cand_t*>(smem + 2 * plane); // [cand_cap]
    int* s_order = reinterpret_cast<int*>(s_cand + cand_cap);     // [cand_cap] sorted position -> candidate
    // the up-sampling tables of geom_t, staged once per block (round 6): every sample of every candidate pair used to read seven of them
    // from global memory - 140 dependent L1 / L2 round trips per pair; a frame of dense maps (the network's own output under random weights:
    // ~55 peaks per part, ~2 900 pairs per limb) spent 259 us here against 12 us on a frame with people.  Same values, same expressions.
    int* s_ofs_x = s_order + cand_cap;                 // [UW]
    float* s_c0_x = reinterpret_cast<float*>(s_ofs_x + g.UW), *s_c1_x = s_c0_x + g.UW;
    int* s_ofs_y0 = reinterpret_cast<int*>(s_c1_x + g.UW), *s_ofs_y1 = s_ofs_y0 + g.UH; // [UH]
    float* s_c0_y = reinterpret_cast<float*>(s_ofs_y1 + g.UH), *s_c1_y = s_c0_y + g.UH;
    if (n1 > 0 && n2 > 0) {
        for (int i = tid; i < g.UW; i += LIMB_THREADS)
            s_ofs_x[i] = g.ofs_x[i], s_c0_x[i] = g.c0_x[i], s_c1_x[i] = g.c1_x[i];
        for (int i = tid; i < g.UH; i += LIMB_THREADS)
            s_ofs_y0[i] = g.ofs_y0[i], s_ofs_y1[i] = g.ofs_y1[i], s_c0_y[i] = g.c0_y[i], s_c1_y[i] = g.c1_y[i];
    }

    if (tid == 0)
        s_ncand = 0;
    if (n1 > 0 && n2 > 0) {
        // the two planes in as few memory round trips as possible: 16-byte loads, four per thread and plane in flight (the simple loop paid
        // one dependent round trip per 256 floats: most of this kernel's time)
        const float* src = paf + (size_t)f * g.L2 * plane;
        const float* sx = src + (size_t)ch1 * plane;
        const float* sy = src + (size_t)ch2 * plane;
        if (plane % 4 == 0 && ((size_t)paf & 15) == 0) {
            const int n4 = plane / 4;
            for (int base = 0; base < n4; base += 4 * LIMB_THREADS) {
                float4 vx[4], vy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = min(base + k * LIMB_THREADS + tid, n4 - 1);
                    vx[k] = reinterpret_cast<const float4*>(sx)[i];
                    vy[k] = reinterpret_cast<const float4*>(sy)[i];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = base + k * LIMB_THREADS + tid;
                    if (i < n4) {
                        reinterpret_cast<float4*>(s_px)[i] = vx[k];
                        reinterpret_cast<float4*>(s_py)[i] = vy[k];
                    }
                }
            }
        } else {
            for (int base = 0; base < plane; base += 4 * LIMB_THREADS) {
                float vx[4], vy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = min(base + k * LIMB_THREADS + tid, plane - 1);
                    vx[k] = sx[i], vy[k] = sy[i];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = base + k * LIMB_THREADS + tid;
                    if (i < plane)
                        s_px[i] = vx[k], s_py[i] = vy[k];
                }
            }
        }
    }
    __syncthreads();

    const dpeak* A = sorted + ((size_t)f * HP_COCO_N_PARTS + p1) * peak_cap;
    const dpeak* B = sorted + ((size_t)f * HP_COCO_N_PARTS + p2) * peak_cap;
    const int npairs = n1 * n2;
    for (int idx = tid; idx < npairs; idx += blockDim.x) {
        const int ia = idx / n2, ib = idx - ia * n2;
        const dpeak a = A[ia], b = B[ib];
        const int dx = b.x - a.x, dy = b.y - a.y;
        const float norm = sqrtf((float)(dx * dx + dy * dy)); // std::sqrt(int l2) narrowed to float, paf.cpp:104
        if (norm < 1e-12)
            continue;
        float vx = (float)dx, vy = (float)dy;
        vx /= norm;
        vy /= norm;
        const float step_x = (b.x - a.x) / float(STEP_PAF);
        const float step_y = (b.y - a.y) / float(STEP_PAF);
        float scores = 0.0f;
        int criterion1 = 0;
#pragma unroll
        for (int i = 0; i < STEP_PAF; ++i) {
            const int lx = static_cast<int>(a.x + i * step_x + 0.5); // roundpaf, paf.cpp:74
            const int ly = static_cast<int>(a.y + i * step_y + 0.5);
            // up_at(s_px, 0, g, ly, lx) and up_at(s_py, 0, g, ly, lx) with the tables in LDS and read once for both planes
            const int sxo = s_ofs_x[lx], r0 = s_ofs_y0[ly] * g.Cc + sxo, r1 = s_ofs_y1[ly] * g.Cc + sxo;
            const float cy0 = s_c0_y[ly], cy1 = s_c1_y[ly];
            float px, py;
            if (lx < g.vmax_x) {
                const float a0 = s_c0_x[lx], a1 = s_c1_x[lx];
                const float hx0 = s_px[r0] * a0 + s_px[r0 + 1] * a1, hx1 = s_px[r1] * a0 + s_px[r1 + 1] * a1;
                const float hy0 = s_py[r0] * a0 + s_py[r0 + 1] * a1, hy1 = s_py[r1] * a0 + s_py[r1 + 1] * a1;
                px = hx0 * cy0 + hx1 * cy1, py = hy0 * cy0 + hy1 * cy1;
            } else {
                const float hx0 = s_px[r0] * 1.f, hx1 = s_px[r1] * 1.f, hy0 = s_py[r0] * 1.f, hy1 = s_py[r1] * 1.f;
                px = hx0 * cy0 + hx1 * cy1, py = hy0 * cy0 + hy1 * cy1;
            }
            const float score = vx * px + vy * py;
            scores += score;
            if (score > paf_thresh)
                criterion1 += 1;
        }
        const float criterion2 = scores / STEP_PAF + fmin(0.0, 0.5 * g.feat_height / norm - 1.0); // paf.cpp:129
        if (criterion1 > THRESH_VECTOR_CNT1 && criterion2 > 0) {
            const int pos = atomicAdd(&s_ncand, 1);
            if (pos < cand_cap) {
                s_cand[pos].score = criterion2;
                s_cand[pos].ab = (ia << 16) | ib;
                s_cand[pos].seq = idx;
            }
        }
    }
    __syncthreads();
    int n = s_ncand;
    if (n > cand_cap) {
        if (tid == 0)
            atomicOr(flags + f, 2);
        n = cand_cap;
    }

    // std::sort(..., std::greater) (paf.cpp:249): rank by (score desc, generation order asc).  Without equal scores that IS the
    // result of any correct sort.  With equal scores the reference's order is whatever libstdc++'s std::sort leaves: up to 16
    // candidates it is a plain insertion sort (stable = generation order, what the ranks give); beyond that introsort's
    // partitioning decides, and the limb is re-sorted below by the same algorithm on one lane.
    __shared__ int s_ties;
    if (tid == 0)
        s_ties = 0;
    __syncthreads();
    bool tie = false;
    for (int i = tid; i < n; i += blockDim.x) {
        const float sc = s_cand[i].score;
        const int sq = s_cand[i].seq;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s_cand[j].score;
            rank += (sj > sc) || (sj == sc && s_cand[j].seq < sq);
            tie |= sj == sc && j != i;
        }
        s_order[rank] = i;
    }
    if (tie)
        s_ties = 1;
    __syncthreads();
    if (s_ties && n > 16) {
        // generation order first (the vector std::sort receives, paf.cpp:108-141), then libstdc++'s algorithm on it
        for (int i = tid; i < n; i += blockDim.x) {
            const int sq = s_cand[i].seq;
            int rank = 0;
            for (int j = 0; j < n; ++j)
                rank += s_cand[j].seq < sq;
            s_order[rank] = i;
        }
        __syncthreads();
        if (tid == 0 && !libstdcxx_sort_greater(s_order, n, s_cand))
            atomicOr(flags + f, 8); // (unreachable: the explicit stack of the restated introsort cannot overflow)
        __syncthreads();
    }

    // greedy assignment (paf.cpp:252-270) by wavefront 0: "used" bitmaps live in registers, one 32-bit word
    // per lane (covers 2048 peaks per part); conflicts inside a 64-candidate chunk are resolved leader by leader.
    if (tid < 64) {
        const int lane = tid;
        unsigned used1 = 0, used2 = 0;
        int nconn = 0;
        dconn* out = conns + ((size_t)f * HP_COCO_N_PAIRS + pair_id) * peak_cap;
        const int max_conn = min(n1, n2);
        for (int base = 0; base < n && nconn < max_conn; base += 64) {
            const int i = base + lane;
            const bool valid = i < n;
            int a = 0, b = 0;
            float sc = 0.f;
            if (valid) {
                const cand_t c = s_cand[s_order[i]];
                a = c.ab >> 16;
                b = c.ab & 0xffff;
                sc = c.score;
            }
            const unsigned w1 = __shfl(used1, a >> 5), w2 = __shfl(used2, b >> 5);
            bool alive = valid && !((w1 >> (a & 31)) & 1u) && !((w2 >> (b & 31)) & 1u);
            unsigned long long m;
            while ((m = __ballot(alive)) != 0ull) {
                const int leader = __ffsll((long long)m) - 1;
                const int la = __builtin_amdgcn_readlane(a, leader), lb = __builtin_amdgcn_readlane(b, leader); // (uniform leader)
                const float ls = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), leader));
                if (lane == 0) {
                    dconn c;
                    c.cid1 = start1 + la;
                    c.cid2 = start2 + lb;
                    c.score = ls;
                    out[nconn] = c;
                }
                if (lane == (la >> 5))
                    used1 |= 1u << (la & 31);
                if (lane == (lb >> 5))
                    used2 |= 1u << (lb & 31);
                ++nconn;
                alive = alive && lane != leader && a != la && b != lb;
            }
        }
        if (lane == 0)
            conn_count[f * HP_COCO_N_PAIRS + pair_id] = nconn;
    }
}

// ---------------------------------------------------------------------------------------------------
// 4. get_humans (paf.cpp:146-232) + the human_t emission of paf::process (:359-372); one wavefront per frame.
__device__ __forceinline__ dpeak peak_by_id(const dpeak* __restrict__ sorted_f, const int* s_start, int peak_cap, int id)
{
    dpeak z;
    z.x = 0, z.y = 0, z.score = 0.f, z.lin = 0;
    if (id < 0 || id >= s_start[HP_COCO_N_PARTS])
        return z; // the reference would index out of bounds here (paf.cpp:193 can synthesise such ids)
    int c = 0;
    while (id >= s_start[c + 1])
        ++c;
    return sorted_f[(size_t)c * peak_cap + (id - s_start[c])];
}

// The walk over the connections is a sequential dependency chain; the dependent global loads inside it (the connection
// itself and two peak scores per step) are staged in LDS first - every connection of the frame in walk order and the
// score of every peak by id - so that the chain only touches LDS (126 -> 94 us per batch; the rest is LDS latency of
// the human tables: a register-resident variant was tried and drowned in code size / scratch).
constexpr int ASM_CONN_CAP = 2560;  // connections of one frame staged in LDS (30 KB); the rest is read from global
constexpr int ASM_PEAK_CAP = 4096;  // peak scores staged by id (16 KB)
// The results go straight to the (pinned, device-mapped) host buffers - `humans`, `n_humans` and `host_flags` are host
// pointers - so no copy kernels follow; and the block leaves this frame's peak counters and overflow flags zeroed for the
// next batch (it is their last reader), so no memset precedes the next one.
__global__ __launch_bounds__(64) void paf_assemble_kernel(const dpeak* __restrict__ sorted, int* __restrict__ pcount,
    int peak_cap, const dconn* __restrict__ conns, const int* __restrict__ conn_count, int res_w, int res_h,
    hp_human* __restrict__ humans, int* __restrict__ n_humans, int human_cap, int* __restrict__ flags, int* __restrict__ host_flags,
    int* __restrict__ pcount_last)
{
    __shared__ int s_parts[MAXH * HP_COCO_N_PARTS];
    __shared__ float s_score[MAXH];
    __shared__ int s_n[MAXH];
    __shared__ int s_start[HP_COCO_N_PARTS + 1];
    __shared__ int s_keep[MAXH];
    __shared__ int s_cstart[HP_COCO_N_PAIRS + 1];
    __shared__ dconn s_conn[ASM_CONN_CAP];
    __shared__ float s_pscore[ASM_PEAK_CAP];
    __shared__ int s_owner[ASM_PEAK_CAP];                 // peak id held at the current limb's first part -> human (parallel limbs)
    __shared__ int s_att_cid[64], s_new_c1[64], s_new_c2[64]; // hand-over between connection lanes and human lanes
    __shared__ float s_att_add[64], s_new_sc[64];
    const int lane = threadIdx.x;
    const int f = blockIdx.x;
    const dpeak* sorted_f = sorted + (size_t)f * HP_COCO_N_PARTS * peak_cap;
    for (int i = lane; i < ASM_PEAK_CAP; i += 64)
        s_owner[i] = -1;

    { // prefix sums of the per-part peak counts and the per-limb connection counts: one load per lane, a wave scan
        int np = 0, ncn = 0;
        if (lane < HP_COCO_N_PARTS) {
            const int raw = pcount[f * HP_COCO_N_PARTS + lane];
            if (raw > peak_cap)
                atomicOr(flags + f, 1);
            np = min(raw, peak_cap);
        }
        if (lane < HP_COCO_N_PAIRS)
            ncn = conn_count[f * HP_COCO_N_PAIRS + lane];
        int ip = np, ic = ncn; // inclusive scans
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int tp = __shfl_up(ip, off), tc = __shfl_up(ic, off);
            if (lane >= off)
                ip += tp, ic += tc;
        }
        if (lane <= HP_COCO_N_PARTS)
            s_start[lane] = ip - np; // exclusive; [N_PARTS] = the total (that lane's own count is 0)
        if (lane <= HP_COCO_N_PAIRS)
            s_cstart[lane] = ic - ncn;
    }
    __syncthreads();
    // stage: connections in walk order, peak scores by id.  One flat index space per table (every lane finds its limb / part from the
    // prefix sums with 18 independent LDS reads) and four requests per lane in flight: the 19 + 18 per-list loops this replaces paid a
    // dependent memory round trip each, more than the whole walk below.
    {
        const int total_c = min(s_cstart[HP_COCO_N_PAIRS], ASM_CONN_CAP);
        for (int base = 0; base < total_c; base += 256) {
            dconn v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(base + k * 64 + lane, total_c - 1);
                int l = 0;
#pragma unroll
                for (int q = 1; q < HP_COCO_N_PAIRS; ++q)
                    l += i >= s_cstart[q];
                v[k] = conns[((size_t)f * HP_COCO_N_PAIRS + l) * peak_cap + (i - s_cstart[l])];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (base + k * 64 + lane < total_c)
                    s_conn[base + k * 64 + lane] = v[k];
        }
        const int total_p = min(s_start[HP_COCO_N_PARTS], ASM_PEAK_CAP);
        for (int base = 0; base < total_p; base += 256) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(base + k * 64 + lane, total_p - 1);
                int c = 0;
#pragma unroll
                for (int q = 1; q < HP_COCO_N_PARTS; ++q)
                    c += i >= s_start[q];
                v[k] = sorted_f[(size_t)c * peak_cap + (i - s_start[c])].score;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (base + k * 64 + lane < total_p)
                    s_pscore[base + k * 64 + lane] = v[k];
        }
    }
    __syncthreads();
    auto peak_score = [&](int part, int id) { // all_peaks[id].score for an id of part `part`
        return id < ASM_PEAK_CAP ? s_pscore[id] : sorted_f[(size_t)part * peak_cap + (id - s_start[part])].score;
    };

    int nh = 0;
    bool overflow = false;
    // ---- the walk with the humans IN REGISTERS (human h = lane h: its 18 part ids, part count and score): the "which humans touch this
    // connection" test is two compares, an update is a v_cndmask, nothing on the chain goes through LDS.  The limb loop is unrolled so that
    // the part indices are register numbers.  Holds while at most 64 skeleton fragments are alive or merged (any real frame); a frame
    // that needs a 65th starts over on the LDS tables below.  Same operations in the same order: the results are those of the LDS walk.
    bool reg_done = false;
    {
        int parts[HP_COCO_N_PARTS];
#pragma unroll
        for (int r = 0; r < HP_COCO_N_PARTS; ++r)
            parts[r] = -1;
        int hn = 0;
        float hscore = 0.f;
        int rnh = 0;
        bool too_many = false;
        auto walk_pair = [&](auto PAIR) {
            constexpr int pair_id = decltype(PAIR)::value, p1 = K_PAIRS[pair_id][0], p2 = K_PAIRS[pair_id][1];
            const int cbase = s_cstart[pair_id], nc = s_cstart[pair_id + 1] - cbase;
            const dconn* cl = conns + ((size_t)f * HP_COCO_N_PAIRS + pair_id) * peak_cap;
            if (nc == 0 || too_many)
                return;
            // ---- the whole limb in parallel when its connections cannot influence each other: no human holds anything at part p2 yet
            // (then a connection can only be touched through p1; the connections of a limb have distinct peaks at either end, so an
            // attach or a new human never changes what a later connection of the same limb sees), and no two humans hold the same peak at
            // p1 (then every connection is touched by at most one human: no merge).  True for 15-17 of the 19 limbs of a real frame.
            // Connection c lives in lane c, human h in lane h; they meet through small LDS tables; new humans get their indices from a
            // prefix count in connection order, i.e. the indices the sequential walk would give them.
            bool par_ok;
            const int myid = lane < rnh ? parts[p1] : -1;
            {
                const bool p2_set = lane < rnh && parts[p2] != -1;
                const bool far = myid >= ASM_PEAK_CAP;
                par_ok = __ballot(p2_set | far) == 0ull;
                if (par_ok) {
                    if (myid >= 0)
                        s_owner[myid] = lane;
                    __builtin_amdgcn_wave_barrier();
                    const bool dup = myid >= 0 && s_owner[myid] != lane;
                    par_ok = __ballot(dup) == 0ull;
                }
            }
            for (int cb = 0; cb < nc && !too_many; cb += 64) {
                dconn mine{ 0, 0, 0.f };
                float my_sc1 = 0.f, my_sc2 = 0.f;
                if (cb + lane < nc) {
                    mine = cbase + cb + lane < ASM_CONN_CAP ? s_conn[cbase + cb + lane] : cl[cb + lane];
                    my_sc1 = peak_score(p1, mine.cid1), my_sc2 = peak_score(p2, mine.cid2);
                }
                const int nb = min(64, nc - cb);
                if (par_ok) {
                    s_att_cid[lane] = -2;
                    __builtin_amdgcn_wave_barrier();
                    const bool cv = lane < nb;
                    int h = -1;
                    if (cv && mine.cid1 >= 0 && mine.cid1 < ASM_PEAK_CAP)
                        h = s_owner[mine.cid1];
                    const bool isnew = cv && h < 0 && pair_id <= 16; // !is_virtual_pair (coco.hpp:6)
                    const unsigned long long nm = __ballot(isnew);
                    const int nnew = __popcll(nm);
                    if (rnh + nnew > 64) {
                        too_many = true;
                        break;
                    }
                    if (cv && h >= 0)
                        s_att_cid[h] = mine.cid2, s_att_add[h] = my_sc2 + mine.score;
                    if (isnew) {
                        const int k = __popcll(nm & ((1ull << lane) - 1ull));
                        s_new_c1[k] = mine.cid1, s_new_c2[k] = mine.cid2, s_new_sc[k] = my_sc1 + my_sc2 + mine.score;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rnh) {
                        const int a = s_att_cid[lane];
                        if (a != -2) { // humans[h].parts[p2] (still unset) != cid2: attach (paf.cpp:170-176)
                            parts[p2] = a;
                            ++hn;
                            hscore += s_att_add[lane];
                        }
                    } else if (lane < rnh + nnew) {
                        const int k = lane - rnh;
                        const int c1 = s_new_c1[k], c2 = s_new_c2[k];
#pragma unroll
                        for (int r = 0; r < HP_COCO_N_PARTS; ++r)
                            parts[r] = r == p1 ? c1 : (r == p2 ? c2 : -1);
                        hn = 2;
                        hscore = s_new_sc[k];
                    }
                    rnh += nnew;
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                int ci = 0;
                while (ci < nb) {
                    // fast loop: connections that exactly one human touches (the common case) as straight-line, predicated code with
                    // nothing else merging into it - a one-wavefront walk is issue-bound, every instruction and taken branch counts
                    for (; ci < nb; ++ci) {
                        const int cid1 = __builtin_amdgcn_readlane(mine.cid1, ci), cid2 = __builtin_amdgcn_readlane(mine.cid2, ci);
                        const bool in = lane < rnh;
                        const bool t2 = in & (parts[p2] == cid2);
                        const unsigned long long m = __ballot(t2 | (in & (parts[p1] == cid1)));
                        if (__popcll(m) != 1)
                            break;
                        const unsigned long long m2 = __ballot(t2);
                        const float add = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc2), ci))
                            + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.score), ci)); // sc2 + conn.score
                        const bool attach = (m2 == 0ull) & ((m >> lane) & 1ull); // the one human's parts[p2] != cid2 (paf.cpp:170-176)
                        parts[p2] = attach ? cid2 : parts[p2];
                        hn += attach ? 1 : 0;
                        hscore = attach ? hscore + add : hscore;
                    }
                    if (ci >= nb)
                        break;
                    // slow step: no human (a new one) or several (merge / attach to the first)
                    const int cid1 = __builtin_amdgcn_readlane(mine.cid1, ci), cid2 = __builtin_amdgcn_readlane(mine.cid2, ci);
                    const float cscore = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.score), ci));
                    const float sc1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc1), ci));
                    const float sc2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc2), ci));
                    ++ci;
                    const bool in = lane < rnh;
                    unsigned long long m = __ballot(in & ((parts[p2] == cid2) | (parts[p1] == cid1)));
                    const int total = __popcll(m);
                    if (total >= 2) {
                        const int first = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int second = __ffsll((long long)m) - 1;
                        bool membership = false;
#pragma unroll
                        for (int r = 0; r < HP_COCO_N_PARTS; ++r)
                            membership |= __builtin_amdgcn_readlane(parts[r], first) > 0 && __builtin_amdgcn_readlane(parts[r], second) > 0; // paf.cpp:185
                        if (!membership) {
#pragma unroll
                            for (int r = 0; r < HP_COCO_N_PARTS; ++r) {
                                const int other = __builtin_amdgcn_readlane(parts[r], second);
                                if (lane == first)
                                    parts[r] += other + 1; // paf.cpp:193
                                if (lane == second)
                                    parts[r] = -1; // erased (paf.cpp:202): never touches again
                            }
                            const int on = __builtin_amdgcn_readlane(hn, second);
                            const float os = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hscore), second));
                            if (lane == first) {
                                hn += on;
                                hscore += os;
                                hscore += cscore;
                            }
                            if (lane == second)
                                hn = -(1 << 20); // erased: fails the n_parts filter
                        } else if (lane == first) {
                            parts[p2] = cid2;
                            hn += 1;
                            hscore += sc2 + cscore;
                        }
                    } else if (total == 0 && pair_id <= 16) { // !is_virtual_pair (coco.hpp:6): a new human
                        if (rnh < 64) {
                            if (lane == rnh) {
#pragma unroll
                                for (int r = 0; r < HP_COCO_N_PARTS; ++r)
                                    parts[r] = r == p1 ? cid1 : (r == p2 ? cid2 : -1);
                                hn = 2;
                                hscore = sc1 + sc2 + cscore;
                            }
                            ++rnh;
                        } else {
                            too_many = true;
                            break;
                        }
                    }
                }
            }
            if (myid >= 0 && myid < ASM_PEAK_CAP) // leave the table clean for the next limb (also when the limb went the sequential way)
                s_owner[myid] = -1;
            __builtin_amdgcn_wave_barrier();
        };
#define HP_WP(K) walk_pair(std::integral_constant<int, K>{});
        HP_WP(0) HP_WP(1) HP_WP(2) HP_WP(3) HP_WP(4) HP_WP(5) HP_WP(6) HP_WP(7) HP_WP(8) HP_WP(9) HP_WP(10) HP_WP(11) HP_WP(12) HP_WP(13) HP_WP(14)
        HP_WP(15) HP_WP(16) HP_WP(17) HP_WP(18)
#undef HP_WP
        static_assert(HP_COCO_N_PAIRS == 19, "limb list");
        if (!too_many) { // hand the tables to the common filter / emission code
            if (lane < rnh) {
#pragma unroll
                for (int r = 0; r < HP_COCO_N_PARTS; ++r)
                    s_parts[lane * HP_COCO_N_PARTS + r] = parts[r];
                s_n[lane] = hn;
                s_score[lane] = hscore;
            }
            nh = rnh;
            reg_done = true;
        }
    }
    for (int pair_id = 0; pair_id < HP_COCO_N_PAIRS && !reg_done; ++pair_id) {
        const int p1 = c_pairs[pair_id][0], p2 = c_pairs[pair_id][1];
        const int cbase = s_cstart[pair_id], nc = s_cstart[pair_id + 1] - cbase;
        const dconn* cl = conns + ((size_t)f * HP_COCO_N_PAIRS + pair_id) * peak_cap;
        for (int cb = 0; cb < nc; cb += 64) {
            // 64 connections at a time: lane i fetches connection cb + i and the scores of its two peaks, so that the serial walk
            // below reads them with v_readlane instead of paying an LDS round trip per dependent access
            dconn mine{ 0, 0, 0.f };
            float my_sc1 = 0.f, my_sc2 = 0.f;
            if (cb + lane < nc) {
                mine = cbase + cb + lane < ASM_CONN_CAP ? s_conn[cbase + cb + lane] : cl[cb + lane];
                my_sc1 = peak_score(p1, mine.cid1), my_sc2 = peak_score(p2, mine.cid2); // all_peaks[cid].score
            }
            const int nb = min(64, nc - cb);
            for (int ci = 0; ci < nb; ++ci) {
                dconn conn;
                conn.cid1 = __builtin_amdgcn_readlane(mine.cid1, ci), conn.cid2 = __builtin_amdgcn_readlane(mine.cid2, ci);
                conn.score = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.score), ci));
                const float sc1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc1), ci));
                const float sc2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc2), ci));
                // which humans touch this connection (paf.cpp:164-168), lowest index first
                int total = 0, first = -1, second = -1;
                bool first_has_c2 = false;
#pragma unroll
                for (int s = 0; s < MAXH / 64; ++s) {
                    if (s * 64 >= nh) // uniform: no human lives in this slice yet
                        break;
                    const int h = s * 64 + lane;
                    const bool in = h < nh;
                    // both reads unconditional and together (`&&` / `||` made them two dependent, exec-masked LDS round trips)
                    const int hs = in ? h : 0;
                    const int part1 = s_parts[hs * HP_COCO_N_PARTS + p1], part2 = s_parts[hs * HP_COCO_N_PARTS + p2];
                    const bool t2 = in & (part2 == conn.cid2);
                    const bool t = t2 | (in & (part1 == conn.cid1));
                    unsigned long long m = __ballot(t);
                    const unsigned long long m2 = __ballot(t2);
                    total += __popcll(m);
                    if (m && first < 0) {
                        const int bit = __ffsll((long long)m) - 1;
                        first = s * 64 + bit;
                        first_has_c2 = (m2 >> bit) & 1ull;
                        m &= m - 1;
                    }
                    if (m && second < 0)
                        second = s * 64 + __ffsll((long long)m) - 1;
                }
                if (total == 1) {
                    if (lane == 0 && !first_has_c2) { // humans[first].parts[p2] != cid2
                        s_parts[first * HP_COCO_N_PARTS + p2] = conn.cid2;
                        // ds_add_u32 / ds_add_f32 without return: the same integer / IEEE single add as `+=`, but nothing waits for an
                        // LDS read on the walk's critical path
                        atomicAdd(&s_n[first], 1);
                        atomicAdd(&s_score[first], sc2 + conn.score);
                    }
                } else if (total >= 2) {
                    bool both = false;
                    if (lane < HP_COCO_N_PARTS)
                        both = s_parts[first * HP_COCO_N_PARTS + lane] > 0 && s_parts[second * HP_COCO_N_PARTS + lane] > 0; // paf.cpp:185
                    const bool membership = __ballot(both) != 0ull;
                    if (!membership) {
                        if (lane < HP_COCO_N_PARTS) {
                            s_parts[first * HP_COCO_N_PARTS + lane] += s_parts[second * HP_COCO_N_PARTS + lane] + 1; // paf.cpp:193
                            s_parts[second * HP_COCO_N_PARTS + lane] = -1; // erased (paf.cpp:202): never touches again
                        }
                        if (lane == 0) {
                            s_n[first] += s_n[second];
                            s_score[first] += s_score[second];
                            s_score[first] += conn.score;
                            s_n[second] = -(1 << 20); // erased: fails the n_parts filter
                        }
                    } else if (lane == 0) {
                        s_parts[first * HP_COCO_N_PARTS + p2] = conn.cid2;
                        s_n[first] += 1;
                        s_score[first] += sc2 + conn.score;
                    }
                } else if (pair_id <= 16) { // !is_virtual_pair, coco.hpp:6
                    if (nh < MAXH) {
                        if (lane < HP_COCO_N_PARTS)
                            s_parts[nh * HP_COCO_N_PARTS + lane] = lane == p1 ? conn.cid1 : (lane == p2 ? conn.cid2 : -1);
                        if (lane == 0) {
                            s_n[nh] = 2;
                            s_score[nh] = sc1 + sc2 + conn.score;
                        }
                        ++nh;
                    } else
                        overflow = true;
                }
                // one wavefront, DS operations retire in order: the next connection's reads see these writes without a barrier
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    if (overflow && lane == 0)
        atomicOr(flags + f, 4);

    // remove_if (paf.cpp:226-230), survivors keep their relative order
    int kept = 0;
#pragma unroll
    for (int s = 0; s < MAXH / 64; ++s) {
        const int h = s * 64 + lane;
        bool keep = false;
        if (h < nh) {
            const int np = s_n[h];
            keep = !(np < THRESH_PART_CNT || s_score[h] / np < THRESH_HUMAN_SCORE);
        }
        const unsigned long long m = __ballot(keep);
        if (keep)
            s_keep[kept + __popcll(m & ((1ull << lane) - 1ull))] = h;
        kept += __popcll(m);
    }
    __syncthreads();
    if (lane == 0)
        n_humans[f] = kept;
    // emission: (human, part) pairs spread over the wavefront
    hp_human* out = humans + (size_t)f * human_cap;
    const int nout = min(kept, human_cap);
    // four (human, part) items per lane at a time: their peaks are requested together (one memory round trip per 256 items instead of one
    // per 64), the part of each id is found with 17 independent LDS reads instead of a search loop
    const int total_ids = s_start[HP_COCO_N_PARTS];
    for (int base = 0; base < nout * HP_COCO_N_PARTS; base += 4 * 64) {
        int ids[4];
        dpeak pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = base + q * 64 + lane;
            ids[q] = -1;
            if (e < nout * HP_COCO_N_PARTS) {
                const int i = e / HP_COCO_N_PARTS, part = e - i * HP_COCO_N_PARTS;
                ids[q] = s_parts[s_keep[i] * HP_COCO_N_PARTS + part];
            }
            // peak_by_id: an id outside the table reads as a zero peak (the reference would index out of bounds, paf.cpp:193 can synthesise such ids)
            const bool valid = ids[q] >= 0 && ids[q] < total_ids;
            const int idc = valid ? ids[q] : 0;
            int c = 0;
#pragma unroll
            for (int k = 1; k < HP_COCO_N_PARTS; ++k)
                c += idc >= s_start[k];
            pk[q] = sorted_f[(size_t)c * peak_cap + (idc - s_start[c])]; // (an empty table: element 0 of part 0, allocated, unused)
            if (!valid)
                pk[q].x = 0, pk[q].y = 0, pk[q].score = 0.f, pk[q].lin = 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = base + q * 64 + lane;
            if (e >= nout * HP_COCO_N_PARTS)
                continue;
            const int i = e / HP_COCO_N_PARTS, part = e - i * HP_COCO_N_PARTS;
            hp_body_part bp;
            bp.has_value = 0;
            bp.x = 0.f, bp.y = 0.f, bp.score = 0.f;
            if (ids[q] != -1) {
                bp.has_value = 1;
                bp.score = pk[q].score;
                bp.x = static_cast<float>(pk[q].x) / res_w; // paf.cpp:367-368
                bp.y = static_cast<float>(pk[q].y) / res_h;
            }
            out[i].parts[part] = bp;
        }
    }
    for (int i = lane; i < nout; i += 64)
        out[i].score = s_score[s_keep[i]];
    __syncthreads();
    if (lane == 0)
        host_flags[f] = atomicExch(flags + f, 0);
    if (lane < HP_COCO_N_PARTS) {
        pcount_last[f * HP_COCO_N_PARTS + lane] = pcount[f * HP_COCO_N_PARTS + lane]; // kept for hp_paf_debug_peaks
        pcount[f * HP_COCO_N_PARTS + lane] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// resize tables on the host: identical arithmetic to oracle/paf_oracle.cpp::make_tab (OpenCV 4.4.0 resize.cpp).
struct host_tab {
    std::vector<int> ofs, ofs1;
    std::vector<float> c0, c1;
    int vmax;
};

host_tab make_tab(int ssize, int dsize)
{
    host_tab t;
    t.ofs.resize(dsize), t.ofs1.resize(dsize), t.c0.resize(dsize), t.c1.resize(dsize);
    t.vmax = dsize;
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        int s = (int)std::floor(d * scale);
        float f = (float)((d + 1) - (s + 1) * inv_scale);
        f = f <= 0 ? 0.f : f - (float)std::floor(f);
        if (s < 0)
            f = 0, s = 0;
        if (s + 1 >= ssize) {
            t.vmax = std::min(t.vmax, d);
            if (s >= ssize - 1)
                f = 0, s = ssize - 1;
        }
        t.ofs[d] = s;
        t.ofs1[d] = std::min(s + 1, ssize - 1);
        t.c0[d] = 1.f - f;
        t.c1[d] = f;
    }
    return t;
}

gauss_t make_gauss(double sigma)
{
    // OpenCV 4.4.0 getGaussianKernel(17, 3.0, CV_32F): double exp, normalised, narrowed to float
    gauss_t gk;
    double v[KSIZE];
    const int n2 = (KSIZE - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0;
    for (int i = 0, x = 1 - KSIZE; i < n2; ++i, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2x);
        sum += v[i];
    }
    sum = sum * 2 + 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; ++i)
        gk.k[i] = gk.k[KSIZE - 1 - i] = (float)(v[i] * mul1);
    gk.k[n2] = (float)(1.0 * mul1);
    return gk;
}

} // namespace

// =====================================================================================================
struct hp_paf {
    float conf_thresh, paf_thresh;
    int res_w, res_h;
    int max_batch;
    int peak_cap = 512;   // peaks per part per frame
    int cand_cap = 2048;  // candidates per limb per frame that survive the two criteria
    int human_cap = 128;  // humans per frame copied back

    bool shaped = false;
    geom_t g{};
    gauss_t gk{};
    int BH = 0, CW = 0, strips = 0, bands = 0, src_rows_cap = 0; // peaks kernel tiling
    size_t peaks_lds = 0, limbs_lds = 0;

    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hp::dev_buf tables, plist, sorted, pcount, pcount_last, conns, conn_count, flags, in_conf, in_paf;
    hp::host_buf h_humans, h_counts; // h_counts: [n_humans(max_batch) | flags(max_batch)], written by the assemble kernel
    int pending = 0; // frames of the enqueued, not yet collected batch
    int last_n = 0;  // frames of the last completed batch (debug taps)
    const float *last_conf = nullptr, *last_paf = nullptr; // inputs of the batch in flight (re-parsed with larger lists on overflow)

    int shape(const int conf_shape[3], const int paf_shape[3]);
    int alloc_lists();
    int launch(int n, const float* dev_conf, const float* dev_paf, hipStream_t s);
};

// The reference's lists are std::vectors (src/post_process.hpp:171-193, src/paf.cpp:108-141): they grow.  Here they start at sizes
// that fit every realistic frame (512 peaks per part, 2048 candidates per limb, 128 humans) and hp_paf_collect re-parses a batch
// with doubled lists when a frame overflowed one.  Hard limits (reported as HP_ERR_CAPACITY, results truncated): 2048 peaks per
// part (the greedy pass keeps its "used" sets in 64 x 32-bit lane registers), the candidates that fit the CU's LDS next to the two PAF
// planes (~8000 at 46x54), 1024 skeleton fragments alive or merged per frame (MAXH), 1024 humans returned.
constexpr int PEAK_CAP_MAX = 2048, HUMAN_CAP_MAX = 1024;
int hp_paf::alloc_lists()
{
    limbs_lds = (size_t)4 * 2 * g.R * g.Cc + (size_t)cand_cap * (sizeof(cand_t) + sizeof(int)) + (size_t)4 * (3 * g.UW + 4 * g.UH); // planes, candidates, up-sampling tables
    HP_REQUIRE(limbs_lds <= 160 * 1024, HP_ERR_INVALID, "paf: feature map too large for LDS tiling");
    HP_HIP_TRY(hipFuncSetAttribute((const void*)paf_limbs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)limbs_lds));
    const size_t B = max_batch;
    HP_TRY(plist.alloc(B * HP_COCO_N_PARTS * peak_cap * sizeof(dpeak)));
    HP_TRY(sorted.alloc(B * HP_COCO_N_PARTS * peak_cap * sizeof(dpeak)));
    HP_TRY(conns.alloc(B * HP_COCO_N_PAIRS * peak_cap * sizeof(dconn)));
    HP_TRY(h_humans.alloc(B * human_cap * sizeof(hp_human)));
    return HP_OK;
}

int hp_paf::shape(const int cs[3], const int ps[3])
{
    HP_REQUIRE(cs && ps, HP_ERR_INVALID, "paf: null shape");
    HP_REQUIRE(cs[0] > 0 && cs[1] > 0 && cs[2] > 0 && ps[0] > 0, HP_ERR_INVALID, "paf: bad shape");
    HP_REQUIRE(cs[1] == ps[1] && cs[2] == ps[2], HP_ERR_INVALID,
        "paf: conf [%d,%d,%d] and paf [%d,%d,%d] disagree (reference asserts, src/paf.cpp:318-319)", cs[0], cs[1], cs[2], ps[0], ps[1], ps[2]);
    if (shaped) {
        HP_REQUIRE(cs[0] == g.J && cs[1] == g.R && cs[2] == g.Cc && ps[0] == g.L2, HP_ERR_STATE,
            "paf: feature-map shape changed after the first call (the reference allocates once, src/paf.cpp:321-332)");
        return HP_OK;
    }
    HP_REQUIRE(cs[0] >= HP_COCO_N_PARTS && ps[0] >= 2 * HP_COCO_N_PAIRS, HP_ERR_INVALID,
        "paf: need >= 18 conf and >= 38 paf channels, got %d / %d", cs[0], ps[0]);
    g.J = cs[0], g.L2 = ps[0], g.R = cs[1], g.Cc = cs[2];
    // src/paf.cpp:311-315: `auto [n, fw, fh] = dims()` => fw = rows, fh = cols; default Size(fw*4, fh*4)
    const int fw = g.R, fh = g.Cc;
    if (res_w == -1 || res_h == -1)
        res_w = fw * 4, res_h = fh * 4;
    HP_REQUIRE(res_w > 0 && res_h > 0, HP_ERR_INVALID, "paf: bad resolution %dx%d", res_w, res_h);
    HP_REQUIRE(!(res_h == g.R && res_w == g.Cc), HP_ERR_INVALID,
        "paf: resolution == feature size: the reference's resize_area returns without writing (post_process.hpp:31-32)");
    HP_REQUIRE(res_w < 65536 / 2 && res_h < 65536 / 2, HP_ERR_INVALID, "paf: resolution too large");
    g.UH = res_h, g.UW = res_w;
    g.feat_height = fh; // m_feature_size = Size(fw, fh) -> .height (paf.cpp:329, :354)

    const host_tab tx = make_tab(g.Cc, g.UW), ty = make_tab(g.R, g.UH);
    g.vmax_x = tx.vmax;
    gk = make_gauss(3.0);
    // one table buffer: ofs_x | c0_x | c1_x | ofs_y0 | ofs_y1 | c0_y | c1_y
    std::vector<int> blob((size_t)3 * g.UW + 4 * g.UH);
    int* b = blob.data();
    memcpy(b, tx.ofs.data(), g.UW * 4), memcpy(b + g.UW, tx.c0.data(), g.UW * 4), memcpy(b + 2 * g.UW, tx.c1.data(), g.UW * 4);
    int* by = b + 3 * g.UW;
    memcpy(by, ty.ofs.data(), g.UH * 4), memcpy(by + g.UH, ty.ofs1.data(), g.UH * 4);
    memcpy(by + 2 * g.UH, ty.c0.data(), g.UH * 4), memcpy(by + 3 * g.UH, ty.c1.data(), g.UH * 4);
    HP_TRY(tables.alloc(blob.size() * 4));
    // (on this parser's own non-blocking stream: a plain hipMemcpy goes through the legacy stream, which another thread's graph capture -
    // the engine of the same pipeline recording its schedule - turns into an error for both)
    HP_HIP_TRY(hipMemcpyAsync(tables.p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice, stream));
    HP_HIP_TRY(hipStreamSynchronize(stream));
    const int* d = tables.as<int>();
    g.ofs_x = d, g.c0_x = (const float*)(d + g.UW), g.c1_x = (const float*)(d + 2 * g.UW);
    const int* dy = d + 3 * g.UW;
    g.ofs_y0 = dy, g.ofs_y1 = dy + g.UH, g.c0_y = (const float*)(dy + 2 * g.UH), g.c1_y = (const float*)(dy + 3 * g.UH);

    // tiling of the up-sampled map for the peaks kernel: strips of <= 184 columns (four wavefronts x 46 columns, each with its own
    // 9-column halo), bands of ~40 rows
    strips = hp::ceil_div(g.UW, PEAK_BCOLS);
    CW = hp::ceil_div(g.UW, strips);
    // (bands of ~40 rows: 18 halo rows are recomputed per band, but shorter bands are skipped more often - real maps are empty almost
    // everywhere - and 2.8 k wavefronts balance better over the 1024 SIMDs than 2.3 k: 56.6 -> 50.4 us per batch of 8 against 54-row bands)
    const int band_rows = 40;
    bands = std::max(1, (g.UH + band_rows / 2) / band_rows);
    BH = hp::ceil_div(g.UH, bands);
    bands = hp::ceil_div(g.UH, BH);
    src_rows_cap = std::min(g.R, (int)std::ceil((BH + 2 * (KR + 1)) * (double)g.R / g.UH) + 3);
    peaks_lds = (size_t)4 * (((size_t)src_rows_cap * g.Cc + 1) / 2 * 2 + 4 * (BH + 2 * (KR + 1) + 6)); // source rows + the four row tables
    HP_REQUIRE(peaks_lds <= 160 * 1024, HP_ERR_INVALID, "paf: feature map too large for LDS tiling");
    HP_HIP_TRY(hipFuncSetAttribute((const void*)paf_peaks_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)peaks_lds));
    HP_HIP_TRY(hipFuncSetAttribute((const void*)paf_peaks_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)peaks_lds));
    HP_TRY(alloc_lists());

    const size_t B = max_batch;
    HP_TRY(pcount.alloc(B * HP_COCO_N_PARTS * sizeof(int)));
    HP_TRY(pcount_last.alloc(B * HP_COCO_N_PARTS * sizeof(int)));
    HP_TRY(conn_count.alloc(B * HP_COCO_N_PAIRS * sizeof(int)));
    HP_TRY(flags.alloc(B * sizeof(int)));
    HP_TRY(h_counts.alloc(2 * B * sizeof(int)));
    // invariant between batches: peak counters and overflow flags are zero (the assemble kernel restores it)
    HP_HIP_TRY(hipMemsetAsync(pcount.p, 0, B * HP_COCO_N_PARTS * sizeof(int), stream)); // (own stream: see the table upload above)
    HP_HIP_TRY(hipMemsetAsync(flags.p, 0, B * sizeof(int), stream));
    HP_HIP_TRY(hipStreamSynchronize(stream));
    memset(h_counts.p, 0, 2 * B * sizeof(int));
    shaped = true;
    return HP_OK;
}

extern "C" {

int hp_paf_create(hp_paf** out, float conf_thresh, float paf_thresh, int res_w, int res_h, int max_batch)
{
    HP_REQUIRE(out, HP_ERR_INVALID, "hp_paf_create: null out");
    HP_REQUIRE(max_batch >= 1 && max_batch <= 65535, HP_ERR_INVALID, "hp_paf_create: max_batch %d", max_batch);
    hp_paf* p = new hp_paf();
    p->conf_thresh = conf_thresh, p->paf_thresh = paf_thresh, p->res_w = res_w, p->res_h = res_h, p->max_batch = max_batch;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        hp::set_error("hp_paf_create: %s", hipGetErrorString(e));
        delete p;
        return HP_ERR_HIP;
    }
    *out = p;
    return HP_OK;
}

void hp_paf_destroy(hp_paf* p)
{
    if (!p)
        return;
    if (p->stream)
        (void)hipStreamSynchronize(p->stream);
    if (p->done)
        (void)hipEventDestroy(p->done);
    if (p->stream)
        (void)hipStreamDestroy(p->stream);
    delete p;
}

void* hp_paf_stream(hp_paf* p) { return p ? (void*)p->stream : nullptr; }

int hp_paf_set_conf_thresh(hp_paf* p, float thresh)
{
    HP_REQUIRE(p, HP_ERR_INVALID, "null parser");
    p->conf_thresh = thresh;
    return HP_OK;
}

int hp_paf_set_paf_thresh(hp_paf* p, float thresh)
{
    HP_REQUIRE(p, HP_ERR_INVALID, "null parser");
    p->paf_thresh = thresh;
    return HP_OK;
}

static int launch_peaks(hp_paf* p, int n, const float* dev_conf, hipStream_t s, float* dump_up, float* dump_smooth, int channels)
{
    dim3 grid(p->bands * p->strips, channels, n);
    if (dump_up || dump_smooth)
        hipLaunchKernelGGL(paf_peaks_kernel<true>, grid, dim3(PEAK_THREADS), p->peaks_lds, s, dev_conf, p->g, p->gk, p->conf_thresh,
            p->BH, p->CW, p->strips, p->src_rows_cap, p->plist.as<dpeak>(), p->pcount.as<int>(), p->peak_cap, dump_up, dump_smooth);
    else
        hipLaunchKernelGGL(paf_peaks_kernel<false>, grid, dim3(PEAK_THREADS), p->peaks_lds, s, dev_conf, p->g, p->gk, p->conf_thresh,
            p->BH, p->CW, p->strips, p->src_rows_cap, p->plist.as<dpeak>(), p->pcount.as<int>(), p->peak_cap, dump_up, dump_smooth);
    HP_HIP_TRY(hipGetLastError());
    return HP_OK;
}

} // extern "C"

int hp_paf::launch(int n, const float* dev_conf, const float* dev_paf, hipStream_t s)
{
    hp_paf* p = this;
    // (pcount / flags are zero here: zeroed at creation and re-zeroed by every assemble launch)
    HP_TRY(launch_peaks(p, n, dev_conf, s, nullptr, nullptr, HP_COCO_N_PARTS));
    hipLaunchKernelGGL(paf_sort_kernel, dim3(HP_COCO_N_PARTS, n), dim3(256), p->peak_cap * sizeof(int), s,
        p->plist.as<dpeak>(), p->pcount.as<int>(), p->peak_cap, p->sorted.as<dpeak>());
    hipLaunchKernelGGL(paf_limbs_kernel, dim3(HP_COCO_N_PAIRS, n), dim3(LIMB_THREADS), p->limbs_lds, s, dev_paf, p->g, p->paf_thresh,
        p->sorted.as<dpeak>(), p->pcount.as<int>(), p->peak_cap, p->cand_cap, p->conns.as<dconn>(), p->conn_count.as<int>(),
        p->flags.as<int>());
    hipLaunchKernelGGL(paf_assemble_kernel, dim3(n), dim3(64), 0, s, p->sorted.as<dpeak>(), p->pcount.as<int>(), p->peak_cap,
        p->conns.as<dconn>(), p->conn_count.as<int>(), p->res_w, p->res_h, p->h_humans.as<hp_human>(), p->h_counts.as<int>(),
        p->human_cap, p->flags.as<int>(), p->h_counts.as<int>() + p->max_batch, p->pcount_last.as<int>());
    HP_HIP_TRY(hipGetLastError());
    HP_HIP_TRY(hipEventRecord(p->done, s));
    return HP_OK;
}

extern "C" {

int hp_paf_enqueue(hp_paf* p, int n, const float* dev_conf, const int conf_shape[3], const float* dev_paf,
    const int paf_shape[3], void* stream)
{
    HP_REQUIRE(p && dev_conf && dev_paf, HP_ERR_INVALID, "hp_paf_enqueue: null argument");
    HP_REQUIRE(n >= 1 && n <= p->max_batch, HP_ERR_CAPACITY, "hp_paf_enqueue: batch %d > max_batch %d", n, p->max_batch);
    HP_REQUIRE(p->pending == 0, HP_ERR_STATE, "hp_paf_enqueue: previous batch not collected");
    HP_TRY(p->shape(conf_shape, paf_shape));
    HP_TRY(p->launch(n, dev_conf, dev_paf, stream ? (hipStream_t)stream : p->stream));
    p->last_conf = dev_conf, p->last_paf = dev_paf; // must stay valid until hp_paf_collect (a frame that overflows a list is re-parsed)
    p->pending = n;
    return HP_OK;
}

int hp_paf_collect(hp_paf* p, hp_human* out, int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_paf_collect: null argument");
    HP_REQUIRE(p->pending > 0, HP_ERR_STATE, "hp_paf_collect: nothing enqueued");
    HP_HIP_TRY(hipEventSynchronize(p->done));
    const int n = p->pending;
    p->pending = 0;
    p->last_n = n;
    int fl = 0, max_h = 0;
    for (int round = 0;; ++round) {
        fl = 0, max_h = 0;
        for (int f = 0; f < n; ++f) {
            fl |= p->h_counts.as<int>()[p->max_batch + f];
            max_h = std::max(max_h, p->h_counts.as<int>()[f]);
        }
        // a list overflowed somewhere in the batch: grow what can grow and parse the batch again (the reference's vectors just grow)
        const size_t planes = (size_t)4 * 2 * p->g.R * p->g.Cc + (size_t)4 * (3 * p->g.UW + 4 * p->g.UH), per_cand = sizeof(cand_t) + sizeof(int); // (+ the up-sampling tables)
        const int cand_max = (int)((156 * 1024 - planes) / per_cand);
        // the new capacities are computed into locals and committed only together with the buffers that match them: leaving the loop
        // (round limit) or failing to allocate must not leave caps larger than plist / sorted / conns / h_humans
        if (round >= 6)
            break;
        int peak_cap = p->peak_cap, cand_cap = p->cand_cap, human_cap = p->human_cap;
        bool grown = false;
        if ((fl & 1) && peak_cap < PEAK_CAP_MAX)
            peak_cap = std::min(peak_cap * 2, PEAK_CAP_MAX), grown = true;
        if ((fl & 2) && cand_cap < cand_max)
            cand_cap = std::min(cand_cap * 2, cand_max), grown = true;
        if (max_h > human_cap && human_cap < HUMAN_CAP_MAX) {
            while (human_cap < max_h && human_cap < HUMAN_CAP_MAX)
                human_cap *= 2;
            grown = true;
        }
        if (!grown)
            break;
        const int old_peak = p->peak_cap, old_cand = p->cand_cap, old_human = p->human_cap;
        p->peak_cap = peak_cap, p->cand_cap = cand_cap, p->human_cap = human_cap;
        if (const int rc_alloc = p->alloc_lists(); rc_alloc != HP_OK) {
            p->peak_cap = old_peak, p->cand_cap = old_cand, p->human_cap = old_human; // (buffers already re-allocated are only larger)
            (void)p->alloc_lists();
            return rc_alloc;
        }
        HP_TRY(p->launch(n, p->last_conf, p->last_paf, p->stream));
        HP_HIP_TRY(hipEventSynchronize(p->done));
    }
    int rc = HP_OK;
    for (int f = 0; f < n; ++f) {
        const int nh = p->h_counts.as<int>()[f];
        const int ff = p->h_counts.as<int>()[p->max_batch + f] & 7;
        n_out[f] = nh;
        if (ff) { // per-frame status: the other frames of the batch are complete, this one is truncated at a hard limit
            hp::set_error("paf: frame %d exceeds a hard list limit (flags=%d: 1=peaks/part>%d, 2=candidates/limb>%d, 4=skeleton fragments>%d); its result is truncated",
                f, ff, p->peak_cap, p->cand_cap, MAXH);
            rc = HP_ERR_CAPACITY;
        }
        if (nh > cap_per_frame || nh > p->human_cap) {
            hp::set_error("paf: frame %d has %d humans, capacity %d", f, nh, std::min(cap_per_frame, p->human_cap));
            rc = HP_ERR_CAPACITY;
        }
        if (out)
            memcpy(out + (size_t)f * cap_per_frame, p->h_humans.as<hp_human>() + (size_t)f * p->human_cap,
                sizeof(hp_human) * std::min(std::min(nh, cap_per_frame), p->human_cap));
    }
    return rc;
}

int hp_paf_process_batch(hp_paf* p, int n, const float* conf, const int conf_shape[3], const float* paf,
    const int paf_shape[3], int on_device, hp_human* out, int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && conf && paf && conf_shape && paf_shape, HP_ERR_INVALID, "hp_paf_process_batch: null argument");
    HP_REQUIRE(n >= 1 && n <= p->max_batch, HP_ERR_CAPACITY, "hp_paf_process_batch: batch %d > max_batch %d", n, p->max_batch);
    const float *dc = conf, *dp = paf;
    if (!on_device) {
        const size_t cb = (size_t)n * conf_shape[0] * conf_shape[1] * conf_shape[2] * sizeof(float);
        const size_t pb = (size_t)n * paf_shape[0] * paf_shape[1] * paf_shape[2] * sizeof(float);
        if (p->in_conf.bytes < cb)
            HP_TRY(p->in_conf.alloc((size_t)p->max_batch * conf_shape[0] * conf_shape[1] * conf_shape[2] * sizeof(float)));
        if (p->in_paf.bytes < pb)
            HP_TRY(p->in_paf.alloc((size_t)p->max_batch * paf_shape[0] * paf_shape[1] * paf_shape[2] * sizeof(float)));
        HP_HIP_TRY(hipMemcpyAsync(p->in_conf.p, conf, cb, hipMemcpyHostToDevice, p->stream));
        HP_HIP_TRY(hipMemcpyAsync(p->in_paf.p, paf, pb, hipMemcpyHostToDevice, p->stream));
        dc = p->in_conf.as<float>(), dp = p->in_paf.as<float>();
    }
    HP_TRY(hp_paf_enqueue(p, n, dc, conf_shape, dp, paf_shape, nullptr));
    return hp_paf_collect(p, out, cap_per_frame, n_out);
}

int hp_paf_debug_peaks(hp_paf* p, int frame, hp_peak* out, int cap, int* n)
{
    HP_REQUIRE(p && n && p->shaped, HP_ERR_INVALID, "hp_paf_debug_peaks: bad argument");
    HP_REQUIRE(frame >= 0 && frame < p->last_n && p->pending == 0, HP_ERR_STATE, "hp_paf_debug_peaks: no completed batch holds frame %d", frame);
    std::vector<int> cnt(HP_COCO_N_PARTS);
    HP_HIP_TRY(hipMemcpy(cnt.data(), p->pcount_last.as<int>() + frame * HP_COCO_N_PARTS, cnt.size() * 4, hipMemcpyDeviceToHost));
    std::vector<dpeak> buf(p->peak_cap);
    int id = 0;
    for (int k = 0; k < HP_COCO_N_PARTS; ++k) {
        const int nk = std::min(cnt[k], p->peak_cap);
        HP_HIP_TRY(hipMemcpy(buf.data(), p->sorted.as<dpeak>() + ((size_t)frame * HP_COCO_N_PARTS + k) * p->peak_cap, nk * sizeof(dpeak), hipMemcpyDeviceToHost));
        for (int i = 0; i < nk; ++i, ++id)
            if (out && id < cap)
                out[id] = hp_peak{ k, buf[i].x, buf[i].y, buf[i].score, id };
    }
    *n = id;
    return HP_OK;
}

int hp_paf_debug_conns(hp_paf* p, int frame, hp_conn* out, int cap, int* n)
{
    HP_REQUIRE(p && n && p->shaped, HP_ERR_INVALID, "hp_paf_debug_conns: bad argument");
    HP_REQUIRE(frame >= 0 && frame < p->last_n && p->pending == 0, HP_ERR_STATE, "hp_paf_debug_conns: no completed batch holds frame %d", frame);
    std::vector<int> cnt(HP_COCO_N_PAIRS);
    HP_HIP_TRY(hipMemcpy(cnt.data(), p->conn_count.as<int>() + frame * HP_COCO_N_PAIRS, cnt.size() * 4, hipMemcpyDeviceToHost));
    std::vector<dconn> buf(p->peak_cap);
    int id = 0;
    for (int l = 0; l < HP_COCO_N_PAIRS; ++l) {
        HP_HIP_TRY(hipMemcpy(buf.data(), p->conns.as<dconn>() + ((size_t)frame * HP_COCO_N_PAIRS + l) * p->peak_cap, cnt[l] * sizeof(dconn), hipMemcpyDeviceToHost));
        for (int i = 0; i < cnt[l]; ++i, ++id)
            if (out && id < cap)
                out[id] = hp_conn{ l, buf[i].cid1, buf[i].cid2, buf[i].score };
    }
    *n = id;
    return HP_OK;
}

int hp_paf_debug_sort(const float* host_scores, int n, int* host_order, int* used_heap)
{
    HP_REQUIRE(host_scores && host_order && n >= 0 && n <= (1 << 20), HP_ERR_INVALID, "hp_paf_debug_sort: bad argument");
    if (used_heap)
        *used_heap = 0;
    if (n == 0)
        return HP_OK;
    hp::dev_buf dsc, dcand, dord, dflag;
    HP_TRY(dsc.alloc((size_t)n * 4));
    HP_TRY(dcand.alloc((size_t)n * sizeof(cand_t)));
    HP_TRY(dord.alloc((size_t)n * 4));
    HP_TRY(dflag.alloc(4));
    HP_HIP_TRY(hipMemcpy(dsc.p, host_scores, (size_t)n * 4, hipMemcpyHostToDevice));
    HP_HIP_TRY(hipMemset(dflag.p, 0, 4));
    hipLaunchKernelGGL(paf_debug_sort_kernel, dim3(1), dim3(64), 0, 0, dsc.as<float>(), n, dcand.as<cand_t>(), dord.as<int>(), dflag.as<int>());
    HP_HIP_TRY(hipGetLastError());
    HP_HIP_TRY(hipDeviceSynchronize());
    HP_HIP_TRY(hipMemcpy(host_order, dord.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    int fl = 0;
    HP_HIP_TRY(hipMemcpy(&fl, dflag.p, 4, hipMemcpyDeviceToHost));
    if (used_heap)
        *used_heap = fl & 1;
    HP_REQUIRE(!(fl & 2), HP_ERR_STATE, "hp_paf_debug_sort: restated introsort stack overflow");
    return HP_OK;
}

int hp_paf_debug_maps(hp_paf* p, const float* host_conf, const int conf_shape[3], float* host_up, float* host_smoothed)
{
    HP_REQUIRE(p && host_conf && conf_shape, HP_ERR_INVALID, "hp_paf_debug_maps: null argument");
    HP_REQUIRE(p->pending == 0, HP_ERR_STATE, "hp_paf_debug_maps: a batch is in flight");
    int ps[3] = { 2 * HP_COCO_N_PAIRS, conf_shape[1], conf_shape[2] };
    if (p->shaped)
        ps[0] = p->g.L2;
    HP_TRY(p->shape(conf_shape, ps));
    const size_t in_b = (size_t)p->g.J * p->g.R * p->g.Cc * 4, out_b = (size_t)p->g.J * p->g.UH * p->g.UW * 4;
    hp::dev_buf din, dup, dsm;
    HP_TRY(din.alloc(in_b));
    HP_TRY(dup.alloc(out_b));
    HP_TRY(dsm.alloc(out_b));
    HP_HIP_TRY(hipMemcpy(din.p, host_conf, in_b, hipMemcpyHostToDevice));
    HP_HIP_TRY(hipMemsetAsync(p->pcount.p, 0, (size_t)HP_COCO_N_PARTS * sizeof(int), p->stream));
    HP_TRY(launch_peaks(p, 1, din.as<float>(), p->stream, dup.as<float>(), dsm.as<float>(), p->g.J));
    HP_HIP_TRY(hipMemsetAsync(p->pcount.p, 0, (size_t)HP_COCO_N_PARTS * sizeof(int), p->stream)); // restore the invariant
    HP_HIP_TRY(hipStreamSynchronize(p->stream));
    if (host_up)
        HP_HIP_TRY(hipMemcpy(host_up, dup.p, out_b, hipMemcpyDeviceToHost));
    if (host_smoothed)
        HP_HIP_TRY(hipMemcpy(host_smoothed, dsm.p, out_b, hipMemcpyDeviceToHost));
    return HP_OK;
}

} // extern "C"
