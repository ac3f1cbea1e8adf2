// pifpaf_parser.hip — hyperpose::parser::pifpaf on gfx950 (replaces reference src/pifpaf.cpp and
// src/pifpaf_decoder/openpifpaf_postprocessor.cpp).
//
// What the reference does per frame on one CPU thread: allocate and zero four [17,H_hr,W_hr] float maps (40 MB at
// 385x385, postprocessor.cpp:295-298,650-654), scatter-add a clamped Gaussian per confident PIF cell into them
// (:194-248, only `targetsCoreOnly` is ever read back), look the map up once per seed candidate (:679-706) and
// twice per CAF candidate (:715-762), then grow skeletons seed by seed (:457-572) and soft-NMS them (:574-635).
//
// Split used here:
//   GPU  pp_cells_kernel  : per (frame, key-point field) ordered compaction of the confident PIF cells together
//                           with their Gaussian footprint (the inputs of scalarSquareAddGaussWitMax).
//        pp_seeds_kernel / pp_caf_kernel : the hi-res map is NEVER materialised — the few thousand look-ups a
//                           frame needs are evaluated on demand by folding the field's compacted cell list in
//                           order (`v = min(1, v + contribution)`), which is exactly the value the reference's
//                           scatter leaves in that pixel.  Seeds are appended with an atomic (they are sorted by
//                           their full tuple afterwards); the 19x2 CAF lists are compacted order-preserving with
//                           wave ballots because growConnectionBlend's tie-breaking depends on entry order.
//        pp_offsets_kernel / pp_pack_kernel : pack all lists of the batch into one dense arena -> two D2H copies.
//   host the data-dependent tail on the compacted lists (KBs): seed-ordered greedy grow with its priority
//        queue, occupancy, soft-NMS, thresholding, 17 -> 18 key-point remap (src/pifpaf.cpp:52-92).
// Compulsory HBM traffic is one read of pif + paf (2.46 MB/frame) instead of >= 80 MB/frame of memset + copies.
//
// Compiled with -ffp-contract=off; every float/double expression keeps the reference's types and operand order.
#include "hp_common.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <numeric>
#include <optional>
#include <queue>
#include <set>
#include <tuple>
#include <vector>

namespace {

constexpr int NK = 17;  // N_PIFPAF_KEYPOINTS
constexpr int NB = 19;  // N_PIFPAF_BONES
constexpr float STRIDE = 8.0f;            // postprocessor.cpp:139
constexpr float SEED_THRESHOLD = 0.3f;    // :140
constexpr float INSTANCE_THRESHOLD = 0.2f; // :143
constexpr float PAF_SCORE_THRE = 0.2;     // :717
constexpr float CIF_FLOOR = 0.1;          // :718
constexpr int HDR = 64; // ints per frame: [0] seeds, [1..38] caf list sizes (bone*2 + dir; dir 0 = forward), [39] arena offset (floats), [40] flags

// bones (1-based joint ids), postprocessor.cpp:64-84
__constant__ int c_bones[NB][2] = { { 16, 14 }, { 14, 12 }, { 17, 15 }, { 15, 13 }, { 12, 13 }, { 6, 12 }, { 7, 13 }, { 6, 7 }, { 6, 8 },
    { 7, 9 }, { 8, 10 }, { 9, 11 }, { 2, 3 }, { 1, 2 }, { 1, 3 }, { 2, 4 }, { 3, 5 }, { 4, 6 }, { 5, 7 } };
const int h_bones[NB][2] = { { 16, 14 }, { 14, 12 }, { 17, 15 }, { 15, 13 }, { 12, 13 }, { 6, 12 }, { 7, 13 }, { 6, 7 }, { 6, 8 },
    { 7, 9 }, { 8, 10 }, { 9, 11 }, { 2, 3 }, { 1, 2 }, { 1, 3 }, { 2, 4 }, { 3, 5 }, { 4, 6 }, { 5, 7 } };
__constant__ int c_bwd_idx[9] = { 0, 3, 4, 1, 2, 6, 5, 8, 7 }; // BACKWARD_IDX :739

struct pp_cell { // one confident PIF cell = one call slice of scalarSquareAddGaussWitMax (:194-248)
    float cx, cy, sigma, v16;
    int minx, maxx, miny, maxy;
};

struct pp_geom {
    int H, W, H_hr, W_hr;
    float maxx, maxy; // W_hr - 0.51, H_hr - 0.51 as float (:682)
};

__device__ __forceinline__ float clipf(float val, float low, float high) { return fmaxf(low, fminf(high, val)); }

// value the reference's targetsCoreOnly holds at (yy, xx) of one field: fold the field's cells in cell order
__device__ float pifhr_at(const pp_cell* __restrict__ cells, int n, long long yy, long long xx)
{
    float val = 0.f;
    for (int i = 0; i < n; ++i) {
        const pp_cell c = cells[i];
        if (xx < c.minx || xx >= c.maxx || yy < c.miny || yy >= c.maxy)
            continue;
        const float dx2 = ((float)xx - c.cx) * ((float)xx - c.cx);
        const float dy2 = ((float)yy - c.cy) * ((float)yy - c.cy);
        const float tc = c.sigma * 1.0f;
        if (dx2 + dy2 > tc * tc)
            continue;
        float vv;
        if (dx2 < 0.25 && dy2 < 0.25)
            vv = c.v16;
        else {
            float x = (float)(-0.5 * (double)(dx2 + dy2) / (double)(c.sigma * c.sigma));
            if (x > 2 || x < -2)
                x = 0.f;
            else {
                x = 1.f + x / 8;
                x *= x;
                x *= x;
                x *= x;
            }
            vv = c.v16 * x;
        }
        val += vv;
        val = fminf(1.0f, val);
    }
    return val;
}

__global__ __launch_bounds__(64) void pp_cells_kernel(const float* __restrict__ pif, pp_geom g, pp_cell* __restrict__ cells, int* __restrict__ ncells)
{
    const int f = blockIdx.x, fr = blockIdx.y, lane = threadIdx.x;
    const int HW = g.H * g.W;
    const float* p = pif + ((size_t)fr * NK + f) * 5 * HW;
    pp_cell* out = cells + ((size_t)fr * NK + f) * HW;
    int n = 0;
    for (int base = 0; base < HW; base += 64) {
        const int j = base + lane;
        bool hit = false;
        pp_cell c;
        if (j < HW) {
            const float v = p[j];
            if (v > 0.1f) { // v_th, postprocessor.hpp:93
                hit = true;
                c.cx = p[HW + j] * STRIDE;
                c.cy = p[2 * HW + j] * STRIDE;
                c.sigma = (float)fmax(1., 0.5 * (double)p[4 * HW + j] * (double)STRIDE); // :327
                c.v16 = v * (1.0f / 16.0f);                                              // vsmul(v, 1/PIF_NN), :344
                const float tc = c.sigma * 1.0f;
                const long long minx = (long long)clipf(c.cx - tc, 0, g.W_hr - 1);
                const long long maxx = (long long)clipf(c.cx + tc + 1, minx + 1, g.W_hr);
                const long long miny = (long long)clipf(c.cy - tc, 0, g.H_hr - 1);
                const long long maxy = (long long)clipf(c.cy + tc + 1, miny + 1, g.H_hr);
                c.minx = (int)minx, c.maxx = (int)maxx, c.miny = (int)miny, c.maxy = (int)maxy;
            }
        }
        const unsigned long long m = __ballot(hit);
        if (hit)
            out[n + __popcll(m & ((1ull << lane) - 1ull))] = c;
        n += __popcll(m);
    }
    if (lane == 0)
        ncells[fr * NK + f] = n;
}

struct pp_seed {
    float v;
    int f;
    float x, y, s;
};

__global__ __launch_bounds__(64) void pp_seeds_kernel(const float* __restrict__ pif, pp_geom g, const pp_cell* __restrict__ cells,
    const int* __restrict__ ncells, pp_seed* __restrict__ seeds, int seed_cap, int* __restrict__ hdr)
{
    const int f = blockIdx.x, fr = blockIdx.y, lane = threadIdx.x;
    const int HW = g.H * g.W;
    const float* p = pif + ((size_t)fr * NK + f) * 5 * HW;
    const pp_cell* fc = cells + ((size_t)fr * NK + f) * HW;
    const int nc = ncells[fr * NK + f];
    for (int j = lane; j < HW; j += 64) {
        const float c = p[j];
        if (!(c > SEED_THRESHOLD))
            continue;
        const float x = p[HW + j], y = p[2 * HW + j], s = p[4 * HW + j];
        if (x < -0.49 || y < -0.49 || x > g.maxx || y > g.maxy) // :689 (field-unit x against hi-res bounds, as written)
            continue;
        // (size_t)(y * STRIDE + 0.5): float product, double add, truncation; clamped where the reference would index out of bounds
        long long iy = (long long)((double)(y * STRIDE) + 0.5), ix = (long long)((double)(x * STRIDE) + 0.5);
        iy = min(max(iy, 0ll), (long long)g.H_hr - 1), ix = min(max(ix, 0ll), (long long)g.W_hr - 1);
        float v = pifhr_at(fc, nc, iy, ix);
        v = (float)(0.9 * (double)v + 0.1 * (double)c); // :697
        if (v > SEED_THRESHOLD) {
            const int pos = atomicAdd(&hdr[fr * HDR + 0], 1);
            if (pos < seed_cap)
                seeds[(size_t)fr * seed_cap + pos] = pp_seed{ v, f, x * STRIDE, y * STRIDE, s * STRIDE };
        }
    }
}

// CAF scoring (:715-762) of one bone of one frame; lists [frame][bone][dir][HW][9], dir 0 = forward, 1 = backward
__global__ __launch_bounds__(64) void pp_caf_kernel(const float* __restrict__ paf, pp_geom g, const pp_cell* __restrict__ cells,
    const int* __restrict__ ncells, float* __restrict__ lists, int* __restrict__ hdr)
{
    const int b = blockIdx.x, fr = blockIdx.y, lane = threadIdx.x;
    const int HW = g.H * g.W;
    const float* p = paf + ((size_t)fr * NB + b) * 9 * HW;
    const int pf_bwd = c_bones[b][0] - 1, pf_fwd = c_bones[b][1] - 1;
    float* l_fwd = lists + (((size_t)fr * NB + b) * 2 + 0) * (size_t)HW * 9;
    float* l_bwd = lists + (((size_t)fr * NB + b) * 2 + 1) * (size_t)HW * 9;
    int n_fwd = 0, n_bwd = 0;
    for (int base = 0; base < HW; base += 64) {
        const int j = base + lane;
        float ch[9];
        bool hit_b = false, hit_f = false;
        float nv_b = 0.f, nv_f = 0.f;
        if (j < HW) {
            const float conf = p[j];
            if (conf > PAF_SCORE_THRE) {
                ch[0] = conf;
#pragma unroll
                for (int c = 1; c < 9; ++c)
                    ch[c] = p[(size_t)c * HW + j] * STRIDE;
#pragma unroll
                for (int dir = 0; dir < 2; ++dir) { // backward pass first, then forward (:764-765); the lists are independent
                    const float x = dir == 0 ? ch[1] : ch[3], y = dir == 0 ? ch[2] : ch[4]; // this_ch[idx_mapping[3]], [4]
                    const int pfield = dir == 0 ? pf_bwd : pf_fwd;
                    if (!(x < -0.49 || y < -0.49 || x > g.maxx || y > g.maxy)) {
                        long long iy = (long long)((double)y + 0.5), ix = (long long)((double)x + 0.5);
                        iy = min(max(iy, 0ll), (long long)g.H_hr - 1), ix = min(max(ix, 0ll), (long long)g.W_hr - 1);
                        const float cifhr_t = pifhr_at(cells + ((size_t)fr * NK + pfield) * HW, ncells[fr * NK + pfield], iy, ix);
                        const float new_v = ch[0] * (CIF_FLOOR + (1 - CIF_FLOOR) * cifhr_t);
                        if (new_v > PAF_SCORE_THRE) {
                            if (dir == 0)
                                hit_b = true, nv_b = new_v;
                            else
                                hit_f = true, nv_f = new_v;
                        }
                    }
                }
            }
        }
        const unsigned long long mb = __ballot(hit_b), mf = __ballot(hit_f);
        if (hit_b) {
            float* e = l_bwd + (size_t)(n_bwd + __popcll(mb & ((1ull << lane) - 1ull))) * 9;
#pragma unroll
            for (int c = 0; c < 9; ++c)
                e[c] = ch[c_bwd_idx[c]];
            e[0] = nv_b;
        }
        if (hit_f) {
            float* e = l_fwd + (size_t)(n_fwd + __popcll(mf & ((1ull << lane) - 1ull))) * 9;
#pragma unroll
            for (int c = 0; c < 9; ++c)
                e[c] = ch[c];
            e[0] = nv_f;
        }
        n_bwd += __popcll(mb);
        n_fwd += __popcll(mf);
    }
    if (lane == 0) {
        hdr[fr * HDR + 1 + b * 2 + 0] = n_fwd;
        hdr[fr * HDR + 1 + b * 2 + 1] = n_bwd;
    }
}

// frame f's arena slice: [seeds x 5 floats][list 0 x 9 floats][list 1]...; offsets are a prefix over frames
__global__ void pp_offsets_kernel(int n, int seed_cap, long long arena_cap, int* __restrict__ hdr, int* __restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    long long off = 0;
    for (int f = 0; f < n; ++f) {
        int* h = hdr + f * HDR;
        if (h[0] > seed_cap) {
            h[40] |= 1;
            h[0] = seed_cap;
        }
        h[39] = (int)off;
        long long sz = (long long)h[0] * 5;
        for (int l = 0; l < 2 * NB; ++l)
            sz += (long long)h[1 + l] * 9;
        if (off + sz > arena_cap) { // the frame's lists do not fit: it is reported, not packed
            h[40] |= 2;
            sz = 0;
            h[0] = 0;
            for (int l = 0; l < 2 * NB; ++l)
                h[1 + l] = 0;
        }
        off += sz;
    }
    *total = (int)off;
}

__global__ __launch_bounds__(256) void pp_pack_kernel(pp_geom g, int seed_cap, const int* __restrict__ hdr, const pp_seed* __restrict__ seeds,
    const float* __restrict__ lists, float* __restrict__ arena)
{
    const int item = blockIdx.x, fr = blockIdx.y; // item 0 = seeds, 1..38 = caf lists
    const int* h = hdr + fr * HDR;
    const int HW = g.H * g.W;
    long long off = h[39];
    if (item == 0) {
        const float* src = reinterpret_cast<const float*>(seeds + (size_t)fr * seed_cap);
        for (int i = threadIdx.x; i < h[0] * 5; i += 256)
            arena[off + i] = src[i];
        return;
    }
    off += (long long)h[0] * 5;
    for (int l = 0; l < item - 1; ++l)
        off += (long long)h[1 + l] * 9;
    const float* src = lists + ((size_t)fr * 2 * NB + (item - 1)) * (size_t)HW * 9;
    for (int i = threadIdx.x; i < h[item] * 9; i += 256)
        arena[off + i] = src[i];
}

// =====================================================================================================
// host tail
struct caf_list {
    const float* e = nullptr; // n x 9, entry-major
    int n = 0;
    float at(int c, int i) const { return e[(size_t)i * 9 + c]; }
};

struct annotation {
    float kp[NK * 3];
    float scale[NK];
    annotation(int j, float x, float y, float v)
    {
        std::memset(kp, 0, sizeof(kp));
        std::memset(scale, 0, sizeof(scale));
        kp[j * 3] = x, kp[j * 3 + 1] = y, kp[j * 3 + 2] = v;
    }
    float score() const // postprocessor.hpp:72-84
    {
        float maxv = 0.0f, vv = 0.0f;
        for (int k = 0; k < NK; ++k) {
            const float v = kp[k * 3 + 2];
            if (v > maxv)
                maxv = v;
            vv += v * v;
        }
        return 0.1f * maxv + 0.9f * vv / (float)NK;
    }
};

// Occupancy (:22-59): byte map indexed [field][y][x]; only the cells that were set are cleared between frames
struct occupancy {
    static constexpr float reduction = 2.f;
    static constexpr float min_scale_reduced = 4.f / reduction;
    size_t d0 = 0, d1 = 0, d2 = 0;
    std::vector<uint8_t> view;
    std::vector<size_t> touched;
    void reset(size_t a, size_t b, size_t c)
    {
        if (a * b * c > view.size())
            view.assign(a * b * c, 0);
        else
            for (size_t i : touched)
                view[i] = 0;
        touched.clear();
        d0 = a, d1 = b, d2 = c;
    }
    bool get(size_t f, size_t y, size_t x) const { return view[(d1 * d2) * f + d2 * y + x]; }
    void set(size_t f, size_t y, size_t x)
    {
        const size_t i = (d1 * d2) * f + d2 * y + x;
        if (!view[i]) {
            view[i] = 1;
            touched.push_back(i);
        }
    }
    bool fuzz_get(size_t f, float y, float x) const
    {
        if (f >= d0)
            return true;
        const float xx = std::min((float)d2 - 1, std::max(0.f, x / reduction));
        const float yy = std::min((float)d1 - 1, std::max(0.f, y / reduction));
        return get(f, (size_t)yy, (size_t)xx);
    }
    // scalarSquareAddSingle (:250-282)
    void add_square(int f, int fieldH, int fieldW, float x, float y, float width, float red = 1.0, float min_scaled = 0.0)
    {
        if (red != 1.0) {
            x /= red;
            y /= red;
            width = std::max(min_scaled, width / red);
        }
        const int minx = std::min(fieldW - 1, std::max(0, (int)(x - width)));
        const int miny = std::min(fieldH - 1, std::max(0, (int)(y - width)));
        const int maxx = std::min(fieldW, std::max(minx + 1, std::min(fieldW, (int)(x + width) + 1)));
        const int maxy = std::min(fieldH, std::max(miny + 1, std::min(fieldH, (int)(y + width) + 1)));
        for (int yy = miny; yy < maxy; ++yy)
            for (int xx = minx; xx < maxx; ++xx)
                set(f, yy, xx);
    }
};

struct link_t {
    int end, caf, forward;
};

// BY_SOURCE_MAP (:91-137) derived from the bone table: bone k = (j1, j2) links j1-1 -> j2-1 forward and back;
// the reference iterates a std::map with std::greater, i.e. by DESCENDING end joint.
const std::array<std::vector<link_t>, NK>& by_source()
{
    static const std::array<std::vector<link_t>, NK> m = [] {
        std::array<std::vector<link_t>, NK> r;
        for (int k = 0; k < NB; ++k) {
            r[h_bones[k][0] - 1].push_back({ h_bones[k][1] - 1, k, 1 });
            r[h_bones[k][1] - 1].push_back({ h_bones[k][0] - 1, k, 0 });
        }
        for (auto& v : r)
            std::sort(v.begin(), v.end(), [](const link_t& a, const link_t& b) { return a.end > b.end; });
        return r;
    }();
    return m;
}

using xysv_t = std::tuple<float, float, float, float>;

// growConnectionBlend (:382-437)
xysv_t connection_blend(float x, float y, float s, const caf_list& L)
{
    const float sigma = 2.0 * s;
    const float sigma2 = 0.25 * s * s;
    size_t i1 = 0, i2 = 0;
    float s1 = 0, s2 = 0;
    for (int i = 0; i < L.n; ++i) {
        const float px = L.at(1, i), py = L.at(2, i);
        if ((px < x - sigma) || (px > x + sigma) || (py < y - sigma) || (py > y + sigma))
            continue;
        const float d2 = (px - x) * (px - x) + (py - y) * (py - y);
        const float score = std::exp(-0.5 * d2 / sigma2) * L.at(0, i);
        if (score >= s1) {
            i2 = i1, s2 = s1;
            i1 = i, s1 = score;
        } else if (score > s2) {
            i2 = i, s2 = score;
        }
    }
    if (s1 == 0)
        return { 0, 0, 0, 0 };
    const float ex1 = L.at(3, i1), ey1 = L.at(4, i1), es1 = L.at(8, i1);
    if (s2 < 0.01 || s2 < 0.5 * s1)
        return { ex1, ey1, es1, (float)(s1 * 0.5) };
    const float ex2 = L.at(3, i2), ey2 = L.at(4, i2), es2 = L.at(8, i2);
    const float blend_d2 = (ex1 - ex2) * (ex1 - ex2) + (ey1 - ey2) * (ey1 - ey2);
    if (blend_d2 > ((es1 * es1) / 4))
        return { ex1, ey1, es1, (float)(s1 * 0.5) };
    return { (s1 * ex1 + s2 * ex2) / (s1 + s2), (s1 * ey1 + s2 * ey2) / (s1 + s2), (s1 * es1 + s2 * es2) / (s1 + s2),
        (float)(0.5 * (s1 + s2)) };
}

struct frontier_item {
    float key; // -score
    std::optional<xysv_t> val;
    int start, end;
};
// std::priority_queue<queue_item, std::deque<queue_item>, std::greater<>> with operator> defined as >= (:441-455)
struct frontier_cmp {
    bool operator()(const frontier_item& l, const frontier_item& r) const { return l.key >= r.key; }
};

// grow (:457-572)
void grow(annotation& ann, const caf_list* fwd, const caf_list* bwd, float keypoint_threshold)
{
    std::set<std::pair<int, int>> in_frontier;
    std::priority_queue<frontier_item, std::deque<frontier_item>, frontier_cmp> frontier;
    auto add_to_frontier = [&](int start) {
        for (const link_t& lk : by_source()[start]) {
            if (ann.kp[3 * lk.end + 2] > 0.0)
                continue;
            if (in_frontier.count({ start, lk.end }))
                continue;
            const float max_possible = std::sqrt(ann.kp[3 * start + 2]);
            frontier.push(frontier_item{ -max_possible, std::nullopt, start, lk.end });
            in_frontier.emplace(start, lk.end);
        }
    };
    auto connection_value = [&](int start, int end) -> std::optional<xysv_t> {
        const link_t* lk = nullptr;
        for (const link_t& c : by_source()[start])
            if (c.end == end)
                lk = &c;
        const caf_list& caf_f = lk->forward ? fwd[lk->caf] : bwd[lk->caf];
        const caf_list& caf_b = lk->forward ? bwd[lk->caf] : fwd[lk->caf];
        const float x = ann.kp[start * 3], y = ann.kp[start * 3 + 1], v = ann.kp[start * 3 + 2];
        const float scale_s = std::max(0.f, ann.scale[start]);
        const auto [nx, ny, ns, nv] = connection_blend(x, y, scale_s, caf_f);
        if (nv == 0)
            return std::nullopt;
        const float kscore = std::sqrt(nv * v);
        if (kscore < keypoint_threshold)
            return std::nullopt;
        constexpr float rel = 0.5;
        if (kscore < v * rel)
            return std::nullopt;
        const float scale_t = std::max(0.f, ns);
        const auto [rx, ry, rs, rv] = connection_blend(nx, ny, scale_t, caf_b);
        (void)rv;
        if (rs == 0 || std::abs(x - rx) + std::abs(y - ry) > scale_s)
            return std::nullopt;
        return std::make_tuple(nx, ny, ns, kscore);
    };
    auto frontier_get = [&]() -> std::optional<frontier_item> {
        while (!frontier.empty()) {
            frontier_item entry = frontier.top();
            frontier.pop();
            if (entry.val.has_value())
                return entry;
            if (ann.kp[entry.end * 3 + 2] > 0.0)
                continue;
            const auto nv = connection_value(entry.start, entry.end);
            if (!nv.has_value())
                continue;
            frontier.push(frontier_item{ -std::get<3>(*nv), nv, entry.start, entry.end });
        }
        return std::nullopt;
    };
    for (int j = 0; j < NK; ++j)
        if (ann.kp[3 * j + 2] != 0.0)
            add_to_frontier(j);
    while (true) {
        const auto entry = frontier_get();
        if (!entry.has_value())
            break;
        const int jt = entry->end;
        if (ann.kp[jt * 3 + 2] > 0.0)
            continue;
        const auto [nx, ny, ns, nv] = *entry->val;
        ann.kp[jt * 3] = nx, ann.kp[jt * 3 + 1] = ny, ann.kp[jt * 3 + 2] = nv;
        ann.scale[jt] = ns;
        add_to_frontier(jt);
    }
}

// softNMS (:574-635)
std::vector<annotation> soft_nms(std::vector<annotation>& anns, occupancy& occ)
{
    float maxx = 0.0f, maxy = 0.0f;
    for (auto& a : anns)
        for (int k = 0; k < NK; ++k) {
            if (a.kp[k * 3] > maxx)
                maxx = a.kp[k * 3];
            if (a.kp[k * 3 + 1] > maxy)
                maxy = a.kp[k * 3 + 1];
        }
    const int h = (int)(maxy + 1), w = (int)(maxx + 1);
    occ.reset(17, h, w);
    std::vector<int> order(anns.size());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&anns](int const& a, int const& b) { return anns[a].score() > anns[b].score(); });
    for (int a : order) {
        annotation& ann = anns[a];
        for (int k = 0; k < NK; ++k) {
            const float x = ann.kp[k * 3], y = ann.kp[k * 3 + 1], v = ann.kp[k * 3 + 2];
            if (v == 0)
                continue;
            const int i = std::min(std::max(0, (int)std::round(x)), w - 1);
            const int j = std::min(std::max(0, (int)std::round(y)), h - 1);
            if (occ.fuzz_get(k, j, i)) // reads through the /2 coordinates although the writes below are full-res (:614,617)
                ann.kp[k * 3 + 2] = 0.0f;
            else
                occ.add_square(k, h, w, x, y, ann.scale[k]);
        }
    }
    std::vector<annotation> filtered;
    for (auto& ann : anns)
        for (int k = 0; k < NK; ++k)
            if (ann.kp[k * 3 + 2] > 0.0f) {
                filtered.push_back(ann);
                break;
            }
    return filtered;
}

// postprocess (:657-927) from the seeds / CAF lists onwards, then the 17 -> 18 remap of src/pifpaf.cpp:52-92
void decode_frame(const pp_geom& g, const int* hdr, const float* arena, float keypoint_threshold, int net_w, int net_h, occupancy& occ,
    std::vector<hp_human>& out)
{
    const float* p = arena + hdr[39];
    std::vector<std::tuple<float, int, float, float, float>> seeds;
    for (int i = 0; i < hdr[0]; ++i, p += 5) {
        int f;
        std::memcpy(&f, p + 1, 4);
        seeds.emplace_back(p[0], f, p[2], p[3], p[4]);
    }
    caf_list fwd[NB], bwd[NB];
    for (int b = 0; b < NB; ++b) {
        fwd[b].e = p, fwd[b].n = hdr[1 + b * 2];
        p += (size_t)fwd[b].n * 9;
        bwd[b].e = p, bwd[b].n = hdr[1 + b * 2 + 1];
        p += (size_t)bwd[b].n * 9;
    }
    std::sort(seeds.begin(), seeds.end(), std::greater{}); // :772

    occ.reset(NK, g.H_hr, g.W_hr);
    std::vector<annotation> anns;
    for (const auto& [v, f, x, y, s] : seeds) {
        if (occ.fuzz_get(f, y, x))
            continue;
        annotation ann(f, x, y, v);
        ann.scale[f] = s;
        grow(ann, fwd, bwd, keypoint_threshold);
        anns.push_back(ann);
        for (int i = 0; i < NK; ++i) {
            if (ann.kp[i * 3 + 2] == 0)
                continue;
            occ.add_square(i, g.H_hr, g.W_hr, ann.kp[i * 3], ann.kp[i * 3 + 1], ann.scale[i], occupancy::reduction, occupancy::min_scale_reduced);
        }
    }
    if (!anns.empty())
        anns = soft_nms(anns, occ);
    std::vector<annotation> kept;
    for (auto& ann : anns) {
        for (int k = 0; k < NK; ++k)
            if (ann.kp[k * 3 + 2] < keypoint_threshold)
                ann.kp[k * 3 + 2] = 0.0f;
        if (ann.score() >= INSTANCE_THRESHOLD)
            kept.push_back(ann);
    }
    std::sort(kept.begin(), kept.end(), [](const annotation& a, const annotation& b) { return a.score() > b.score(); });

    static const int from_index[16] = { 6, 8, 10, 5, 7, 9, 12, 14, 16, 11, 13, 15, 2, 1, 4, 3 }; // src/pifpaf.cpp:72-76
    for (const annotation& ann : kept) {
        hp_human man;
        std::memset(&man, 0, sizeof(man));
        man.score = ann.score();
        auto p2p = [&](int src, hp_body_part& dst) {
            const int x = ann.kp[src * 3], y = ann.kp[src * 3 + 1]; // truncated to int like Landmark.position (:888-889)
            if (ann.kp[src * 3 + 2] > 0.) {
                dst.score = 1;
                dst.x = x / (float)net_w;
                dst.y = y / (float)net_h;
                dst.has_value = 1;
            }
        };
        p2p(0, man.parts[0]);
        for (int i = 0; i < 16; ++i)
            p2p(from_index[i], man.parts[i + 2]);
        if (man.parts[2].has_value && man.parts[5].has_value) {
            man.parts[1].x = (man.parts[2].x + man.parts[5].x) / 2;
            man.parts[1].y = (man.parts[2].y + man.parts[5].y) / 2;
            man.parts[1].has_value = 1;
            man.parts[1].score = (man.parts[2].score + man.parts[5].score) / 2;
        }
        out.push_back(man);
    }
}

} // namespace

struct hp_pifpaf {
    int net_h, net_w, max_batch;
    float thresh;
    int seed_cap = 4096;
    bool shaped = false;
    pp_geom g{};
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hp::dev_buf cells, ncells, seeds, lists, hdr, total, in_paf, in_pif;
    hp::host_buf h_hdr, h_arena; // the pack kernel writes the dense arena straight into pinned host memory
    size_t arena_cap = 0;        // floats
    std::vector<occupancy> occ;  // one per pool worker
    int pending = 0;
};

extern "C" {

int hp_pifpaf_create(hp_pifpaf** out, int net_h, int net_w, float thresh, int max_batch)
{
    HP_REQUIRE(out && net_h > 0 && net_w > 0 && max_batch >= 1, HP_ERR_INVALID, "hp_pifpaf_create: bad argument");
    std::unique_ptr<hp_pifpaf> p(new hp_pifpaf());
    p->net_h = net_h, p->net_w = net_w, p->thresh = thresh, p->max_batch = max_batch;
    HP_HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HP_HIP_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    p->occ.resize(hp::frame_pool::instance().workers());
    *out = p.release();
    return HP_OK;
}

void hp_pifpaf_destroy(hp_pifpaf* p)
{
    if (!p)
        return;
    if (p->stream) {
        (void)hipStreamSynchronize(p->stream);
        (void)hipStreamDestroy(p->stream);
    }
    if (p->done) {
        (void)hipEventSynchronize(p->done);
        (void)hipEventDestroy(p->done);
    }
    delete p;
}

void* hp_pifpaf_stream(hp_pifpaf* p) { return p ? (void*)p->stream : nullptr; }

static int pifpaf_launch(hp_pifpaf* p, int n, const float* paf, const float* pif, int fh, int fw, int on_device, hipStream_t s)
{
    HP_REQUIRE(p && paf && pif, HP_ERR_INVALID, "hp_pifpaf: null argument");
    HP_REQUIRE(n >= 1 && n <= p->max_batch, HP_ERR_CAPACITY, "hp_pifpaf: batch %d > max_batch %d", n, p->max_batch);
    HP_REQUIRE(fh >= 2 && fw >= 2, HP_ERR_INVALID, "pifpaf: bad field size %dx%d", fh, fw);
    HP_REQUIRE(p->pending == 0, HP_ERR_STATE, "hp_pifpaf: a batch is already in flight, collect it first");
    if (p->shaped)
        HP_REQUIRE(p->g.H == fh && p->g.W == fw, HP_ERR_STATE, "pifpaf: field size changed after the first call");
    const size_t HW = (size_t)fh * fw, B = p->max_batch;
    if (!p->shaped) {
        p->g.H = fh, p->g.W = fw;
        p->g.H_hr = (fh - 1) * (int)STRIDE + 1, p->g.W_hr = (fw - 1) * (int)STRIDE + 1; // initTensors :641-642
        p->g.maxx = p->g.W_hr - 0.51, p->g.maxy = p->g.H_hr - 0.51;                        // :682
        HP_TRY(p->cells.alloc(B * NK * HW * sizeof(pp_cell)));
        HP_TRY(p->ncells.alloc(B * NK * sizeof(int)));
        HP_TRY(p->seeds.alloc(B * p->seed_cap * sizeof(pp_seed)));
        HP_TRY(p->lists.alloc(B * NB * 2 * HW * 9 * sizeof(float)));
        HP_TRY(p->hdr.alloc(B * HDR * sizeof(int)));
        HP_TRY(p->total.alloc(sizeof(int)));
        p->arena_cap = B * ((size_t)p->seed_cap * 5 + 4 * HW * 9); // generous; a frame that does not fit is flagged by pp_offsets_kernel
        HP_TRY(p->h_hdr.alloc(p->hdr.bytes));
        HP_TRY(p->h_arena.alloc(p->arena_cap * sizeof(float)));
        p->shaped = true;
    }
    const float *dpaf = paf, *dpif = pif;
    if (!on_device) {
        const size_t pb = (size_t)NB * 9 * HW * sizeof(float), ib = (size_t)NK * 5 * HW * sizeof(float);
        if (p->in_paf.bytes < pb * B)
            HP_TRY(p->in_paf.alloc(pb * B));
        if (p->in_pif.bytes < ib * B)
            HP_TRY(p->in_pif.alloc(ib * B));
        HP_HIP_TRY(hipMemcpyAsync(p->in_paf.p, paf, pb * n, hipMemcpyHostToDevice, s));
        HP_HIP_TRY(hipMemcpyAsync(p->in_pif.p, pif, ib * n, hipMemcpyHostToDevice, s));
        dpaf = p->in_paf.as<float>(), dpif = p->in_pif.as<float>();
    }
    HP_HIP_TRY(hipMemsetAsync(p->hdr.p, 0, (size_t)n * HDR * sizeof(int), s));
    hipLaunchKernelGGL(pp_cells_kernel, dim3(NK, n), dim3(64), 0, s, dpif, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>());
    hipLaunchKernelGGL(pp_seeds_kernel, dim3(NK, n), dim3(64), 0, s, dpif, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>(), p->seeds.as<pp_seed>(),
        p->seed_cap, p->hdr.as<int>());
    hipLaunchKernelGGL(pp_caf_kernel, dim3(NB, n), dim3(64), 0, s, dpaf, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>(), p->lists.as<float>(),
        p->hdr.as<int>());
    hipLaunchKernelGGL(pp_offsets_kernel, dim3(1), dim3(1), 0, s, n, p->seed_cap, (long long)p->arena_cap, p->hdr.as<int>(), p->total.as<int>());
    hipLaunchKernelGGL(pp_pack_kernel, dim3(1 + 2 * NB, n), dim3(256), 0, s, p->g, p->seed_cap, p->hdr.as<int>(), p->seeds.as<pp_seed>(),
        p->lists.as<float>(), p->h_arena.as<float>());
    HP_HIP_TRY(hipGetLastError());
    HP_HIP_TRY(hipMemcpyAsync(p->h_hdr.p, p->hdr.p, (size_t)n * HDR * sizeof(int), hipMemcpyDeviceToHost, s));
    HP_HIP_TRY(hipEventRecord(p->done, s));
    p->pending = n;
    return HP_OK;
}

int hp_pifpaf_enqueue(hp_pifpaf* p, int n, const float* dev_paf, const float* dev_pif, int fh, int fw, void* stream)
{
    return pifpaf_launch(p, n, dev_paf, dev_pif, fh, fw, 1, stream ? (hipStream_t)stream : (p ? p->stream : nullptr));
}

namespace {
struct pifpaf_job {
    hp_pifpaf* p;
    hp_human* out;
    int cap;
    int* n_out;
    std::vector<int> rc;
};
void pifpaf_frame(int f, int worker, void* ctx)
{
    pifpaf_job& j = *static_cast<pifpaf_job*>(ctx);
    hp_pifpaf* p = j.p;
    const int* hdr = p->h_hdr.as<int>() + (size_t)f * HDR;
    if (hdr[40] != 0)
        j.rc[f] = hdr[40] & 2 ? 3 : 1;
    std::vector<hp_human> humans;
    decode_frame(p->g, hdr, p->h_arena.as<float>(), p->thresh, p->net_w, p->net_h, p->occ[worker], humans);
    j.n_out[f] = (int)humans.size();
    if ((int)humans.size() > j.cap && !j.rc[f])
        j.rc[f] = 2;
    if (j.out)
        std::copy(humans.begin(), humans.begin() + std::min<size_t>(humans.size(), j.cap), j.out + (size_t)f * j.cap);
}
} // namespace

int hp_pifpaf_collect(hp_pifpaf* p, hp_human* out, int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_pifpaf_collect: null argument");
    HP_REQUIRE(p->pending > 0, HP_ERR_STATE, "hp_pifpaf_collect: nothing was enqueued");
    const int n = p->pending;
    p->pending = 0;
    HP_HIP_TRY(hipEventSynchronize(p->done));
    pifpaf_job job{ p, out, cap_per_frame, n_out, std::vector<int>(n, 0) };
    hp::frame_pool::instance().run(n, pifpaf_frame, &job);
    int rc = HP_OK;
    for (int f = 0; f < n; ++f)
        if (job.rc[f] == 1) {
            hp::set_error("pifpaf: frame %d has more than %d seeds", f, p->seed_cap);
            rc = HP_ERR_CAPACITY;
        } else if (job.rc[f] == 3) {
            hp::set_error("pifpaf: the compacted lists of frame %d exceed the arena (%zu floats for the batch)", f, p->arena_cap);
            rc = HP_ERR_CAPACITY;
        } else if (job.rc[f] == 2) {
            hp::set_error("pifpaf: frame %d has %d humans, capacity %d", f, n_out[f], cap_per_frame);
            rc = HP_ERR_CAPACITY;
        }
    return rc;
}

int hp_pifpaf_process_batch(hp_pifpaf* p, int n, const float* paf, const float* pif, int fh, int fw, int on_device, hp_human* out,
    int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_pifpaf_process_batch: null argument");
    HP_TRY(pifpaf_launch(p, n, paf, pif, fh, fw, on_device, p->stream));
    return hp_pifpaf_collect(p, out, cap_per_frame, n_out);
}

} // extern "C"
