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
//        pp_seed_rank_kernel / pp_decode_kernel : the data-dependent tail - seed order, greedy grow with its priority
//                           queue, occupancy, soft-NMS, thresholding, 17 -> 18 key-point remap (src/pifpaf.cpp:52-92) -
//                           one wavefront per frame (see "Device decoder" below).
//   host the same tail in C++ (decode_frame) for the frames the device decoder declines (capacities, a score too
//        close to a float rounding boundary to be libm-independent) and behind HP_PIFPAF_HOST_TAIL=1; only those
//        frames are packed and copied.
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

// the same value for ONE point (uniform yy, xx), computed by the whole wavefront: every lane tests one cell of a 64-cell slice, the
// contributing cells are then folded in cell order (the running clamp makes the sum order-dependent), so the result is the sequential
// fold's, bit for bit, at 1/64 of its dependent memory round trips
constexpr int PP_LDS_CELLS = 512; // cells of one field staged in LDS (16 KB); a denser field is read from global memory
__device__ __forceinline__ const pp_cell* pp_stage_cells(const pp_cell* __restrict__ cells, int n, pp_cell* lds, int lane)
{
    if (n > PP_LDS_CELLS)
        return cells;
    for (int i = lane; i < n * 2; i += 64) // 32-byte cells as two 16-byte halves
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(cells)[i];
    __syncthreads();
    return lds;
}
__device__ float pifhr_at_wave(const pp_cell* cells, int n, int yy, int xx, int lane)
{
    float val = 0.f;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        // branch-free up to the one uniform test: the cell is read unconditionally (clamped index), the footprint and radius tests are
        // bit operations, and the expensive part (a double division) runs only when some lane's cell covers the point
        const pp_cell c = cells[min(i, n - 1)];
        const bool inb = (i < n) & !((xx < c.minx) | (xx >= c.maxx) | (yy < c.miny) | (yy >= c.maxy));
        const float dx2 = ((float)xx - c.cx) * ((float)xx - c.cx);
        const float dy2 = ((float)yy - c.cy) * ((float)yy - c.cy);
        const float tc = c.sigma * 1.0f;
        const bool on = inb & !(dx2 + dy2 > tc * tc);
        unsigned long long m = __ballot(on);
        if (!m)
            continue;
        float vv = c.v16;
        if (!(dx2 < 0.25 && dy2 < 0.25)) {
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
        while (m) {
            const int l = __ffsll(m) - 1;
            m &= m - 1;
            val += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), l)); // (uniform l: v_readlane, not an LDS-crossbar shuffle per contribution)
            val = fminf(1.0f, val);
        }
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
    __shared__ pp_cell s_cells[PP_LDS_CELLS];
    const int nc = ncells[fr * NK + f];
    const pp_cell* fc = pp_stage_cells(cells + ((size_t)fr * NK + f) * HW, nc, s_cells, lane);
    for (int base = 0; base < HW; base += 64) {
        const int j = base + lane;
        bool cand = false;
        float c = 0.f, x = 0.f, y = 0.f, s = 0.f;
        int iy = 0, ix = 0;
        if (j < HW) {
            c = p[j];
            if (c > SEED_THRESHOLD) {
                x = p[HW + j], y = p[2 * HW + j], s = p[4 * HW + j];
                if (!(x < -0.49 || y < -0.49 || x > g.maxx || y > g.maxy)) { // :689 (field-unit x against hi-res bounds, as written)
                    // (size_t)(y * STRIDE + 0.5): float product, double add, truncation; clamped where the reference would index out of bounds
                    const long long ly = (long long)((double)(y * STRIDE) + 0.5), lx = (long long)((double)(x * STRIDE) + 0.5);
                    iy = (int)min(max(ly, 0ll), (long long)g.H_hr - 1), ix = (int)min(max(lx, 0ll), (long long)g.W_hr - 1);
                    cand = true;
                }
            }
        }
        float v = 0.f;
        for (unsigned long long m = __ballot(cand); m; m &= m - 1) { // one wave-wide look-up per candidate
            const int l = __ffsll(m) - 1;
            const float r = pifhr_at_wave(fc, nc, __builtin_amdgcn_readlane(iy, l), __builtin_amdgcn_readlane(ix, l), lane);
            if (lane == l)
                v = r;
        }
        if (cand) {
            v = (float)(0.9 * (double)v + 0.1 * (double)c); // :697
            if (v > SEED_THRESHOLD) {
                const int pos = atomicAdd(&hdr[fr * HDR + 0], 1);
                if (pos < seed_cap)
                    seeds[(size_t)fr * seed_cap + pos] = pp_seed{ v, f, x * STRIDE, y * STRIDE, s * STRIDE };
            }
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
    __shared__ pp_cell s_cells[2][PP_LDS_CELLS];
    const int nc_d[2] = { ncells[fr * NK + pf_bwd], ncells[fr * NK + pf_fwd] };
    const pp_cell* fc_d[2] = { pp_stage_cells(cells + ((size_t)fr * NK + pf_bwd) * HW, nc_d[0], s_cells[0], lane),
        pp_stage_cells(cells + ((size_t)fr * NK + pf_fwd) * HW, nc_d[1], s_cells[1], lane) };
    int n_fwd = 0, n_bwd = 0;
    for (int base = 0; base < HW; base += 64) {
        const int j = base + lane;
        float ch[9] = {};
        bool hit_b = false, hit_f = false, conf_ok = false;
        float nv_b = 0.f, nv_f = 0.f;
        if (j < HW) {
            const float conf = p[j];
            if (conf > PAF_SCORE_THRE) {
                conf_ok = true;
                ch[0] = conf;
#pragma unroll
                for (int c = 1; c < 9; ++c)
                    ch[c] = p[(size_t)c * HW + j] * STRIDE;
            }
        }
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) { // backward pass first, then forward (:764-765); the lists are independent
            const float x = dir == 0 ? ch[1] : ch[3], y = dir == 0 ? ch[2] : ch[4]; // this_ch[idx_mapping[3]], [4]
            const pp_cell* fc = fc_d[dir];
            const int nc = nc_d[dir];
            bool cand = false;
            int iy = 0, ix = 0;
            if (conf_ok && !(x < -0.49 || y < -0.49 || x > g.maxx || y > g.maxy)) {
                const long long ly = (long long)((double)y + 0.5), lx = (long long)((double)x + 0.5);
                iy = (int)min(max(ly, 0ll), (long long)g.H_hr - 1), ix = (int)min(max(lx, 0ll), (long long)g.W_hr - 1);
                cand = true;
            }
            float cifhr_t = 0.f;
            for (unsigned long long m = __ballot(cand); m; m &= m - 1) { // one wave-wide look-up per candidate
                const int l = __ffsll(m) - 1;
                const float r = pifhr_at_wave(fc, nc, __builtin_amdgcn_readlane(iy, l), __builtin_amdgcn_readlane(ix, l), lane);
                if (lane == l)
                    cifhr_t = r;
            }
            if (cand) {
                const float new_v = ch[0] * (CIF_FLOOR + (1 - CIF_FLOOR) * cifhr_t);
                if (new_v > PAF_SCORE_THRE) {
                    if (dir == 0)
                        hit_b = true, nv_b = new_v;
                    else
                        hit_f = true, nv_f = new_v;
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
// (a frame the device decoder finished - dflags[f] == 0 - is not packed at all: h[41] = 1)
__global__ void pp_offsets_kernel(int n, int seed_cap, long long arena_cap, int* __restrict__ hdr, int* __restrict__ total,
    const int* __restrict__ dflags)
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
        if (dflags && dflags[f] == 0) {
            h[41] = 1;
            continue;
        }
        long long sz = (long long)h[0] * 5;
        for (int l = 0; l < 2 * NB; ++l)
            sz += (long long)h[1 + l] * 9;
        h[42] = (int)sz; // what the frame needs (hp_pifpaf_collect grows the arena by it and parses the batch again)
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
    if (h[41])
        return;
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

// =====================================================================================================
// Device decoder: the data-dependent tail (seed order, grow, occupancy, soft-NMS, score / sort, 17 -> 18 remap;
// openpifpaf_postprocessor.cpp:382-635, 764-851, src/pifpaf.cpp:52-92) as ONE WAVEFRONT PER FRAME.  The walk itself is sequential by
// definition (every accepted key-point changes what the next step may do), so the wavefront executes it as uniform scalar code over
// state in LDS, and spends its 64 lanes where the work is: growConnectionBlend's scan of a CAF list (scores in parallel, then the
// reference's order-dependent top-2 rule evaluated in closed form from prefix / suffix maxima), the occupancy squares, the sorts.
// Order-sensitive library behaviour is restated, not approximated: libstdc++'s priority_queue (push_heap / pop_heap with the
// reference's NON-strict comparator, :441-455) and std::sort (introsort + final insertion sort) decide ties exactly as on the host.
// std::exp is evaluated in double as the reference does (`std::exp(-0.5 * d2 / sigma2)`: double arguments) and rounded to float after
// the multiplication; a double-precision exp that differs from glibc's in its last bit changes that float only when the product lies
// within 2^-29 relative of a rounding boundary.
namespace {

constexpr int PD_MAXA = 1024;  // annotations per frame on the device path (more: the frame is decoded by the host tail).  256 until round 6: the sort keys (8 KB of LDS now), 272 B per annotation in HBM and 292 B per human in pinned memory are what it costs; a 49 x 49 map has 2 401 cells per joint type
constexpr int PD_NLINK = 2 * NB;

struct pd_ann {
    float kp[NK * 3];
    float scale[NK];
};
struct pd_links { // BY_SOURCE_MAP (:91-137): per start joint its links by DESCENDING end joint (std::map with std::greater)
    int first[64];  // [j] = first link of start joint j, [NK] = PD_NLINK
    int packed[64]; // [l] = start | end << 8 | caf << 16 | forward << 24
};
__constant__ pd_links c_links;

pd_links make_links()
{
    pd_links L{};
    int at = 0;
    for (int j = 0; j < NK; ++j) {
        L.first[j] = at;
        for (int e = NK - 1; e >= 0; --e)
            for (int k = 0; k < NB; ++k) {
                if (h_bones[k][0] - 1 == j && h_bones[k][1] - 1 == e)
                    L.packed[at++] = j | e << 8 | k << 16 | 1 << 24;
                else if (h_bones[k][1] - 1 == j && h_bones[k][0] - 1 == e)
                    L.packed[at++] = j | e << 8 | k << 16;
            }
    }
    L.first[NK] = at;
    return L;
}

struct pd_occ { // Occupancy (:22-59) with logical extent (d1 rows, d2 cols) and a stored window of (oh x ow) cells per field
    unsigned char* v;
    int oh, ow; // stored; every READ of the decoder lands inside (coordinates are halved), writes outside are dropped
    int d1, d2; // the reference's extent (clamping arithmetic)
};
__device__ __forceinline__ bool occ_get(const pd_occ& o, int f, int y, int x)
{
    return (y < o.oh && x < o.ow) ? o.v[((size_t)f * o.oh + y) * o.ow + x] != 0 : false;
}
__device__ __forceinline__ bool occ_fuzz_get(const pd_occ& o, int f, float y, float x)
{
    if (f >= NK)
        return true;
    const float xx = fminf((float)o.d2 - 1, fmaxf(0.f, x / 2.f));
    const float yy = fminf((float)o.d1 - 1, fmaxf(0.f, y / 2.f));
    return occ_get(o, f, (int)yy, (int)xx);
}
// scalarSquareAddSingle (:250-282); the cells of the square are set by the lanes
__device__ void occ_add_square(const pd_occ& o, int lane, int f, int fieldH, int fieldW, float x, float y, float width, bool reduce)
{
    if (reduce) {
        x /= 2.f;
        y /= 2.f;
        width = fmaxf(2.f, width / 2.f);
    }
    const int minx = min(fieldW - 1, max(0, (int)(x - width)));
    const int miny = min(fieldH - 1, max(0, (int)(y - width)));
    const int maxx = min(fieldW, max(minx + 1, min(fieldW, (int)(x + width) + 1)));
    const int maxy = min(fieldH, max(miny + 1, min(fieldH, (int)(y + width) + 1)));
    const int w = min(maxx, o.ow) - minx, h = min(maxy, o.oh) - miny;
    if (w <= 0 || h <= 0)
        return;
    for (int i = lane; i < w * h; i += 64)
        o.v[((size_t)f * o.oh + miny + i / w) * o.ow + minx + i % w] = 1;
}

__device__ __forceinline__ float ann_score(const float* kp) // postprocessor.hpp:72-84
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

// ---- std::priority_queue<queue_item, std::deque<queue_item>, std::greater<>> with operator> := (>=) on the key (:441-455):
// libstdc++'s push_heap / pop_heap (bits/stl_heap.h) restated, comp(a, b) = a.key >= b.key.  The heap array lives ACROSS THE LANES of the
// wavefront (element i = lane i's (key, id) registers; at most 38 entries are alive: every directed link enters once and is replaced by
// at most one evaluated entry), every index is wave-uniform and elements are read with v_readlane: a sift step costs a few ALU
// instructions instead of an LDS round trip per level.
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
struct pd_heap {
    float k; // this lane's element: key
    int id;  //                      link index (+ 64 once the link carries its evaluated value)
    int n;   // uniform
};
__device__ __forceinline__ void pd_push_heap(pd_heap& h, int lane, int hole, float vk, int vid) // __push_heap(first, hole, 0, value)
{
    int parent = (hole - 1) / 2;
    while (hole > 0) {
        const float pk = rl_f(h.k, parent);
        if (!(pk >= vk))
            break;
        const int pid = rl_i(h.id, parent);
        if (lane == hole)
            h.k = pk, h.id = pid;
        hole = parent;
        parent = (hole - 1) / 2;
    }
    if (lane == hole)
        h.k = vk, h.id = vid;
}
__device__ __forceinline__ void pd_heap_push(pd_heap& h, int lane, float vk, int vid)
{
    ++h.n; // push_back, then push_heap(first, last)
    pd_push_heap(h, lane, h.n - 1, vk, vid);
}
__device__ __forceinline__ void pd_heap_pop(pd_heap& h, int lane) // pop_heap + pop_back
{
    if (h.n > 1) {
        const float vk = rl_f(h.k, h.n - 1); // __pop_heap: value = *result; *result = *first; __adjust_heap(first, 0, len, value)
        const int vid = rl_i(h.id, h.n - 1);
        const int len = h.n - 1;
        int hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (rl_f(h.k, second) >= rl_f(h.k, second - 1))
                --second;
            const float ck = rl_f(h.k, second);
            const int cid = rl_i(h.id, second);
            if (lane == hole)
                h.k = ck, h.id = cid;
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            const float ck = rl_f(h.k, second - 1);
            const int cid = rl_i(h.id, second - 1);
            if (lane == hole)
                h.k = ck, h.id = cid;
            hole = second - 1;
        }
        pd_push_heap(h, lane, hole, vk, vid);
    }
    --h.n;
}

// ---- std::sort on an index array, keys DESCENDING (comp(a, b) = key[a] > key[b]): libstdc++'s introsort + final insertion sort, as in
// paf_parser.hip::libstdcxx_sort_greater.  Returns false when the depth limit ran out (heap-sort fall-back not restated).
__device__ bool pd_sort_desc(int* v, int n, const float* key)
{
#define PD_GT(i, j) (key[v[i]] > key[v[j]])
#define PD_SWAP(i, j)                                                                                             \
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
    int stk_f[48], stk_l[48], stk_d[48], sp = 0;
    stk_f[0] = 0, stk_l[0] = n, stk_d[0] = 2 * lg, sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_f[sp], last = stk_l[sp], depth = stk_d[sp];
        while (last - first > 16) {
            if (depth == 0) {
                ok = false;
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            if (PD_GT(a, b)) {
                if (PD_GT(b, c))
                    PD_SWAP(first, b)
                else if (PD_GT(a, c))
                    PD_SWAP(first, c)
                else
                    PD_SWAP(first, a)
            } else if (PD_GT(a, c))
                PD_SWAP(first, a)
            else if (PD_GT(b, c))
                PD_SWAP(first, c)
            else
                PD_SWAP(first, b)
            int lo = first + 1, hi = last;
            for (;;) {
                while (PD_GT(lo, first))
                    ++lo;
                --hi;
                while (PD_GT(first, hi))
                    --hi;
                if (!(lo < hi))
                    break;
                PD_SWAP(lo, hi)
                ++lo;
            }
            if (sp < 47)
                stk_f[sp] = first, stk_l[sp] = lo, stk_d[sp] = depth, ++sp;
            else
                ok = false;
            first = lo;
        }
    }
    const int head = n > 16 ? 16 : n;
    for (int i = 1; i < head; ++i) {
        const int val = v[i];
        if (key[val] > key[v[0]]) {
            for (int k = i; k > 0; --k)
                v[k] = v[k - 1];
            v[0] = val;
        } else {
            int k = i;
            while (key[val] > key[v[k - 1]]) {
                v[k] = v[k - 1];
                --k;
            }
            v[k] = val;
        }
    }
    for (int i = head; i < n; ++i) {
        const int val = v[i];
        int k = i;
        while (k > 0 && key[val] > key[v[k - 1]]) {
            v[k] = v[k - 1];
            --k;
        }
        v[k] = val;
    }
#undef PD_GT
#undef PD_SWAP
    return ok;
}

struct pd_xysv {
    float x, y, s, v;
};
// score of an entry (conf, px, py) inside the box of (x, y).  `amb` is raised when the double product lies so close to a float rounding
// boundary that a last-bit difference between this device's exp() and the host libm's could round it the other way: such a frame is
// decoded by the host tail instead, so the device result never depends on libm agreement.
__device__ __forceinline__ float pd_score(float cf, float px, float py, float x, float y, float sigma2, bool& amb)
{
    const float d2 = (px - x) * (px - x) + (py - y) * (py - y);
    const double pr = exp(-0.5 * (double)d2 / (double)sigma2) * (double)cf;
    const float sc = (float)pr;
    if ((float)(pr * (1.0 + 0x1p-50)) != sc || (float)(pr * (1.0 - 0x1p-50)) != sc)
        amb = true;
    return sc;
}
// One 128-entry slice of a CAF list as a group of 16 lanes holds it: entry base + gl + 16 k in slot k.  All 24 loads are issued before
// anything is used, so a slice costs one memory round trip.
constexpr int PD_SLOTS = 8;
struct pd_slice {
    float cf[PD_SLOTS], px[PD_SLOTS], py[PD_SLOTS];
};
__device__ __forceinline__ pd_slice pd_load_slice(const float* __restrict__ L, int n, int base, int gl)
{
    pd_slice r;
#pragma unroll
    for (int k = 0; k < PD_SLOTS; ++k) {
        const int i = base + gl + 16 * k;
        const bool in = i < n;
        const float* e = L + (size_t)(in ? i : 0) * 9;
        r.cf[k] = in ? e[0] : 0.f, r.px[k] = in ? e[1] : 0.f, r.py[k] = in ? e[2] : 0.f;
    }
    return r;
}
constexpr int PD_Q = 1024; // entries of one list inside one box (more: the frame goes to the host tail); 4 x 20 KB of LDS (256 until round 6)
struct pd_queue {         // per group of 16 lanes, in LDS: the entries that passed the box test, in list order
    int i[PD_Q];
    float cf[PD_Q], px[PD_Q], py[PD_Q], sc[PD_Q];
};
// growConnectionBlend (:382-437) over one CAF list (n entries x 9 floats, entry-major), evaluated by a GROUP of 16 lanes (gl = lane
// within the group, gs = the group's first lane): four links are scored side by side, one per group, when a joint enters the annotation.
// Group-uniform control flow; the shuffles stay inside the (16-aligned) group.
//   1. box test of every entry (cheap), the entries inside compacted into the group's queue in list order;
//   2. the double-precision scores of the queued entries only, one per lane (the expensive part runs once, not once per slot);
//   3. the streaming top-2 rule (`>=` replaces the best and demotes it, `>` replaces the second) in closed form:
//        best   = the maximum, at the LAST index attaining it;
//        second = the previous occurrence of the maximum if there is one; otherwise max(prefix maximum at its last index,
//                 suffix maximum at its FIRST index) with the prefix winning ties (what was best before the maximum arrived has been
//                 demoted to second and only a strictly larger later score replaces it); (0, index 0) when nothing else passed.
// `first` is the list's first slice, loaded by the caller ahead of time (both lists of a link are requested together).
__device__ pd_xysv pd_connection_blend(int gl, int gs, float x, float y, float s, const float* __restrict__ L, int n, const pd_slice& first,
    pd_queue& Q, bool& amb, bool& overflow)
{
    const float sigma = 2.0 * s;
    const float sigma2 = 0.25 * s * s;
    const float x0 = x - sigma, x1 = x + sigma, y0 = y - sigma, y1 = y + sigma;
    int np = 0;
    for (int base = 0; base < n; base += 16 * PD_SLOTS) {
        pd_slice t;
        if (base == 0)
            t = first;
        else
            t = pd_load_slice(L, n, base, gl);
#pragma unroll
        for (int k = 0; k < PD_SLOTS; ++k) {
            const int i = base + gl + 16 * k;
            const bool in = i < n && !((t.px[k] < x0) || (t.px[k] > x1) || (t.py[k] < y0) || (t.py[k] > y1));
            const unsigned m16 = (unsigned)(__ballot(in) >> gs) & 0xffffu;
            if (in) {
                const int pos = np + __popc(m16 & ((1u << gl) - 1u));
                if (pos < PD_Q)
                    Q.i[pos] = i, Q.cf[pos] = t.cf[k], Q.px[pos] = t.px[k], Q.py[pos] = t.py[k];
            }
            np += __popc(m16);
        }
    }
    if (np > PD_Q)
        np = PD_Q, overflow = true;
    __builtin_amdgcn_wave_barrier();
    // the maximum, at the last position attaining it (a score is NaN only when s == 0: such an entry never wins on the host either)
    float m = -1.f;
    int mq = -1;
    for (int q = gl; q < np; q += 16) {
        const float sc = pd_score(Q.cf[q], Q.px[q], Q.py[q], x, y, sigma2, amb);
        Q.sc[q] = sc;
        if (sc >= 0.f && sc >= m)
            m = sc, mq = q; // within a lane the positions ascend: `>=` keeps the last
    }
    // reductions over the group's 16 lanes with DPP (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: after the four steps
    // every lane holds the result) - register moves instead of four LDS-crossbar round trips per value; the combine steps are
    // commutative and associative (a total order on (score, position)), so the butterfly shape does not matter
#define PD_DPP_F(V, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, 0xf, 0xf, false))
#define PD_DPP_I(V, CTRL) __builtin_amdgcn_update_dpp(0, V, CTRL, 0xf, 0xf, false)
#define PD_RED1(CTRL)                                                                                             \
    {                                                                                                             \
        const float om = PD_DPP_F(m, CTRL);                                                                       \
        const int oq = PD_DPP_I(mq, CTRL);                                                                        \
        if (om > m || (om == m && oq > mq))                                                                       \
            m = om, mq = oq;                                                                                      \
    }
    PD_RED1(0xB1) PD_RED1(0x4E) PD_RED1(0x141) PD_RED1(0x140)
#undef PD_RED1
    // (the reference starts from score_1 = 0 and replaces it with `>=`; score_1 == 0 afterwards: nothing passed, or only zeros)
    if (mq < 0 || m == 0.f)
        return pd_xysv{ 0.f, 0.f, 0.f, 0.f };
    // previous occurrence of the maximum, prefix maximum (last position), suffix maximum (first position); every lane re-reads the
    // scores it wrote itself
    int dup = -1;
    float pm = -1.f, sm = -1.f;
    int pq = -1, sq = 0x7fffffff;
    for (int q = gl; q < np; q += 16) {
        const float sc = Q.sc[q];
        if (q == mq || !(sc >= 0.f))
            continue;
        if (q < mq) {
            if (sc == m)
                dup = max(dup, q);
            if (sc >= pm)
                pm = sc, pq = q;
        } else if (sc > sm)
            sm = sc, sq = q;
    }
#define PD_RED2(CTRL)                                                                                             \
    {                                                                                                             \
        dup = max(dup, PD_DPP_I(dup, CTRL));                                                                      \
        const float opm = PD_DPP_F(pm, CTRL), osm = PD_DPP_F(sm, CTRL);                                           \
        const int opq = PD_DPP_I(pq, CTRL), osq = PD_DPP_I(sq, CTRL);                                             \
        if (opm > pm || (opm == pm && opq > pq))                                                                  \
            pm = opm, pq = opq;                                                                                   \
        if (osm > sm || (osm == sm && osq < sq))                                                                  \
            sm = osm, sq = osq;                                                                                   \
    }
    PD_RED2(0xB1) PD_RED2(0x4E) PD_RED2(0x141) PD_RED2(0x140)
#undef PD_RED2
#undef PD_DPP_F
#undef PD_DPP_I
    float s2 = 0.f;
    int q2 = -1; // (-1: the reference's initial second, score 0 at list index 0)
    if (dup >= 0)
        s2 = m, q2 = dup;
    else {
        if (pq >= 0) // what was best before the maximum arrived has been demoted to second
            s2 = pm, q2 = pq;
        if (sq != 0x7fffffff && sm > s2)
            s2 = sm, q2 = sq;
    }
    // both end points requested together: one more round trip
    const int i1 = Q.i[mq], i2 = q2 >= 0 ? Q.i[q2] : 0;
    const float s1 = m;
    const float ex1 = L[(size_t)i1 * 9 + 3], ey1 = L[(size_t)i1 * 9 + 4], es1 = L[(size_t)i1 * 9 + 8];
    const float ex2 = L[(size_t)i2 * 9 + 3], ey2 = L[(size_t)i2 * 9 + 4], es2 = L[(size_t)i2 * 9 + 8];
    if (s2 < 0.01 || s2 < 0.5 * s1)
        return pd_xysv{ ex1, ey1, es1, (float)(s1 * 0.5) };
    const float blend_d2 = (ex1 - ex2) * (ex1 - ex2) + (ey1 - ey2) * (ey1 - ey2);
    if (blend_d2 > ((es1 * es1) / 4))
        return pd_xysv{ ex1, ey1, es1, (float)(s1 * 0.5) };
    return pd_xysv{ (s1 * ex1 + s2 * ex2) / (s1 + s2), (s1 * ey1 + s2 * ey2) / (s1 + s2), (s1 * es1 + s2 * es2) / (s1 + s2),
        (float)(0.5 * (s1 + s2)) };
}

struct pd_params {
    pp_geom g;
    int seed_cap, oh, ow;
    size_t occ_stride; // bytes per frame (multiple of 16)
    float keypoint_threshold;
    int net_w, net_h;
    int decline_odd; // test hook (HP_PIFPAF_DECLINE_ODD=1): odd frames are flagged 64 and go to the host tail, even ones stay on the device
};

// seeds ranked by their full tuple, descending (std::sort(seeds, std::greater{}) on (v, f, x, y, s) tuples, :772): a total order, so
// any correct sort gives the reference's sequence (identical tuples are interchangeable)
__global__ __launch_bounds__(256) void pp_seed_rank_kernel(const int* __restrict__ hdr, const pp_seed* __restrict__ seeds, int seed_cap,
    int* __restrict__ seed_order)
{
    __shared__ __attribute__((aligned(16))) float t_v[256];
    __shared__ float t_x[256], t_y[256], t_s[256];
    __shared__ int t_f[256];
    const int fr = blockIdx.y;
    const int ns = min(hdr[fr * HDR], seed_cap);
    const pp_seed* S = seeds + (size_t)fr * seed_cap;
    // a few blocks per frame, each walking over its share of the seeds (thousands of mostly empty blocks cost more to launch than the
    // ranking itself)
    for (int i0 = blockIdx.x * 256; i0 < ns; i0 += gridDim.x * 256) {
    const int i = i0 + threadIdx.x;
    const pp_seed a = i < ns ? S[i] : pp_seed{ 0.f, 0, 0.f, 0.f, 0.f };
    int rank = 0;
    for (int base = 0; base < ns; base += 256) {
        __syncthreads();
        {
            const bool in = base + (int)threadIdx.x < ns;
            const pp_seed b = in ? S[base + threadIdx.x] : pp_seed{ 0.f, 0, 0.f, 0.f, 0.f };
            t_v[threadIdx.x] = in ? b.v : -INFINITY; // never above, never equal to a real seed value
            t_f[threadIdx.x] = b.f, t_x[threadIdx.x] = b.x, t_y[threadIdx.x] = b.y, t_s[threadIdx.x] = b.s;
        }
        __syncthreads();
        // the seed value decides almost every comparison: a branch-free count, four values per LDS read; the full tuple is looked at
        // only when some value of the tile ties with this seed's (beyond the seed itself)
        int eq = 0;
#pragma unroll 8
        for (int q = 0; q < 256; q += 4) {
            const float4 bv = *reinterpret_cast<const float4*>(&t_v[q]);
            rank += (bv.x > a.v) + (bv.y > a.v) + (bv.z > a.v) + (bv.w > a.v);
            eq += (bv.x == a.v) + (bv.y == a.v) + (bv.z == a.v) + (bv.w == a.v);
        }
        if (eq > (i >= base && i < base + 256 ? 1 : 0))
            for (int u = 0; u < 256; ++u) {
                if (t_v[u] != a.v)
                    continue;
                bool gt; // b sorts before a
                if (t_f[u] != a.f)
                    gt = t_f[u] > a.f;
                else if (t_x[u] != a.x)
                    gt = t_x[u] > a.x;
                else if (t_y[u] != a.y)
                    gt = t_y[u] > a.y;
                else if (t_s[u] != a.s)
                    gt = t_s[u] > a.s;
                else
                    gt = base + u < i;
                rank += gt;
            }
    }
    if (i < ns)
        seed_order[(size_t)fr * seed_cap + rank] = i;
    }
}

// flags per frame (0 = decoded here; anything else: the host tail decodes the frame from the packed lists):
//   1 more than PD_MAXA annotations      2 soft-NMS extent beyond the stored occupancy window     4 a sort ran out of its depth limit
//   8 more seeds than seed_cap           16 frontier overflow                                      32 rounding-ambiguous score
__global__ __launch_bounds__(64) void pp_decode_kernel(pd_params P, const int* __restrict__ hdr, const pp_seed* __restrict__ seeds,
    const float* __restrict__ lists, const int* __restrict__ seed_order, pd_ann* __restrict__ anns_g, unsigned char* __restrict__ occ_g,
    hp_human* __restrict__ humans, int* __restrict__ n_humans, int* __restrict__ dflags)
{
    __shared__ float s_key[PD_MAXA];
    __shared__ int s_idx[PD_MAXA];
    __shared__ pd_queue s_q[4];
    const int fr = blockIdx.x, lane = threadIdx.x;
    const int HW = P.g.H * P.g.W;
    const int* h = hdr + fr * HDR;
    int flags = (P.decline_odd && (fr & 1)) ? 64 : 0;
    int ns = h[0];
    if (ns > P.seed_cap)
        ns = P.seed_cap, flags |= 8;
    const pp_seed* S = seeds + (size_t)fr * P.seed_cap;
    const int* order = seed_order + (size_t)fr * P.seed_cap;
    pd_ann* A = anns_g + (size_t)fr * PD_MAXA;
    // two occupancy windows (seed loop, soft-NMS), both zeroed by a memset ahead of the kernel
    pd_occ occ{ occ_g + (size_t)fr * 2 * P.occ_stride, P.oh, P.ow, P.g.H_hr, P.g.W_hr };
    // lane l: directed link l (start | end << 8 | caf << 16 | forward << 24), its list sizes; lane j: first link of start joint j
    const int lk = c_links.packed[lane], lk_first = c_links.first[lane];
    int lk_nf = 0, lk_nb = 0;
    if (lane < PD_NLINK) {
        const int caf = (lk >> 16) & 255, d_f = (lk >> 24) ? 0 : 1;
        lk_nf = h[1 + caf * 2 + d_f], lk_nb = h[1 + caf * 2 + (1 - d_f)];
    }
    // the frame's CAF lists are about to be read with one dependent round trip per grown joint: pull them into this XCD's L2 first
    {
        float sink = 0.f;
        for (int l = 0; l < PD_NLINK; l += 2) { // (forward list of bone b = link data of both directions: 38 lists = 19 bones x 2)
            const int nl = h[1 + l] * 9;
            const float* Lp = lists + ((size_t)fr * 2 * NB + l) * (size_t)HW * 9;
            const int nl2 = h[2 + l] * 9;
            const float* Lq = lists + ((size_t)fr * 2 * NB + l + 1) * (size_t)HW * 9;
            for (int i = lane * 16; i < nl; i += 64 * 16)
                sink += Lp[i];
            for (int i = lane * 16; i < nl2; i += 64 * 16)
                sink += Lq[i];
        }
        if (sink == 12345.678f) // (never: keeps the loads)
            flags |= 64;
    }

    int na = 0;
    int si = 0;
    while (si < ns) {
        // the next seed whose cell is free: 64 candidates tested at once (occupancy only grows, so the first free one is the next the
        // sequential loop would accept, and everything before it stays rejected)
        bool free_ = false;
        if (si + lane < ns) {
            const pp_seed c = S[order[si + lane]];
            free_ = !occ_fuzz_get(occ, c.f, c.y, c.x);
        }
        const unsigned long long fm = __ballot(free_);
        if (!fm) {
            si += 64;
            continue;
        }
        si += __ffsll((unsigned long long)fm) - 1;
        const pp_seed sd = S[order[si]];
        ++si;
        if (na >= PD_MAXA) {
            flags |= 1;
            break;
        }
        // Annotation(f, x, y, v), jointScales[f] = s: joint j lives in lane j
        const int sf = __builtin_amdgcn_readfirstlane(sd.f);
        float a_x = 0.f, a_y = 0.f, a_v = 0.f, a_s = 0.f;
        if (lane == sf)
            a_x = sd.x, a_y = sd.y, a_v = sd.v, a_s = sd.s;
        // ---- grow (:457-572): a wave-uniform walk with its whole state in registers (joints, link values, frontier heap across the
        // lanes).  connection_value(start, end) depends only on the start joint, which never changes once it is set (an accepted
        // joint has v > 0 and the walk never overwrites a positive joint), so it is evaluated when the link ENTERS the frontier - up
        // to four links of the new joint at once, one per group of 16 lanes - instead of when the queue first pops it (:503-528): the
        // same values, most of the memory round trips off the serial path.  A link whose end joint is set before its turn is simply
        // never looked at, as in the reference.
        unsigned long long in_frontier = 0ull; // bit = directed link index (one per (start, end) pair)
        pd_heap hp{ 0.f, 0, 0 };
        int v_has = 0; // lane l: connection value of link l
        float v_x = 0.f, v_y = 0.f, v_s = 0.f, v_v = 0.f;
        const int gl = lane & 15, grp = lane >> 4;
        auto add_to_frontier = [&](int start) {
            int ls[4] = { -1, -1, -1, -1 }, nl = 0; // (no joint has more than four links)
            const int l0 = rl_i(lk_first, start), l1 = rl_i(lk_first, start + 1);
            for (int l = l0; l < l1; ++l) {
                const int end = (rl_i(lk, l) >> 8) & 255;
                if (rl_f(a_v, end) > 0.0)
                    continue;
                if ((in_frontier >> l) & 1ull)
                    continue;
                if (nl < 4)
                    ls[nl] = l;
                ++nl;
                in_frontier |= 1ull << l;
            }
            if (nl == 0)
                return;
            if (nl > 4 || hp.n + nl > 64) {
                flags |= 16;
                return;
            }
            const float x = rl_f(a_x, start), y = rl_f(a_y, start), v = rl_f(a_v, start);
            const float scale_s = fmaxf(0.f, rl_f(a_s, start));
            const int my = grp == 0 ? ls[0] : grp == 1 ? ls[1] : grp == 2 ? ls[2] : ls[3];
            bool amb = false, ovf = false;
            int r_has = 0;
            float r_x = 0.f, r_y = 0.f, r_s = 0.f, r_v = 0.f;
            // (`my` differs between the groups: a real cross-lane gather, taken while every lane is active)
            const int pk = __shfl(lk, max(my, 0)), nf = __shfl(lk_nf, max(my, 0)), nb = __shfl(lk_nb, max(my, 0));
            if (my >= 0) { // connection_value(start, end) (:503-528)
                const int caf = (pk >> 16) & 255, d_f = (pk >> 24) ? 0 : 1;
                const float* Lf = lists + (((size_t)fr * NB + caf) * 2 + d_f) * (size_t)HW * 9;
                const float* Lb = lists + (((size_t)fr * NB + caf) * 2 + (1 - d_f)) * (size_t)HW * 9;
                const pd_slice sf0 = pd_load_slice(Lf, nf, 0, gl), sb0 = pd_load_slice(Lb, nb, 0, gl); // both lists in flight
                const pd_xysv nw = pd_connection_blend(gl, grp * 16, x, y, scale_s, Lf, nf, sf0, s_q[grp], amb, ovf);
                if (nw.v != 0) {
                    const float kscore = sqrtf(nw.v * v);
                    if (!(kscore < P.keypoint_threshold) && !(kscore < v * 0.5f)) {
                        const float scale_t = fmaxf(0.f, nw.s);
                        __builtin_amdgcn_wave_barrier();
                        const pd_xysv rv = pd_connection_blend(gl, grp * 16, nw.x, nw.y, scale_t, Lb, nb, sb0, s_q[grp], amb, ovf);
                        if (!(rv.s == 0 || fabsf(x - rv.x) + fabsf(y - rv.y) > scale_s))
                            r_has = 1, r_x = nw.x, r_y = nw.y, r_s = nw.s, r_v = kscore;
                    }
                }
            }
            if (__any(amb))
                flags |= 32;
            if (__any(ovf))
                flags |= 16;
            const float max_possible = sqrtf(v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= nl)
                    break;
                const int l = ls[q];
                const int t_has = rl_i(r_has, q * 16);
                const float t_x = rl_f(r_x, q * 16), t_y = rl_f(r_y, q * 16), t_s = rl_f(r_s, q * 16), t_v = rl_f(r_v, q * 16);
                if (lane == l)
                    v_has = t_has, v_x = t_x, v_y = t_y, v_s = t_s, v_v = t_v;
                pd_heap_push(hp, lane, -max_possible, l);
            }
        };
        add_to_frontier(sf); // (`for j: if v[j] != 0`: only the seed joint is set)
        for (;;) {
            // frontier_get
            bool have = false;
            int link = 0;
            while (hp.n > 0) {
                const int id = rl_i(hp.id, 0);
                pd_heap_pop(hp, lane);
                link = id & 63;
                if (id >= 64) {
                    have = true;
                    break;
                }
                const int end = (rl_i(lk, link) >> 8) & 255;
                if (rl_f(a_v, end) > 0.0)
                    continue;
                if (!rl_i(v_has, link))
                    continue;
                pd_heap_push(hp, lane, -rl_f(v_v, link), 64 + link); // (replaces the entry just popped)
            }
            if (!have)
                break;
            const int jt = (rl_i(lk, link) >> 8) & 255;
            if (rl_f(a_v, jt) > 0.0)
                continue;
            const float e_x = rl_f(v_x, link), e_y = rl_f(v_y, link), e_s = rl_f(v_s, link), e_v = rl_f(v_v, link);
            if (lane == jt)
                a_x = e_x, a_y = e_y, a_v = e_v, a_s = e_s;
            add_to_frontier(jt);
        }
        // annotations.push_back(ann); occupancy of its joints (:797: reduction 2, min scale 2)
        if (lane < NK) {
            A[na].kp[lane * 3] = a_x, A[na].kp[lane * 3 + 1] = a_y, A[na].kp[lane * 3 + 2] = a_v;
            A[na].scale[lane] = a_s;
        }
        ++na;
        for (int i = 0; i < NK; ++i) {
            const float jv = rl_f(a_v, i);
            if (jv == 0)
                continue;
            occ_add_square(occ, lane, i, P.g.H_hr, P.g.W_hr, rl_f(a_x, i), rl_f(a_y, i), rl_f(a_s, i), true);
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- softNMS (:574-635)
    int nkept = 0;
    if (na > 0 && !(flags & 1)) {
        float maxx = 0.0f, maxy = 0.0f;
        for (int i = lane; i < na * NK; i += 64) {
            const int a = i / NK, k = i - a * NK;
            maxx = fmaxf(maxx, A[a].kp[k * 3]);
            maxy = fmaxf(maxy, A[a].kp[k * 3 + 1]);
        }
        for (int off = 32; off >= 1; off >>= 1) {
            maxx = fmaxf(maxx, __shfl_xor(maxx, off));
            maxy = fmaxf(maxy, __shfl_xor(maxy, off));
        }
        const int hh = (int)(maxy + 1), ww = (int)(maxx + 1);
        if ((ww - 1) / 2 >= P.ow || (hh - 1) / 2 >= P.oh)
            flags |= 2; // a read of the soft-NMS map would fall outside the stored window
        pd_occ occ2{ occ.v + P.occ_stride, P.oh, P.ow, hh, ww };
        for (int a = lane; a < na; a += 64) {
            s_key[a] = ann_score(A[a].kp);
            s_idx[a] = a;
        }
        __syncthreads();
        if (lane == 0 && !pd_sort_desc(s_idx, na, s_key))
            dflags[fr] = 4; // (merged with `flags` below)
        __syncthreads();
        for (int oi = 0; oi < na; ++oi) {
            pd_ann& ann = A[s_idx[oi]];
            // joint k of this annotation in lane k: one memory round trip per annotation instead of one per joint
            float jx = 0.f, jy = 0.f, jv = 0.f, js = 0.f;
            if (lane < NK)
                jx = ann.kp[lane * 3], jy = ann.kp[lane * 3 + 1], jv = ann.kp[lane * 3 + 2], js = ann.scale[lane];
            for (int k = 0; k < NK; ++k) {
                const float x = rl_f(jx, k), y = rl_f(jy, k), v = rl_f(jv, k);
                if (v == 0)
                    continue;
                const int i = min(max(0, (int)roundf(x)), ww - 1);
                const int j = min(max(0, (int)roundf(y)), hh - 1);
                const bool taken = occ_fuzz_get(occ2, k, (float)j, (float)i);
                __syncthreads();
                if (taken) {
                    if (lane == 0)
                        ann.kp[k * 3 + 2] = 0.0f;
                } else
                    occ_add_square(occ2, lane, k, hh, ww, x, y, rl_f(js, k), false);
                __syncthreads();
            }
        }
        // filtered = annotations with a positive joint, in their ORIGINAL order; then threshold, score filter, std::sort by score
        for (int a = 0; a < na; ++a) {
            bool any = false;
            for (int k = 0; k < NK; ++k)
                any |= A[a].kp[k * 3 + 2] > 0.0f;
            if (!any)
                continue;
            __syncthreads();
            if (lane < NK && A[a].kp[lane * 3 + 2] < P.keypoint_threshold)
                A[a].kp[lane * 3 + 2] = 0.0f;
            __syncthreads();
            const float sc = ann_score(A[a].kp);
            if (sc >= INSTANCE_THRESHOLD) {
                if (lane == 0)
                    s_key[a] = sc, s_idx[nkept] = a;
                ++nkept;
            }
        }
        __syncthreads();
        if (lane == 0 && !pd_sort_desc(s_idx, nkept, s_key))
            dflags[fr] = 4;
        __syncthreads();
    }
    // ---- 17 -> 18 key-point remap (src/pifpaf.cpp:52-92), straight into the pinned host array
    const int from_index[16] = { 6, 8, 10, 5, 7, 9, 12, 14, 16, 11, 13, 15, 2, 1, 4, 3 };
    hp_human* out = humans + (size_t)fr * PD_MAXA;
    for (int i = lane; i < nkept; i += 64) {
        const pd_ann& ann = A[s_idx[i]];
        hp_human man = {};
        man.score = ann_score(ann.kp);
        auto p2p = [&](int src, hp_body_part& dst) {
            const int x = ann.kp[src * 3], y = ann.kp[src * 3 + 1];
            if (ann.kp[src * 3 + 2] > 0.) {
                dst.score = 1;
                dst.x = x / (float)P.net_w;
                dst.y = y / (float)P.net_h;
                dst.has_value = 1;
            }
        };
        p2p(0, man.parts[0]);
        for (int q = 0; q < 16; ++q)
            p2p(from_index[q], man.parts[q + 2]);
        if (man.parts[2].has_value && man.parts[5].has_value) {
            man.parts[1].x = (man.parts[2].x + man.parts[5].x) / 2;
            man.parts[1].y = (man.parts[2].y + man.parts[5].y) / 2;
            man.parts[1].has_value = 1;
            man.parts[1].score = (man.parts[2].score + man.parts[5].score) / 2;
        }
        out[i] = man;
    }
    __syncthreads();
    if (lane == 0) {
        n_humans[fr] = nkept;
        dflags[fr] = (dflags[fr] & 4) | flags;
    }
}

} // namespace

struct hp_pifpaf {
    int net_h, net_w, max_batch;
    float thresh;
    int seed_cap = 0; // NK x H x W once the field size is known
    bool shaped = false;
    pp_geom g{};
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hp::dev_buf cells, ncells, seeds, lists, hdr, total, in_paf, in_pif;
    hp::host_buf h_hdr, h_arena; // the pack kernel writes the dense arena straight into pinned host memory
    size_t arena_cap = 0;        // floats
    int arena_growths = 0;       // times hp_pifpaf_collect enlarged the arena and parsed a batch again (the reference's vectors just grow)
    struct { int n = 0, fh = 0, fw = 0; const float *paf = nullptr, *pif = nullptr; hipStream_t s = nullptr; } last; // the batch in flight, on the device
    std::vector<occupancy> occ;  // one per pool worker
    int pending = 0;
    // device decoder (pp_decode_kernel); HP_PIFPAF_HOST_TAIL=1 keeps every frame on the host tail
    bool device_decode = true;
    pd_params dp{};
    hp::dev_buf seed_order, anns, occ_dev, n_humans, dflags;
    hp::host_buf h_humans, h_counts; // humans written by the kernel; h_counts = [n_humans x B][dflags x B]
    std::vector<int> last_flags;
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
    if (const char* e = std::getenv("HP_PIFPAF_HOST_TAIL"))
        p->device_decode = !(e[0] && e[0] != '0');
    p->dp.decline_odd = std::getenv("HP_PIFPAF_DECLINE_ODD") && std::atoi(std::getenv("HP_PIFPAF_DECLINE_ODD")) != 0; // (test hook)
    static const int links_once = [] { // BY_SOURCE_MAP into constant memory
        const pd_links L = make_links();
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_links), &L, sizeof(L));
    }();
    HP_REQUIRE(links_once == 0, HP_ERR_HIP, "pifpaf: link table upload failed");
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
        p->seed_cap = NK * (int)HW; // every cell of every field can seed at most once: the seed list cannot overflow
        HP_TRY(p->cells.alloc(B * NK * HW * sizeof(pp_cell)));
        HP_TRY(p->ncells.alloc(B * NK * sizeof(int)));
        HP_TRY(p->seeds.alloc(B * p->seed_cap * sizeof(pp_seed)));
        HP_TRY(p->lists.alloc(B * NB * 2 * HW * 9 * sizeof(float)));
        HP_TRY(p->hdr.alloc(B * HDR * sizeof(int)));
        HP_TRY(p->total.alloc(sizeof(int)));
        p->arena_cap = B * ((size_t)std::min(p->seed_cap, 4096) * 5 + 4 * HW * 9); // generous; a frame that does not fit is flagged by pp_offsets_kernel
        HP_TRY(p->h_hdr.alloc(p->hdr.bytes));
        HP_TRY(p->h_arena.alloc(p->arena_cap * sizeof(float)));
        if (p->device_decode) {
            pd_params& d = p->dp;
            d.g = p->g, d.seed_cap = p->seed_cap;
            // stored occupancy window: every read of the seed loop lies below H_hr / 2 (+1); the soft-NMS pass reads up to
            // (max joint coordinate) / 2, so the window leaves room for joints up to 1.5x outside the image
            d.oh = p->g.H_hr * 3 / 4 + 2, d.ow = p->g.W_hr * 3 / 4 + 2;
            d.occ_stride = ((size_t)NK * d.oh * d.ow + 15) / 16 * 16;
            d.keypoint_threshold = p->thresh, d.net_w = p->net_w, d.net_h = p->net_h;
            HP_TRY(p->seed_order.alloc(B * p->seed_cap * sizeof(int)));
            HP_TRY(p->anns.alloc(B * PD_MAXA * sizeof(pd_ann)));
            HP_TRY(p->occ_dev.alloc(B * 2 * d.occ_stride));
            HP_TRY(p->n_humans.alloc(B * sizeof(int)));
            HP_TRY(p->dflags.alloc(B * sizeof(int)));
            HP_TRY(p->h_humans.alloc(B * PD_MAXA * sizeof(hp_human)));
            HP_TRY(p->h_counts.alloc(2 * B * sizeof(int)));
        }
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
    p->last.n = n, p->last.fh = fh, p->last.fw = fw, p->last.paf = dpaf, p->last.pif = dpif, p->last.s = s;
    HP_HIP_TRY(hipMemsetAsync(p->hdr.p, 0, (size_t)n * HDR * sizeof(int), s));
    hipLaunchKernelGGL(pp_cells_kernel, dim3(NK, n), dim3(64), 0, s, dpif, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>());
    hipLaunchKernelGGL(pp_seeds_kernel, dim3(NK, n), dim3(64), 0, s, dpif, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>(), p->seeds.as<pp_seed>(),
        p->seed_cap, p->hdr.as<int>());
    hipLaunchKernelGGL(pp_caf_kernel, dim3(NB, n), dim3(64), 0, s, dpaf, p->g, p->cells.as<pp_cell>(), p->ncells.as<int>(), p->lists.as<float>(),
        p->hdr.as<int>());
    if (p->device_decode) {
        HP_HIP_TRY(hipMemsetAsync(p->dflags.p, 0, (size_t)n * sizeof(int), s));
        HP_HIP_TRY(hipMemsetAsync(p->occ_dev.p, 0, (size_t)n * 2 * p->dp.occ_stride, s));
        hipLaunchKernelGGL(pp_seed_rank_kernel, dim3(8, n), dim3(256), 0, s, p->hdr.as<int>(), p->seeds.as<pp_seed>(),
            p->seed_cap, p->seed_order.as<int>());
        hipLaunchKernelGGL(pp_decode_kernel, dim3(n), dim3(64), 0, s, p->dp, p->hdr.as<int>(), p->seeds.as<pp_seed>(), p->lists.as<float>(),
            p->seed_order.as<int>(), p->anns.as<pd_ann>(), p->occ_dev.as<unsigned char>(), p->h_humans.as<hp_human>(), p->n_humans.as<int>(),
            p->dflags.as<int>());
        HP_HIP_TRY(hipMemcpyAsync(p->h_counts.p, p->n_humans.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
        HP_HIP_TRY(hipMemcpyAsync(p->h_counts.as<int>() + p->max_batch, p->dflags.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    hipLaunchKernelGGL(pp_offsets_kernel, dim3(1), dim3(1), 0, s, n, p->seed_cap, (long long)p->arena_cap, p->hdr.as<int>(), p->total.as<int>(),
        p->device_decode ? p->dflags.as<int>() : (const int*)nullptr);
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
    if (p->device_decode && p->last_flags[f] == 0) { // decoded on the device: the kernel wrote the humans into pinned memory
        const int nh = p->h_counts.as<int>()[f];
        j.n_out[f] = nh;
        if (nh > j.cap)
            j.rc[f] = 2;
        if (j.out)
            std::copy_n(p->h_humans.as<hp_human>() + (size_t)f * PD_MAXA, std::min(nh, j.cap), j.out + (size_t)f * j.cap);
        return;
    }
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
    // The reference's lists are std::vectors (openpifpaf_postprocessor.cpp:764-851).  The pinned arena that carries the frames the host tail
    // decodes starts at a size that fits ordinary frames; when a frame's lists did not fit (h[40] & 2) it grows to what the batch needs
    // (h[42] per frame, + 25 %) - at most the size at which no list can overflow - and the batch is parsed again from the inputs, which
    // the API keeps valid until collect returns (device inputs) or which were copied to the parser's own buffers (host inputs).
    for (int round = 0; round < 4; ++round) {
        size_t need = 0;
        bool over = false;
        for (int f = 0; f < n; ++f) {
            const int* h = p->h_hdr.as<int>() + (size_t)f * HDR;
            over |= (h[40] & 2) != 0;
            if (!h[41])
                need += (size_t)h[42];
        }
        const size_t HW = (size_t)p->g.H * p->g.W, cap_max = (size_t)p->max_batch * HW * (NK * 5 + NB * 2 * 9);
        if (!over || p->arena_cap >= cap_max)
            break;
        const size_t cap = std::min(cap_max, std::max(p->arena_cap * 2, need + need / 4));
        HP_TRY(p->h_arena.alloc(cap * sizeof(float)));
        p->arena_cap = cap, ++p->arena_growths;
        HP_TRY(pifpaf_launch(p, p->last.n, p->last.paf, p->last.pif, p->last.fh, p->last.fw, 1, p->last.s));
        p->pending = 0;
        HP_HIP_TRY(hipEventSynchronize(p->done));
    }
    p->last_flags.assign(n, -1);
    if (p->device_decode)
        std::copy_n(p->h_counts.as<int>() + p->max_batch, n, p->last_flags.begin());
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

int hp_pifpaf_decode_flags(const hp_pifpaf* p, int* flags, int n)
{
    HP_REQUIRE(p && flags && n >= 0, HP_ERR_INVALID, "hp_pifpaf_decode_flags: bad argument");
    HP_REQUIRE(n <= (int)p->last_flags.size(), HP_ERR_INVALID, "hp_pifpaf_decode_flags: the last batch had %zu frames", p->last_flags.size());
    std::copy_n(p->last_flags.begin(), n, flags);
    return HP_OK;
}

int hp_pifpaf_process_batch(hp_pifpaf* p, int n, const float* paf, const float* pif, int fh, int fw, int on_device, hp_human* out,
    int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_pifpaf_process_batch: null argument");
    HP_TRY(pifpaf_launch(p, n, paf, pif, fh, fw, on_device, p->stream));
    return hp_pifpaf_collect(p, out, cap_per_frame, n_out);
}

} // extern "C"
