// ppn_parser.hip — hyperpose::parser::pose_proposal on gfx950 (replaces reference src/pose_proposal.cpp).
//
// Split: everything that touches the 855 KB/frame of network output runs on the GPU, one block
// per frame of the batch —
//   phase 1  threshold the [K,gh,gw] key-point confidences, decode the integer boxes (pose_proposal.cpp:144-158),
//            order-preserving ballot compaction per key-point class;
//   phase 2  per-class greedy IoU NMS with the reference's exact sequential semantics, including its
//            erase-without-index-fix-up (:110-133) — one lane per class, lists live in LDS;
//   phase 3  limb candidates: for every surviving "from" box, gather its 81 edge confidences from the
//            [E,nh,nw,gh,gw] tensor and match the aimed cell against the surviving "to" boxes (:172-221);
// the compacted lists stay in device memory, and the order-dependent tail runs on the device as well (ppn_assemble_kernel, one
// wavefront per frame): libstdc++'s std::sort restated step by step (equal limb confidences must come out in ITS order) + the
// pop-back selection with the root rule (:224-270), the 64x64 hash merge with its stale indices (:275-325), the score filter
// (:329-332); humans go straight to pinned host memory.  hp_ppn_collect does no per-frame work.  The same statements as host
// C++ (assemble_frame) remain for the frames the kernel declines (more than 2048 skeleton fragments or 8192 hash entries: reported
// by hp_ppn_decode_flags) and behind HP_PPN_HOST_TAIL=1, which the tests use to compare the two.
//
// Compiled with -ffp-contract=off; the float expressions keep the reference's operand order.
#include "hp_common.hpp"
#include "libstdcxx_sort.hpp"

#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <vector>

namespace {

constexpr int PPN_K = 18;       // key-point classes the tail indexes (human_t has 18 parts)
constexpr int PPN_LIMBS = 17;   // COCOPAIR_STD.size(), pose_proposal.cpp:23-41
constexpr int PPN_MAXG = 256;   // grid cells per map supported (12x12 = 144 in the reference model)
constexpr int PPN_MAXB = 144;   // NMS survivors kept per class (a 12x12 grid cannot yield more)
constexpr int PPN_MAXC = 2048;  // limb candidates kept per limb
constexpr int HDR = 64;         // ints per frame: [0,18) survivors, [18,35) candidates, [35] flags
constexpr int PPN_MAXP = 256;   // skeleton fragments (poses before the merge) the device tail keeps in LDS per frame
constexpr int PPN_MAXH = 2048;  // ... and in all: fragment i >= PPN_MAXP lives in a per-frame HBM scratch list (round 6: the reference's vector grows,
                                // src/pose_proposal.cpp:167-336; until round 6 a 257th fragment sent the frame to the host statements)
constexpr int PPN_MAXE = 8192;  // entries of the 64 x 64 spatial hash the device tail holds per frame (2048 until round 6)
constexpr int PPN_FLAG_POSES = 4, PPN_FLAG_HASH = 8, PPN_FLAG_OUT = 16; // device-tail decline reasons (on top of 1 / 2 from the extract kernel)

// pose_proposal.cpp:23-41
__constant__ int c_pair_std[PPN_LIMBS][2] = {
    { 1, 8 }, { 8, 9 }, { 9, 10 }, { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 2 }, { 2, 3 }, { 3, 4 }, { 1, 5 },
    { 5, 6 }, { 6, 7 }, { 1, 0 }, { 0, 14 }, { 0, 15 }, { 14, 16 }, { 15, 17 },
};
const int h_pair_std[PPN_LIMBS][2] = {
    { 1, 8 }, { 8, 9 }, { 9, 10 }, { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 2 }, { 2, 3 }, { 3, 4 }, { 1, 5 },
    { 5, 6 }, { 6, 7 }, { 1, 0 }, { 0, 14 }, { 0, 15 }, { 14, 16 }, { 15, 17 },
};

struct ppn_box {
    int grid;
    float conf;
    int x, y, w, h;
};
struct ppn_cand {
    int from, to;
    float conf;
};

struct ppn_geom {
    int K, gh, gw, E, nh, nw;
    int net_w, net_h;
    float point_thresh, limb_thresh, nms_thresh;
};

__device__ __forceinline__ float box_iou(const ppn_box& l, const ppn_box& r)
{
    // cv::Rect operator& + the lambda at pose_proposal.cpp:117-122
    const int x1 = max(l.x, r.x), y1 = max(l.y, r.y);
    const int iw = min(l.x + l.w, r.x + r.w) - x1, ih = min(l.y + l.h, r.y + r.h) - y1;
    const float int_area = (iw <= 0 || ih <= 0) ? 0 : iw * ih;
    const float union_area = l.w * l.h + r.w * r.h - int_area;
    return int_area / union_area;
}

__global__ __launch_bounds__(256) void ppn_extract_kernel(const float* __restrict__ conf_point, const float* __restrict__ tx,
    const float* __restrict__ ty, const float* __restrict__ tw, const float* __restrict__ th, const float* __restrict__ edge,
    ppn_geom g, int* __restrict__ hdr, ppn_box* __restrict__ out_boxes, ppn_cand* __restrict__ out_cands)
{
    // dynamic LDS: s_box [K][G] (thresholded boxes in cell order, later the NMS work list) | s_surv [K][MAXB]
    extern __shared__ __attribute__((aligned(16))) unsigned char ppn_smem[];
    __shared__ int s_cnt[PPN_K];
    __shared__ int s_nsurv[PPN_K];
    __shared__ int s_flags;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = g.gh * g.gw;
    ppn_box* const box_base = reinterpret_cast<ppn_box*>(ppn_smem);
    ppn_box* const surv_base = box_base + (size_t)PPN_K * G;
#define s_box(i) (box_base + (size_t)(i) * G)
#define s_surv(i) (surv_base + (size_t)(i) * PPN_MAXB)
    const size_t map_off = (size_t)f * g.K * G;
    if (tid == 0)
        s_flags = 0;
    __syncthreads();

    // ---- phase 1: threshold + box decode, one wave per class, cells in order
    for (int i = wave; i < PPN_K && i < g.K; i += 4) {
        int n = 0;
        for (int base = 0; base < G; base += 64) {
            const int j = base + lane;
            bool hit = false;
            ppn_box b;
            if (j < G) {
                const size_t idx = map_off + (size_t)i * G + j;
                const float c = conf_point[idx];
                if (g.point_thresh < c) {
                    hit = true;
                    const float x = tx[idx], y = ty[idx], w = tw[idx], h = th[idx];
                    b.grid = j;
                    b.conf = c;
                    b.x = max(min(g.net_w, static_cast<int>(x - w / 2)), 0);
                    b.y = max(min(g.net_h, static_cast<int>(y - h / 2)), 0);
                    b.w = max(min(g.net_w, static_cast<int>(w)), 0);
                    b.h = max(min(g.net_h, static_cast<int>(h)), 0);
                }
            }
            const unsigned long long m = __ballot(hit);
            if (hit)
                s_box(i)[n + __popcll(m & ((1ull << lane) - 1ull))] = b;
            n += __popcll(m);
        }
        if (lane == 0)
            s_cnt[i] = n;
    }
    __syncthreads();

    // ---- phase 2: NMS, one lane per class (spread over the waves), sequential semantics of :110-133
    if ((tid & 7) == 0 && (tid >> 3) < PPN_K && (tid >> 3) < g.K) {
        const int i = tid >> 3;
        ppn_box* arr = s_box(i);
        int m = s_cnt[i];
        // std::sort ascending by conf (stable insertion sort: equal confidences keep cell order)
        for (int a = 1; a < m; ++a) {
            const ppn_box key = arr[a];
            int q = a - 1;
            while (q >= 0 && key.conf < arr[q].conf) {
                arr[q + 1] = arr[q];
                --q;
            }
            arr[q + 1] = key;
        }
        int ns = 0;
        while (m > 0) {
            const ppn_box cur = arr[m - 1];
            --m;
            if (ns < PPN_MAXB)
                s_surv(i)[ns] = cur;
            else
                atomicOr(&s_flags, 1);
            ++ns;
            for (int k = 0; k < m; ++k)
                if (box_iou(cur, arr[k]) >= g.nms_thresh) {
                    for (int q = k; q < m - 1; ++q) // boxes.erase(begin + k); the loop index is NOT decremented
                        arr[q] = arr[q + 1];
                    --m;
                }
        }
        s_nsurv[i] = min(ns, PPN_MAXB);
    }
    __syncthreads();

    // ---- phase 3: limb candidates in (from_index, neighbour j) order, one wave per limb
    const int NB = g.nh * g.nw;
    const int n_range = min(g.E, PPN_LIMBS);
    for (int l = wave; l < PPN_LIMBS; l += 4) {
        int nc = 0;
        if (l < n_range) {
            const int p1 = c_pair_std[l][0], p2 = c_pair_std[l][1];
            const int nf = s_nsurv[p1], nt = s_nsurv[p2];
            const float* e_l = edge + ((size_t)f * g.E + l) * (size_t)NB * G;
            for (int fi = 0; fi < nf; ++fi) {
                const int from_grid = s_surv(p1)[fi].grid;
                const int fy = from_grid / g.gw, fx = from_grid - fy * g.gw;
                for (int base = 0; base < NB; base += 64) {
                    const int j = base + lane;
                    bool hit = false;
                    ppn_cand c;
                    if (j < NB) {
                        const int ny = j / g.nw, nx = j - g.nw * ny;
                        const int ay = fy + ny - g.nh / 2, ax = fx + nx - g.nw / 2; // size_t wrap-around == range check
                        const float e = e_l[(size_t)j * G + from_grid];
                        if (ax >= 0 && ax < g.gw && ay >= 0 && ay < g.gh && e > g.limb_thresh) {
                            const int aim = ay * g.gw + ax;
                            for (int ti = 0; ti < nt; ++ti)
                                if (s_surv(p2)[ti].grid == aim) { // at most one box per cell and class
                                    hit = true;
                                    c.from = fi, c.to = ti, c.conf = e;
                                    break;
                                }
                        }
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const int pos = nc + __popcll(m & ((1ull << lane) - 1ull));
                        if (pos < PPN_MAXC)
                            out_cands[((size_t)f * PPN_LIMBS + l) * PPN_MAXC + pos] = c;
                        else
                            atomicOr(&s_flags, 2);
                    }
                    nc += __popcll(m);
                }
            }
        }
        if (lane == 0)
            hdr[f * HDR + PPN_K + l] = min(nc, PPN_MAXC);
    }
    __syncthreads();
    for (int i = tid; i < PPN_K * PPN_MAXB; i += 256) {
        const int c = i / PPN_MAXB, k = i - c * PPN_MAXB;
        if (c < g.K && k < s_nsurv[c])
            out_boxes[((size_t)f * PPN_K + c) * PPN_MAXB + k] = s_surv(c)[k];
    }
    if (tid < PPN_K)
        hdr[f * HDR + tid] = tid < g.K ? s_nsurv[tid] : 0;
    if (tid == 0)
        hdr[f * HDR + PPN_K + PPN_LIMBS] = s_flags;
#undef s_box
#undef s_surv
}

// ---- device tail: pose_proposal.cpp:224-332 on the compacted lists, ONE WAVEFRONT PER FRAME ------------------------------
// Every step mutates what the next one reads (roots, the pose list, the hash buckets with their stale indices), so the walk is one
// lane's; the other 63 lanes take the bulk work around it: staging a limb's confidences, shifting the pose list after an erase,
// the final filter + copy to pinned host memory.  State in LDS: roots [18][144], poses [256] x 292 B, the hash as per-bucket
// singly linked lists in insertion order (the reference's `std::vector<uint16_t>` buckets, traversed front to back).
__global__ __launch_bounds__(64) void ppn_assemble_kernel(const int* __restrict__ hdr, const ppn_box* __restrict__ boxes,
    const ppn_cand* __restrict__ cands, int net_w, int net_h, int n_key_points, hp_human* __restrict__ out, int out_cap, int* __restrict__ out_n,
    int* __restrict__ out_flags, hp_human* __restrict__ spill)
{
    __shared__ int s_root[PPN_K][PPN_MAXB];
    __shared__ hp_human s_pose[PPN_MAXP];
    __shared__ int s_ord[PPN_MAXC];
    __shared__ float s_conf[PPN_MAXC];
    __shared__ unsigned short s_head[64 * 64], s_tail[64 * 64], s_val[PPN_MAXE], s_next[PPN_MAXE];
    __shared__ int s_np, s_flags, s_cmd, s_arg, s_i, s_ne;
    // ~150 KB of static LDS (the first 256 fragments alone are 256 x 292 B): this kernel is written for gfx950's 160 KB per CU (one block per CU)
    static_assert(sizeof(s_root) + sizeof(s_pose) + sizeof(s_ord) + sizeof(s_conf) + sizeof(s_head) + sizeof(s_tail) + sizeof(s_val) + sizeof(s_next) + 64
            <= 160 * 1024, "ppn_assemble_kernel needs the 160 KB LDS of gfx950");
    const int f = blockIdx.x, lane = threadIdx.x;
    // fragment i: LDS below PPN_MAXP, this frame's slice of the HBM scratch list above (one wavefront works on a frame: program order and the
    // barriers below order its own reads and writes there).  A spilled slot is zeroed when it is handed out.
    hp_human* const sp = spill + (size_t)f * (PPN_MAXH - PPN_MAXP);
    auto pose = [&](int i) -> hp_human& { return i < PPN_MAXP ? s_pose[i] : sp[i - PPN_MAXP]; };
    auto pose_words = [&](int i) -> unsigned* { return reinterpret_cast<unsigned*>(i < PPN_MAXP ? &s_pose[i] : &sp[i - PPN_MAXP]); };
    const int* const h = hdr + (size_t)f * HDR;
    const ppn_box* const fb = boxes + (size_t)f * PPN_K * PPN_MAXB;
    const ppn_cand* const fc = cands + (size_t)f * PPN_LIMBS * PPN_MAXC;
    for (int i = lane; i < PPN_K * PPN_MAXB; i += 64)
        (&s_root[0][0])[i] = -1;
    for (int i = lane; i < (int)(sizeof(s_pose) / 4); i += 64)
        reinterpret_cast<unsigned*>(s_pose)[i] = 0u;
    for (int i = lane; i < 64 * 64; i += 64)
        s_head[i] = 0xffff, s_tail[i] = 0xffff;
    if (lane == 0)
        s_np = 0, s_flags = h[PPN_K + PPN_LIMBS], s_ne = 0, s_i = 0;
    __syncthreads();
    if (s_flags) { // the extract kernel overflowed a list: nothing to assemble (hp_ppn_collect reports the frame)
        if (lane == 0)
            out_n[f] = 0, out_flags[f] = s_flags;
        return;
    }
    auto set_part = [&](hp_human& hm, int part, const ppn_box& b) {
        hm.parts[part].has_value = 1;
        hm.parts[part].x = (float)(b.x + b.w / 2) / net_w; // integer /2, pose_proposal.cpp:252-253
        hm.parts[part].y = (float)(b.y + b.h / 2) / net_h;
        hm.parts[part].score = b.conf;
    };

    // ---- limbs (:172-270): std::sort ascending by confidence, pop from the back, root rule
    for (int l = 0; l < PPN_LIMBS; ++l) {
        const int nc = h[PPN_K + l];
        const ppn_cand* const lc = fc + (size_t)l * PPN_MAXC;
        for (int i = lane; i < nc; i += 64)
            s_conf[i] = lc[i].conf, s_ord[i] = i;
        __syncthreads();
        if (lane == 0 && nc > 0) {
            const float* const cf = s_conf;
            hp::libstdcxx_sort(s_ord, nc, [cf](int a, int b) { return cf[a] < cf[b]; });
            const int p1 = c_pair_std[l][0], p2 = c_pair_std[l][1];
            int np = s_np;
            for (int q = nc - 1; q >= 0; --q) {
                const ppn_cand cur = lc[s_ord[q]];
                int& fr = s_root[p1][cur.from];
                int& tr = s_root[p2][cur.to];
                int root;
                if ((fr != -1) == (tr != -1)) { // both rooted OR both free -> a new pose (:241-244)
                    if (np >= PPN_MAXH) {
                        s_flags |= PPN_FLAG_POSES;
                        break;
                    }
                    root = np++;
                    if (root >= PPN_MAXP) { // a spilled slot: zero it like the LDS slots were
                        unsigned* const z = pose_words(root);
                        for (int k = 0; k < (int)(sizeof(hp_human) / 4); ++k)
                            z[k] = 0u;
                    }
                } else
                    root = fr != -1 ? fr : tr;
                hp_human& hm = pose(root);
                if (!hm.parts[p1].has_value) {
                    set_part(hm, p1, fb[(size_t)p1 * PPN_MAXB + cur.from]);
                    fr = root;
                    hm.score += 1.;
                }
                if (!hm.parts[p2].has_value) {
                    set_part(hm, p2, fb[(size_t)p2 * PPN_MAXB + cur.to]);
                    tr = root;
                    hm.score += 1.;
                }
            }
            s_np = np;
        }
        __syncthreads();
        if (s_flags)
            break;
    }

    // ---- merge pass (:275-325).  Lane 0 walks the poses; when one is to be erased it hands the shift of the tail of the list to all
    // lanes (s_cmd = 1, s_arg = the index) and resumes at the same index, as the reference's `--i; break;` + `++i` does.
    auto bucket_of = [](const hp_body_part& p) {
        size_t xi = p.x * 64, yi = p.y * 64; // size_t x_ind = part.x * grid_size (:279-283)
        xi = xi == 64 ? 63 : xi;
        yi = yi == 64 ? 63 : yi;
        return (int)(min(xi, (size_t)63) * 64 + min(yi, (size_t)63)); // (coordinates beyond 1.0: the reference indexes out of bounds)
    };
    auto push_back = [&](int bucket, int value) { // hash_table[..][..].push_back(value)
        const int e = s_ne;
        if (e >= PPN_MAXE) {
            s_flags |= PPN_FLAG_HASH;
            return;
        }
        s_val[e] = (unsigned short)value, s_next[e] = 0xffff;
        if (s_tail[bucket] == 0xffff)
            s_head[bucket] = (unsigned short)e;
        else
            s_next[s_tail[bucket]] = (unsigned short)e;
        s_tail[bucket] = (unsigned short)e;
        s_ne = e + 1;
    };
    while (!s_flags) {
        if (lane == 0) {
            s_cmd = 0;
            int i = s_i;
            int np = s_np;
            for (; i < np && !s_flags; ++i) {
                hp_human& cur = pose(i);
                if (cur.score > n_key_points - 0.1)
                    continue;
                bool remove_cur = false;
                for (int j = 0; j < HP_COCO_N_PARTS && !remove_cur; ++j) {
                    if (!cur.parts[j].has_value)
                        continue;
                    const hp_body_part this_part = cur.parts[j];
                    const int bk = bucket_of(this_part);
                    for (int e = s_head[bk]; e != 0xffff; e = s_next[e]) {
                        const int pid = s_val[e];
                        if (pid == i || pid >= np) // (an index past the end: the reference reads out of bounds)
                            continue;
                        hp_human& other = pose(pid);
                        if (other.parts[j].y != this_part.y || other.parts[j].x != this_part.x)
                            continue;
                        remove_cur = true;
                        for (int u = 0; u < HP_COCO_N_PARTS; ++u)
                            if (cur.parts[u].has_value && !other.parts[u].has_value) {
                                other.parts[u] = cur.parts[u];
                                other.score += 1.0;
                                push_back(bucket_of(cur.parts[u]), i); // the reference pushes `i`, not the survivor (:310)
                            }
                        break;
                    }
                    if (!remove_cur)
                        push_back(bk, i);
                }
                if (remove_cur) {
                    s_cmd = 1, s_arg = i; // ret_poses.erase(begin + i); --i; (the loop's ++i lands on the same index)
                    break;
                }
            }
            s_i = i;
        }
        __syncthreads();
        if (s_cmd != 1)
            break;
        { // erase pose s_arg: shift the tail of the list down by one, word by word, in index order
            const int e = s_arg, np = s_np;
            constexpr int WPP = (int)(sizeof(hp_human) / 4);
            // (every lane moves the same two word columns of every row: a row's words are read one iteration before they are
            // overwritten, by the same lane - no cross-lane hazard, no barrier inside the loop)
            for (int k = e; k + 1 < np; ++k) {
                unsigned* const wd = pose_words(k);
                const unsigned* const ws = pose_words(k + 1);
                if (lane < WPP)
                    wd[lane] = ws[lane];
                if (lane + 64 < WPP)
                    wd[lane + 64] = ws[lane + 64];
            }
            if (lane == 0)
                s_np = np - 1;
        }
        __syncthreads();
    }

    // ---- score filter (:329-332), order kept; humans -> pinned host memory
    int n_out = 0;
    if (!s_flags) {
        const int np = s_np;
        constexpr int WPP = (int)(sizeof(hp_human) / 4);
        unsigned* const o = reinterpret_cast<unsigned*>(out + (size_t)f * out_cap);
        for (int k = 0; k < np; ++k) {
            if (!(pose(k).score <= 3)) { // uniform
                if (n_out < out_cap) {
                    const unsigned* const w = pose_words(k);
                    if (lane < WPP)
                        o[n_out * WPP + lane] = w[lane];
                    if (lane + 64 < WPP)
                        o[n_out * WPP + lane + 64] = w[lane + 64];
                } else if (lane == 0)
                    s_flags |= PPN_FLAG_OUT;
                ++n_out;
            }
        }
    }
    __syncthreads();
    if (lane == 0) {
        out_n[f] = n_out;
        out_flags[f] = s_flags;
    }
}

// ---- host tail: pose_proposal.cpp:167-336 on the compacted lists -----------------------------------------
struct kp_t {
    ppn_box box;
    int root = -1;
};

int assemble_frame(const int* hdr, const ppn_box* boxes, const ppn_cand* cands, int net_w, int net_h, int n_key_points,
    std::vector<hp_human>& poses)
{
    std::vector<std::vector<kp_t>> key_points(PPN_K);
    for (int c = 0; c < PPN_K; ++c)
        for (int k = 0; k < hdr[c]; ++k)
            key_points[c].push_back(kp_t{ boxes[(size_t)c * PPN_MAXB + k], -1 });

    auto blank = [] {
        hp_human h;
        std::memset(&h, 0, sizeof(h));
        return h;
    };
    auto set_part = [&](hp_human& h, int part, const kp_t& k) {
        h.parts[part].has_value = 1;
        h.parts[part].x = (float)(k.box.x + k.box.w / 2) / net_w; // integer /2, pose_proposal.cpp:252-253
        h.parts[part].y = (float)(k.box.y + k.box.h / 2) / net_h;
        h.parts[part].score = k.box.conf;
    };

    for (int l = 0; l < PPN_LIMBS; ++l) {
        const int p1 = h_pair_std[l][0], p2 = h_pair_std[l][1];
        std::vector<ppn_cand> lc(cands + (size_t)l * PPN_MAXC, cands + (size_t)l * PPN_MAXC + hdr[PPN_K + l]);
        std::sort(lc.begin(), lc.end(), [](const ppn_cand& a, const ppn_cand& b) { return a.conf < b.conf; }); // :224
        while (!lc.empty()) {
            const ppn_cand cur = lc.back();
            lc.pop_back();
            kp_t& fv = key_points[p1][cur.from];
            kp_t& tv = key_points[p2][cur.to];
            size_t root;
            if ((fv.root != -1) == (tv.root != -1)) { // both rooted OR both free -> a new pose (:241-244)
                poses.push_back(blank());
                root = poses.size() - 1;
            } else
                root = fv.root != -1 ? fv.root : tv.root;
            if (!poses[root].parts[p1].has_value) {
                set_part(poses[root], p1, fv);
                fv.root = (int)root;
                poses[root].score += 1.;
            }
            if (!poses[root].parts[p2].has_value) {
                set_part(poses[root], p2, tv);
                tv.root = (int)root;
                poses[root].score += 1.;
            }
        }
    }

    // merge pass (:275-325): 64x64 spatial hash of part positions; indices in the buckets go stale after an erase,
    // exactly like the reference (an index past the end is skipped here; the reference reads out of bounds).
    constexpr size_t GS = 64;
    std::vector<std::vector<uint16_t>> table(GS * GS);
    auto bucket = [&](const hp_body_part& p) -> std::vector<uint16_t>& {
        size_t xi = p.x * GS, yi = p.y * GS;
        xi = xi == GS ? GS - 1 : xi;
        yi = yi == GS ? GS - 1 : yi;
        return table[std::min(xi, GS - 1) * GS + std::min(yi, GS - 1)];
    };
    for (size_t i = 0; i < poses.size(); ++i) {
        if (poses[i].score > n_key_points - 0.1)
            continue;
        for (size_t j = 0; j < HP_COCO_N_PARTS; ++j) {
            if (!poses[i].parts[j].has_value)
                continue;
            const hp_body_part this_part = poses[i].parts[j];
            std::vector<uint16_t>& maybes = bucket(this_part);
            bool remove_cur = false;
            for (size_t q = 0; q < maybes.size(); ++q) {
                const uint16_t pid = maybes[q];
                if (pid == i || pid >= poses.size())
                    continue;
                hp_human& other = poses[pid];
                if (other.parts[j].y != this_part.y || other.parts[j].x != this_part.x)
                    continue;
                remove_cur = true;
                const hp_human cur = poses[i];
                for (size_t u = 0; u < HP_COCO_N_PARTS; ++u)
                    if (cur.parts[u].has_value && !other.parts[u].has_value) {
                        other.parts[u] = cur.parts[u];
                        other.score += 1.0;
                        bucket(cur.parts[u]).push_back((uint16_t)i); // the reference pushes `i`, not the survivor (:310)
                    }
                poses.erase(poses.begin() + i);
                --i;
                break;
            }
            if (remove_cur)
                break;
            maybes.push_back((uint16_t)i);
        }
    }
    poses.erase(std::remove_if(poses.begin(), poses.end(), [](const hp_human& p) { return p.score <= 3; }), poses.end()); // :329-332
    return HP_OK;
}

} // namespace

struct hp_ppn {
    int net_w, net_h, max_batch;
    float point_thresh, limb_thresh, nms_thresh;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hp::dev_buf in[7];
    // compacted lists: device memory (read by ppn_assemble_kernel); the host copies are filled on demand only (a declined frame,
    // HP_PPN_HOST_TAIL=1)
    hp::dev_buf d_hdr, d_boxes, d_cands;
    std::vector<int> h_hdr;
    std::vector<ppn_box> h_boxes;
    std::vector<ppn_cand> h_cands;
    hp::host_buf h_humans, h_counts; // pinned: [B][PPN_MAXH] humans, [n_humans(B) | flags(B)], written by ppn_assemble_kernel
    hp::dev_buf d_spill;             // [B][PPN_MAXH - PPN_MAXP] fragments beyond the LDS list
    std::vector<int> last_flags;
    bool host_tail = false; // HP_PPN_HOST_TAIL=1: the device tail is not launched, every frame takes the host statements
    int pending = 0;        // frames enqueued and not yet collected
    int K = 0;
};

extern "C" {

int hp_ppn_create(hp_ppn** out, int net_w, int net_h, float point_thresh, float limb_thresh, float nms_thresh, int max_batch)
{
    HP_REQUIRE(out && net_w > 0 && net_h > 0 && max_batch >= 1, HP_ERR_INVALID, "hp_ppn_create: bad argument");
    std::unique_ptr<hp_ppn> p(new hp_ppn());
    p->net_w = net_w, p->net_h = net_h, p->max_batch = max_batch;
    p->point_thresh = point_thresh, p->limb_thresh = limb_thresh, p->nms_thresh = nms_thresh;
    HP_HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HP_HIP_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    const size_t B = max_batch;
    HP_TRY(p->d_hdr.alloc(B * HDR * sizeof(int)));
    HP_TRY(p->d_boxes.alloc(B * PPN_K * PPN_MAXB * sizeof(ppn_box)));
    HP_TRY(p->d_cands.alloc(B * PPN_LIMBS * PPN_MAXC * sizeof(ppn_cand)));
    HP_TRY(p->h_humans.alloc(B * PPN_MAXH * sizeof(hp_human)));
    HP_TRY(p->d_spill.alloc(B * (PPN_MAXH - PPN_MAXP) * sizeof(hp_human)));
    HP_TRY(p->h_counts.alloc(2 * B * sizeof(int)));
    p->h_hdr.assign(B * HDR, 0);
    p->last_flags.assign(B, 0);
    p->host_tail = getenv("HP_PPN_HOST_TAIL") && atoi(getenv("HP_PPN_HOST_TAIL")) != 0;
    *out = p.release();
    return HP_OK;
}

void hp_ppn_destroy(hp_ppn* p)
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

void* hp_ppn_stream(hp_ppn* p) { return p ? (void*)p->stream : nullptr; }

int hp_ppn_set_thresholds(hp_ppn* p, float point_thresh, float limb_thresh, float nms_thresh)
{
    HP_REQUIRE(p, HP_ERR_INVALID, "null parser");
    p->point_thresh = point_thresh, p->limb_thresh = limb_thresh, p->nms_thresh = nms_thresh;
    return HP_OK;
}

static int ppn_launch(hp_ppn* p, int n, const float* const tensors[7], const int conf_shape[3], const int edge_shape[5], int on_device,
    hipStream_t s)
{
    HP_REQUIRE(p && tensors && conf_shape && edge_shape, HP_ERR_INVALID, "hp_ppn: null argument");
    HP_REQUIRE(n >= 1 && n <= p->max_batch, HP_ERR_CAPACITY, "hp_ppn: batch %d > max_batch %d", n, p->max_batch);
    HP_REQUIRE(p->pending == 0, HP_ERR_STATE, "hp_ppn: a batch is already in flight, collect it first");
    ppn_geom g;
    g.K = conf_shape[0], g.gh = conf_shape[1], g.gw = conf_shape[2];
    g.E = edge_shape[0], g.nh = edge_shape[1], g.nw = edge_shape[2];
    g.net_w = p->net_w, g.net_h = p->net_h;
    g.point_thresh = p->point_thresh, g.limb_thresh = p->limb_thresh, g.nms_thresh = p->nms_thresh;
    HP_REQUIRE(g.K >= 1 && g.K <= PPN_K && g.gh * g.gw <= PPN_MAXG && g.gh >= 1 && g.gw >= 1, HP_ERR_INVALID,
        "ppn: conf shape [%d,%d,%d] unsupported (<= %d classes, <= %d cells)", g.K, g.gh, g.gw, PPN_K, PPN_MAXG);
    HP_REQUIRE(edge_shape[3] == g.gh && edge_shape[4] == g.gw && g.nh >= 1 && g.nw >= 1, HP_ERR_INVALID, "ppn: edge shape does not match the grid");
    const float* d[7];
    for (int t = 0; t < 7; ++t) {
        HP_REQUIRE(tensors[t], HP_ERR_INVALID, "ppn: tensor %d is null", t);
        d[t] = tensors[t];
    }
    if (!on_device) {
        const size_t map_b = (size_t)g.K * g.gh * g.gw * sizeof(float), edge_b = (size_t)g.E * g.nh * g.nw * g.gh * g.gw * sizeof(float);
        for (int t = 0; t < 7; ++t) {
            const size_t per = t == 6 ? edge_b : map_b;
            if (p->in[t].bytes < per * p->max_batch)
                HP_TRY(p->in[t].alloc(per * p->max_batch));
            HP_HIP_TRY(hipMemcpyAsync(p->in[t].p, tensors[t], per * n, hipMemcpyHostToDevice, s));
            d[t] = p->in[t].as<float>();
        }
    }
    // tensor order: 0 conf_point, 1 conf_iou (ignored, pose_proposal.cpp:74), 2 x, 3 y, 4 w, 5 h, 6 edge
    const size_t lds = ((size_t)PPN_K * g.gh * g.gw + (size_t)PPN_K * PPN_MAXB) * sizeof(ppn_box);
    HP_REQUIRE(lds <= 150 * 1024, HP_ERR_INVALID, "ppn: grid too large for the LDS work lists");
    HP_HIP_TRY(hipFuncSetAttribute((const void*)ppn_extract_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ppn_extract_kernel, dim3(n), dim3(256), lds, s, d[0], d[2], d[3], d[4], d[5], d[6], g, p->d_hdr.as<int>(),
        p->d_boxes.as<ppn_box>(), p->d_cands.as<ppn_cand>());
    HP_HIP_TRY(hipGetLastError());
    if (!p->host_tail) {
        hipLaunchKernelGGL(ppn_assemble_kernel, dim3(n), dim3(64), 0, s, p->d_hdr.as<int>(), p->d_boxes.as<ppn_box>(), p->d_cands.as<ppn_cand>(),
            p->net_w, p->net_h, g.K, p->h_humans.as<hp_human>(), PPN_MAXH, p->h_counts.as<int>(), p->h_counts.as<int>() + p->max_batch, p->d_spill.as<hp_human>());
        HP_HIP_TRY(hipGetLastError());
    }
    HP_HIP_TRY(hipEventRecord(p->done, s));
    p->pending = n;
    p->K = g.K;
    return HP_OK;
}

int hp_ppn_enqueue(hp_ppn* p, int n, const float* const dev_tensors[7], const int conf_shape[3], const int edge_shape[5], void* stream)
{
    return ppn_launch(p, n, dev_tensors, conf_shape, edge_shape, 1, stream ? (hipStream_t)stream : (p ? p->stream : nullptr));
}

namespace {
struct ppn_job {
    hp_ppn* p;
    hp_human* out;
    int cap;
    int* n_out;
    std::vector<int> frames; // the frames that take the host statements
    std::vector<int> rc;
};
void ppn_frame(int k, int, void* ctx)
{
    ppn_job& j = *static_cast<ppn_job*>(ctx);
    hp_ppn* p = j.p;
    const int f = j.frames[k];
    const int* hdr = p->h_hdr.data() + (size_t)f * HDR;
    std::vector<hp_human> poses;
    assemble_frame(hdr, p->h_boxes.data() + (size_t)f * PPN_K * PPN_MAXB, p->h_cands.data() + (size_t)f * PPN_LIMBS * PPN_MAXC, p->net_w, p->net_h,
        p->K, poses);
    j.n_out[f] = (int)poses.size();
    if ((int)poses.size() > j.cap)
        j.rc[f] = 2;
    if (j.out)
        std::copy(poses.begin(), poses.begin() + std::min<size_t>(poses.size(), j.cap), j.out + (size_t)f * j.cap);
}
} // namespace

int hp_ppn_collect(hp_ppn* p, hp_human* out, int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_ppn_collect: null argument");
    HP_REQUIRE(p->pending > 0, HP_ERR_STATE, "hp_ppn_collect: nothing was enqueued");
    const int n = p->pending;
    p->pending = 0;
    HP_HIP_TRY(hipEventSynchronize(p->done));
    ppn_job job{ p, out, cap_per_frame, n_out, {}, std::vector<int>(n, 0) };
    const int* counts = p->h_counts.as<int>();
    const int* flags = counts + p->max_batch;
    for (int f = 0; f < n; ++f) {
        const int fl = p->host_tail ? 0 : flags[f];
        p->last_flags[f] = p->host_tail ? -1 : fl;
        if (p->host_tail || (fl & (PPN_FLAG_POSES | PPN_FLAG_HASH | PPN_FLAG_OUT))) {
            job.frames.push_back(f); // declined by the device tail (or HP_PPN_HOST_TAIL=1): the host statements take the frame
            continue;
        }
        if (fl) { // an extract-kernel list overflowed
            job.rc[f] = 1;
            n_out[f] = 0;
            continue;
        }
        const int nh = counts[f];
        n_out[f] = nh;
        if (nh > cap_per_frame)
            job.rc[f] = 2;
        if (out)
            memcpy(out + (size_t)f * cap_per_frame, p->h_humans.as<hp_human>() + (size_t)f * PPN_MAXH, sizeof(hp_human) * std::min(nh, cap_per_frame));
    }
    if (!job.frames.empty()) {
        // rare path: fetch the compacted lists of the batch and run the reference's statements on the host
        if (p->h_boxes.empty())
            p->h_boxes.resize((size_t)p->max_batch * PPN_K * PPN_MAXB), p->h_cands.resize((size_t)p->max_batch * PPN_LIMBS * PPN_MAXC);
        // (on the parser's own stream, then one wait: a plain hipMemcpy runs on the legacy stream and fails - in this thread AND in the
        //  capturing one - while another pipe's engine is recording its hipGraph; the batch's kernels completed before collect got here)
        HP_HIP_TRY(hipMemcpyAsync(p->h_hdr.data(), p->d_hdr.p, (size_t)n * HDR * sizeof(int), hipMemcpyDeviceToHost, p->stream));
        HP_HIP_TRY(hipMemcpyAsync(p->h_boxes.data(), p->d_boxes.p, (size_t)n * PPN_K * PPN_MAXB * sizeof(ppn_box), hipMemcpyDeviceToHost, p->stream));
        HP_HIP_TRY(hipMemcpyAsync(p->h_cands.data(), p->d_cands.p, (size_t)n * PPN_LIMBS * PPN_MAXC * sizeof(ppn_cand), hipMemcpyDeviceToHost, p->stream));
        HP_HIP_TRY(hipStreamSynchronize(p->stream));
        std::vector<int> todo;
        for (int f : job.frames) {
            if (p->h_hdr[(size_t)f * HDR + PPN_K + PPN_LIMBS] != 0) { // (host-tail mode: the extract kernel's own overflow flags)
                job.rc[f] = 1;
                n_out[f] = 0;
            } else
                todo.push_back(f);
        }
        job.frames = todo;
        hp::frame_pool::instance().run((int)job.frames.size(), ppn_frame, &job);
    }
    int rc = HP_OK;
    for (int f = 0; f < n; ++f)
        if (job.rc[f] == 1) {
            hp::set_error("ppn: frame %d overflowed a device list (1=survivors/class>%d, 2=candidates/limb>%d)", f, PPN_MAXB, PPN_MAXC);
            rc = HP_ERR_CAPACITY;
        } else if (job.rc[f] == 2) {
            hp::set_error("ppn: frame %d has %d humans, capacity %d", f, n_out[f], cap_per_frame);
            rc = HP_ERR_CAPACITY;
        }
    return rc;
}

int hp_ppn_decode_flags(hp_ppn* p, int* flags, int n)
{
    HP_REQUIRE(p && flags && n >= 0 && n <= p->max_batch, HP_ERR_INVALID, "hp_ppn_decode_flags: bad argument");
    for (int f = 0; f < n; ++f)
        flags[f] = p->last_flags[f];
    return HP_OK;
}

int hp_ppn_process_batch(hp_ppn* p, int n, const float* const tensors[7], const int conf_shape[3], const int edge_shape[5],
    int on_device, hp_human* out, int cap_per_frame, int* n_out)
{
    HP_REQUIRE(p && n_out, HP_ERR_INVALID, "hp_ppn_process_batch: null argument");
    HP_TRY(ppn_launch(p, n, tensors, conf_shape, edge_shape, on_device, p->stream));
    return hp_ppn_collect(p, out, cap_per_frame, n_out);
}

} // extern "C"
