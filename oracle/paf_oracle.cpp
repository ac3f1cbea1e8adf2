/* oracle/paf_oracle.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference PAF parser:
 *   hyperpose::parser::paf::process           src/paf.cpp:300-375
 *   resize_area / smooth / same_max_pool_3x3  src/post_process.hpp:26-102
 *   peak_finder_t::find_peak_coords/group_by  src/post_process.hpp:147-205
 *   get_paf_vectors / get_connection_candidates / get_connections / get_humans   src/paf.cpp:67-272
 *   COCOPAIRS / COCOPAIRS_NET / is_virtual_pair                                   src/coco.hpp:6-51
 *
 * Status (round 3): the reference's own src/paf.cpp + src/post_process.hpp ARE compiled here, verbatim, behind the
 * container / header shims of oracle/shim (oracle/ref_paf_wrap.cpp -> oracle/_ref/libhp_ref.so: `ref_paf_process`), and that
 * is what the GPU parity tests and the golden vectors compare with.  This file has two jobs left:
 *  (1) the two OpenCV 4.4.0 calls of that path - the ONLY third-party ARITHMETIC in it, OpenCV is not in the image - are
 *      restated here and the shim's cv::resize / cv::GaussianBlur forward to them (oracle_resize_area_1ch /
 *      oracle_gaussian_blur_1ch):
 *        cv::resize(..., INTER_AREA) when up-scaling   (imgproc/resize.cpp: the "area_mode" branch of the linear
 *            resizer: HResizeLinear + VResizeLinear, float)
 *        cv::GaussianBlur(k=17x17, sigma=3)            (imgproc/smooth.dispatch.cpp -> sepFilter2D:
 *            RowFilter<float,float> + SymmColumnFilter<float>, BORDER_REFLECT_101)
 *      with NON-fused fp32 arithmetic in the scalar (non-SIMD) evaluation order.  PARITY UNPINNED for these two calls only:
 *      the reference ships no golden vectors for this path and OpenCV is not available to cross-check
 *      (tests/test_paf_envelope.py bounds what an FMA build of OpenCV could change: HP_ORACLE_VARIANT below);
 *  (2) a line-by-line restatement of the rest of the parser (oracle_paf_process), kept as an independent second checker and
 *      asserted byte-equal to the reference-compiled one on every fixture (tests/test_paf_oracle.py).
 *
 * Build with: g++ -O2 -ffp-contract=off (strict IEEE; the reference ships -Ofast).
 */
#include "oracle_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <utility>
#include <vector>

/* HP_ORACLE_VARIANT: how a real OpenCV 4.4.0 binary may round the same published algorithm (tests/test_paf_envelope.py):
 *   0  scalar, non-fused: a*b + c*d, s += k*x                                (the parity oracle; -ffp-contract=off)
 *   1  OpenCV's universal-intrinsic forms on an FMA target (AVX2 dispatch): VResizeLinearVec_32f = v_muladd(S0, b0, S1*b1),
 *      RowVec_32f / SymmColumnVec_32f = v_muladd(x, k, s), HResizeLinear (plain C++) contracted by the compiler the same way
 *   2  the other association a compiler may pick when contracting a*b + c*d: fma(c, d, a*b)
 * Variants 1 and 2 are compiled with -mfma; nothing but the two OpenCV restatements depends on the variant. */
#ifndef HP_ORACLE_VARIANT
#define HP_ORACLE_VARIANT 0
#endif

namespace {

#if HP_ORACLE_VARIANT == 0
inline float madd2(float a, float b, float c, float d) { return a * b + c * d; }
inline float macc(float k, float x, float s) { return s + k * x; }
#elif HP_ORACLE_VARIANT == 1
inline float madd2(float a, float b, float c, float d) { return std::fma(a, b, c * d); }
inline float macc(float k, float x, float s) { return std::fma(k, x, s); }
#else
inline float madd2(float a, float b, float c, float d) { return std::fma(c, d, a * b); }
inline float macc(float k, float x, float s) { return std::fma(k, x, s); }
#endif

/* src/coco.hpp:10-30 */
const int COCOPAIRS_NET[19][2] = {
    { 12, 13 }, { 20, 21 }, { 14, 15 }, { 16, 17 }, { 22, 23 }, { 24, 25 }, { 0, 1 }, { 2, 3 },
    { 4, 5 }, { 6, 7 }, { 8, 9 }, { 10, 11 }, { 28, 29 }, { 30, 31 }, { 34, 35 }, { 32, 33 },
    { 36, 37 }, { 18, 19 }, { 26, 27 },
};
/* src/coco.hpp:32-51 */
const int COCOPAIRS[19][2] = {
    { 1, 2 }, { 1, 5 }, { 2, 3 }, { 3, 4 }, { 5, 6 }, { 6, 7 }, { 1, 8 }, { 8, 9 }, { 9, 10 },
    { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 0 }, { 0, 14 }, { 14, 16 }, { 0, 15 }, { 15, 17 },
    { 2, 16 }, { 5, 17 },
};
inline bool is_virtual_pair(int pair_id) { return pair_id > 16; } /* src/coco.hpp:6 */

/* src/paf.cpp:57-60 */
constexpr int THRESH_VECTOR_CNT1 = 8;
constexpr int THRESH_PART_CNT = 4;
constexpr float THRESH_HUMAN_SCORE = 0.4;
constexpr int STEP_PAF = 10;

/* ---- OpenCV 4.4.0 cv::resize INTER_AREA, linear ("area_mode") branch, CV_32FC1 ------------------
 * cv::resize: inv_scale = dsize/ssize (double); hal::resize: scale = 1./inv_scale.
 * Tables (resize.cpp, "for( dx = 0; dx < dsize.width; dx++ )", area_mode branch):
 *   sx = cvFloor(dx*scale_x); fx = (float)((dx+1) - (sx+1)*inv_scale_x); fx = fx<=0 ? 0 : fx - cvFloor(fx);
 *   if (sx + 1 >= ssize.width) { xmax = min(xmax, dx); if (sx >= ssize.width-1) fx = 0, sx = ssize.width-1; }
 *   alpha = {1.f - fx, fx}; same for rows with beta; rows are clipped to [0, ssize.height-1].
 * HResizeLinear: D[dx] = S[sx]*a0 + S[sx+1]*a1 for dx < xmax, else S[sx]*1.f
 * VResizeLinear: dst = S0*b0 + S1*b1
 */
struct resize_tab {
    std::vector<int> ofs;
    std::vector<float> c0, c1;
    int vmax; /* first index from which the 1-tap copy is used (xmax) */
};

resize_tab make_tab(int ssize, int dsize)
{
    resize_tab t;
    t.ofs.resize(dsize);
    t.c0.resize(dsize);
    t.c1.resize(dsize);
    t.vmax = dsize;
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        int s = (int)std::floor(d * scale);
        float f = (float)((d + 1) - (s + 1) * inv_scale);
        f = f <= 0 ? 0.f : f - (float)std::floor(f);
        if (s < 0) {
            f = 0;
            s = 0;
        }
        if (s + 1 >= ssize) {
            t.vmax = std::min(t.vmax, d);
            if (s >= ssize - 1) {
                f = 0;
                s = ssize - 1;
            }
        }
        t.ofs[d] = s;
        t.c0[d] = 1.f - f;
        t.c1[d] = f;
    }
    return t;
}

/* one channel; src [sh][sw] -> dst [dh][dw].  Returns false when the reference early-returns. */
bool resize_area_1ch(const float* src, int sh, int sw, float* dst, int dh, int dw)
{
    const resize_tab tx = make_tab(sw, dw);
    const resize_tab ty = make_tab(sh, dh);
    std::vector<float> rows((size_t)sh * dw);
    for (int r = 0; r < sh; ++r) {
        const float* S = src + (size_t)r * sw;
        float* D = rows.data() + (size_t)r * dw;
        int dx = 0;
        for (; dx < tx.vmax; ++dx) {
            const int sx = tx.ofs[dx];
            D[dx] = madd2(S[sx], tx.c0[dx], S[sx + 1], tx.c1[dx]);
        }
        for (; dx < dw; ++dx)
            D[dx] = S[tx.ofs[dx]] * 1.f;
    }
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = ty.ofs[dy];
        const int sy1 = std::min(sy0 + 1, sh - 1);
        const float b0 = ty.c0[dy], b1 = ty.c1[dy];
        const float* S0 = rows.data() + (size_t)sy0 * dw;
        const float* S1 = rows.data() + (size_t)sy1 * dw;
        float* D = dst + (size_t)dy * dw;
        for (int x = 0; x < dw; ++x)
            D[x] = madd2(S0[x], b0, S1[x], b1);
    }
    return true;
}

/* ---- OpenCV 4.4.0 getGaussianKernel(17, 3.0, CV_32F): double exp(-x^2/(2 sigma^2)), normalised, -> float */
void gaussian_kernel(int ksize, double sigma, float* out)
{
    std::vector<double> v(ksize);
    const int n2 = (ksize - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0;
    for (int i = 0, x = 1 - ksize; i < n2; ++i, x += 2) {
        const double t = std::exp((double)(x * x) * scale2x);
        v[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; ++i) {
        out[i] = (float)(v[i] * mul1);
        out[ksize - 1 - i] = out[i];
    }
    out[n2] = (float)(1.0 * mul1);
}

inline int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

/* one channel separable blur, RowFilter (sequential taps) then SymmColumnFilter (centre + symmetric pairs) */
void gaussian_blur_1ch(const float* src, int h, int w, int ksize, const float* kern, float* dst)
{
    const int r = ksize / 2;
    std::vector<float> rowf((size_t)h * w);
    for (int y = 0; y < h; ++y) {
        const float* S = src + (size_t)y * w;
        float* D = rowf.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float s = kern[0] * S[reflect101(x - r, w)];
            for (int k = 1; k < ksize; ++k)
                s = macc(kern[k], S[reflect101(x - r + k, w)], s);
            D[x] = s;
        }
    }
    const float* ky = kern + r;
    for (int y = 0; y < h; ++y) {
        float* D = dst + (size_t)y * w;
        const float* C = rowf.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float s = macc(ky[0], C[x], 0.f);
            for (int k = 1; k <= r; ++k) {
                const float* S = rowf.data() + (size_t)reflect101(y + k, h) * w;
                const float* S2 = rowf.data() + (size_t)reflect101(y - k, h) * w;
                s = macc(ky[k], S[x] + S2[x], s);
            }
            D[x] = s;
        }
    }
}

/* src/post_process.hpp:71-92 */
void same_max_pool_3x3_2d(int height, int width, const float* input, float* output)
{
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            float max_val = input[i * width + j];
            for (int dx = 0; dx < 3; ++dx)
                for (int dy = 0; dy < 3; ++dy) {
                    const int nx = i + dx - 1;
                    const int ny = j + dy - 1;
                    if (0 <= nx && nx < height && 0 <= ny && ny < width)
                        max_val = std::max(max_val, input[nx * width + ny]);
                }
            output[i * width + j] = max_val;
        }
}

/* src/paf.cpp:41-51 */
struct connection_candidate {
    int idx1;
    int idx2;
    float score;
    float etc;
};
inline bool operator>(const connection_candidate& a, const connection_candidate& b) { return a.score > b.score; }

/* src/paf.cpp:19-37 */
struct human_ref_t {
    int id;
    int parts[O_COCO_N_PARTS];
    float score;
    int n_parts;
    human_ref_t()
        : id(-1)
        , score(0)
        , n_parts(0)
    {
        for (int& p : parts)
            p = -1;
    }
    bool touches(int p_first, int p_second, const o_conn& conn) const
    {
        return parts[p_first] == conn.cid1 || parts[p_second] == conn.cid2;
    }
};

struct paf_state {
    int C, H, W; /* upsampled conf dims */
    std::vector<float> up_conf, up_paf, smoothed, pooled;
};

/* src/paf.cpp:67-144, one limb */
std::vector<connection_candidate> get_connection_candidates(const float* pafmap, int ph, int pw,
    const std::vector<o_peak>& all_peaks, const std::vector<int>& idx1, const std::vector<int>& idx2,
    int ch1, int ch2, int height, float paf_thresh)
{
    std::vector<connection_candidate> candidates;
    const size_t plane = (size_t)ph * pw;
    for (int id1 : idx1)
        for (int id2 : idx2) {
            const o_peak& a = all_peaks[id1];
            const o_peak& b = all_peaks[id2];
            const int dx = b.x - a.x, dy = b.y - a.y;
            const float norm = std::sqrt((float)(dx * dx + dy * dy)); /* std::sqrt(int) -> double in C++; see note */
            if (norm < 1e-12)
                continue;
            float vx = (float)dx, vy = (float)dy;
            vx /= norm;
            vy /= norm;
            const float STEP_X = (b.x - a.x) / float(STEP_PAF);
            const float STEP_Y = (b.y - a.y) / float(STEP_PAF);
            float scores = 0.0f;
            int criterion1 = 0;
            for (int i = 0; i < STEP_PAF; ++i) {
                const int lx = static_cast<int>(a.x + i * STEP_X + 0.5);
                const int ly = static_cast<int>(a.y + i * STEP_Y + 0.5);
                const float px = pafmap[ch1 * plane + (size_t)ly * pw + lx];
                const float py = pafmap[ch2 * plane + (size_t)ly * pw + lx];
                const float score = vx * px + vy * py;
                scores += score;
                if (score > paf_thresh)
                    criterion1 += 1;
            }
            float criterion2 = scores / STEP_PAF + std::min(0.0, 0.5 * height / norm - 1.0);
            if (criterion1 > THRESH_VECTOR_CNT1 && criterion2 > 0)
                candidates.push_back({ a.id, b.id, criterion2, criterion2 + a.score + b.score });
        }
    return candidates;
}

} // namespace

extern "C" {

/* resize_area of src/post_process.hpp:26-52 on [C,sh,sw] -> [C,dh,dw]; returns 1 if it wrote, 0 on the
 * reference's early return (equal dims, post_process.hpp:31-32). */
int oracle_resize_area(const float* src, int C, int sh, int sw, float* dst, int dh, int dw)
{
    if (sh == dh && sw == dw)
        return 0;
    for (int k = 0; k < C; ++k)
        resize_area_1ch(src + (size_t)k * sh * sw, sh, sw, dst + (size_t)k * dh * dw, dh, dw);
    return 1;
}

void oracle_gaussian_kernel(int ksize, double sigma, float* out) { gaussian_kernel(ksize, sigma, out); }

/* The two OpenCV calls on ONE plane: what oracle/shim/opencv2/opencv.hpp's cv::resize(INTER_AREA) and
 * cv::GaussianBlur forward to when the reference's own src/paf.cpp is compiled (oracle/ref_paf_wrap.cpp). */
void oracle_resize_area_1ch(const float* src, int sh, int sw, float* dst, int dh, int dw)
{
    resize_area_1ch(src, sh, sw, dst, dh, dw);
}

void oracle_gaussian_blur_1ch(const float* src, int h, int w, int ksize, double sigma, float* dst)
{
    std::vector<float> kern(ksize);
    gaussian_kernel(ksize, sigma, kern.data());
    gaussian_blur_1ch(src, h, w, ksize, kern.data(), dst);
}

/* smooth of src/post_process.hpp:54-69 (sigma fixed 3.0) */
void oracle_smooth(const float* src, int C, int h, int w, int ksize, float* dst)
{
    std::vector<float> kern(ksize);
    gaussian_kernel(ksize, 3.0, kern.data());
    for (int k = 0; k < C; ++k) {
        if (ksize > 1)
            gaussian_blur_1ch(src + (size_t)k * h * w, h, w, ksize, kern.data(), dst + (size_t)k * h * w);
    }
}

void oracle_max_pool_3x3(const float* src, int C, int h, int w, float* dst)
{
    for (int k = 0; k < C; ++k)
        same_max_pool_3x3_2d(h, w, src + (size_t)k * h * w, dst + (size_t)k * h * w);
}

/* Full parser::paf::process (src/paf.cpp:300-375).
 * conf [J,fh,fw] (rows=fh, cols=fw in memory; the reference NAMES them (fw=rows, fh=cols), see below),
 * paf [2L,fh,fw].  res_w/res_h = m_resolution_size (-1 -> default).  Outputs are optional (NULL ok).
 * Returns number of humans, or -1 on error.  *n_peaks / *n_conns receive the totals (even if > cap). */
int oracle_paf_process(const float* conf, int J, int rows, int cols, const float* paf, int L2,
    float conf_thresh, float paf_thresh, int res_w, int res_h,
    o_human* out_humans, int cap_humans,
    o_peak* out_peaks, int cap_peaks, int* n_peaks,
    o_conn* out_conns, int cap_conns, int* n_conns)
{
    /* paf.cpp:311-315: `auto [n, fw, fh] = dims()` binds fw = dim1 (rows), fh = dim2 (cols). */
    const int fw_paf = rows, fh_paf = cols;
    if (res_w == -1 || res_h == -1) {
        res_w = fw_paf * 4;
        res_h = fh_paf * 4;
    }
    const int feature_height = fh_paf; /* m_feature_size = Size(fw, fh) -> .height = fh = cols (paf.cpp:329,354) */
    const int UH = res_h, UW = res_w; /* buffers (C, height, width) paf.cpp:326-327 */

    std::vector<float> up_conf((size_t)J * UH * UW, 0.f), up_paf((size_t)L2 * UH * UW, 0.f);
    /* resize_area: input (channel, height=rows, width=cols) -> (channel, UH, UW)  post_process.hpp:34-40 */
    oracle_resize_area(conf, J, rows, cols, up_conf.data(), UH, UW);
    oracle_resize_area(paf, L2, rows, cols, up_paf.data(), UH, UW);

    std::vector<float> smoothed((size_t)J * UH * UW, 0.f), pooled((size_t)J * UH * UW, 0.f);
    oracle_smooth(up_conf.data(), J, UH, UW, 17, smoothed.data());
    oracle_max_pool_3x3(smoothed.data(), J, UH, UW, pooled.data());

    /* post_process.hpp:171-193 */
    std::vector<o_peak> all_peaks;
    {
        size_t off = 0;
        for (int k = 0; k < J; ++k)
            for (int i = 0; i < UH; ++i)
                for (int j = 0; j < UW; ++j) {
                    if (k < O_COCO_N_PARTS && smoothed[off] > conf_thresh && smoothed[off] == pooled[off]) {
                        const int idx = (int)all_peaks.size();
                        all_peaks.push_back(o_peak{ k, j, i, up_conf[off], idx });
                    }
                    ++off;
                }
    }
    /* group_by, post_process.hpp:197-205 */
    std::vector<std::vector<int>> by_channel(O_COCO_N_PARTS);
    for (const o_peak& pi : all_peaks)
        by_channel[pi.part_id].push_back(pi.id);

    /* get_connections, paf.cpp:234-272 */
    std::vector<std::vector<o_conn>> all_connections;
    for (int pair_id = 0; pair_id < O_COCO_N_PAIRS; ++pair_id) {
        std::vector<connection_candidate> candidates = get_connection_candidates(up_paf.data(), UH, UW, all_peaks,
            by_channel[COCOPAIRS[pair_id][0]], by_channel[COCOPAIRS[pair_id][1]],
            COCOPAIRS_NET[pair_id][0], COCOPAIRS_NET[pair_id][1], feature_height, paf_thresh);
        std::sort(candidates.begin(), candidates.end(), std::greater<connection_candidate>());
        std::vector<o_conn> conns;
        for (const auto& candidate : candidates) {
            bool assigned = false;
            for (const auto& conn : conns)
                if (conn.cid1 == candidate.idx1 || conn.cid2 == candidate.idx2) {
                    assigned = true;
                    break;
                }
            if (!assigned)
                conns.push_back(o_conn{ pair_id, candidate.idx1, candidate.idx2, candidate.score });
        }
        all_connections.push_back(conns);
    }

    /* get_humans, paf.cpp:146-232 */
    std::vector<human_ref_t> human_refs;
    for (int pair_id = 0; pair_id < O_COCO_N_PAIRS; ++pair_id) {
        const int part_id1 = COCOPAIRS[pair_id][0];
        const int part_id2 = COCOPAIRS[pair_id][1];
        for (const o_conn& conn : all_connections[pair_id]) {
            std::vector<int> hr_ids;
            for (auto hr : human_refs)
                if (hr.touches(part_id1, part_id2, conn))
                    hr_ids.push_back(hr.id);

            if (hr_ids.size() == 1) {
                auto& hr1 = human_refs[hr_ids[0]];
                if (hr1.parts[part_id2] != conn.cid2) {
                    hr1.parts[part_id2] = conn.cid2;
                    ++hr1.n_parts;
                    hr1.score += all_peaks[conn.cid2].score + conn.score;
                }
            } else if (hr_ids.size() >= 2) {
                auto& hr1 = human_refs[hr_ids[0]];
                auto& hr2 = human_refs[hr_ids[1]];
                int membership = 0;
                for (int i = 0; i < O_COCO_N_PARTS; ++i)
                    if (hr1.parts[i] > 0 && hr2.parts[i] > 0)
                        membership = 2;
                if (membership == 0) {
                    for (int i = 0; i < O_COCO_N_PARTS; i++)
                        hr1.parts[i] += hr2.parts[i] + 1;
                    hr1.n_parts += hr2.n_parts;
                    hr1.score += hr2.score;
                    hr1.score += conn.score;
                    size_t delete_id = hr_ids[1];
                    human_refs.erase(human_refs.begin() + delete_id);
                    for (auto& hr_ref : human_refs)
                        if ((size_t)hr_ref.id > delete_id)
                            --hr_ref.id;
                } else {
                    hr1.parts[part_id2] = conn.cid2;
                    hr1.n_parts += 1;
                    hr1.score += all_peaks[conn.cid2].score + conn.score;
                }
            } else if (hr_ids.size() == 0 && !is_virtual_pair(pair_id)) {
                human_ref_t h;
                h.parts[part_id1] = conn.cid1;
                h.parts[part_id2] = conn.cid2;
                h.n_parts = 2;
                h.score = all_peaks[conn.cid1].score + all_peaks[conn.cid2].score + conn.score;
                h.id = (int)human_refs.size();
                human_refs.push_back(h);
            }
        }
    }
    human_refs.erase(std::remove_if(human_refs.begin(), human_refs.end(),
                         [&](const human_ref_t& hr) {
                             return (hr.n_parts < THRESH_PART_CNT || hr.score / hr.n_parts < THRESH_HUMAN_SCORE);
                         }),
        human_refs.end());

    /* paf.cpp:359-372 */
    int n_h = 0;
    for (const auto& hr : human_refs) {
        if (out_humans && n_h < cap_humans) {
            o_human human;
            std::memset(&human, 0, sizeof(human));
            human.score = hr.score;
            for (int i = 0; i < O_COCO_N_PARTS; ++i)
                if (hr.parts[i] != -1) {
                    human.parts[i].has_value = 1;
                    const o_peak p = all_peaks[hr.parts[i]];
                    human.parts[i].score = p.score;
                    human.parts[i].x = static_cast<float>(p.x) / res_w;
                    human.parts[i].y = static_cast<float>(p.y) / res_h;
                }
            out_humans[n_h] = human;
        }
        ++n_h;
    }

    if (n_peaks)
        *n_peaks = (int)all_peaks.size();
    if (out_peaks)
        for (size_t i = 0; i < all_peaks.size() && (int)i < cap_peaks; ++i)
            out_peaks[i] = all_peaks[i];
    int nc = 0;
    for (const auto& v : all_connections)
        for (const auto& c : v) {
            if (out_conns && nc < cap_conns)
                out_conns[nc] = c;
            ++nc;
        }
    if (n_conns)
        *n_conns = nc;
    return n_h;
}

/* std::sort(candidates.begin(), candidates.end(), std::greater<connection_candidate>()) of src/paf.cpp:249 alone: n scores in
 * generation order -> order[i] = index of the candidate that ends at position i (this libstdc++'s choice among equal scores). */
void oracle_std_sort_greater(const float* scores, int n, int* order)
{
    std::vector<connection_candidate> c(n);
    for (int i = 0; i < n; ++i)
        c[i] = connection_candidate{ i, i, scores[i], 0.f };
    std::sort(c.begin(), c.end(), std::greater<connection_candidate>());
    for (int i = 0; i < n; ++i)
        order[i] = c[i].idx1;
}

/* M. D. McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against THIS libstdc++'s std::sort: the comparator decides
 * the values lazily so that every pivot lands near an end -> quadratic partitioning -> introsort's depth limit -> its heap-sort
 * fall-back.  out[i] = a score sequence (distinct values, to be sorted with std::greater) that reproduces that behaviour. */
void oracle_sort_killer(int n, float* out)
{
    std::vector<int> val(n, n - 1), ptr(n); /* gas = n - 1 */
    const int gas = n - 1;
    int nsolid = 0, candidate = 0;
    for (int i = 0; i < n; ++i)
        ptr[i] = i;
    std::sort(ptr.begin(), ptr.end(), [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) {
            if (x == candidate)
                val[x] = nsolid++;
            else
                val[y] = nsolid++;
        }
        if (val[x] == gas)
            candidate = x;
        else if (val[y] == gas)
            candidate = y;
        return val[x] < val[y];
    });
    /* the adversary played "less"; the parser sorts with "greater": negate so that a > b <=> val[a] < val[b] */
    for (int i = 0; i < n; ++i)
        out[i] = (float)(-val[i]);
}

/* nhwc_images_append_nchw_batch, src/data.cpp:21-51: u8 HWC -> f32 CHW, `(*line++)[c] * factor` is
 * int * double -> double, narrowed to float by push_back; channel order {2,1,0} when flip_rb. */
void oracle_nhwc_u8_to_nchw_f32(const uint8_t* images, int n, int h, int w, double factor, int flip_rb, float* out)
{
    const size_t plane = (size_t)h * w;
    size_t o = 0;
    for (int b = 0; b < n; ++b) {
        const uint8_t* img = images + (size_t)b * plane * 3;
        for (int ci = 0; ci < 3; ++ci) {
            const int c = flip_rb ? 2 - ci : ci;
            for (size_t p = 0; p < plane; ++p)
                out[o++] = (float)(img[p * 3 + c] * factor);
        }
    }
}

} // extern "C"
