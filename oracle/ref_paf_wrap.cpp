// oracle/ref_paf_wrap.cpp — TEST INFRASTRUCTURE ONLY.
//
// The reference's OWN PAF parser, compiled from where it lies: this translation unit textually includes
// /root/reference/src/paf.cpp (and through it src/post_process.hpp, src/coco.hpp, src/cudnn*.hpp) at COMPILE time
// (`-I$(REF)`, oracle/Makefile `ref`; nothing is copied into the repo) and adds extern "C" entry points behind it.
// Including the .cpp rather than compiling it as its own object gives the wrapper access to the file-static
// functions (get_connections, get_humans: src/paf.cpp:146-272) so that peaks and connections can be exposed for the
// stage-by-stage parity tests, not only the final humans.
//
// Third-party pieces absent from /root/reference and what stands in for them (oracle/shim/):
//   ttl/*  (stdtensor v0.9.1)     containers / views / range only - no arithmetic            shim/ttl/*
//   cuda_runtime.h, cudnn.h       the cuDNN max-pool member is constructed but never run     shim/cuda_runtime.h, shim/cudnn.h
//                                 (paf::process passes use_gpu = false, src/paf.cpp:345)
//   OpenCV 4.4.0                  cv::resize(INTER_AREA), cv::GaussianBlur on float planes    forwarders below ->
//                                 oracle/paf_oracle.cpp (restated; the ONLY unpinned arithmetic of the PAF path)
// Everything else - max-pool, peak scan, line integrals, std::sort + greedy assignment, human assembly, output
// normalisation - is the reference's code, compiled by the same g++ / libstdc++.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include <src/paf.cpp> // == /root/reference/src/paf.cpp via -I$(REF)

#include "oracle_common.h"

extern "C" {
void oracle_resize_area_1ch(const float* src, int sh, int sw, float* dst, int dh, int dw);
void oracle_gaussian_blur_1ch(const float* src, int h, int w, int ksize, double sigma, float* dst);
}

namespace cv {

void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation)
{
    if (interpolation != INTER_AREA || fx != 0 || fy != 0 || src.type() != DataType<float>::type
        || dst.type() != DataType<float>::type || dsize != dst.size()) {
        std::fprintf(stderr, "oracle shim cv::resize: only the call of src/post_process.hpp:50 is provided\n");
        std::abort();
    }
    oracle_resize_area_1ch(static_cast<const float*>(src.ptr()), src.size().height, src.size().width,
        static_cast<float*>(dst.ptr()), dsize.height, dsize.width);
}

void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigmaX, double sigmaY, int borderType)
{
    if (ksize.width != ksize.height || (ksize.width & 1) == 0 || sigmaX <= 0 || sigmaY != 0 || borderType != 4
        || src.type() != DataType<float>::type || src.size() != dst.size()) {
        std::fprintf(stderr, "oracle shim cv::GaussianBlur: only the call of src/post_process.hpp:66 is provided\n");
        std::abort();
    }
    oracle_gaussian_blur_1ch(static_cast<const float*>(src.ptr()), src.size().height, src.size().width, ksize.width,
        sigmaX, static_cast<float*>(dst.ptr()));
}

} // namespace cv

namespace {

hyperpose::feature_map_t make_map3(const char* name, const float* data, int c, int h, int w)
{
    const size_t n = (size_t)c * h * w;
    std::unique_ptr<char[]> buf(new char[n * sizeof(float)]);
    std::memcpy(buf.get(), data, n * sizeof(float));
    return hyperpose::feature_map_t(name, std::move(buf), std::vector<int>{ c, h, w });
}

} // namespace

extern "C" {

/* hyperpose::parser::paf::process (reference src/paf.cpp:300-375) on one frame: conf [J,rows,cols], paf [L2,rows,cols];
 * res_w / res_h = the constructor's resolution_size (-1,-1 = the default 4x, src/paf.cpp:316-317).
 * Returns the number of humans (all of them counted, at most `cap` written). */
int ref_paf_process(const float* conf, int J, int rows, int cols, const float* paf, int L2, float conf_thresh,
    float paf_thresh, int res_w, int res_h, o_human* out, int cap)
{
    static_assert(sizeof(hyperpose::human_t) == sizeof(o_human), "human_t layout");
    hyperpose::parser::paf parser(conf_thresh, paf_thresh, cv::Size(res_w, res_h));
    const auto humans = parser.process(make_map3("conf", conf, J, rows, cols), make_map3("paf", paf, L2, rows, cols));
    int n = 0;
    for (const auto& h : humans) {
        if (out && n < cap) {
            o_human o;
            std::memset(&o, 0, sizeof(o));
            o.score = h.score;
            for (int i = 0; i < O_COCO_N_PARTS; ++i) {
                o.parts[i].has_value = h.parts[i].has_value ? 1 : 0;
                o.parts[i].x = h.parts[i].x;
                o.parts[i].y = h.parts[i].y;
                o.parts[i].score = h.parts[i].score;
            }
            out[n] = o;
        }
        ++n;
    }
    return n;
}

/* The intermediate lists of the same frame, produced by the reference's own functions called in the order
 * paf::process calls them (src/paf.cpp:336-357): resize_area x2, peak_finder_t::find_peak_coords / group_by,
 * get_connections per limb.  Counts are totals (even if > cap).  Returns the number of humans get_humans keeps. */
int ref_paf_debug(const float* conf, int J, int rows, int cols, const float* paf, int L2, float conf_thresh,
    float paf_thresh, int res_w, int res_h, o_peak* out_peaks, int cap_peaks, int* n_peaks, o_conn* out_conns,
    int cap_conns, int* n_conns)
{
    using namespace hyperpose;
    using namespace hyperpose::parser;
    const ttl::tensor_view<float, 3> conf_t(conf, J, rows, cols), paf_t(paf, L2, rows, cols);
    /* src/paf.cpp:311-317: `auto [n, fw, fh] = dims()` (fw = rows, fh = cols), default resolution (fw*4, fh*4) */
    auto [n2, fw_paf, fh_paf] = paf_t.dims();
    (void)n2;
    cv::Size resolution(res_w, res_h);
    if (res_w == -1 || res_h == -1)
        resolution = cv::Size(fw_paf * 4, fh_paf * 4);
    ttl::tensor<float, 3> up_conf(J, resolution.height, resolution.width), up_paf(L2, resolution.height, resolution.width);
    const cv::Size feature_size(fw_paf, fh_paf);
    peak_finder_t<float> finder(J, resolution.height, resolution.width, 17);
    resize_area(conf_t, ttl::ref(up_conf));
    resize_area(paf_t, ttl::ref(up_paf));
    const auto all_peaks = finder.find_peak_coords(ttl::view(up_conf), conf_thresh, false);
    const auto by_channel = finder.group_by(all_peaks);
    const ttl::tensor_view<float, 3>& pafmap = ttl::view(up_paf);
    std::vector<std::vector<connection>> all_connections;
    for (int pair_id = 0; pair_id < COCO_N_PAIRS; pair_id++)
        all_connections.push_back(get_connections(pafmap, all_peaks, by_channel, pair_id, feature_size.height, paf_thresh));
    const auto human_refs = get_humans(all_peaks, all_connections);

    if (n_peaks)
        *n_peaks = (int)all_peaks.size();
    if (out_peaks)
        for (size_t i = 0; i < all_peaks.size() && (int)i < cap_peaks; ++i)
            out_peaks[i] = o_peak{ all_peaks[i].part_id, all_peaks[i].pos.x, all_peaks[i].pos.y, all_peaks[i].score, all_peaks[i].id };
    int nc = 0;
    for (int pair_id = 0; pair_id < COCO_N_PAIRS; pair_id++)
        for (const connection& c : all_connections[pair_id]) {
            if (out_conns && nc < cap_conns)
                out_conns[nc] = o_conn{ pair_id, c.cid1, c.cid2, c.score };
            ++nc;
        }
    if (n_conns)
        *n_conns = nc;
    return (int)human_refs.size();
}

} // extern "C"
