// examples/cli.cpp — `hyperpose-cli` on the MI355X engine.  What is kept from the reference's examples/cli.cpp is its INTERFACE: the flag
// names and defaults (:15-35: model / post / w / h / max_batch_size / source / runtime / keep_ratio / alpha / saving_prefix / logging /
// imshow), the values `operator` / `stream` and `paf` / `ppn` / `pifpaf`, the model-file suffix rule (.onnx / .uff / anything else =
// serialized engine) and the order of the work (resize + inference, parse, resume_ratio, draw, blend with weight alpha, write).  The
// program itself is written for this engine; the pieces that need gflags and OpenCV are replaced by what this image has:
//   * flags: `--name=value`, `--name value`, `--flag` / `--noflag` (gflags syntax), parsed below;
//   * media: binary PPM (P6) images - one file, or every *.ppm of a directory - and `synthetic:<n>:<w>x<h>` (seeded frames); results are
//     written as `<saving_prefix>_<id>.ppm` with the skeletons drawn (hp::draw_human) and blended with weight alpha.  With OpenCV
//     (-DHYPERPOSE_USE_OPENCV) any cv::imread format works; videos / the camera need cv::VideoCapture and are refused without it.
//   * `--model`: .onnx, a serialized engine (anything else), or `builtin:<arch>` (hp_model_archs(); synthetic weights).
// build: g++ -std=c++17 -O2 -Iinclude examples/cli.cpp -Lhyperpose_amd -lhp_hip -lpthread -Wl,-rpath,$PWD/hyperpose_amd -o hyperpose-cli
#include <hyperpose/hyperpose.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <dirent.h>
#include <fstream>
#include <iostream>
#include <map>
#include <string_view>
#include <variant>

#define kOPERATOR "operator"
#define kSTREAM "stream"
#define kPAF "paf"
#define kPPN "ppn"
#define kPIFPAF "pifpaf"

namespace hp = hyperpose;

// ---- flags (defaults of examples/cli.cpp:15-35)
static std::string FLAGS_model = "builtin:lw_openpose_mobilenet";
static std::string FLAGS_post = kPAF;
static int FLAGS_w = 368, FLAGS_h = 342, FLAGS_max_batch_size = 6;
static bool FLAGS_imshow = true;
static std::string FLAGS_source = "synthetic:4:640x480";
static std::string FLAGS_runtime = kOPERATOR;
static bool FLAGS_keep_ratio = true;
static double FLAGS_alpha = 0.5;
static std::string FLAGS_saving_prefix = "output";
static bool FLAGS_logging = false;
static bool FLAGS_half = false; // addition: data_type::kHALF engines (the reference CLI always builds data_type::kFLOAT ones)

static std::ostream& cli_log() { return std::cout << "[HyperPose::CLI] "; }

static bool parse_flags(int argc, char** argv)
{
    std::map<std::string, std::string*> sflags = { { "model", &FLAGS_model }, { "post", &FLAGS_post }, { "source", &FLAGS_source },
        { "runtime", &FLAGS_runtime }, { "saving_prefix", &FLAGS_saving_prefix } };
    std::map<std::string, int*> iflags = { { "w", &FLAGS_w }, { "h", &FLAGS_h }, { "max_batch_size", &FLAGS_max_batch_size } };
    std::map<std::string, bool*> bflags = { { "imshow", &FLAGS_imshow }, { "keep_ratio", &FLAGS_keep_ratio }, { "logging", &FLAGS_logging }, { "half", &FLAGS_half } };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0 && a.rfind("-", 0) == 0)
            a = "-" + a; // gflags accepts -flag too
        if (a.rfind("--", 0) != 0) {
            cli_log() << "ERROR: unexpected argument " << a << "\n";
            return false;
        }
        a = a.substr(2);
        std::string name = a, value;
        bool has_value = false;
        if (const auto eq = a.find('='); eq != std::string::npos)
            name = a.substr(0, eq), value = a.substr(eq + 1), has_value = true;
        if (bflags.count(name) || (name.rfind("no", 0) == 0 && bflags.count(name.substr(2)))) {
            const bool neg = !bflags.count(name);
            bool v = !neg;
            if (has_value)
                v = (value == "true" || value == "1" || value == "yes") != neg;
            *bflags[neg ? name.substr(2) : name] = v;
            continue;
        }
        if (!has_value) {
            if (i + 1 >= argc) {
                cli_log() << "ERROR: flag --" << name << " needs a value\n";
                return false;
            }
            value = argv[++i];
        }
        if (sflags.count(name))
            *sflags[name] = value;
        else if (iflags.count(name))
            *iflags[name] = std::atoi(value.c_str());
        else if (name == "alpha")
            FLAGS_alpha = std::atof(value.c_str());
        else {
            cli_log() << "ERROR: unknown command line flag '" << name << "'\n";
            return false;
        }
    }
    return true;
}

// ---- media
static bool read_ppm(const std::string& path, cv::Mat& out)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    auto token = [&](auto& v) {
        for (;;) {
            f >> std::ws;
            if (f.peek() == '#') {
                std::string line;
                std::getline(f, line);
                continue;
            }
            f >> v;
            return;
        }
    };
    token(magic), token(w), token(h), token(maxv);
    if (!f || magic != "P6" || w <= 0 || h <= 0 || maxv != 255 || (size_t)w * h > (size_t)1 << 28)
        return false;
    f.get(); // the single whitespace after maxval
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    f.read((char*)rgb.data(), rgb.size());
    if (!f)
        return false;
    out = cv::Mat(h, w, CV_8UC3);
    uint8_t* d = const_cast<uint8_t*>(hp::detail::mat_data(out));
    for (size_t i = 0; i < (size_t)w * h; ++i)
        d[i * 3] = rgb[i * 3 + 2], d[i * 3 + 1] = rgb[i * 3 + 1], d[i * 3 + 2] = rgb[i * 3]; // RGB file -> BGR cv::Mat
    return true;
}
static bool write_ppm(const std::string& path, const cv::Mat& m)
{
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << m.cols << " " << m.rows << "\n255\n";
    const uint8_t* d = hp::detail::mat_data(m);
    std::vector<uint8_t> rgb((size_t)m.rows * m.cols * 3);
    for (size_t i = 0; i < (size_t)m.rows * m.cols; ++i)
        rgb[i * 3] = d[i * 3 + 2], rgb[i * 3 + 1] = d[i * 3 + 1], rgb[i * 3 + 2] = d[i * 3];
    f.write((const char*)rgb.data(), rgb.size());
    return (bool)f;
}
static cv::Mat clone(const cv::Mat& m)
{
    cv::Mat c(m.rows, m.cols, CV_8UC3);
    std::memcpy(const_cast<uint8_t*>(hp::detail::mat_data(c)), hp::detail::mat_data(m), (size_t)m.rows * m.cols * 3);
    return c;
}
// cv::addWeighted(mat, alpha, background, 1 - alpha, 0, mat) (examples/cli.cpp:213-215): saturate_cast<uchar>(round(a * x + b * y))
static void add_weighted(cv::Mat& mat, double alpha, const cv::Mat& background)
{
    uint8_t* d = const_cast<uint8_t*>(hp::detail::mat_data(mat));
    const uint8_t* b = hp::detail::mat_data(background);
    for (size_t i = 0; i < (size_t)mat.rows * mat.cols * 3; ++i) {
        const double v = std::nearbyint(d[i] * alpha + b[i] * (1 - alpha));
        d[i] = (uint8_t)std::min(255.0, std::max(0.0, v));
    }
}
static std::vector<cv::Mat> load_source()
{
    std::vector<cv::Mat> images;
    auto match_suffix = [](std::string_view suffix) {
        return FLAGS_source.size() >= suffix.size() && std::equal(suffix.crbegin(), suffix.crend(), FLAGS_source.crbegin());
    };
    if (FLAGS_source.rfind("synthetic:", 0) == 0) {
        int n = 0, w = 0, h = 0;
        if (std::sscanf(FLAGS_source.c_str(), "synthetic:%d:%dx%d", &n, &w, &h) != 3 || n <= 0 || w <= 0 || h <= 0 || n > 4096)
            return {};
        unsigned s = 20240;
        for (int i = 0; i < n; ++i) {
            cv::Mat m(h, w, CV_8UC3);
            uint8_t* d = const_cast<uint8_t*>(hp::detail::mat_data(m));
            for (size_t k = 0; k < (size_t)w * h * 3; ++k)
                s = s * 1664525u + 1013904223u, d[k] = (uint8_t)(s >> 24);
            images.push_back(m);
        }
        return images;
    }
#ifdef HYPERPOSE_USE_OPENCV
    if (match_suffix(".jpg") || match_suffix(".jpeg") || match_suffix(".png"))
        return { cv::imread(FLAGS_source) };
#endif
    if (match_suffix(".ppm")) {
        cv::Mat m;
        if (read_ppm(FLAGS_source, m))
            images.push_back(m);
        return images;
    }
    if (DIR* dir = opendir(FLAGS_source.c_str())) { // glob_images (examples/utils.cpp)
        std::vector<std::string> names;
        while (dirent* e = readdir(dir))
            names.push_back(e->d_name);
        closedir(dir);
        std::sort(names.begin(), names.end());
        for (const auto& nm : names) {
            cv::Mat m;
            const std::string path = FLAGS_source + "/" + nm;
            if (nm.size() > 4 && nm.substr(nm.size() - 4) == ".ppm" && read_ppm(path, m))
                images.push_back(m);
#ifdef HYPERPOSE_USE_OPENCV
            else if (!(m = cv::imread(path)).empty())
                images.push_back(m);
#endif
        }
    }
    return images;
}

// the three post-processing operators behind one value (the `--post` flag picks the alternative)
using any_parser = std::variant<hp::parser::paf, hp::parser::pose_proposal, hp::parser::pifpaf>;

static bool has_suffix(const std::string& text, std::string_view suffix)
{
    return text.size() >= suffix.size() && text.compare(text.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// `--model`: built-in topology with synthetic weights, ONNX file, UFF file (TensorFlow frozen graphs: refused by the engine with its
// own message), or - any other name - an engine saved with tensorrt::save
static hp::dnn::tensorrt build_engine()
{
    const cv::Size net_size(FLAGS_w, FLAGS_h);
    cli_log() << "engine: model '" << FLAGS_model << "', network input " << FLAGS_w << " x " << FLAGS_h << " (w x h), batches of up to "
              << FLAGS_max_batch_size << (FLAGS_keep_ratio ? ", aspect ratio kept (letter-box)\n" : ", frames stretched to the network size\n");
    const hp::data_type dtype = FLAGS_half ? hp::data_type::kHALF : hp::data_type::kFLOAT;
    if (FLAGS_model.rfind("builtin:", 0) == 0)
        return hp::dnn::tensorrt(hp::dnn::builtin_model{ FLAGS_model.substr(8), {}, 20241 }, net_size, FLAGS_max_batch_size, FLAGS_keep_ratio, dtype);
    if (has_suffix(FLAGS_model, ".onnx"))
        return hp::dnn::tensorrt(hp::dnn::onnx{ FLAGS_model }, net_size, FLAGS_max_batch_size, FLAGS_keep_ratio, dtype);
    if (has_suffix(FLAGS_model, ".uff"))
        return hp::dnn::tensorrt(hp::dnn::uff{ FLAGS_model, "image", { "outputs/conf", "outputs/paf" } }, net_size, FLAGS_max_batch_size, FLAGS_keep_ratio);
    cli_log() << "'" << FLAGS_model << "' is neither .onnx nor .uff: loading it as a serialized engine\n";
    return hp::dnn::tensorrt(hp::dnn::tensorrt_serialized{ FLAGS_model }, net_size, FLAGS_max_batch_size, FLAGS_keep_ratio);
}

static any_parser build_parser(hp::dnn::tensorrt& engine) // (input_size() is non-const in the reference's class, tensorrt.hpp:98)
{
    const cv::Size in = engine.input_size();
    if (FLAGS_post == kPPN)
        return any_parser{ std::in_place_type<hp::parser::pose_proposal>, in };
    if (FLAGS_post == kPIFPAF)
        return any_parser{ std::in_place_type<hp::parser::pifpaf>, in.height, in.width };
    if (FLAGS_post != kPAF) {
        cli_log() << "ERROR: --post=" << FLAGS_post << " is not one of " kPAF ", " kPPN ", " kPIFPAF "\n";
        std::exit(-1);
    }
    return any_parser{ std::in_place_type<hp::parser::paf> };
}

int main(int argc, char** argv)
{
    if (!parse_flags(argc, argv))
        return 1;
    if (FLAGS_logging)
        cli_log() << "--logging: the engine reports through hp_last_error(); nothing more to switch on\n";
    if (FLAGS_alpha < 0 || FLAGS_alpha > 1) {
        const double inside = std::min(1.0, std::max(0.0, FLAGS_alpha));
        cli_log() << "WARNING: --alpha=" << FLAGS_alpha << " is outside [0, 1]; using " << inside << "\n";
        FLAGS_alpha = inside;
    }
    if (hp_init(0) != HP_OK) {
        cli_log() << "ERROR: " << hp_last_error() << "\n";
        return 2;
    }
    auto images = load_source();
    if (images.empty()) {
        cli_log() << "ERROR: no frames from --source=" << FLAGS_source << " (PPM files / directories and synthetic:<n>:<w>x<h> are supported"
#ifndef HYPERPOSE_USE_OPENCV
                  << "; videos and the camera need a build with OpenCV"
#endif
                  << ")" << std::endl;
        std::exit(-1);
    }
    if (FLAGS_imshow) {
        FLAGS_imshow = false;
        cli_log() << "--imshow needs a display and OpenCV's highgui: results go to files only\n";
    }

    auto engine = build_engine();
    any_parser parser = build_parser(engine);
    if (FLAGS_runtime != kOPERATOR && FLAGS_runtime != kSTREAM) {
        cli_log() << "WARNING: --runtime=" << FLAGS_runtime << " is neither " kOPERATOR " nor " kSTREAM "; using " kOPERATOR "\n";
        FLAGS_runtime = kOPERATOR;
    }

    using clk_t = std::chrono::high_resolution_clock;
    size_t n_humans = 0, n_written = 0;
    auto render = [&](cv::Mat& img, const std::vector<hp::human_t>& poses, bool resume) {
        cv::Mat background;
        if (FLAGS_alpha > 0)
            background = clone(img);
        for (auto pose : poses) {
            if (resume)
                hp::resume_ratio(pose, img.size(), engine.input_size());
            hp::draw_human(img, pose);
        }
        if (FLAGS_alpha > 0)
            add_weighted(img, FLAGS_alpha, background);
        n_humans += poses.size();
        n_written += write_ppm(FLAGS_saving_prefix + "_" + std::to_string(n_written) + ".ppm", img);
    };

    auto beg = clk_t::now();
    if (FLAGS_runtime == kOPERATOR) {
        // operator API: one batch at a time - engine.inference(batch) -> one internal_t per frame -> parser.process(internal_t)
        const size_t step = (size_t)std::max(1, FLAGS_max_batch_size);
        for (size_t first = 0; first < images.size(); first += step) {
            std::vector<cv::Mat> batch(images.begin() + first, images.begin() + std::min(images.size(), first + step));
            const auto maps = engine.inference(batch);
            for (size_t k = 0; k < batch.size(); ++k) {
                const auto poses = std::visit([&](auto& op) { return op.process(maps[k]); }, parser);
                render(batch[k], poses, FLAGS_keep_ratio);
            }
        }
    } else {
        // stream API: make_stream(engine, parser, use_original_resolution = true, keep_ratio); frames in, (frame, poses) out in order
        std::visit(
            [&](auto& op) {
                auto stream = hp::make_stream(engine, op, true, FLAGS_keep_ratio);
                stream.async() << images;
                auto sink = [&](size_t, const cv::Mat& frame, const std::vector<hp::human_t>& poses) {
                    cv::Mat img = clone(frame);
                    render(img, poses, false); // the stream already applied resume_ratio
                };
                stream.sync() >> sink;
            },
            parser);
    }
    const auto ms = std::chrono::duration<double, std::milli>(clk_t::now() - beg).count();
    std::cout << images.size() << " images got processed in " << ms << " ms, FPS = " << 1000. * images.size() / ms << " (" << n_humans
              << " humans, " << n_written << " files written as " << FLAGS_saving_prefix << "_<id>.ppm)\n";
    return n_written == images.size() ? 0 : 3;
}
