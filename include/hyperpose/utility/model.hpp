// hyperpose::dnn model descriptors — reference include/hyperpose/utility/model.hpp:16-32, same names and fields, so that
// `tensorrt(onnx{path}, ...)`, `tensorrt(uff{path, input, outputs}, ...)` and `tensorrt(tensorrt_serialized{path}, ...)` call
// sites compile unchanged.  `builtin_model` is the addition of this implementation: one of the topologies libhp_hip.so restates
// from hyperpose/Model/<arch>.py (hp_model_archs()) with caller-provided or deterministic synthetic weights.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace hyperpose {
namespace dnn {

    /// Uff models are a TensorRT-only format: the descriptor exists so that reference code compiles; constructing an engine from
    /// it fails at run time with the reference's error behaviour (log + exit), see operator/dnn/tensorrt.hpp.
    struct uff {
        std::string model_path;
        std::string input_name;
        std::vector<std::string> output_names;
    };

    /// ONNX model file (hp_model_from_onnx_file).
    struct onnx {
        std::string model_path;
    };

    /// The file written by `tensorrt::save` (hp_engine_save / hp_engine_load): topology + pre-processing + fp32 weights.
    struct tensorrt_serialized {
        std::string model_path;
    };

    struct builtin_model {
        std::string arch;           // see hp_model_archs()
        std::vector<float> weights; // empty: deterministic synthetic weights (seed below)
        uint64_t seed = 20241;
    };

} // namespace dnn
} // namespace hyperpose
