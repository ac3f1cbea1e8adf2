// Round-1 names of the engine mirror, kept as aliases: the class is hyperpose::dnn::tensorrt (operator/dnn/tensorrt.hpp), exactly
// the reference's name and constructor signatures.
#pragma once
#include "tensorrt.hpp"

namespace hyperpose {
namespace dnn {
    using hip_engine = tensorrt;
    using serialized_model = tensorrt_serialized;
} // namespace dnn
} // namespace hyperpose
