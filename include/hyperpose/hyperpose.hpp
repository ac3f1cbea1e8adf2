// Umbrella header, mirroring the reference's include/hyperpose/hyperpose.hpp.
#pragma once
#include "operator/dnn/tensorrt.hpp"
#include "operator/parser/paf.hpp"
#include "operator/parser/pifpaf.hpp"
#include "operator/parser/proposal_network.hpp"
#include "stream/stream.hpp"
#include "utility/data.hpp"
#include "utility/human.hpp"
