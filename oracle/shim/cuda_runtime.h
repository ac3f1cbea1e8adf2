/* oracle/shim/cuda_runtime.h — TEST INFRASTRUCTURE ONLY.  Empty: src/post_process.hpp:8 includes it, nothing on the
 * CPU path of the reference's PAF parser uses a CUDA runtime symbol. */
#pragma once
