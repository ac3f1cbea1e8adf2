"""Build libhp_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m hyperpose_amd.build [--force]

Each translation unit is compiled separately (cached by mtime) and linked into
``hyperpose_amd/libhp_hip.so``.  The parser kernels are compiled with ``-ffp-contract=off``: their results
must be bit-identical to the CPU code they replace ("Parser numerics").  hipcc cross-compiles
without a GPU, so this runs in the GPU-less build container; the .so travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libhp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(HERE, "..", "include")]
# (source, extra flags)
UNITS = [
    ("hp_runtime.cpp", []),
    ("preproc.hip", ["-ffp-contract=off"]),
    ("resize.hip", ["-ffp-contract=off"]),
    ("paf_parser.hip", ["-ffp-contract=off"]),
    ("ppn_parser.hip", ["-ffp-contract=off"]),
    ("pifpaf_parser.hip", ["-ffp-contract=off"]),
    ("conv_kernels.hip", []),
    ("conv_chain.hip", []),
    ("conv_bottleneck.hip", []),
    # -fno-slp-vectorize: no v_pk_fma_f32 in this file's vector kernels.  first_conv32_kernel built with the packed FMAs hipcc's SLP pass forms
    # (a pixel value broadcast against pairs of weights) returns different bits for the same inputs - recomputed in the same thread - while
    # fp16-MFMA kernels of another stream share its CUs (an fp16 engine or an HP_DTYPE_F32S engine next to it): DESIGN.md section 7B.8,
    # tools/r6_two_engines_debug.py, HP_FIRST_CONV_VERIFY.  Without them it is exact; the stem costs 25.3 instead of 24.5 us.
    ("conv_fp32.hip", ["-fno-slp-vectorize"]),
    ("conv32_direct.hip", []),
    ("conv32_winograd.hip", []),
    ("conv32_winograd3.hip", []),
    ("conv32_head.hip", []),
    ("engine.cpp", []),
    ("models.cpp", []),
    ("onnx_import.cpp", []),
    ("pipeline.cpp", []),
    ("dist.cpp", []),
]


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "hp_hip.h"))
    objs, changed = [], False
    procs = []
    for src, extra in UNITS:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(OBJ, src + ".o")
        objs.append(obj)
        if force or _newer(path, obj, headers):
            cmd = [HIPCC, "-x", "hip", *COMMON, *extra, "-c", path, "-o", obj]
            if verbose:
                print("[hyperpose_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
            changed = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"]
        if verbose:
            print("[hyperpose_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


BENCH_BIN = os.path.join(HERE, "operator_api_bench.bin")


def build_operator_bench(verbose: bool = True) -> str:
    """examples/operator_api_bench.cpp (the reference's operator-API loop, timed) against the C++ mirror headers and libhp_hip.so, with plain
    g++: a host program, no device code.  bench.py runs it for `operator_api_fps`; the binary travels to the GPU box with the snapshot."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "operator_api_bench.cpp")
    inc = os.path.join(root, "include")
    deps = [LIB] + [os.path.join(dp, f) for dp, _, fs in os.walk(inc) for f in fs]
    if _newer(src, BENCH_BIN, deps):
        cmd = ["g++", "-std=c++17", "-O2", "-I" + inc, src, "-L" + HERE, "-lhp_hip", "-lpthread", "-Wl,-rpath," + HERE, "-o", BENCH_BIN]
        if verbose:
            print("[hyperpose_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return BENCH_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_operator_bench()
