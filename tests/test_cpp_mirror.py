"""The C++ mirror headers (include/hyperpose/) compile against the C ABI with plain g++ (CPU) and the
reference's operator-API flow runs through them on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "operator_api_paf.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "operator_api_paf.bin")


SRC2 = os.path.join(ROOT, "tests", "cpp", "reference_call_sites.cpp")
BIN2 = os.path.join(ROOT, "tests", "cpp", "reference_call_sites.bin")


def _build(src=SRC, out=BIN):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "hyperpose_amd"), "-lhp_hip", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "hyperpose_amd"), "-o", out])


def test_mirror_headers_compile():
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_operator_api_flow_runs():
    _build()
    out = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden", "onnx", "mobile_paf.onnx")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    tag, n_packets, humans, threw = out.stdout.split()[-4:]
    assert tag == "OK" and int(n_packets) == 3 and int(threw) == 1


def test_reference_call_sites_compile():
    """The reference's own engine-construction / inference / make_stream lines (examples/operator_api_batched_images_paf.example.cpp:36-74,
    examples/stream_api_video_paf.example.cpp:74-88), pasted verbatim into tests/cpp/reference_call_sites.cpp, compile against the mirror."""
    _build(SRC2, BIN2)
    assert os.path.exists(BIN2)


@pytest.mark.gpu
def test_reference_call_sites_run():
    _build(SRC2, BIN2)
    # a real Lightweight-OpenPose graph (19 conf / 38 paf channels) exported by PyTorch's ONNX serializer at the test's 48 x 64 input
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_from_layers as T
    from hyperpose_amd import engine as E
    m = E.Model("lw_openpose_mobilenet", 48, 64)
    path = os.path.join(tempfile.mkdtemp(), "lw_openpose.onnx")
    T.export(m.layers, m.outputs, m.init_weights(3), 64, 48, path, m.mean, m.inv_std)
    out = subprocess.run([BIN2, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    tag, humans, frames, stream_humans, threw = out.stdout.split()[-5:]
    assert tag == "OK" and int(frames) == 11 and int(threw) == 1


# ---- the reference's stream operator, unchanged, over this engine and parser
SRC3 = os.path.join(ROOT, "tests", "cpp", "reference_stream_unchanged.cpp")
BIN3 = os.path.join(ROOT, "tests", "cpp", "reference_stream_unchanged.bin")
REF = os.environ.get("HP_REFERENCE", "/root/reference")


def _build_reference_stream():
    """An include tree of SYMLINKS: the reference's stream.hpp / thread_pool.hpp / thread_safe_queue.hpp / logging.hpp as they lie, this
    repo's data / human / model / operator headers under the names the reference header asks for, the OpenCV stand-in of
    tests/cpp/cv_stream_shim.hpp; compiled together with the reference's src/stream.cpp, src/thread_pool.cpp, src/logging.cpp."""
    import shutil
    import tempfile
    from oracle import loader
    loader.build(ref=False)
    tree = tempfile.mkdtemp(prefix="hp_stream_tree_")
    try:
        os.makedirs(os.path.join(tree, "hyperpose", "stream"))
        os.makedirs(os.path.join(tree, "hyperpose", "utility"))
        os.makedirs(os.path.join(tree, "opencv2"))
        os.symlink(os.path.join(REF, "include", "hyperpose", "stream", "stream.hpp"), os.path.join(tree, "hyperpose", "stream", "stream.hpp"))
        for f in ("thread_pool.hpp", "thread_safe_queue.hpp", "logging.hpp"):
            os.symlink(os.path.join(REF, "include", "hyperpose", "utility", f), os.path.join(tree, "hyperpose", "utility", f))
        for f in ("data.hpp", "human.hpp", "model.hpp", "cv_min.hpp"):
            os.symlink(os.path.join(ROOT, "include", "hyperpose", "utility", f), os.path.join(tree, "hyperpose", "utility", f))
        os.symlink(os.path.join(ROOT, "include", "hyperpose", "operator"), os.path.join(tree, "hyperpose", "operator"))
        os.symlink(os.path.join(ROOT, "include", "hp_hip.h"), os.path.join(tree, "hp_hip.h"))
        os.symlink(os.path.join(ROOT, "tests", "cpp", "cv_stream_shim.hpp"), os.path.join(tree, "opencv2", "opencv.hpp"))
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + tree, "-I" + os.path.join(REF, "src"), SRC3,
                               os.path.join(REF, "src", "stream.cpp"), os.path.join(REF, "src", "thread_pool.cpp"), os.path.join(REF, "src", "logging.cpp"),
                               "-L" + os.path.join(ROOT, "hyperpose_amd"), "-lhp_hip", "-L" + os.path.join(ROOT, "oracle", "_build"), "-loracle", "-lpthread",
                               "-Wl,-rpath," + os.path.join(ROOT, "hyperpose_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build"), "-o", BIN3])
    finally:
        shutil.rmtree(tree, ignore_errors=True)


def test_reference_stream_header_compiles_unchanged():
    """include/hyperpose/stream/stream.hpp + src/stream.cpp of the reference, byte for byte as they lie under /root/reference, compile and
    link against this repo's engine / parser / data-type mirrors (only where the reference is mounted; the binary then travels to the
    GPU box with the snapshot)."""
    if not os.path.isdir(REF):
        pytest.skip("/root/reference not mounted here")
    _build_reference_stream()
    assert os.path.exists(BIN3)


@pytest.mark.gpu
def test_reference_stream_runs_unchanged():
    """The reference's four stage threads and bounded queues push 70 frames of three sizes through hyperpose::dnn::tensorrt::inference and
    hyperpose::parser::paf::process of this repo and hand all of them, at their original sizes, to the writer."""
    if not os.path.exists(BIN3):
        if not os.path.isdir(REF):
            pytest.skip("tests/cpp/reference_stream_unchanged.bin was not built (needs /root/reference at build time)")
        _build_reference_stream()
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_from_layers as T
    from hyperpose_amd import engine as E
    m = E.Model("lw_openpose_mobilenet", 64, 48)
    path = os.path.join(tempfile.mkdtemp(), "lw_openpose.onnx")
    T.export(m.layers, m.outputs, m.init_weights(3), 48, 64, path, m.mean, m.inv_std)
    out = subprocess.run([BIN3, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    tag, n, ok = out.stdout.split()[-3:]
    assert tag == "OK" and int(n) == 70 and int(ok) == 1
