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
