"""The C++ mirror headers (include/hyperpose/) compile against the C ABI with plain g++ (CPU) and the
reference's operator-API flow runs through them on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "operator_api_paf.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "operator_api_paf.bin")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + os.path.join(ROOT, "hyperpose_amd"), "-lhp_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "hyperpose_amd"), "-o", BIN])


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
