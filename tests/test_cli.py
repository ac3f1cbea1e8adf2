"""examples/cli.cpp (the reference's `hyperpose-cli` flag surface, examples/cli.cpp:15-35) builds against the mirror headers and runs
on the GPU in both runtimes with all three parsers; draw_human changes pixels."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "cli.cpp")
BIN = os.path.join(ROOT, "examples", "hyperpose-cli.bin")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + os.path.join(ROOT, "hyperpose_amd"),
                           "-lhp_hip", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "hyperpose_amd"), "-o", BIN])


def test_cli_builds_and_rejects_unknown_flags():
    _build()
    r = subprocess.run([BIN, "--no_such_flag=1"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown command line flag" in r.stdout
    r = subprocess.run([BIN, "--w"], capture_output=True, text=True)
    assert r.returncode == 1 and "needs a value" in r.stdout


def _read_ppm(path):
    raw = open(path, "rb").read()
    head, rest = raw.split(b"\n255\n", 1)
    w, h = (int(v) for v in head.split()[1:3])
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("post,model,runtime", [("paf", "builtin:lw_openpose_mobilenet", "operator"), ("paf", "builtin:lw_openpose_mobilenet", "stream"),
                                                ("ppn", "builtin:pose_proposal_resnet50", "operator"), ("pifpaf", "builtin:pifpaf_resnet50", "stream")])
def test_cli_runs(tmp_path, post, model, runtime):
    _build()
    prefix = str(tmp_path / "out")
    size = ["--w", "161", "--h=129"] if post == "pifpaf" else ["--w", "160", "--h=128"]
    r = subprocess.run([BIN, "--model", model, "--post=" + post, *size, "--max_batch_size", "3", "--source=synthetic:5:200x150",
                        "--runtime", runtime, "--nokeep_ratio" if post == "ppn" else "--keep_ratio", "--alpha=0.5", "--saving_prefix", prefix, "--noimshow"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "5 images got processed" in r.stdout
    for i in range(5):
        img = _read_ppm(f"{prefix}_{i}.ppm")
        assert img.shape == (150, 200, 3)


@pytest.mark.gpu
def test_draw_human_and_ppm_source(tmp_path):
    """A PPM directory as the source; with alpha = 1 and loose thresholds nothing but the drawn skeletons may differ from the input."""
    _build()
    rng = np.random.default_rng(3)
    src_dir = tmp_path / "media"
    src_dir.mkdir()
    frames = []
    for i in range(2):
        a = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
        frames.append(a)
        with open(src_dir / f"f{i}.ppm", "wb") as f:
            f.write(b"P6\n# comment line\n160 120\n255\n" + a.tobytes())
    prefix = str(tmp_path / "o")
    r = subprocess.run([BIN, "--model=builtin:lw_openpose_mobilenet", "--w=160", "--h=128", "--source", str(src_dir), "--alpha", "0",
                        "--saving_prefix", prefix, "--nokeep_ratio"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(2):
        out = _read_ppm(f"{prefix}_{i}.ppm")
        changed = np.any(out != frames[i], axis=2)
        assert changed.mean() < 0.5   # alpha = 0 keeps the skeleton pixels as drawn, everything else untouched
