"""GPU: imported ONNX models run by the HIP engine vs (a) PyTorch's fp32 outputs of the exported module (fp16 storage bound:
1e-2 of the output range) and (b) the torch oracle evaluated with the engine's fp16 storage points (2e-3: summation order only)."""
import os

import numpy as np
import pytest

from hyperpose_amd import engine as E
from oracle import ref_net

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onnx")


@pytest.mark.parametrize("name", ["mobile_paf", "resnet_ppn", "vgg_stages", "unfolded", "small_upsample"])
def test_imported_model_on_the_engine(hp, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = E.Model.from_onnx(os.path.join(GOLD, name + ".onnx"))
    eng = E.Engine.from_model(m, m.weights, max_batch=2)
    got = eng.inference_f32(z["image"])
    tight = ref_net.run(m.layers, m.outputs, m.weights, frames_f32=z["image"], mean=m.mean, inv_std=m.inv_std, match_fp16=True)
    for b in range(2):
        assert [nm for nm, _ in got[b]] == sorted(tight)
        for nm, arr in got[b]:
            scale = np.abs(z[nm]).max()
            assert np.abs(arr - z[nm][b]).max() <= 1e-2 * scale + 1e-3, (nm, "vs PyTorch fp32")
            assert np.abs(arr - tight[nm][b]).max() <= 2e-3 * scale + 1e-3, (nm, "vs fp16-matched oracle")


def test_imported_model_u8_frames_and_serialized_engine(hp, tmp_path):
    """The whole reference flow on an imported file: u8 BGR frames -> (flip, 1/255, in-graph normalisation) -> outputs; then
    tensorrt::save / tensorrt_serialized (examples/gen_serialized_engine.example.cpp:30-60) round-trips it."""
    m = E.Model.from_onnx(os.path.join(GOLD, "vgg_stages.onnx"))
    frames = np.random.default_rng(5).integers(0, 256, (2, m.in_h, m.in_w, 3), dtype=np.uint8)
    eng = E.Engine.from_model(m, m.weights, max_batch=2)
    got = eng.inference(frames)
    ref = ref_net.run(m.layers, m.outputs, m.weights, frames_u8=frames, mean=m.mean, inv_std=m.inv_std, match_fp16=True)
    for b in range(2):
        for nm, arr in got[b]:
            assert np.abs(arr - ref[nm][b]).max() <= 2e-3 * np.abs(ref[nm]).max() + 1e-3, nm
    path = str(tmp_path / "vgg_stages.hpeng")
    eng.save(path)
    again = E.Engine.load(path).inference(frames)
    for b in range(2):
        for (n0, a0), (n1, a1) in zip(got[b], again[b]):
            assert n0 == n1 and np.array_equal(a0, a1)


def test_config1_model_from_onnx_runs_the_same_schedule(hp, tmp_path):
    """BASELINE.json config[1] (LW-OpenPose / MobilenetDilated at 432 x 368) arriving as an ONNX file: the imported engine
    launches the same kernels as the built-in topology (same per-step tile codes: fused separable blocks, fused heads, direct
    3x3) and its outputs are bit-identical."""
    import torch_from_layers as T
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    blob = m.init_weights(20240)
    path = str(tmp_path / "lw_openpose.onnx")
    T.export(m.layers, m.outputs, blob, 368, 432, path, m.mean, m.inv_std)
    im = E.Model.from_onnx(path, 432, 368)
    assert T.signature(im.layers) == T.signature(m.layers)
    frames = np.random.default_rng(1).integers(0, 256, (2, 368, 432, 3), dtype=np.uint8)
    a = E.Engine.from_model(m, blob, max_batch=2)
    b = E.Engine.from_model(im, im.weights, max_batch=2)
    ga, gb = a.inference(frames), b.inference(frames)
    for f in range(2):
        for (n0, x0), (n1, x1) in zip(ga[f], gb[f]):
            assert n0 == n1 and np.array_equal(x0, x1)
    assert [(t["op"], t["tile"]) for t in a.profile(2, 1)] == [(t["op"], t["tile"]) for t in b.profile(2, 1)]


def test_post_processing_operators_run(hp):
    """Split / Reshape / Transpose-pair post-processing (test_onnx_import._post_op_model): values == the torch evaluation of the same graph."""
    import torch
    import test_onnx_import as T
    m = E.Model.from_onnx(T._post_op_model())
    eng = E.Engine.from_model(m, m.weights, max_batch=2, factor=1.0, flip_rgb=False)
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 256, (2, 6, 8, 3), dtype=np.uint8)
    got = eng.inference(fr)
    x = torch.from_numpy(fr.astype(np.float32)).permute(0, 3, 1, 2)
    w = torch.tensor([0.01 * (i + 1) for i in range(36)], dtype=torch.float32).view(12, 3, 1, 1)
    b = torch.tensor([0.1 * i for i in range(12)], dtype=torch.float32)
    y = torch.sigmoid(torch.nn.functional.conv2d(x, w, b)).numpy()
    for f in range(2):
        named = dict(got[f])
        assert np.allclose(named["out_a"], y[f, :4], atol=2e-3)
        assert np.allclose(named["out_b"], y[f, 4:], atol=2e-3)   # [8, 6, 8] = the [2, 4, 6, 8] view's memory


def test_tensorflow_export_idioms_run_on_both_engines(hp):
    """tests/test_onnx_import.py::tf2onnx_like_model (N,H,W,3 input + Transpose, auto_pad SAME_UPPER, un-folded BatchNormalization, Clip with
    input bounds, an activation inside a Transpose sandwich, Sigmoid -> Split -> restore_coor arithmetic) through the importer and the
    engine in both precisions, against the PyTorch evaluation of the same graph."""
    import test_onnx_import as T
    raw, ref = T.tf2onnx_like_model()
    m = E.Model.from_onnx(raw)
    x = np.random.default_rng(2).random((2, 64, 96, 3), dtype=np.float32)
    want = ref(x)
    nchw = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    for dtype, rel in (("f32", 1e-4), ("f16", 5e-3)):
        eng = E.Engine.from_model(m, m.weights, max_batch=2, dtype=dtype)
        got = eng.inference_f32(nchw)
        for f in range(2):
            named = dict(got[f])
            assert sorted(named) == sorted(want)
            for k, v in want.items():
                assert np.abs(named[k] - v[f]).max() <= rel * np.abs(v).max() + rel, (dtype, k)
