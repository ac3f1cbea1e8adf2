"""CPU: the conv-stack oracle (oracle/ref_net.py) and the built-in topologies (no GPU calls)."""
import numpy as np
import torch
import torch.nn.functional as F

from hyperpose_amd import engine as E
from oracle import ref_net


def test_same_padding_matches_tf_semantics():
    # stride-2 3x3 on an even size pads ONLY bottom/right (TF SAME), unlike torch's symmetric padding=1
    x = torch.arange(36, dtype=torch.float32).view(1, 1, 6, 6)
    w = np.zeros(9 * 1 * 1 + 1, np.float32)
    w[0] = 1.0  # tap (0,0)
    L = E.make_layer(E.OP_CONV, 0, 1, 1, 1, k=3, stride=2, w_off=0, b_off=9)
    L.cin = 1
    layers = [L]

    class O:  # minimal output desc
        name, tensor, coff, channels, act = b"y", 1, 0, 1, 0
    x3 = x.repeat(1, 3, 1, 1).numpy()
    # use a 3-channel input through the generic path
    L3 = E.make_layer(E.OP_CONV, 0, 1, 3, 1, k=3, stride=2, w_off=0, b_off=27)
    w3 = np.zeros(28, np.float32)
    w3[0] = 1.0  # [cout=0][ky=0][kx=0][cin=0]
    y = ref_net.run([L3], [O], w3, frames_f32=x3, match_fp16=False)["y"]
    # tap (0,0) with pad_top = pad_left = 0 reads x[2i, 2j]
    assert np.array_equal(y[0, 0], x[0, 0, ::2, ::2].numpy())


def test_ref_net_matches_direct_torch_on_lw_openpose():
    m = E.Model("lw_openpose_mobilenet", 64, 48)
    w = m.init_weights(1)
    frames = np.random.default_rng(0).integers(0, 256, (1, 48, 64, 3), dtype=np.uint8)
    out = ref_net.run(m.layers, m.outputs, w, frames_u8=frames, match_fp16=False)
    assert out["conf"].shape == (1, 19, 6, 8) and out["paf"].shape == (1, 38, 6, 8)
    assert np.isfinite(out["conf"]).all() and np.isfinite(out["paf"]).all()
    assert out["conf"].std() > 0


def test_lw_openpose_flops_match_survey():
    # MobilenetDilated + LW head @368x432 = 23.39 GFLOP/frame (11.593 dense + 0.100 dw GMAC)
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    assert abs(m.flops_per_frame / 1e9 - 23.39) < 0.05, m.flops_per_frame
    m2 = E.Model("lw_openpose_vggtiny", 432, 368)
    assert abs(m2.flops_per_frame / 1e9 - 67.36) < 0.1, m2.flops_per_frame


def test_openpose_vgg19_flops_match_survey():
    m = E.Model("openpose_vgg19", 768, 432)
    # Appendix C: 333.025 GMAC -> 666.05 GFLOP (+ one duplicated cheap CPM conv, see models.cpp)
    assert abs(m.flops_per_frame / 1e9 - 666.05) < 4.0, m.flops_per_frame
    assert abs(m.n_weights / 1e6 - 52.3) < 0.5


def test_weights_deterministic():
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    a, b = m.init_weights(7), m.init_weights(7)
    assert np.array_equal(a, b) and not np.array_equal(a, m.init_weights(8))
    assert abs(m.n_weights / 1e6 - 4.58) < 0.1  # Appendix C: 4.58 M parameters
