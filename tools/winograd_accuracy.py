#!/usr/bin/env python
"""tools/winograd_accuracy.py — what would F(4x4,3x3) cost in accuracy?  (CPU only, no GPU needed.)

The fp32 engine runs its 3 x 3 stride-1 layers in Winograd's F(2x2,3x3) form (csrc/conv32_winograd.hip: 16 products per 2 x 2 tile).  F(4x4,3x3)
needs 36 per 4 x 4 tile - 1.78 x fewer matrix-pipe cycles - but its transforms have entries up to 8 (At), 5 (Bt) and 1/24 (G), so the
cancellation error grows.  This script evaluates LW-OpenPose (configs[1]) at full size on the drift test's frames and weights
(tests/test_pipeline_gpu.py::_engine_vs_fp32_oracle_keypoint_drift) with every 3 x 3 stride-1 layer of >= 16 input channels computed
  (a) directly in fp32 (PyTorch CPU: the tests' oracle),  (b) in F(2x2,3x3) fp32,  (c) in F(4x4,3x3) fp32,  (d) directly in fp64 (the yardstick),
all transforms and products in fp32 arithmetic (U = G g Gt formed in fp64 and rounded once, like conv32_winograd_pack), and reports the
heat-map error of each against (d) and against (a) - the engine tests' 1e-4-of-scale bound is against (a) - and the peaks / humans the
reference-compiled PAF parser finds on each (flipped peaks = the drift test's zero-flip criterion).

    python tools/winograd_accuracy.py [frames=4] > profiles/r06_winograd_f43_accuracy.txt
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperpose_amd import engine as E  # noqa: E402
from oracle import loader, ref_net  # noqa: E402

MATS = {
    2: dict(Bt=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
            G=[[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
            At=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    # F(3 x 3, 3 x 3): Cook-Toom at the points 0, 1, -1, 2, inf (25 products per 3 x 3 tile; entries up to 3 (Bt), 4 (At), 1/6 .. 2/3 (G))
    3: dict(Bt=[[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]],
            G=[[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]],
            At=[[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]]),
    4: dict(Bt=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
            G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
            At=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def winograd_conv(xp, wt, bias, m):
    """valid 3 x 3 convolution of the already padded xp [N,C,H+2,W+2] in F(m x m, 3 x 3), fp32 arithmetic throughout"""
    mats = MATS[m]
    Bt = torch.tensor(mats["Bt"], dtype=torch.float32)
    At = torch.tensor(mats["At"], dtype=torch.float32)
    G = torch.tensor(mats["G"], dtype=torch.float64)
    n, c, hp, wp = xp.shape
    h, w = hp - 2, wp - 2
    th, tw = -(-h // m), -(-w // m)
    xp = F.pad(xp, (0, tw * m - w, 0, th * m - h))
    a = m + 2
    d = xp.unfold(2, a, m).unfold(3, a, m)                       # [N, C, th, tw, a, a]
    # V = Bt d B: two passes of fp32 additions (rows, then columns), as a kernel would do them
    V = torch.einsum("ij,nctujk->nctuik", Bt, d)
    V = torch.einsum("nctuik,lk->nctuil", V, Bt)
    U = torch.einsum("ij,kcjl,ml->kcim", G, wt.double(), G).float()  # [K, C, a, a], rounded to fp32 once
    M = torch.einsum("nctuij,kcij->nktuij", V, U)                # fp32 products and sums over c
    Y = torch.einsum("pi,nktuij->nktupj", At, M)
    Y = torch.einsum("nktupj,qj->nktupq", Y, At)                 # [N, K, th, tw, m, m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, wt.shape[0], th * m, tw * m)[:, :, :h, :w]
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def run_variant(model, weights, frames, mode):
    real = F.conv2d

    def conv2d(xp, wt, b=None, stride=1, dilation=1, groups=1, **kw):
        if mode == "f64":
            return real(xp.double(), wt.double(), None if b is None else b.double(), stride=stride, dilation=dilation, groups=groups).float()
        if mode in (2, 3, 4) and groups == 1 and wt.shape[2:] == (3, 3) and stride == 1 and dilation == 1 and wt.shape[1] >= 16:
            return winograd_conv(xp, wt, b, mode)
        return real(xp, wt, b, stride=stride, dilation=dilation, groups=groups)

    ref_net.F.conv2d = conv2d
    try:
        return ref_net.run(model.layers, model.outputs, weights, frames_u8=frames, match_fp16=False)
    finally:
        ref_net.F.conv2d = real


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    in_w, in_h = 432, 368
    m = E.Model("lw_openpose_mobilenet", in_w, in_h)
    w = m.init_weights(11)
    for L in m.layers:
        if L.op == E.OP_CONV and L.cout in (19, 38) and L.out in [o.tensor for o in m.outputs]:
            w[L.w_off:L.w_off + L.cout * L.cin] *= 400.0
    frames = np.random.default_rng(21).integers(0, 256, (8, in_h, in_w, 3), dtype=np.uint8)[:B]
    n33 = sum(1 for L in m.layers if L.op == E.OP_CONV and L.kh == 3 and L.stride == 1 and L.dil == 1 and L.cin >= 16)
    print(f"LW-OpenPose @ {in_h}x{in_w}, {B} frames of the drift test, {n33} 3x3 stride-1 layers in the form under test; heads x400 as in the drift test")
    res = {k: run_variant(m, w, frames, k) for k in ("f64", "direct", 2, 3, 4)}
    thr = 0.05
    peaks = {}
    for k, r in res.items():
        pk, nh = [], 0
        for b in range(B):
            oh, op, _ = loader.ref_paf_process(r["conf"][b], r["paf"][b], thr, -1e9, cap_humans=256, cap_peaks=65536, cap_conns=65536)
            pk.append({(int(p["part_id"]), int(p["y"]), int(p["x"])) for p in op})
            nh += len(oh)
        peaks[k] = (pk, nh)
    for k in ("direct", 2, 3, 4):
        name = {"direct": "direct fp32 (torch CPU)", 2: "F(2x2,3x3) fp32", 3: "F(3x3,3x3) fp32", 4: "F(4x4,3x3) fp32"}[k]
        for base in ("f64", "direct"):
            if base == k:
                continue
            err = max(float(np.abs(res[k][n] - res[base][n]).max() / np.abs(res[base][n]).max()) for n in ("conf", "paf"))
            same = sum(len(a & b_) for a, b_ in zip(peaks[k][0], peaks[base][0]))
            tot = sum(len(b_) for b_ in peaks[base][0])
            mine = sum(len(a) for a in peaks[k][0])
            print(f"{name:26s} vs {base:6s}: heat-map max err / scale {err:.3e}; peaks {same} of {tot} identical ({mine} found); humans {peaks[k][1]} vs {peaks[base][1]}")


if __name__ == "__main__":
    main()
