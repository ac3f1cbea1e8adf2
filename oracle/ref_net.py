"""oracle/ref_net.py — TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 (CPU) evaluation of an ``hp_layer`` list: the conv-stack oracle for the engine.  The
reference's engine is TensorRT running an exported TensorFlow graph (src/tensorrt.cpp:393); neither is
available here and no reference activations exist ("parity unpinned"), so the oracle is the
textbook definition of every layer with TensorFlow's "SAME" padding (pad_total = max((ceil(i/s)-1)*s +
(k-1)*d + 1 - i, 0), the extra pixel at the bottom/right), cross-checked op by op against
``torch.nn.functional.conv2d``.  Pre-processing follows src/data.cpp:21-51 (oracle/paf_oracle.cpp).

``match_fp16=True`` rounds weights and every stored activation to fp16 exactly where the engine does (HBM
storage), keeping fp32 accumulation: the comparison then isolates kernel bugs from quantisation noise.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import loader

OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_UPSAMPLE = 1, 2, 3, 4


def _same_pad(size, k, s, d):
    out = -(-size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return out, total // 2, total - total // 2


def _act(x, act, param, alpha):
    if act == 1:
        return F.relu(x)
    if act == 2:
        return torch.clamp(x, 0, 6)
    if act == 3:
        return torch.where(x > 0, x, x * param)
    if act == 4:
        return torch.where(x > 0, x, x * alpha.view(1, -1, 1, 1))
    if act == 5:
        return torch.sigmoid(x)
    if act == 6:
        return F.softplus(x)
    return x


def _q(x, on):
    return x.half().float() if on else x


def run(layers, outputs, weights: np.ndarray, frames_u8: np.ndarray = None, frames_f32: np.ndarray = None,
        factor=1.0 / 255, flip_rb=True, mean=(0, 0, 0), inv_std=(1, 1, 1), match_fp16=True, return_tensors=False, device="cpu"):
    """Returns {name: [n,C,H,W] float32} (and the dict of internal tensors when return_tensors).  ``device="cuda"`` evaluates the
    same fp32 definition with PyTorch's own GPU kernels (MIOpen / rocBLAS - an implementation independent of libhp_hip.so) so that
    the full-size BASELINE configurations can be checked in seconds; TF32-style shortcuts are switched off."""
    torch.set_num_threads(max(1, torch.get_num_threads()))
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.backends.cudnn.allow_tf32 = False
        torch.backends.cuda.matmul.allow_tf32 = False
    w = torch.from_numpy(np.ascontiguousarray(weights, np.float32)).to(dev)
    if frames_u8 is not None:
        x0 = torch.from_numpy(loader.nhwc_u8_to_nchw_f32(frames_u8, factor, flip_rb))
    else:
        x0 = torch.from_numpy(np.ascontiguousarray(frames_f32, np.float32))
    x0 = x0.to(dev)
    x0 = (x0 - torch.tensor(mean, dtype=torch.float32, device=dev).view(1, 3, 1, 1)) * torch.tensor(inv_std, dtype=torch.float32, device=dev).view(1, 3, 1, 1)
    tensors = {0: x0}
    for L in layers:
        x = tensors[L.in_][:, L.in_coff:L.in_coff + L.cin]
        if L.op == OP_UPSAMPLE:  # integer scale in `stride`; kh = 0 nearest, 1 bilinear with half-pixel centres
            y = F.interpolate(x, scale_factor=L.stride, mode="nearest") if L.kh == 0 else \
                F.interpolate(x, scale_factor=L.stride, mode="bilinear", align_corners=False)
            oh, ow = y.shape[2], y.shape[3]
            pt = pl = pb = pr = 0
        elif getattr(L, "pad_explicit", 0):
            pt, pl, pb, pr = (int(v) for v in L.pad)
            oh = (x.shape[2] + pt + pb - ((L.kh - 1) * L.dil + 1)) // L.stride + 1
            ow = (x.shape[3] + pl + pr - ((L.kw - 1) * L.dil + 1)) // L.stride + 1
        else:
            oh, pt, pb = _same_pad(x.shape[2], L.kh, L.stride, L.dil)
            ow, pl, pr = _same_pad(x.shape[3], L.kw, L.stride, L.dil)
        first = (L.in_ == 0)
        if L.op == OP_UPSAMPLE:
            pass
        elif L.op == OP_MAXPOOL:
            xp = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
            y = F.max_pool2d(xp, L.kh, L.stride)
        else:
            xp = F.pad(x, (pl, pr, pt, pb))
            if L.op == OP_CONV:
                wt = w[L.w_off:L.w_off + L.cout * L.kh * L.kw * L.cin].view(L.cout, L.kh, L.kw, L.cin).permute(0, 3, 1, 2)
                groups = 1
            else:
                wt = w[L.w_off:L.w_off + L.cin * L.kh * L.kw].view(L.cin, 1, L.kh, L.kw)
                groups = L.cin
            # the first layer: on the fp16 matrix pipe (normalised input and weights rounded to fp16, first_conv_f16_kernel) for the
            # common 3 x 3 / 7 x 7 stems with <= 64 output channels, all-fp32 otherwise
            first_f16 = first and L.op == OP_CONV and L.kh == L.kw and L.kh in (3, 7) and L.cout % 8 == 0 and L.cout <= 64 and L.stride in (1, 2)
            wt = _q(wt, match_fp16 and (not first or first_f16))
            if first_f16:
                xp = _q(xp, match_fp16)
            b = w[L.b_off:L.b_off + L.cout] if L.b_off >= 0 else None
            y = F.conv2d(xp, wt.contiguous(), b, stride=L.stride, dilation=L.dil, groups=groups)
            alpha = w[L.alpha_off:L.alpha_off + L.cout] if L.alpha_off >= 0 else None
            if L.res >= 0:
                r = tensors[L.res][:, :L.cout]
                y = _act(y + r, L.act, L.act_param, alpha) if L.res_before_act else _act(y, L.act, L.act_param, alpha) + r
            else:
                y = _act(y, L.act, L.act_param, alpha)
        assert y.shape[2] == oh and y.shape[3] == ow
        full = tensors.get(L.out)
        need = L.out_coff + L.cout
        if full is None:
            full = torch.zeros(y.shape[0], need, oh, ow, device=dev)
        elif full.shape[1] < need:
            full = torch.cat([full, torch.zeros(y.shape[0], need - full.shape[1], oh, ow, device=dev)], 1)
        full = full.clone()
        full[:, L.out_coff:need] = y  # un-rounded copy kept for fused fp32 outputs
        tensors[L.out] = full
        tensors[("raw", L.out, L.out_coff)] = y
        # what later layers read back from HBM is the fp16-rounded value
        tensors[L.out][:, L.out_coff:need] = _q(y, match_fp16)
    result = {}
    for o in outputs:
        name = o.name.decode() if isinstance(o.name, bytes) else o.name
        shuffle = getattr(o, "shuffle", 0) or 1
        group = getattr(o, "group", 0)
        scale = getattr(o, "scale", 0.0) or 1.0
        grid = getattr(o, "grid", 0)
        out_h, out_w = getattr(o, "out_h", 0), getattr(o, "out_w", 0)
        plain = shuffle == 1 and group == 0 and scale == 1.0 and grid == 0 and not out_h and not out_w
        raw = tensors.get(("raw", o.tensor, o.coff))
        if plain and raw is not None and raw.shape[1] == o.channels and o.act == 0:
            v = raw  # conv epilogue writes the fp32 accumulator result directly
        else:
            v = tensors[o.tensor][:, o.coff:o.coff + o.channels]
            if shuffle == 2:  # hyperpose/Model/pifpaf/utils.py:371-379
                v = F.pixel_shuffle(v, 2)
            if out_h or out_w:
                v = v[:, :, :out_h or None, :out_w or None]
            if group:
                v = v.clone()
                for comp in range(group):
                    if (getattr(o, "sigmoid_mask", 0) >> comp) & 1:
                        v[:, comp::group] = torch.sigmoid(v[:, comp::group])
                    elif (getattr(o, "softplus_mask", 0) >> comp) & 1:
                        v[:, comp::group] = F.softplus(v[:, comp::group])
            else:
                v = _act(v, o.act, 0.0, None)
            if grid == 1:
                v = v + torch.arange(v.shape[3], dtype=torch.float32, device=dev).view(1, 1, 1, -1)
            elif grid == 2:
                v = v + torch.arange(v.shape[2], dtype=torch.float32, device=dev).view(1, 1, -1, 1)
            v = v * scale
        result[name] = v.cpu().numpy()
    if return_tensors:
        return result, {k: v.cpu().numpy() for k, v in tensors.items() if isinstance(k, int)}
    return result
