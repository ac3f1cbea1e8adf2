"""Test helper: an ``hp_layer`` list + weight blob as a torch.nn.Module, and its export through PyTorch's ONNX serializer.
Used to check that a model of the reference's real topologies (hyperpose/Model/*.py, restated by hp_model_build) that
arrives as an ONNX file is lowered by hp_model_from_onnx to the SAME network the built-in topology describes."""
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

OP_CONV, OP_DWCONV, OP_MAXPOOL = 1, 2, 3


def _same(size, k, s, d):
    out = -(-size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return total // 2, total - total // 2


class LayerNet(nn.Module):
    def __init__(self, layers, outputs, blob, in_h, in_w, mean=(0, 0, 0), inv_std=(1, 1, 1)):
        super().__init__()
        self.layers, self.outs, self.in_h, self.in_w = layers, outputs, in_h, in_w
        w = torch.from_numpy(np.ascontiguousarray(blob, np.float32))
        self.ws, self.bs, self.alphas = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        for L in layers:
            if L.op == OP_CONV:
                wt = w[L.w_off:L.w_off + L.cout * L.kh * L.kw * L.cin].view(L.cout, L.kh, L.kw, L.cin).permute(0, 3, 1, 2)
            elif L.op == OP_DWCONV:
                wt = w[L.w_off:L.w_off + L.cin * L.kh * L.kw].view(L.cin, 1, L.kh, L.kw)
            else:
                wt = torch.zeros(1)
            self.ws.append(nn.Parameter(wt.contiguous().clone()))
            self.bs.append(nn.Parameter(w[L.b_off:L.b_off + L.cout].clone() if L.b_off >= 0 else torch.zeros(1)))
            self.alphas.append(nn.Parameter(w[L.alpha_off:L.alpha_off + L.cout].clone() if L.alpha_off >= 0 else torch.zeros(1)))
        self.normalise = any(m != 0 for m in mean) or any(s != 1 for s in inv_std)
        self.register_buffer("mean", torch.tensor(list(mean), dtype=torch.float32).view(1, 3, 1, 1))
        self.register_buffer("inv_std", torch.tensor(list(inv_std), dtype=torch.float32).view(1, 3, 1, 1))

    def forward(self, x):
        if self.normalise:
            x = (x - self.mean) * self.inv_std
        pieces = {0: [(0, 3, x)]}  # tensor id -> [(coff, channels, value)]
        sizes = {0: (self.in_h, self.in_w)}

        def read(t, coff, c):
            ps = sorted(pieces[t], key=lambda p: p[0])
            for o, n, v in ps:
                if o == coff and n == c:
                    return v
            assert coff == 0 and sum(p[1] for p in ps) == c, "reads must cover one writer or the whole concatenation"
            return torch.cat([p[2] for p in ps], 1)

        for i, L in enumerate(self.layers):
            v = read(L.in_, L.in_coff, L.cin)
            H, W = sizes[L.in_]
            (pt, pb), (pl, pr) = _same(H, L.kh, L.stride, L.dil), _same(W, L.kw, L.stride, L.dil)
            if L.op == OP_MAXPOOL:
                if pt or pb or pl or pr:
                    v = F.pad(v, (pl, pr, pt, pb), value=float("-inf"))
                y = F.max_pool2d(v, L.kh, L.stride)
            else:
                if pt == pb and pl == pr:
                    padding = (pt, pl)
                else:
                    v, padding = F.pad(v, (pl, pr, pt, pb)), 0
                y = F.conv2d(v, self.ws[i], self.bs[i] if L.b_off >= 0 else None, L.stride, padding, L.dil,
                             L.cin if L.op == OP_DWCONV else 1)
                act = {0: lambda t: t, 1: F.relu, 2: lambda t: torch.clamp(t, 0.0, 6.0), 3: lambda t: F.leaky_relu(t, L.act_param),
                       4: lambda t: F.prelu(t, self.alphas[i])}[L.act]
                if L.res >= 0:
                    r = read(L.res, 0, L.cout)
                    y = act(y + r) if L.res_before_act else act(y) + r
                else:
                    y = act(y)
            sizes[L.out] = (-(-H // L.stride), -(-W // L.stride))  # python ints: traced shapes would become graph nodes
            kept = [q for q in pieces.get(L.out, []) if q[0] != L.out_coff]  # a later writer of the same channels replaces the earlier
            pieces[L.out] = kept + [(L.out_coff, L.cout, y)]
        res = []
        for o in self.outs:
            v = read(o.tensor, o.coff, o.channels)
            res.append(torch.sigmoid(v) if o.act == 5 else F.softplus(v) if o.act == 6 else v)
        return tuple(res)


def export(layers, outputs, blob, in_h, in_w, path, mean=(0, 0, 0), inv_std=(1, 1, 1), opset=10):
    """opset 10: Pad / Clip carry their constants as attributes (from 11 on PyTorch emits a shape subgraph for F.pad)."""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # the `onnx` package is not installed
    net = LayerNet(layers, outputs, blob, in_h, in_w, mean, inv_std).eval()
    names = [o.name.decode() if isinstance(o.name, bytes) else o.name for o in outputs]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(net, torch.zeros(1, 3, in_h, in_w), path, dynamo=False, opset_version=opset, input_names=["image"],
                          output_names=names, dynamic_axes={"image": {0: "batch"}})
    return net


def signature(layers):
    """Layer list up to tensor renumbering: what each layer computes and how it is wired (by writer index, not tensor id)."""
    writers, sig = {}, []
    for i, L in enumerate(layers):
        src = tuple(sorted((j, layers[j].out_coff) for j in writers.get(L.in_, []))) if L.in_ else "image"
        res = tuple(sorted(writers.get(L.res, []))) if L.res >= 0 else None
        sig.append((L.op, src, L.in_coff, res, L.res_before_act if L.res >= 0 else 0, L.out_coff, L.cin, L.cout, L.kh, L.kw, L.stride,
                    L.dil, L.act, round(L.act_param, 6), L.pad_explicit, L.b_off >= 0))
        writers.setdefault(L.out, []).append(i)
    return sig
