"""Generates tests/golden/onnx/*.onnx + *.npz: small networks with the operator patterns of the reference's released
ONNX models (scripts/downloader.py:12-21: MobileNet / VGG / ResNet-50 backbones, PAF / PoseProposal / PifPaf heads),
serialized by PyTorch's own ONNX exporter (an independent producer: libtorch writes the protobuf), together with the
input and PyTorch's fp32 CPU outputs.  tests/test_onnx_import.py imports the .onnx with hp_model_from_onnx and checks
the lowered network against these outputs.

Run here (no GPU, no network):  python tests/golden/make_onnx_fixtures.py
The `onnx` Python package is not installed; the exporter only needs it for a post-processing step on custom
onnxscript functions, which these models do not have, so that hook is replaced by the identity.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "onnx")


def conv_bn(cin, cout, k, stride=1, dil=1, act=nn.ReLU, pad=None, bias=False, groups=1):
    pad = dil * (k // 2) if pad is None else pad
    mods = [nn.Conv2d(cin, cout, k, stride, pad, dil, groups, bias=bias), nn.BatchNorm2d(cout)]
    if act is not None:
        mods.append(act())
    return nn.Sequential(*mods)


class MobilePaf(nn.Module):
    """MobilenetDilated-style backbone (depthwise + pointwise blocks, one dilated, ReLU6 in one block) + LightWeight-OpenPose
    style heads: initial stage, then a refinement stage on concat(features, conf, paf) with a residual block."""

    def __init__(self):
        super().__init__()
        self.stem = conv_bn(3, 16, 3, 2)
        self.dw1 = conv_bn(16, 16, 3, 1, groups=16)
        self.pw1 = conv_bn(16, 32, 1)
        self.dw2 = conv_bn(32, 32, 3, 2, groups=32, act=nn.ReLU6)
        self.pw2 = conv_bn(32, 64, 1, act=nn.ReLU6)
        self.dw3 = conv_bn(64, 64, 3, 1, 2, groups=64)
        self.pw3 = conv_bn(64, 64, 1)
        self.cpm = conv_bn(64, 32, 1, bias=True)
        self.trunk = nn.Sequential(conv_bn(32, 32, 3), conv_bn(32, 32, 3))
        self.conf0 = nn.Sequential(conv_bn(32, 64, 1), nn.Conv2d(64, 5, 1))
        self.paf0 = nn.Sequential(conv_bn(32, 64, 1), nn.Conv2d(64, 6, 1))
        self.ref_in = conv_bn(32 + 5 + 6, 32, 1)
        self.ref_a = conv_bn(32, 32, 3)
        self.ref_b = conv_bn(32, 32, 3, dil=2)
        self.conf1 = nn.Sequential(conv_bn(32, 32, 1), nn.Conv2d(32, 5, 1))
        self.paf1 = nn.Sequential(conv_bn(32, 32, 1), nn.Conv2d(32, 6, 1))

    def forward(self, x):
        x = self.pw1(self.dw1(self.stem(x)))
        x = self.pw2(self.dw2(x))
        x = self.pw3(self.dw3(x))
        f = self.cpm(x)
        t = self.trunk(f)
        conf0, paf0 = self.conf0(t), self.paf0(t)
        r = self.ref_in(torch.cat([f, conf0, paf0], 1))
        r = r + self.ref_b(self.ref_a(r))
        return self.conf1(r), self.paf1(r)


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride, act=nn.ReLU):
        super().__init__()
        self.a = conv_bn(cin, mid, 1, act=act)
        self.b = conv_bn(mid, mid, 3, stride, act=act)
        self.c = conv_bn(mid, cout, 1, act=None)
        self.short = conv_bn(cin, cout, 1, stride, act=None) if (stride != 1 or cin != cout) else None
        self.act = act()

    def forward(self, x):
        y = self.c(self.b(self.a(x)))
        return self.act(y + (x if self.short is None else self.short(x)))


class ResNetHead(nn.Module):
    """ResNet-50 style: 7x7/2 stem with symmetric padding 3, 3x3/2 max-pool with padding 1, bottlenecks with projection and
    identity shortcuts, LeakyReLU head convolutions (PoseProposal, hyperpose/Model/pose_proposal) and sigmoid outputs."""

    def __init__(self):
        super().__init__()
        self.stem = conv_bn(3, 16, 7, 2)
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.l1 = Bottleneck(16, 8, 32, 1)
        self.l2 = Bottleneck(32, 8, 32, 1)
        self.l3 = Bottleneck(32, 16, 64, 2)
        self.head = conv_bn(64, 64, 3, act=lambda: nn.LeakyReLU(0.1))
        self.out_c = nn.Conv2d(64, 6, 1)
        self.out_e = nn.Conv2d(64, 10, 1)

    def forward(self, x):
        x = self.l3(self.l2(self.l1(self.pool(self.stem(x)))))
        h = self.head(x)
        return torch.sigmoid(self.out_c(h)), self.out_e(h)


class VggStages(nn.Module):
    """VGG-style: in-graph input normalisation, TF 'SAME' stride-2 convolution written as an explicit bottom/right pad,
    2x2 max-pools, PReLU, and two stages that both concatenate the same feature map (OpenPose CMU, openpose.py:13-198)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self.c1 = nn.Conv2d(3, 16, 3, 1, 1)
        self.p1 = nn.PReLU(16)
        self.c2 = nn.Conv2d(16, 32, 3, 2, 0)  # fed by F.pad(.., (0, 1, 0, 1)): TensorFlow's SAME at stride 2
        self.p2 = nn.PReLU(32)
        self.pool = nn.MaxPool2d(2, 2)
        self.c3 = nn.Conv2d(32, 32, 3, 1, 1)
        self.feat = nn.Conv2d(32, 32, 3, 1, 1)
        self.s1 = nn.Sequential(nn.Conv2d(32, 32, 3, 1, 1), nn.ReLU(), nn.Conv2d(32, 8, 1))
        self.s2 = nn.Sequential(nn.Conv2d(40, 32, 5, 1, 2), nn.ReLU(), nn.Conv2d(32, 8, 1))
        self.s3 = nn.Sequential(nn.Conv2d(40, 32, 3, 1, 1), nn.ReLU(), nn.Conv2d(32, 8, 1))

    def forward(self, x):
        x = (x - self.mean) / self.std
        x = self.p1(self.c1(x))
        x = self.p2(self.c2(F.pad(x, (0, 1, 0, 1))))
        x = F.relu(self.c3(self.pool(x)))
        f = F.relu(self.feat(x))
        a = self.s1(f)
        b = self.s2(torch.cat([a, f], 1))
        c = self.s3(torch.cat([b, f], 1))
        return b, c


class SmallUpsample(nn.Module):
    """MobilenetSmall-style feature pyramid (hyperpose/Model/backbones.py:301-341): concat(maxpool(early), middle, upsample(late)),
    the late map up-sampled x2 bilinearly (TensorLayer UpSampling2d) in one head and by nearest neighbour in the other."""

    def __init__(self):
        super().__init__()
        self.c0 = conv_bn(3, 16, 3, 2)
        self.c1 = conv_bn(16, 16, 3, 1)
        self.c2 = conv_bn(16, 32, 3, 2)
        self.c3 = conv_bn(32, 64, 3, 2)
        self.pool = nn.MaxPool2d(2, 2)
        self.up_lin = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self.up_nn = nn.Upsample(scale_factor=2, mode="nearest")
        self.head_a = nn.Conv2d(16 + 32 + 64, 8, 1)
        self.head_b = nn.Conv2d(64, 8, 3, 1, 1)

    def forward(self, x):
        e = self.c1(self.c0(x))
        m = self.c2(e)
        l = self.c3(m)
        a = self.head_a(torch.cat([self.pool(e), m, self.up_lin(l)], 1))
        b = self.head_b(self.up_nn(l))
        return a, b


class Unfolded(nn.Module):
    """Exported WITHOUT the exporter's constant folding: the normalisation constants and the PReLU slope reach the graph
    through Sub / Div / Unsqueeze nodes on initializers."""

    def __init__(self):
        super().__init__()
        self.register_buffer("mean", torch.tensor([0.4, 0.5, 0.6]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.25, 0.2, 0.3]).view(1, 3, 1, 1))
        self.c1 = nn.Conv2d(3, 16, 3, 1, 1)
        self.p1 = nn.PReLU(16)
        self.bn = nn.BatchNorm2d(16)
        self.c2 = nn.Conv2d(16, 8, 1)

    def forward(self, x):
        x = (x - self.mean) / self.std
        return self.c2(F.relu(self.bn(self.p1(self.c1(x)))))


def randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
            m.running_mean.data = 0.2 * torch.randn(m.running_mean.shape, generator=g)
            m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=g)
        elif isinstance(m, nn.Conv2d):
            fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5
            if m.bias is not None:
                m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
        elif isinstance(m, nn.PReLU):
            m.weight.data = 0.1 + 0.3 * torch.rand(m.weight.shape, generator=g)
    return model.eval()


CASES = [
    # name, module, (H, W), output names, exporter options
    ("mobile_paf", MobilePaf, (64, 48), ["conf", "paf"], dict(opset_version=11, dynamic_batch=True)),
    ("resnet_ppn", ResNetHead, (64, 64), ["c", "e"], dict(opset_version=13, dynamic_batch=False)),
    # (opset 10: Pad carries its amounts as an attribute; from opset 11 PyTorch computes them with a shape subgraph)
    ("vgg_stages", VggStages, (48, 64), ["stage2", "stage3"], dict(opset_version=10, dynamic_batch=True)),
    ("unfolded", Unfolded, (24, 32), ["out"], dict(opset_version=13, dynamic_batch=False, fold=False)),
    ("small_upsample", SmallUpsample, (48, 64), ["pyramid", "fine"], dict(opset_version=11, dynamic_batch=False)),
]


def main():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    os.makedirs(OUT, exist_ok=True)
    warnings.simplefilter("ignore")
    for seed, (name, cls, (H, W), out_names, opt) in enumerate(CASES):
        model = randomise(cls(), 100 + seed)
        g = torch.Generator().manual_seed(7 + seed)
        x = torch.rand((2, 3, H, W), generator=g)
        with torch.no_grad():
            ys = model(x)
        ys = ys if isinstance(ys, tuple) else (ys,)
        path = os.path.join(OUT, name + ".onnx")
        torch.onnx.export(model, x[:1], path, dynamo=False, opset_version=opt["opset_version"], input_names=["image"],
                          output_names=out_names, do_constant_folding=opt.get("fold", True),
                          dynamic_axes={"image": {0: "batch"}} if opt["dynamic_batch"] else None)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), image=x.numpy(),
                            **{n: y.numpy() for n, y in zip(out_names, ys)})
        print(name, os.path.getsize(path), "bytes;", {n: tuple(y.shape) for n, y in zip(out_names, ys)})


if __name__ == "__main__":
    main()
