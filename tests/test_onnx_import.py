"""CPU: ONNX import (hp_model_from_onnx, the job of nvonnxparser in src/tensorrt.cpp:162-223) — no GPU needed, the importer
is host code.  The lowered layer list is evaluated by the fp32 torch oracle (oracle/ref_net.py) and must reproduce the
outputs PyTorch computed for the very module the file was exported from (tests/golden/onnx/*.npz), to fp32 round-off:
|err| <= 1e-4 * max|ref|."""
import os
import struct

import numpy as np
import pytest

from hyperpose_amd import engine as E
from hyperpose_amd._lib import HpError
from oracle import ref_net

import onnx_writer as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onnx")
CASES = ["mobile_paf", "resnet_ppn", "vgg_stages", "unfolded", "small_upsample"]


def _oracle(m, image):
    return ref_net.run(m.layers, m.outputs, m.weights, frames_f32=image, mean=m.mean, inv_std=m.inv_std, match_fp16=False)


@pytest.mark.parametrize("name", CASES)
def test_pytorch_exports_match_pytorch(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = E.Model.from_onnx(os.path.join(GOLD, name + ".onnx"))
    assert (m.in_h, m.in_w) == z["image"].shape[2:]
    out = _oracle(m, z["image"])
    assert sorted(out) == sorted(k for k in z.files if k != "image")
    for k, v in out.items():
        assert v.shape == z[k].shape
        assert np.abs(v - z[k]).max() <= 1e-4 * np.abs(z[k]).max(), k


def test_lowering_shapes_the_graph_for_the_engine():
    """What the lowering is expected to produce for the LightWeight-OpenPose-like export: BatchNorm gone, activations fused,
    concat members written in place at channel offsets, the residual add fused into the later convolution."""
    m = E.Model.from_onnx(os.path.join(GOLD, "mobile_paf.onnx"))
    ops = [L.op for L in m.layers]
    assert len(m.layers) == 21 and ops.count(E.OP_DWCONV) == 3 and ops.count(E.OP_MAXPOOL) == 0
    concat_reader = [L for L in m.layers if L.cin == 43]
    assert len(concat_reader) == 1
    writers = sorted((L.out_coff, L.cout) for L in m.layers if L.out == concat_reader[0].in_)
    assert writers == [(0, 32), (32, 5), (37, 6)]
    res = [L for L in m.layers if L.res >= 0]
    assert len(res) == 1 and res[0].res_before_act == 0 and res[0].dil == 2
    assert [L.act for L in m.layers if L.op == E.OP_DWCONV] == [E.ACT_RELU, E.ACT_RELU6, E.ACT_RELU]
    # PyTorch's symmetric padding at stride 2 on an even size is NOT TensorFlow's SAME: kept as explicit pads
    s2 = [L for L in m.layers if L.stride == 2]
    assert all(L.pad_explicit and list(L.pad) == [1, 1, 1, 1] for L in s2) and len(s2) == 2
    assert all(not L.pad_explicit for L in m.layers if L.stride == 1)
    # ResNet export: Add -> Relu becomes act(conv + res)
    r = E.Model.from_onnx(os.path.join(GOLD, "resnet_ppn.onnx"))
    assert [L.res_before_act for L in r.layers if L.res >= 0] == [1, 1, 1]
    assert [o.act for o in r.outputs] == [E.ACT_SIGMOID, E.ACT_NONE]
    # VGG export: Pad(0,1,0,1) + VALID stride-2 conv is recognised as SAME; normalisation lands in mean / inv_std
    v = E.Model.from_onnx(os.path.join(GOLD, "vgg_stages.onnx"))
    assert all(not L.pad_explicit for L in v.layers)
    np.testing.assert_allclose(v.mean, [0.485, 0.456, 0.406], rtol=1e-6)
    np.testing.assert_allclose(v.inv_std, [1 / 0.229, 1 / 0.224, 1 / 0.225], rtol=1e-6)


def test_resize_lowering_and_its_limits():
    """Resize by an integer factor becomes HP_OP_UPSAMPLE writing straight into the concatenation; other Resize forms are refused."""
    m = E.Model.from_onnx(os.path.join(GOLD, "small_upsample.onnx"))
    ups = [L for L in m.layers if L.op == E.OP_UPSAMPLE]
    assert [(L.kh, L.stride, L.cin) for L in ups] == [(1, 2, 64), (0, 2, 64)]
    cat = [L for L in m.layers if L.cin == 112][0]
    assert sorted((L.out_coff, L.cout) for L in m.layers if L.out == cat.in_) == [(0, 16), (16, 32), (48, 64)]
    assert not any(L.op == E.OP_CONV and L.kh == 1 and L.cin == L.cout == 64 for L in m.layers)  # no identity copies
    x = W.value_info("x", ["N", 3, 8, 8])
    conv = W.node("Conv", ["x", "w"], ["c"], [W.attr_ints("kernel_shape", [1, 1])])
    w = W.tensor("w", [8, 3, 1, 1], [0.1] * 24)
    for scales, kw, msg in (([1.0, 1.0, 1.5, 1.5], {}, "integer scales"), ([1.0, 1.0, 2.0, 3.0], {}, "integer scales"),
                            ([1.0, 1.0, 2.0, 2.0], {"ctm": "align_corners"}, "align_corners")):
        attrs = [W.attr_str("mode", "linear"), W.attr_str("coordinate_transformation_mode", kw.get("ctm", "half_pixel"))]
        raw = W.model([conv, W.node("Resize", ["c", "", "s"], ["y"], attrs, name="rz")],
                      [w, W.tensor("s", [4], scales)], [x], [W.value_info("y", ["N", 8, "h", "w"])])
        with pytest.raises(HpError, match=msg):
            E.Model.from_onnx(raw)


def test_bytes_and_input_size_rules():
    raw = open(os.path.join(GOLD, "mobile_paf.onnx"), "rb").read()
    m = E.Model.from_onnx(raw)  # static 64 x 48 in the file
    assert (m.in_w, m.in_h) == (48, 64)
    assert (E.Model.from_onnx(raw, 48, 64).in_w, E.Model.from_onnx(raw, 48, 64).in_h) == (48, 64)
    with pytest.raises(HpError, match="fixed to 48x64"):  # src/tensorrt.cpp:190-205: the profile must fit the network
        E.Model.from_onnx(raw, 64, 64)
    for cut in (10, len(raw) // 2, len(raw) - 7):
        with pytest.raises(HpError):
            E.Model.from_onnx(raw[:cut])
    with pytest.raises(HpError):
        E.Model.from_onnx(b"\x00" * 64)
    with pytest.raises(HpError, match="cannot open"):
        E.Model.from_onnx("/nonexistent/model.onnx")


def _hand_model(extra_nodes=(), extra_inputs=(), h=8, w="W", out="y", opset=11, final="r"):
    rng = np.random.default_rng(3)
    wt = rng.normal(0, 0.3, (8, 3, 3, 3)).astype(np.float32)
    bs = rng.normal(0, 0.1, 8).astype(np.float32)
    nodes = [W.node("Conv", ["x", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_str("auto_pad", "SAME_UPPER"),
                                                      W.attr_ints("strides", [2, 2])], name="conv0"),
             W.node("LeakyRelu", ["c"], ["r"], [W.attr_float("alpha", 0.2)])]
    nodes += list(extra_nodes)
    nodes.append(W.node("Identity", [final], [out]))
    init = [W.tensor("w", wt.shape, wt.ravel().tolist()), W.tensor("b", [8], bs.tolist(), raw=True)]
    inputs = [W.value_info("x", ["N", 3, h, w])] + list(extra_inputs)
    return W.model(nodes, init, inputs, [W.value_info(out, ["N", 8, "h", "w"])], opset), wt, bs


def test_hand_written_file_other_encodings():
    """float_data / unpacked repeated fields / symbolic H,W / auto_pad SAME_UPPER: the encodings PyTorch does not emit."""
    import torch
    import torch.nn.functional as F
    raw, wt, bs = _hand_model()
    with pytest.raises(HpError, match="dynamic"):
        E.Model.from_onnx(raw)
    m = E.Model.from_onnx(raw, 10, 8)
    assert len(m.layers) == 1 and m.layers[0].act == E.ACT_LEAKY and abs(m.layers[0].act_param - 0.2) < 1e-7
    assert not m.layers[0].pad_explicit and m.layers[0].stride == 2
    x = np.random.default_rng(0).random((1, 3, 8, 10), dtype=np.float32)
    ref = F.leaky_relu(F.conv2d(F.pad(torch.from_numpy(x), (0, 1, 0, 1)), torch.from_numpy(wt), torch.from_numpy(bs), stride=2), 0.2)
    np.testing.assert_allclose(_oracle(m, x)["y"], ref.numpy(), atol=1e-5)


def test_unsupported_graphs_fail_with_the_node_named():
    raw, _, _ = _hand_model(extra_nodes=[W.node("Softmax", ["r"], ["s"], [W.attr_int("axis", 1)], name="sm")], final="s")
    with pytest.raises(HpError, match=r"'sm' \(Softmax\): operator not supported"):
        E.Model.from_onnx(raw, 10, 8)
    raw, _, _ = _hand_model(extra_inputs=[W.value_info("x2", ["N", 3, 8, 8])])
    with pytest.raises(HpError, match="2 inputs"):  # src/tensorrt.cpp:179-180
        E.Model.from_onnx(raw, 10, 8)
    raw, _, _ = _hand_model(extra_nodes=[W.node("Sigmoid", ["r"], ["s"]), W.node("Relu", ["s"], ["t"], name="after")], final="t")
    with pytest.raises(HpError, match="after.*Sigmoid / Softplus"):
        E.Model.from_onnx(raw, 10, 8)
    # one channel too many for the network input (src/tensorrt.cpp:192-194)
    bad = W.model([W.node("Identity", ["x"], ["y"])], [], [W.value_info("x", ["N", 4, 8, 8])], [W.value_info("y", ["N", 4, 8, 8])])
    with pytest.raises(HpError, match="channel dimension must be 3"):
        E.Model.from_onnx(bad)


@pytest.mark.parametrize("arch,w,h", [("lw_openpose_mobilenet", 64, 48), ("lw_openpose_vggtiny", 48, 64), ("openpose_vgg19", 64, 64)])
def test_reference_topologies_survive_export_and_import(arch, w, h, tmp_path):
    """The reference's PAF topologies (hyperpose/Model/openpose/model/*.py as restated by hp_model_build) written as a
    torch module, exported by PyTorch to ONNX and imported again give the built-in layer list back — same layers, same
    wiring, same weights — so an imported model runs the same fused schedule as the built-in one."""
    import torch_from_layers as T
    m = E.Model(arch, w, h)
    blob = m.init_weights(11)
    path = str(tmp_path / (arch + ".onnx"))
    T.export(m.layers, m.outputs, blob, h, w, path, m.mean, m.inv_std)
    im = E.Model.from_onnx(path, w, h)
    assert T.signature(im.layers) == T.signature(m.layers)
    assert [(o.name, o.coff, o.channels, o.act) for o in im.outputs] == [(o.name, o.coff, o.channels, o.act) for o in m.outputs]
    np.testing.assert_allclose(im.mean, m.mean, atol=1e-6)
    np.testing.assert_allclose(im.inv_std, m.inv_std, rtol=1e-6)
    for a, b in zip(im.layers, m.layers):  # weights: same values in the engine's layout
        if a.op == E.OP_MAXPOOL:
            continue
        n = a.cout * a.kh * a.kw * (a.cin if a.op == E.OP_CONV else 1)
        assert np.array_equal(im.weights[a.w_off:a.w_off + n], blob[b.w_off:b.w_off + n])
        assert np.array_equal(im.weights[a.b_off:a.b_off + a.cout], blob[b.b_off:b.b_off + b.cout])


def test_corrupted_files_never_crash_the_importer():
    """Model files are untrusted input: any corruption must come back as an error code (or a valid model), never as a crash,
    hang or allocation blow-up.  400 single-byte / truncation / splice mutations of two fixtures."""
    rng = np.random.default_rng(2024)
    ok = bad = 0
    for name in ("unfolded", "resnet_ppn"):
        raw = bytearray(open(os.path.join(GOLD, name + ".onnx"), "rb").read())
        head = min(len(raw), 6000)  # structure lives in the first KBs of unfolded; weights dominate resnet_ppn
        for it in range(200):
            m = bytearray(raw)
            kind = it % 4
            if kind == 0:
                m[int(rng.integers(0, head))] = int(rng.integers(0, 256))
            elif kind == 1:
                for _ in range(8):
                    m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                m = m[:int(rng.integers(1, len(m)))]
            else:
                a, b = sorted(int(x) for x in rng.integers(0, head, 2))
                m = m[:a] + m[b:]
            try:
                E.Model.from_onnx(bytes(m), 32, 24) if name == "unfolded" else E.Model.from_onnx(bytes(m))
                ok += 1
            except HpError:
                bad += 1
    assert ok + bad == 400 and bad > 50


def test_hostile_tensor_dims_are_rejected():
    """TensorProto dims are untrusted: a huge / negative dimension with an empty (or short) raw_data must be an error, not 2^62 reads
    past the buffer (the element-count product used to wrap around the size check)."""
    ow = W

    def conv_model(w_dims, payload):
        t = b"".join(ow._int(1, d) for d in w_dims) + ow._int(2, 1) + ow._len(9, payload) + ow._str(8, "w")
        n = ow.node("Conv", ["x", "w"], ["y"], [ow.attr_ints("kernel_shape", [1, 1])])
        return ow.model([n], [t], [ow.value_info("x", [1, 3, 8, 8])], [ow.value_info("y", [1, 4, 8, 8])])

    good = conv_model([4, 3, 1, 1], struct.pack("<12f", *range(12)))
    assert E.Model.from_onnx(good).layers[0].cout == 4
    for dims, payload in (([1 << 62], b""), ([1 << 62, 4], b""), ([1 << 32, 1 << 32], b""), ([-4, 3, 1, 1], struct.pack("<12f", *range(12))),
                          ([1 << 31, 3, 1, 1], b""), ([4, 3, 1, 1], struct.pack("<11f", *range(11))), ([4, 3, 1 << 40, 1], b""),
                          ([0, 3, 1, 1], b""), ([4, 3, 1, 64], struct.pack("<768f", *([0.0] * 768)))):
        with pytest.raises(HpError):
            E.Model.from_onnx(conv_model(dims, payload))


def _post_op_model():
    """conv 3 -> 12 (1x1) -> Sigmoid -> Split(4 | 8) -> {first: Transpose to NHWC and back; second: Reshape [N, 2, 4, H, W]}: the shape of the
    post-processing the PoseProposal / PifPaf exports end in (split heads, 5-D edge / field tensors)."""
    w = [0.01 * (i + 1) for i in range(36)]
    init = [W.tensor("w", [12, 3, 1, 1], w), W.tensor("b", [12], [0.1 * i for i in range(12)]),
            W.tensor("shape", [5], [-1, 2, 4, 6, 8], int64=True), W.tensor("split", [2], [4, 8], int64=True)]
    nodes = [W.node("Conv", ["x", "w", "b"], ["c"], [W.attr_ints("kernel_shape", [1, 1])]),
             W.node("Sigmoid", ["c"], ["s"]),
             W.node("Split", ["s", "split"], ["head_a", "head_b"], [W.attr_int("axis", 1)]),
             W.node("Transpose", ["head_a"], ["a_nhwc"], [W.attr_ints("perm", [0, 2, 3, 1])]),
             W.node("Transpose", ["a_nhwc"], ["out_a"], [W.attr_ints("perm", [0, 3, 1, 2])]),
             W.node("Reshape", ["head_b", "shape"], ["out_b"])]
    return W.model(nodes, init, [W.value_info("x", ["N", 3, 6, 8])], [W.value_info("out_a", ["N", 4, 6, 8]), W.value_info("out_b", ["N", 2, 4, 6, 8])], opset=13)


def test_post_processing_operators_of_the_exported_heads():
    m = E.Model.from_onnx(_post_op_model())
    assert len(m.layers) == 1 and m.layers[0].cout == 12
    outs = {o.name: (o.coff, o.channels, o.act) for o in m.outputs}
    assert outs == {b"out_a": (0, 4, E.ACT_SIGMOID), b"out_b": (4, 8, E.ACT_SIGMOID)}
    # views cannot feed further layers, and a map left in N,H,W,C order cannot be an output
    bad = W.model([W.node("Conv", ["x", "w"], ["c"], [W.attr_ints("kernel_shape", [1, 1])]),
                   W.node("Transpose", ["c"], ["y"], [W.attr_ints("perm", [0, 2, 3, 1])])],
                  [W.tensor("w", [4, 3, 1, 1], [0.1] * 12)], [W.value_info("x", ["N", 3, 6, 8])], [W.value_info("y", ["N", 6, 8, 4])])
    with pytest.raises(HpError, match="N,H,W,C"):
        E.Model.from_onnx(bad)
    bad2 = W.model([W.node("Conv", ["x", "w"], ["c"], [W.attr_ints("kernel_shape", [1, 1])]),
                    W.node("Reshape", ["c", "shape"], ["r"]), W.node("Relu", ["r"], ["y"])],
                   [W.tensor("w", [4, 3, 1, 1], [0.1] * 12), W.tensor("shape", [3], [-1, 4, 48], int64=True)],
                   [W.value_info("x", ["N", 3, 6, 8])], [W.value_info("y", ["N", 4, 48])])
    with pytest.raises(HpError, match="reshaped"):
        E.Model.from_onnx(bad2)
    bad3 = W.model([W.node("Conv", ["x", "w"], ["c"], [W.attr_ints("kernel_shape", [1, 1])]), W.node("Reshape", ["c", "shape"], ["y"])],
                   [W.tensor("w", [4, 3, 1, 1], [0.1] * 12), W.tensor("shape", [3], [-1, 5, 48], int64=True)],
                   [W.value_info("x", ["N", 3, 6, 8])], [W.value_info("y", ["N", 5, 48])])
    with pytest.raises(HpError):
        E.Model.from_onnx(bad3)


# ---- a TensorFlow-export-shaped graph (every released HyperPose model is a TensorFlow export, /root/reference/scripts/downloader.py:12-21):
# the idioms tf2onnx leaves behind that PyTorch's exporter never emits
def tf2onnx_like_model(h=64, w=96, seed=5):
    """N,H,W,3 input -> Transpose -> Conv(auto_pad=SAME_UPPER, stride 2, no bias) -> BatchNormalization (NOT folded) -> Relu
    -> depthwise Conv (group = C) + BatchNormalization + Clip(min, max as INPUTS: relu6) -> 1x1 Conv + BN -> Transpose to N,H,W,C ->
    Relu (in N,H,W,C) -> Transpose back -> 1x1 head Conv -> Sigmoid -> Split into (pc, px, py, pw) -> restore_coor: px = (px + grid_x) * 32,
    py = (py + grid_y) * 32, pw = pw * W_in  (hyperpose/Model/pose_proposal/model.py:111-119).  Returns (bytes, torch reference)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(seed)
    C0, C1, K = 16, 24, 2
    def bn(c):
        return [rng.uniform(0.5, 1.5, c), rng.normal(0, 0.2, c), rng.normal(0, 0.2, c), rng.uniform(0.5, 2.0, c)]
    w0 = rng.normal(0, 0.3, (C0, 3, 3, 3)).astype(np.float32)
    wd = rng.normal(0, 0.4, (C0, 1, 3, 3)).astype(np.float32)
    w1 = rng.normal(0, 0.3, (C1, C0, 1, 1)).astype(np.float32)
    wh = rng.normal(0, 0.3, (4 * K, C1, 1, 1)).astype(np.float32)
    bh = rng.normal(0, 0.2, 4 * K).astype(np.float32)
    bn0, bnd, bn1 = bn(C0), bn(C0), bn(C1)
    gh, gw = -(-(h // 1) // 2), -(-w // 2)  # SAME, stride 2
    gx = np.tile(np.arange(gw, dtype=np.float32), (gh, 1))
    gy = np.tile(np.arange(gh, dtype=np.float32)[:, None], (1, gw))
    init, nodes = [], []
    def const(name, arr, **kw):
        arr = np.asarray(arr, np.float32)
        init.append(W.tensor(name, list(arr.shape), arr.ravel().tolist(), **kw))
    def bn_node(x, y, name, ps, c):
        for k, tag in enumerate(("scale", "B", "mean", "var")):
            const(f"{name}_{tag}", ps[k])
        nodes.append(W.node("BatchNormalization", [x] + [f"{name}_{t}" for t in ("scale", "B", "mean", "var")], [y], [W.attr_float("epsilon", 1e-3)], name=name))
    const("w0", w0), const("wd", wd, raw=True), const("w1", w1), const("wh", wh), const("bh", bh)
    const("lo", np.zeros(())), const("hi", np.full((), 6.0)), const("gx", gx[None, None]), const("gy", gy[None, None])
    const("s32", np.full((), 32.0)), const("sw", np.full((1,), float(w)))
    init.append(W.tensor("split", [4], [K] * 4, int64=True))
    same = W.attr_str("auto_pad", "SAME_UPPER")
    nodes.append(W.node("Transpose", ["image:0"], ["x_nchw"], [W.attr_ints("perm", [0, 3, 1, 2])]))
    nodes.append(W.node("Conv", ["x_nchw", "w0"], ["c0"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_ints("strides", [2, 2]), same], name="conv0"))
    bn_node("c0", "b0", "bn0", bn0, C0)
    nodes.append(W.node("Relu", ["b0"], ["r0"]))
    nodes.append(W.node("Conv", ["r0", "wd"], ["cd"], [W.attr_ints("kernel_shape", [3, 3]), W.attr_int("group", C0), same], name="dw"))
    bn_node("cd", "bd", "bnd", bnd, C0)
    nodes.append(W.node("Clip", ["bd", "lo", "hi"], ["rd"]))
    nodes.append(W.node("Conv", ["rd", "w1"], ["c1"], [W.attr_ints("kernel_shape", [1, 1])], name="pw"))
    bn_node("c1", "b1", "bn1", bn1, C1)
    nodes.append(W.node("Transpose", ["b1"], ["b1_nhwc"], [W.attr_ints("perm", [0, 2, 3, 1])]))
    nodes.append(W.node("Relu", ["b1_nhwc"], ["r1_nhwc"]))
    nodes.append(W.node("Transpose", ["r1_nhwc"], ["r1"], [W.attr_ints("perm", [0, 3, 1, 2])]))
    nodes.append(W.node("Conv", ["r1", "wh", "bh"], ["head"], [W.attr_ints("kernel_shape", [1, 1])], name="head"))
    nodes.append(W.node("Sigmoid", ["head"], ["sg"]))
    nodes.append(W.node("Split", ["sg", "split"], ["pc", "sx", "sy", "sw_"], [W.attr_int("axis", 1)]))
    nodes.append(W.node("Add", ["sx", "gx"], ["ax"]))
    nodes.append(W.node("Mul", ["ax", "s32"], ["px"]))
    nodes.append(W.node("Add", ["gy", "sy"], ["ay"]))      # constant first
    nodes.append(W.node("Mul", ["s32", "ay"], ["py"]))
    nodes.append(W.node("Mul", ["sw_", "sw"], ["pw"]))
    outs = [W.value_info(nm, ["N", K, gh, gw]) for nm in ("pc", "px", "py", "pw")]
    raw = W.model(nodes, init, [W.value_info("image:0", ["N", h, w, 3])], outs, opset=13)

    def ref(x_nhwc):
        t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
        def bnf(x, ps):
            return F.batch_norm(x, t(ps[2]), t(ps[3]), t(ps[0]), t(ps[1]), False, 0.0, 1e-3)
        x = t(x_nhwc).permute(0, 3, 1, 2)
        x = F.relu(bnf(F.conv2d(F.pad(x, (0, 1, 0, 1)), t(w0), None, stride=2), bn0))
        x = torch.clamp(bnf(F.conv2d(F.pad(x, (1, 1, 1, 1)), t(wd), None, groups=C0), bnd), 0, 6)
        x = F.relu(bnf(F.conv2d(x, t(w1)), bn1))
        sg = torch.sigmoid(F.conv2d(x, t(wh), t(bh)))
        pc, sx, sy, sw_ = torch.split(sg, K, 1)
        return {"pc": pc.numpy(), "px": ((sx + t(gx)) * 32).numpy(), "py": ((sy + t(gy)) * 32).numpy(), "pw": (sw_ * float(w)).numpy()}
    return raw, ref


def test_tensorflow_export_idioms_are_lowered():
    raw, ref = tf2onnx_like_model()
    m = E.Model.from_onnx(raw)
    assert (m.in_w, m.in_h) == (96, 64)
    ops = [(L.op, L.cin, L.cout, L.kh, L.stride, L.act) for L in m.layers]
    # BatchNorm folded away, Clip(0, 6) and the Relu inside the Transpose sandwich fused, no copy layers
    assert ops == [(E.OP_CONV, 3, 16, 3, 2, E.ACT_RELU), (E.OP_DWCONV, 16, 16, 3, 1, E.ACT_RELU6), (E.OP_CONV, 16, 24, 1, 1, E.ACT_RELU),
                   (E.OP_CONV, 24, 8, 1, 1, E.ACT_NONE)], ops
    assert not any(L.pad_explicit for L in m.layers)   # auto_pad SAME_UPPER == the engine's TF "SAME"
    outs = {o.name: (o.coff, o.channels, o.act, o.grid, round(o.scale, 3)) for o in m.outputs}
    assert outs == {b"pc": (0, 2, E.ACT_SIGMOID, 0, 0.0), b"px": (2, 2, E.ACT_SIGMOID, 1, 32.0), b"py": (4, 2, E.ACT_SIGMOID, 2, 32.0),
                    b"pw": (6, 2, E.ACT_SIGMOID, 0, 96.0)}, outs
    x = np.random.default_rng(0).random((2, 64, 96, 3), dtype=np.float32)
    want = ref(x)
    got = ref_net.run(m.layers, m.outputs, m.weights, frames_f32=np.ascontiguousarray(x.transpose(0, 3, 1, 2)), match_fp16=False,
                      mean=m.mean, inv_std=m.inv_std)
    for k, v in want.items():
        np.testing.assert_allclose(got[k], v, rtol=1e-4, atol=1e-4)


def test_arithmetic_after_sigmoid_that_cannot_be_evaluated_is_refused():
    """Only `(sigmoid + cell-index grid) * scalar` is an output post-op (hp_output_desc::grid / scale): anything else after a Sigmoid names its
    node instead of being dropped or mis-evaluated."""
    def model(extra_init, nodes, out):
        init = [W.tensor("w", [4, 3, 1, 1], [0.1] * 12)] + extra_init
        base = [W.node("Conv", ["x", "w"], ["c"], [W.attr_ints("kernel_shape", [1, 1])]), W.node("Sigmoid", ["c"], ["s"])]
        return W.model(base + nodes, init, [W.value_info("x", ["N", 3, 6, 8])], [W.value_info(out, ["N", 4, 6, 8])], opset=13)
    not_grid = [float(i % 5) for i in range(48)]
    with pytest.raises(HpError, match=r"a constant map added after Sigmoid"):
        E.Model.from_onnx(model([W.tensor("g", [1, 1, 6, 8], not_grid)], [W.node("Add", ["s", "g"], ["y"], name="addg")], "y"))
    with pytest.raises(HpError, match=r"after Sigmoid / Softplus only"):   # per-channel factor
        E.Model.from_onnx(model([W.tensor("v", [1, 4, 1, 1], [1.0, 2.0, 3.0, 4.0])], [W.node("Mul", ["s", "v"], ["y"], name="mulv")], "y"))
    with pytest.raises(HpError, match=r"after Sigmoid / Softplus only"):   # the grid AFTER the factor: (s * 32) + grid is not (s + grid) * 32
        gx = [float(i % 8) for i in range(48)]
        E.Model.from_onnx(model([W.tensor("k", [], [32.0]), W.tensor("g", [1, 1, 6, 8], gx)],
                                [W.node("Mul", ["s", "k"], ["m"]), W.node("Add", ["m", "g"], ["y"], name="late")], "y"))
    with pytest.raises(HpError, match=r"Sigmoid / Softplus"):  # a convolution reading the post-processed map
        E.Model.from_onnx(model([W.tensor("k", [], [32.0]), W.tensor("w2", [4, 4, 1, 1], [0.1] * 16)],
                                [W.node("Mul", ["s", "k"], ["m"]), W.node("Conv", ["m", "w2"], ["y"], [W.attr_ints("kernel_shape", [1, 1])], name="conv2")], "y"))
    # the accepted form, column grid then row grid, and Div as the inverse factor
    gx = [float(i % 8) for i in range(48)]
    gy = [float(i // 8) for i in range(48)]
    ok = model([W.tensor("g", [6, 8], gy), W.tensor("k", [1], [4.0])], [W.node("Add", ["s", "g"], ["a"]), W.node("Div", ["a", "k"], ["y"])], "y")
    m = E.Model.from_onnx(ok)
    assert [(o.grid, round(o.scale, 4)) for o in m.outputs] == [(2, 0.25)]
    x = np.random.default_rng(0).random((1, 3, 6, 8), dtype=np.float32)
    want = (1 / (1 + np.exp(-(0.1 * x.sum(1, keepdims=True)))) + np.asarray(gy, np.float32).reshape(1, 1, 6, 8)) / 4.0
    np.testing.assert_allclose(_oracle(m, x)["y"], np.repeat(want, 4, 1), rtol=1e-5, atol=1e-6)
    del gx
