"""GPU: the HIP conv stack (libhp_hip.so through the C ABI) vs the plain PyTorch fp32 oracle (oracle/ref_net.py).

Tolerance: activations/weights are fp16 in HBM with fp32 MFMA accumulation.  Against the oracle evaluated with
the SAME fp16 storage points (match_fp16=True) the only differences are fp32 summation order and rare fp16
rounding flips: |err| <= 2e-3 * max|ref| + 1e-3.  Against the pure fp32 oracle the bound is the fp16 one (1e-2).
"""
import numpy as np
import pytest

from hyperpose_amd import engine as E
from oracle import ref_net

pytestmark = pytest.mark.gpu


class Out:
    def __init__(self, name, tensor, coff, channels, act=0, **kw):
        self.name, self.tensor, self.coff, self.channels, self.act = name.encode(), tensor, coff, channels, act
        self.shuffle, self.group, self.sigmoid_mask, self.softplus_mask = 0, 0, 0, 0
        self.out_h, self.out_w, self.scale, self.grid = 0, 0, 0.0, 0
        for k, v in kw.items():
            setattr(self, k, v)

    def c(self):
        o = E.OutputDesc()
        for f, _ in E.OutputDesc._fields_:
            setattr(o, f, getattr(self, f))
        return o


class Net:
    """Tiny graph builder for kernel-level tests (weights appended to one blob, He-scaled)."""

    def __init__(self, seed=0):
        self.layers, self.w, self.rng, self.nt = [], [], np.random.default_rng(seed), 1

    def _alloc(self, n, std):
        off = sum(len(x) for x in self.w)
        self.w.append((self.rng.normal(0, std, n)).astype(np.float32))
        return off

    def conv(self, in_, cin, cout, k=1, stride=1, dil=1, act=E.ACT_RELU, out=None, out_coff=0, in_coff=0, res=-1,
             res_before_act=0, op=E.OP_CONV, act_param=0.0):
        if out is None:
            out = self.nt
            self.nt += 1
        if op == E.OP_CONV:
            w_off = self._alloc(cout * k * k * cin, np.sqrt(2.0 / (k * k * cin)))
        elif op == E.OP_DWCONV:
            w_off = self._alloc(cin * k * k, np.sqrt(2.0 / (k * k)))
        else:
            w_off = -1
        b_off = self._alloc(cout, 0.1) if op != E.OP_MAXPOOL else -1
        a_off = -1
        if act == E.ACT_PRELU:
            a_off = self._alloc(cout, 0.0)
            self.w[-1][:] = self.rng.uniform(0.1, 0.4, cout)
        self.layers.append(E.make_layer(op, in_, out, cin, cout, k, stride, dil, act, in_coff, out_coff, res,
                                        res_before_act, w_off, b_off, a_off, act_param))
        return out

    def new_tensor(self):
        t = self.nt
        self.nt += 1
        return t

    def blob(self):
        return np.concatenate(self.w) if self.w else np.zeros(1, np.float32)


def _run_both(net, outs, frames, h, w, max_batch=None, f32=False, **kw):
    blob = net.blob()
    eng = E.Engine(net.layers, [o.c() for o in outs], blob, w, h, max_batch or len(frames), **kw)
    if f32:
        got = eng.inference_f32(frames)
        ref = ref_net.run(net.layers, outs, blob, frames_f32=frames, match_fp16=True)
    else:
        got = eng.inference(frames)
        ref = ref_net.run(net.layers, outs, blob, frames_u8=frames, match_fp16=True,
                          factor=kw.get("factor", 1 / 255), flip_rb=kw.get("flip_rgb", True),
                          mean=kw.get("mean", (0, 0, 0)), inv_std=kw.get("inv_std", (1, 1, 1)))
    return eng, got, ref


def _close(got, ref, rel=2e-3, abs_=1e-3):
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= rel * scale + abs_, f"max err {err:.4g} vs scale {scale:.4g}"


def _check(got, ref, n, **tol):
    names = sorted(ref)
    for b in range(n):
        assert [nm for nm, _ in got[b]] == names  # ordered by tensor name, src/tensorrt.cpp:405
        for nm, arr in got[b]:
            _close(arr, ref[nm][b], **tol)


def _frames(n, h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def test_first_conv_u8_and_f32(hp):
    for stride, k, cout in ((2, 3, 32), (1, 3, 64), (2, 7, 64)):
        net = Net(1)
        t = net.conv(0, 3, cout, k, stride)
        fr = _frames(2, 37, 45)
        _, got, ref = _run_both(net, [Out("y", t, 0, cout)], fr, 37, 45, flip_rgb=True, mean=(0.4, 0.45, 0.5), inv_std=(2., 3., 4.))
        _check(got, ref, 2)
    net = Net(2)
    t = net.conv(0, 3, 32, 3, 2)
    x = np.random.default_rng(3).normal(size=(2, 3, 20, 28)).astype(np.float32)
    _, got, ref = _run_both(net, [Out("y", t, 0, 32)], x, 20, 28, f32=True)
    _check(got, ref, 2)


@pytest.mark.parametrize("stride,cout,act,h,w,f32,k", [
    (2, 32, E.ACT_RELU, 64, 96, False, 3),   # whole 8 x 32 tiles
    (2, 16, E.ACT_RELU6, 33, 47, False, 3),  # ragged tiles, half a row tile of output channels
    (1, 24, E.ACT_NONE, 19, 70, False, 3),   # stride 1, 24 channels: three 8-channel groups
    (1, 64, E.ACT_LEAKY, 21, 40, False, 3),  # two row tiles, the general (non-clamp) activation path
    (2, 40, E.ACT_RELU, 30, 34, True, 3),    # f32 NCHW input, second row tile partly filled
    (2, 64, E.ACT_RELU, 97, 129, False, 7),  # the ResNet-50 stem (7x7 stride 2), odd sizes
    (2, 24, E.ACT_LEAKY, 33, 40, False, 7),  # ... one row tile partly filled, general activation
    (1, 48, E.ACT_RELU6, 20, 37, True, 7),   # ... stride 1, f32 input
])
def test_first_conv_matrix_pipe_shapes(hp, stride, cout, act, h, w, f32, k):
    """first_conv_f16_kernel (fp16 matrix pipe, normalised input and weights rounded to fp16 like every other layer's operands) vs the
    fp16-matched oracle AND vs the all-fp32 oracle (match_fp16 = False): the fp16 form differs from fp32 arithmetic by the operand
    rounding only (2^-11 relative per operand)."""
    net = Net(7)
    t = net.conv(0, 3, cout, k, stride, act=act, act_param=0.1)
    z = net.conv(t, cout, 8, 1, act=E.ACT_NONE)
    outs = [Out("y", t, 0, cout), Out("z", z, 0, 8)]
    fr = np.random.default_rng(9).normal(size=(2, 3, h, w)).astype(np.float32) if f32 else _frames(2, h, w, seed=5)
    kw = {} if f32 else dict(mean=(0.485, 0.456, 0.406), inv_std=(4.0, 4.5, 4.4))
    eng, got, ref = _run_both(net, outs, fr, h, w, f32=f32, **kw)
    _check(got, ref, 2)
    ref32 = ref_net.run(net.layers, outs, net.blob(), match_fp16=False, **(dict(frames_f32=fr) if f32 else dict(frames_u8=fr, **kw)))
    for b in range(2):
        for n0, a0 in got[b]:
            scale = float(np.abs(ref32[n0][b]).max()) + 1e-6
            assert np.abs(a0 - ref32[n0][b]).max() <= 1e-2 * scale + 1e-3, (n0, "fp16 pipe vs fp32 arithmetic")


@pytest.mark.parametrize("cin,cout,k,stride,dil", [
    (32, 64, 1, 1, 1), (64, 128, 1, 1, 1), (128, 128, 3, 1, 1), (128, 512, 1, 1, 1), (512, 19, 1, 1, 1),
    (512, 38, 1, 1, 1), (64, 64, 3, 2, 1), (96, 128, 3, 1, 2), (128, 128, 7, 1, 1), (256, 200, 3, 1, 1),
    (64, 256, 1, 2, 1), (64, 64, 3, 1, 1), (128, 64, 3, 1, 1),
    (256, 512, 1, 2, 1), (512, 256, 1, 1, 1), (256, 1024, 1, 1, 1), (512, 128, 1, 3, 1),   # conv1x1_big_kernel: strided (odd map), 128- / 256-row blocks
])
def test_mfma_conv_shapes(hp, cin, cout, k, stride, dil):
    net = Net(cin * 7 + cout)
    t0 = net.conv(0, 3, cin, 3, 1)
    t = net.conv(t0, cin, cout, k, stride, dil, act=E.ACT_RELU)
    fr = _frames(3, 23, 29, seed=cout)
    _, got, ref = _run_both(net, [Out("y", t, 0, cout)], fr, 23, 29)
    _check(got, ref, 3)


@pytest.mark.parametrize("cout,with_res,h,w,n", [(1024, True, 64, 64, 4), (1024, False, 49, 49, 7), (512, True, 64, 67, 8), (256, True, 96, 100, 7)])
def test_pixel_block_gemm_through_the_fast_epilogue(hp, cout, with_res, h, w, n):
    """conv1x1_big_kernel on ResNet-sized expansions (256 input channels), with and without the shortcut, ragged last tile, 1 / 2 / 4
    channel groups.  (`scale = 2` keeps the output from being a plain one: the convolution then writes its fp16 tensor through the fast
    epilogue - a plain network output takes the generic kernel with the fused fp32 copy, which is what test_mfma_conv_shapes reaches.)"""
    net = Net(cout + h)
    t0 = net.conv(0, 3, 256, 3, 1)
    r = net.conv(0, 3, cout, 1, 1) if with_res else -1
    t = net.conv(t0, 256, cout, 1, 1, act=E.ACT_RELU, res=r, res_before_act=1 if with_res else 0)
    fr = _frames(n, h, w, seed=cout + w)
    eng, got, ref = _run_both(net, [Out("y", t, 0, cout, scale=2.0)], fr, h, w)
    _check(got, ref, n)
    assert any(5200000 <= q["tile"] < 5300000 for q in eng.profile(n, 1))
    alone = eng.inference(fr[n - 1:n])[0][0][1]
    assert np.array_equal(alone, got[n - 1][0][1])   # batch invariance, bit for bit


def test_mfma_conv_is_not_transposed(hp):
    """Asymmetric, structured weights (not random): catches row/col or channel-order mix-ups exactly."""
    net = Net(0)
    t0 = net.conv(0, 3, 32, 1, 1, act=E.ACT_NONE)
    t = net.conv(t0, 32, 64, 3, 1, act=E.ACT_NONE)
    w = net.blob()
    L0, L1 = net.layers
    w[:] = 0
    for c in range(32):  # t0[c] = B channel (u8 index 0 after flip: R<-B) * (c+1)/32
        w[L0.w_off + c * 3 + (c % 3)] = (c + 1) / 32.0
    for co in range(64):  # y[co] = tap(ky=co%3, kx=(co//3)%3) of channel co%32 times (1 + co/64)
        w[L1.w_off + ((co * 3 + co % 3) * 3 + (co // 3) % 3) * 32 + co % 32] = 1.0 + co / 64.0
    net.w = [w]
    fr = _frames(2, 17, 21, seed=5)
    _, got, ref = _run_both(net, [Out("y", t, 0, 64)], fr, 17, 21)
    _check(got, ref, 2, rel=1e-3, abs_=1e-4)


def test_depthwise_pool_residual_concat_prelu(hp):
    net = Net(4)
    a = net.conv(0, 3, 32, 3, 2)
    d = net.conv(a, 32, 32, 3, 1, op=E.OP_DWCONV)
    d2 = net.conv(d, 32, 32, 3, 2, op=E.OP_DWCONV)
    d3 = net.conv(d2, 32, 32, 3, 1, dil=2, op=E.OP_DWCONV, act=E.ACT_RELU6)
    p = net.conv(d3, 32, 32, 2, 2, op=E.OP_MAXPOOL, act=E.ACT_NONE)
    p3 = net.conv(d3, 32, 32, 3, 2, op=E.OP_MAXPOOL, act=E.ACT_NONE)
    cat = net.new_tensor()
    x = net.conv(p, 32, 128, 1, out=cat, out_coff=0)
    net.conv(p3, 32, 19, 1, act=E.ACT_NONE, out=cat, out_coff=128)
    net.conv(p3, 32, 38, 3, act=E.ACT_LEAKY, out=cat, out_coff=147, act_param=0.1)
    u = net.conv(cat, 185, 128, 1)
    v = net.conv(u, 128, 128, 3, act=E.ACT_PRELU)
    r = net.conv(v, 128, 128, 3, res=u, res_before_act=0)
    r2 = net.conv(r, 128, 128, 1, res=u, res_before_act=1)
    s = net.conv(cat, 128, 64, 3, in_coff=0, act=E.ACT_NONE)  # sigmoid/softplus are output post-ops
    fr = _frames(2, 64, 80, seed=9)
    outs = [Out("a_r2", r2, 0, 128), Out("b_cat", cat, 0, 185), Out("c_sig", s, 0, 64, act=E.ACT_SIGMOID), Out("d_slice", cat, 128, 57),
            Out("e_softplus", s, 8, 40, act=E.ACT_SOFTPLUS)]
    _, got, ref = _run_both(net, outs, fr, 64, 80)
    _check(got, ref, 2)


@pytest.mark.parametrize("kind,scale,c,h,w", [(0, 2, 32, 11, 9), (1, 2, 32, 11, 9), (1, 3, 16, 7, 10), (0, 4, 8, 5, 6), (1, 2, 72, 23, 27)])
def test_upsample(hp, kind, scale, c, h, w):
    """HP_OP_UPSAMPLE (nearest / bilinear with half-pixel centres) vs torch.nn.functional.interpolate, also into a channel offset."""
    net = Net(seed=scale * 10 + kind)
    t = net.conv(0, 3, c, 3, act=E.ACT_RELU)
    cat = net.new_tensor()
    up = E.make_layer(E.OP_UPSAMPLE, t, cat, c, c, 1, scale, 1, E.ACT_NONE, out_coff=8)
    up.kh = kind
    net.layers.append(up)
    _, got, ref = _run_both(net, [Out("up", cat, 8, c)], _frames(2, h, w, seed=3), h, w)
    assert got[0][0][1].shape == (c, h * scale, w * scale)
    _check(got, ref, 2)


def test_lw_openpose_small_end_to_end(hp):
    m = E.Model("lw_openpose_mobilenet", 96, 80)
    w = m.init_weights(3)
    eng = E.Engine.from_model(m, w, max_batch=4)
    fr = _frames(3, 80, 96, seed=1)
    got = eng.inference(fr)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True)
    assert got[0][0][0] == "conf" and got[0][1][0] == "paf"
    _check(got, ref, 3, rel=5e-3, abs_=2e-3)
    ref32 = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=False)
    _check(got, ref32, 3, rel=3e-2, abs_=5e-3)
    # replay from the captured graph and without it give identical bits
    again = eng.inference(fr)
    eng.set_graph(False)
    plain = eng.inference(fr)
    for b in range(3):
        for i in range(2):
            assert np.array_equal(got[b][i][1], again[b][i][1]) and np.array_equal(got[b][i][1], plain[b][i][1])


def test_lw_openpose_full_size_config1(hp):
    """BASELINE config 1 geometry: 368x432, conf [19,46,54] + paf [38,46,54]."""
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    w = m.init_weights(20241)
    eng = E.Engine.from_model(m, w, max_batch=8)
    fr = _frames(2, 368, 432, seed=2)
    got = eng.inference(fr)
    assert got[0][0][1].shape == (19, 46, 54) and got[0][1][1].shape == (38, 46, 54)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True)
    _check(got, ref, 2, rel=5e-3, abs_=2e-3)


def test_vggtiny_and_batch_overflow(hp):
    m = E.Model("lw_openpose_vggtiny", 64, 48)
    w = m.init_weights(5)
    eng = E.Engine.from_model(m, w, max_batch=2)
    fr = _frames(2, 48, 64, seed=4)
    got = eng.inference(fr)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True)
    _check(got, ref, 2, rel=5e-3, abs_=2e-3)
    with pytest.raises(ValueError):  # reference: std::logic_error, src/tensorrt.cpp:439-443
        eng.inference(_frames(3, 48, 64))
    import ctypes as C
    rc = hp.lib().hp_engine_infer_u8(eng._h, fr.ctypes.data_as(C.POINTER(C.c_uint8)), 3, 0, None)
    assert rc == hp.HP_ERR_CAPACITY


def test_openpose_vgg19_small(hp):
    m = E.Model("openpose_vgg19", 64, 48)
    w = m.init_weights(6)
    eng = E.Engine.from_model(m, w, max_batch=1)
    fr = _frames(1, 48, 64, seed=6)
    got = eng.inference(fr)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True, mean=m.mean, inv_std=m.inv_std)
    _check(got, ref, 1, rel=1e-2, abs_=2e-3)


def test_output_transforms(hp):
    """pixel shuffle x2 + crop + per-component sigmoid/softplus (PifPaf heads) and sigmoid + grid affine (PPN)."""
    net = Net(11)
    a = net.conv(0, 3, 32, 3, 2)
    t = net.conv(a, 32, 40, 1, act=E.ACT_NONE)  # 40 = 2 groups x 5 comps x 4 sub-pixels
    fr = _frames(2, 22, 30, seed=3)
    outs = [Out("a_shuf", t, 0, 40, shuffle=2, group=5, sigmoid_mask=1, softplus_mask=1 << 4, out_h=21, out_w=29),
            Out("b_gridx", t, 0, 16, act=E.ACT_SIGMOID, scale=32.0, grid=1),
            Out("c_gridy", t, 16, 8, act=E.ACT_SIGMOID, scale=8.0, grid=2),
            Out("d_scaled", t, 24, 16, act=E.ACT_SIGMOID, scale=384.0)]
    eng, got, ref = _run_both(net, outs, fr, 22, 30)
    assert [s for _, s, _ in eng.outputs] == [(10, 21, 29), (16, 11, 15), (8, 11, 15), (16, 11, 15)]
    _check(got, ref, 2, rel=2e-3, abs_=2e-3)


def test_pose_proposal_resnet50_end_to_end(hp):
    """BASELINE config 3 topology at reduced size: engine outputs (device-resident) -> PPN parser, against the torch
    oracle for the conv stack and the reference's own parser (oracle/_ref) on the same feature maps."""
    from hyperpose_amd.parser import PoseProposal
    from oracle import loader
    m = E.Model("pose_proposal_resnet50", 160, 128)
    w = m.init_weights(12)
    eng = E.Engine.from_model(m, w, max_batch=2)
    fr = _frames(2, 128, 160, seed=8)
    got = eng.inference(fr)
    assert [n for n, _ in got[0]] == ["0_conf_point", "1_conf_iou", "2_x", "3_y", "4_w", "5_h", "6_edge"]
    assert got[0][0][1].shape == (18, 4, 5) and got[0][6][1].shape == (17 * 81, 4, 5)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True)
    _check(got, ref, 2, rel=2e-2, abs_=5e-3)
    parser = PoseProposal((160, 128), max_batch=2)
    shapes = [s for _, s, _ in eng.outputs]
    humans = parser.process_batch([p for _, _, p in eng.outputs], on_device=True, n=2, conf_shape=shapes[0],
                                  edge_shape=(17, 9, 9) + shapes[0][1:])
    if loader.ref_lib() is not None:
        for b in range(2):
            t = [got[b][i][1] for i in range(6)] + [got[b][6][1].reshape(17, 9, 9, 4, 5)]
            refh = loader.ref_ppn_process(t, 160, 128)
            assert humans[b].tobytes() == refh.tobytes()


def test_pifpaf_resnet50_end_to_end(hp):
    """BASELINE config 4 topology at reduced size (97x97 -> 13x13 fields)."""
    from hyperpose_amd.parser import PifPaf
    from oracle import loader
    m = E.Model("pifpaf_resnet50", 97, 97)
    w = m.init_weights(13)
    eng = E.Engine.from_model(m, w, max_batch=2)
    fr = _frames(2, 97, 97, seed=9)
    got = eng.inference(fr)
    assert [n for n, _ in got[0]] == ["0_paf", "1_pif"]
    assert got[0][0][1].shape == (19 * 9, 13, 13) and got[0][1][1].shape == (17 * 5, 13, 13)
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True, mean=m.mean, inv_std=m.inv_std)
    _check(got, ref, 2, rel=2e-2, abs_=5e-3)
    parser = PifPaf(97, 97, max_batch=2)
    humans = parser.process_batch(eng.outputs[0][2], eng.outputs[1][2], on_device=True, n=2, fh=13, fw=13)
    if loader.ref_lib() is not None:
        for b in range(2):
            refh = loader.ref_pifpaf_process(got[b][0][1].reshape(19, 9, 13, 13), got[b][1][1].reshape(17, 5, 13, 13), 97, 97)
            assert humans[b].tobytes() == refh.tobytes()


@pytest.mark.parametrize("c,cout,stride,dil,h,w", [
    (32, 128, 1, 1, 40, 56),     # 32 -> 128: no fused instance (the 64-channel K chunks do not divide C): depthwise + 1x1 launches
    (64, 128, 2, 1, 45, 61),     # variant 2: stride 2 (odd input: SAME pads 1/1)
    (128, 256, 2, 1, 46, 60),    # variant 3: stride 2 (even input: SAME pads 0/1)
    (256, 256, 1, 1, 23, 27),    # variant 4
    (256, 512, 1, 1, 23, 27),    # variant 5
    (512, 512, 1, 2, 19, 21),    # variant 6: dilation 2
    (96, 72, 1, 1, 21, 30),      # ragged channels (C = 96, Cout = 72): no fused instance either, the generic kernels take it
    (128, 128, 1, 1, 29, 35),    # variant 1 in its half-CU form (C a multiple of 64), ragged edge tiles
    (64, 128, 2, 1, 46, 54),     # variant 2, even input (SAME pads 0/1)
    (32, 64, 1, 1, 37, 45),      # variant 7: the 32-channel block (sepconv_small_kernel), odd sizes
    (32, 40, 1, 1, 16, 24),      # variant 7 with fewer than 64 outputs
])
def test_fused_separable_block(hp, monkeypatch, c, cout, stride, dil, h, w):
    """depthwise 3x3 + pointwise 1x1 as one launch (sepconv_slot_kernel / sepconv_small_kernel): against the oracle, and bit-for-bit
    against the two-launch schedule (same fp32 depthwise arithmetic, same fp16 rounding point, same MFMA order)."""
    net = Net(c + cout)
    a = net.conv(0, 3, c, 3, 1)
    d = net.conv(a, c, c, 3, stride, dil, op=E.OP_DWCONV, act=E.ACT_RELU6)
    y = net.conv(d, c, cout, 1, act=E.ACT_RELU)
    z = net.conv(y, cout, 32, 1, act=E.ACT_NONE)  # the fused block must not be a network output
    fr = _frames(3, h, w, seed=c)
    outs = [Out("z", z, 0, 32)]
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3)
    tiles = [p["tile"] for p in eng.profile(3, 1)]
    fused = c % 64 == 0 or (c == 32 and cout <= 64)
    assert any(4000000 <= t < 5000000 for t in tiles) == fused, tiles  # the fused kernel really ran (where an instance exists)
    mid = eng.debug_tensor(y, 3)
    monkeypatch.setenv("HP_NO_FUSE", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    got2 = eng2.inference(fr)
    assert not any(4000000 <= p["tile"] < 5000000 for p in eng2.profile(3, 1))
    assert np.array_equal(mid, eng2.debug_tensor(y, 3))
    for b in range(3):
        assert np.array_equal(got[b][0][1], got2[b][0][1])
    if fused:
        with pytest.raises(Exception):
            eng.debug_tensor(d, 3)  # never materialised


@pytest.mark.parametrize("c,dil,act,h,w", [
    (512, 1, E.ACT_LEAKY, 23, 27),   # general activation: the epilogue runs after the last interval (sepconv_pipe3_kernel<.., TAIL = false>)
    (256, 1, E.ACT_PRELU, 13, 20),
    (512, 2, E.ACT_LEAKY, 19, 21),
    (512, 1, E.ACT_RELU6, 25, 17),   # relu6 clamp inside the last interval (TAIL = true), ragged tiles both ways
    (384, 1, E.ACT_RELU, 12, 8),     # six 64-channel chunks, exactly one tile
])
def test_512_output_separable_block_tails(hp, monkeypatch, c, dil, act, h, w):
    """sepconv_pipe3_kernel's two tails - the epilogue inside the last interval's MFMA stream (relu / relu6) and the general one behind
    it - against the oracle and bit-for-bit against depthwise + pointwise as two launches."""
    net = Net(c + dil + act)
    a = net.conv(0, 3, c, 3, 1)
    d = net.conv(a, c, c, 3, 1, dil, op=E.OP_DWCONV, act=E.ACT_RELU6)
    y = net.conv(d, c, 512, 1, act=act, act_param=0.1)
    z = net.conv(y, 512, 32, 1, act=E.ACT_NONE)
    fr = _frames(2, h, w, seed=c + h)
    outs = [Out("z", z, 0, 32)]
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 2)
    assert any(t in (4000005, 4000006) for t in [p["tile"] for p in eng.profile(2, 1)])
    mid = eng.debug_tensor(y, 2)
    monkeypatch.setenv("HP_NO_FUSE", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 2)
    got2 = eng2.inference(fr)
    assert np.array_equal(mid, eng2.debug_tensor(y, 2))
    for b in range(2):
        assert np.array_equal(got[b][0][1], got2[b][0][1])


@pytest.mark.parametrize("h,w,stem_stride,act", [
    (46, 54, 1, E.ACT_RELU6),   # even map: SAME pads 0 / 1 on the stride-2 block; 23 x 27 outputs = ragged 4 x 8 tiles both ways
    (37, 45, 1, E.ACT_RELU),    # odd map: pads 1 / 1; 19 x 23 outputs
    (64, 96, 2, E.ACT_RELU6),   # behind a stride-2 stem, as in the network: 32 x 48 map, whole tiles
    (9, 7, 1, E.ACT_RELU),      # smaller than one tile: everything is padding or ragged
])
def test_separable_pair_in_one_launch(hp, monkeypatch, h, w, stem_stride, act):
    """sepconv_pair_kernel (32 -> 64 stride 1 and 64 -> 128 stride 2 in one launch, the tensor between them in LDS only): against the
    oracle and bit-for-bit against one launch per block (HP_NO_SEPPAIR=1) - pixels of the tensor in between that lie outside the image
    must act as the second block's zero padding, ragged tiles must not write outside the map."""
    net = Net(17)
    a = net.conv(0, 3, 32, 3, stem_stride, act=act)
    d1 = net.conv(a, 32, 32, 3, 1, op=E.OP_DWCONV, act=act)
    p1 = net.conv(d1, 32, 64, 1, act=act)
    d2 = net.conv(p1, 64, 64, 3, 2, op=E.OP_DWCONV, act=act)
    p2 = net.conv(d2, 64, 128, 1, act=act)
    z = net.conv(p2, 128, 32, 1, act=E.ACT_NONE)  # the fused blocks must not be a network output
    fr = _frames(3, h, w, seed=h + w)
    outs = [Out("z", z, 0, 32)]
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3)
    tiles = [p["tile"] for p in eng.profile(3, 1)]
    assert tiles.count(4000020) == 1 and sum(4000000 <= t < 5000000 for t in tiles) == 1, tiles  # ONE launch for both blocks
    with pytest.raises(Exception):
        eng.debug_tensor(p1, 3)  # lives in LDS only
    mid = eng.debug_tensor(p2, 3)
    monkeypatch.setenv("HP_NO_SEPPAIR", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    got2 = eng2.inference(fr)
    tiles2 = [p["tile"] for p in eng2.profile(3, 1)]
    assert 4000020 not in tiles2 and sum(4000000 <= t < 5000000 for t in tiles2) == 2, tiles2
    assert np.array_equal(mid, eng2.debug_tensor(p2, 3))
    for b in range(3):
        assert np.array_equal(got[b][0][1], got2[b][0][1])


@pytest.mark.parametrize("variant,h,w,act", [
    ("pair", 52, 68, E.ACT_RELU),        # 13 x 17 map: ragged tiles in both directions (8 x 12 output tiles)
    ("pair_res1", 40, 100, E.ACT_RELU),  # 10 x 25: residual on the FIRST 3x3 (the CPM stage's `x + main_block(x)`, lw_openpose.py:118-121)
    ("pair_res2", 36, 44, E.ACT_RELU6),  # 9 x 11: smaller than one tile in x; residual on the second 3x3; relu6 clamp
    ("block", 92, 108, E.ACT_RELU),      # 23 x 27: refinement block 1x1 -> 3x3 -> 3x3 + (1x1's output), lw_openpose.py:176-191
    ("block_nores", 64, 96, E.ACT_RELU), # 16 x 24: whole tiles, 1x1 -> 3x3 -> 3x3 without the residual
    ("block", 8, 8, E.ACT_RELU),         # 2 x 2 map: everything is halo
])
def test_conv_chain_variants(hp, monkeypatch, variant, h, w, act):
    """conv_chain_kernel ([1x1 ->] 3x3 -> 3x3 [+ residual] on 128 channels, intermediates in LDS) against the torch oracle AND against
    the one-launch-per-layer schedule (HP_NO_CHAIN=1): intermediate pixels outside the image must act as zero padding, partial
    tiles must not write outside the map, every residual placement the LW-OpenPose head uses."""
    net = Net(11)
    t = net.conv(0, 3, 32, 3, 2)
    t = net.conv(t, 32, 128, 3, 2)      # the chain's 128-channel input, 1/4 of the frame
    if variant.startswith("block"):
        u = net.conv(t, 128, 128, 1, act=act)
        v = net.conv(u, 128, 128, 3, act=act)
        y = net.conv(v, 128, 128, 3, act=act, res=u if variant == "block" else -1)
    else:
        v = net.conv(t, 128, 128, 3, act=act, res=t if variant == "pair_res1" else -1)
        y = net.conv(v, 128, 128, 3, act=act, res=t if variant == "pair_res2" else -1)
    z = net.conv(y, 128, 32, 1, act=E.ACT_NONE)  # (the chain's output must not be a network output)
    fr = _frames(3, h, w, seed=h)
    outs = [Out("z", z, 0, 32)]
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3)
    tiles = [p["tile"] for p in eng.profile(3, 1)]
    assert sum(7000000 <= t_ < 8000000 for t_ in tiles) == 1, tiles   # the chain kernel really ran, once
    with pytest.raises(Exception):
        eng.debug_tensor(v, 3)   # lives in LDS only
    mid = eng.debug_tensor(y, 3)
    monkeypatch.setenv("HP_NO_CHAIN", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    assert not any(7000000 <= p["tile"] < 8000000 for p in eng2.profile(3, 1))
    got2 = eng2.inference(fr)
    mid2 = eng2.debug_tensor(y, 3)
    _close(mid, mid2, rel=4e-3, abs_=2e-3)          # same fp16 storage points, fp32 sums in another order
    assert (mid != mid2).mean() < 0.2               # ... most stored values identical
    for b in range(3):
        _close(got[b][0][1], got2[b][0][1])


@pytest.mark.parametrize("m,mr,front,h,w", [
    (64, 64, True, 52, 76),     # 13 x 19 map: partial tiles; 3x3 -> 64 -> 256 + shortcut -> the next block's 256 -> 64
    (64, 128, True, 64, 64),    # stage end: the next stage's reduction 256 -> 128 sits behind its projection shortcut in the schedule
    (64, 64, False, 36, 44),    # no 3x3 in front (its stride is 2): expansion + reduction only
    (64, 0, True, 32, 32),      # last block of the network: nothing to reduce
    (128, 128, True, 52, 76),
    (128, 256, True, 40, 40),
    (128, 128, False, 8, 8),    # 2 x 2 map
    (128, 256, False, 24, 40),  # stage end without a 3x3 in front
    (64, 128, False, 20, 28),
    (128, 0, True, 36, 44),
])
def test_bottleneck_variants(hp, monkeypatch, m, mr, front, h, w):
    """bottleneck_kernel ([3x3 ->] 1x1 expansion + shortcut [-> the next block's 1x1 reduction] in one launch, conv_bottleneck.hip)
    against the torch oracle AND the one-launch-per-layer schedule (HP_NO_BNECK=1): the shortcut is added BEFORE the relu (torchvision
    bottleneck), the 3x3's halo outside the image is zero padding, partial tiles write nothing outside the map, and both tensors
    that leave the block (the 4M-channel sum and the reduced one) land where the per-layer schedule puts them."""
    net = Net(m + mr)
    t = net.conv(0, 3, 32, 3, 2)
    x = net.conv(t, 32, 4 * m, 3, 2)                    # the block input = shortcut, 1/4 of the frame
    if front:
        r = net.conv(x, 4 * m, m, 1)                    # this block's reduction (a launch of its own)
        v = net.conv(r, m, m, 3)
    else:
        v = net.conv(x, 4 * m, m, 3, 1)                 # (stands for the stride-2 3x3 of a stage's first block)
    y = net.conv(v, m, 4 * m, 1, res=x, res_before_act=1)
    outs = []
    if mr == 2 * m:
        pj = net.conv(y, 4 * m, 32, 1, stride=2)        # the next stage's projection shortcut comes first in the schedule
        outs.append(Out("pj", pj, 0, 32))
    if mr:
        z = net.conv(y, 4 * m, mr, 1)
        q = net.conv(z, mr, 32, 3, act=E.ACT_NONE)
        outs.append(Out("q", q, 0, 32))
    s2 = net.conv(y, 4 * m, 32, 1, act=E.ACT_NONE)      # a second reader of the sum (the next block's shortcut in a real network)
    outs.append(Out("s", s2, 0, 32))
    fr = _frames(3, h, w, seed=h + m)
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3, rel=3e-3)
    tiles = [p["tile"] for p in eng.profile(3, 1)]
    want = 9000000 + 1000 * (m // 64) + 10 * (mr // 64) + int(front)
    assert tiles.count(want) == 1, tiles                # the fused kernel really ran, once, in the expected instance
    if front:
        with pytest.raises(Exception):
            eng.debug_tensor(v, 3)                      # the 3x3's output lives in LDS only
    ysum = eng.debug_tensor(y, 3)
    zred = eng.debug_tensor(z, 3) if mr else None
    monkeypatch.setenv("HP_NO_BNECK", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    assert not any(p["tile"] >= 9000000 for p in eng2.profile(3, 1))
    got2 = eng2.inference(fr)
    y2 = eng2.debug_tensor(y, 3)
    _close(ysum, y2, rel=4e-3, abs_=2e-3)               # same fp16 storage points, fp32 sums in another order
    assert (ysum != y2).mean() < 0.2
    if mr:
        _close(zred, eng2.debug_tensor(z, 3), rel=4e-3, abs_=2e-3)
    for b in range(3):
        for (n0, a0), (n1, a1) in zip(got[b], got2[b]):
            assert n0 == n1
            _close(a0, a1, rel=4e-3, abs_=2e-3)


@pytest.mark.parametrize("mr,h,w,own", [(64, 52, 76, True), (128, 40, 40, True), (0, 36, 28, True), (64, 44, 36, False), (128, 36, 52, False),
                                        (64, 40, 44, None), (128, 28, 36, None)])
def test_bottleneck_with_projection_shortcut(hp, monkeypatch, mr, h, w, own):
    """The first block of ResNet's first stage: its shortcut is a 1x1 projection (64 -> 256, no activation) of the block input that only
    this block reads.  bottleneck64_kernel<.., PJ> computes it inside the launch (K = [3x3 output ; block input]) instead of reading
    it: the projection's tensor is never written, and the sum skips the fp16 rounding of the projection - compared with the oracle
    (which rounds it) and the per-layer schedule at the tolerance of one fp16 rounding of an addend.  `own`: the block's own reduction
    (1x1 64 -> 64 of the same input) feeds only the 3x3 and is computed on the 3x3's halo tile as well (pixels outside the image must come
    out as the 3x3's zero padding, not relu(bias)) - the launch then reads nothing but the block input."""
    net = Net(7 + mr)
    t = net.conv(0, 3, 32, 3, 2)
    x = net.conv(t, 32, 64, 3, 2)                       # the block input, 1/4 of the frame
    pj = net.conv(x, 64, 256, 1, act=E.ACT_NONE)        # projection shortcut (conv + BN, no relu)
    if own is None:                                     # (no 3x3 the kernel takes in front of the expansion: projection + expansion [+ reduction] only)
        r = net.conv(x, 64, 96, 1)
        v = net.conv(r, 96, 64, 3)
    else:
        r = net.conv(x, 64, 64, 1)
        v = net.conv(r, 64, 64, 3)
    y = net.conv(v, 64, 256, 1, res=pj, res_before_act=1)
    outs = []
    if mr:
        z = net.conv(y, 256, mr, 1)
        outs.append(Out("q", net.conv(z, mr, 32, 3, act=E.ACT_NONE), 0, 32))
    outs.append(Out("s", net.conv(y, 256, 32, 1, act=E.ACT_NONE), 0, 32))
    if not own:
        outs.append(Out("r2", net.conv(r, 64, 32, 1, act=E.ACT_NONE), 0, 32))   # a second reader keeps the reduction a launch of its own
    fr = _frames(3, h, w, seed=h + mr)
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3, rel=4e-3, abs_=2e-3)
    tiles = [p["tile"] for p in eng.profile(3, 1)]
    assert tiles.count(9000000 + 1000 + (300 if own else 100) + 10 * (mr // 64) + (0 if own is None else 1)) == 1, tiles
    with pytest.raises(Exception):
        eng.debug_tensor(pj, 3)                         # the projection is never materialised
    if own:
        with pytest.raises(Exception):
            eng.debug_tensor(r, 3)                      # ... nor the reduction
    ysum = eng.debug_tensor(y, 3)
    monkeypatch.setenv("HP_NO_BNECK", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    got2 = eng2.inference(fr)
    _close(ysum, eng2.debug_tensor(y, 3), rel=4e-3, abs_=4e-3)
    for b in range(3):
        for (n0, a0), (n1, a1) in zip(got[b], got2[b]):
            assert n0 == n1
            _close(a0, a1, rel=4e-3, abs_=4e-3)


def test_lw_openpose_fused_equals_unfused(hp, monkeypatch):
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    w = m.init_weights(7)
    fr = _frames(2, 368, 432, seed=4)
    eng = E.Engine.from_model(m, w, max_batch=2)
    tiles = [p["tile"] for p in eng.profile(2, 1)]
    assert sum(4000000 <= t < 5000000 for t in tiles) == 10  # every MobileNet separable block (the stem's first two share a launch)
    assert tiles.count(4000020) == 1
    assert sum(6000000 <= t < 7000000 for t in tiles) == 2   # init + refinement stage: conf + paf heads share a launch
    assert sum(7000000 <= t < 8000000 for t in tiles) == 8   # CPM (2) + init stage (1) + five refinement blocks as chained launches
    got = eng.inference(fr)
    monkeypatch.setenv("HP_NO_PAIR_HEADS", "1")              # one launch per head: same bits
    solo = E.Engine.from_model(m, w, max_batch=2)
    assert sum(6000000 <= t < 7000000 for t in [p["tile"] for p in solo.profile(2, 1)]) == 4
    for a, b in zip(got, solo.inference(fr)):
        for (n0, x0), (n1, x1) in zip(a, b):
            assert n0 == n1 and np.array_equal(x0, x1)
    monkeypatch.delenv("HP_NO_PAIR_HEADS")
    monkeypatch.setenv("HP_NO_FUSE_HEAD", "1")               # separable blocks fused, heads as two launches: same bits
    monkeypatch.setenv("HP_NO_CHAIN", "1")                   # (the chained 3x3 convolutions sum in another order: compared below)
    mid = E.Engine.from_model(m, w, max_batch=2).inference(fr)
    monkeypatch.setenv("HP_NO_FUSE", "1")
    ref = E.Engine.from_model(m, w, max_batch=2).inference(fr)
    for b in range(2):
        for i in range(2):
            assert np.array_equal(mid[b][i][1], ref[b][i][1])
            _close(got[b][i][1], ref[b][i][1])


@pytest.mark.parametrize("k1,cout2,h,w", [(128, 19, 46, 54), (128, 38, 23, 29), (64, 64, 17, 12), (256, 5, 20, 31)])
def test_fused_two_layer_head(hp, monkeypatch, k1, cout2, h, w):
    """1x1 K1 -> 512 relu -> 1x1 512 -> cout2 as one launch (mlp_head_kernel): the hidden tensor never leaves the
    registers.  Against the oracle (same fp16 rounding point for the hidden activations) and the two-launch schedule;
    the second GEMM sums K in a different order, so the comparison with the unfused engine is a tolerance, not bits."""
    net = Net(k1 + cout2)
    a = net.conv(0, 3, k1, 3, 1)
    cat = net.new_tensor()
    net.conv(a, k1, 32, 1, out=cat, out_coff=0)
    hid = net.conv(a, k1, 512, 1, act=E.ACT_RELU)
    net.conv(hid, 512, cout2, 1, act=E.ACT_NONE, out=cat, out_coff=32)          # fp16 NHWC slice, read by the next conv
    hid2 = net.conv(cat, 32 + cout2, 512, 1, act=E.ACT_RELU6) if (32 + cout2) in (64, 128, 256) else None
    z = net.conv(cat, 32 + cout2, 24, 3, act=E.ACT_LEAKY, act_param=0.1)
    hid3 = net.conv(a, k1, 512, 1, act=E.ACT_RELU)
    y = net.conv(hid3, 512, cout2, 1, act=E.ACT_NONE)                           # fp32 NCHW network output written by the kernel
    outs = [Out("y", y, 0, cout2), Out("z", z, 0, 24), Out("cat", cat, 0, 32 + cout2)]
    if hid2 is not None:
        y2 = net.conv(hid2, 512, 8, 1, act=E.ACT_LEAKY, act_param=0.2)
        outs.append(Out("y2", y2, 0, 8))
    fr = _frames(3, h, w, seed=k1)
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, 3)
    n_heads = sum(6000000 <= p["tile"] < 7000000 for p in eng.profile(3, 1))
    assert n_heads == (3 if hid2 is not None else 2)
    monkeypatch.setenv("HP_NO_FUSE", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, 3)
    got2 = eng2.inference(fr)
    assert not any(p["tile"] >= 6000000 for p in eng2.profile(3, 1))
    for b in range(3):
        for (n1, a1), (n2, a2) in zip(got[b], got2[b]):
            assert n1 == n2
            _close(a1, a2)


def test_engine_save_load_roundtrip(hp, tmp_path):
    """tensorrt::save / tensorrt_serialized (src/tensorrt.cpp:225-252, :463-471): a saved engine reloads to the same bits."""
    m = E.Model("lw_openpose_vggtiny", 128, 96)
    w = m.init_weights(5)
    eng = E.Engine.from_model(m, w, max_batch=3)
    fr = _frames(3, 96, 128, seed=8)
    got = eng.inference(fr)
    path = str(tmp_path / "engine.hpe")
    eng.save(path)
    eng2 = E.Engine.load(path)
    assert (eng2.in_w, eng2.in_h, eng2.max_batch) == (128, 96, 3) and [o[:2] for o in eng2.outputs] == [o[:2] for o in eng.outputs]
    again = eng2.inference(fr)
    for b in range(3):
        for (n1, a1), (n2, a2) in zip(got[b], again[b]):
            assert n1 == n2 and np.array_equal(a1, a2)
    eng4 = E.Engine.load(path, max_batch=5)
    assert eng4.max_batch == 5
    bad = tmp_path / "bad.hpe"
    bad.write_bytes(b"not an engine")
    with pytest.raises(Exception):
        E.Engine.load(str(bad))


def test_partial_batches_are_frame_independent(hp):
    """A max_batch = 8 engine fed 1, 3, 5 and 8 frames (the stream API's ragged last batch, src/stream.cpp) returns for every frame the
    bits it returns for that frame alone - no kernel of the LW-OpenPose schedule mixes frames or depends on the batch size."""
    m = E.Model("lw_openpose_mobilenet", 96, 80)
    w = m.init_weights(3)
    eng = E.Engine.from_model(m, w, max_batch=8)
    fr = _frames(8, 80, 96, seed=11)
    full = eng.inference(fr)
    for n in (1, 3, 5):
        part = eng.inference(fr[:n])
        assert len(part) == n
        for b in range(n):
            for (n0, a0), (n1, a1) in zip(part[b], full[b]):
                assert n0 == n1 and np.array_equal(a0, a1), (n, b, n0)
    solo = eng.inference(fr[6:7])
    for (n0, a0), (n1, a1) in zip(solo[0], full[6]):
        assert np.array_equal(a0, a1)


def test_profile_in_sequence_reports_every_step(hp):
    """hp_engine_profile_sequence (kernels' own begin / end timestamps, schedule order) lists the same steps as the back-to-back
    profile, with positive durations of the same order of magnitude."""
    m = E.Model("lw_openpose_mobilenet", 96, 80)
    eng = E.Engine.from_model(m, m.init_weights(3), max_batch=2)
    a, b = eng.profile(2, 3), eng.profile(2, 3, in_sequence=True)
    assert [(p["layer"], p["op"], p["tile"]) for p in a] == [(p["layer"], p["op"], p["tile"]) for p in b]
    assert all(p["ms"] > 0 for p in b)
    ta, tb = sum(p["ms"] for p in a), sum(p["ms"] for p in b)
    assert 0.2 * ta < tb < 5 * ta
    # hp_engine_profile_pair: the same steps with a second engine of the same model on its own stream (machine time per launch)
    eng2 = E.Engine.from_model(m, m.init_weights(3), max_batch=2)
    c = eng.profile(2, 3, pair=eng2)
    assert [(p["layer"], p["op"], p["tile"]) for p in a] == [(p["layer"], p["op"], p["tile"]) for p in c]
    tc = sum(p["ms"] for p in c)
    assert all(p["ms"] > 0 for p in c) and 0.02 * ta < tc < 20 * ta, (ta, tc)
    with pytest.raises(Exception):
        eng.profile(2, 3, pair=eng)          # the same engine twice
    other = E.Model("lw_openpose_vggtiny", 96, 80)
    with pytest.raises(Exception):
        eng.profile(1, 1, pair=E.Engine.from_model(other, other.init_weights(1), max_batch=2))   # another schedule


def test_corrupted_engine_files_are_rejected_not_crashed(hp, tmp_path):
    """Serialized engines are files: truncations and flipped header / layer bytes must come back as errors (or as a network that
    still builds), never as a crash or an allocation blow-up."""
    m = E.Model("lw_openpose_vggtiny", 64, 48)
    eng = E.Engine.from_model(m, m.init_weights(1), max_batch=1)
    path = str(tmp_path / "e.hpeng")
    eng.save(path)
    raw = bytearray(open(path, "rb").read())
    rng = np.random.default_rng(1)
    bad = 0
    for it in range(60):
        mut = bytearray(raw)
        if it % 3 == 0:
            mut = mut[:int(rng.integers(1, len(mut)))]
        else:
            for _ in range(4):
                mut[int(rng.integers(0, 4096))] = int(rng.integers(0, 256))  # header + first layers
        p2 = str(tmp_path / "m.hpeng")
        open(p2, "wb").write(bytes(mut))
        try:
            E.Engine.load(p2).close()
        except hp.HpError:
            bad += 1
    assert bad >= 20


def test_hostile_layer_fields_are_rejected(hp):
    """hp_engine_create / hp_engine_load take untrusted descriptions: fields that index device memory (channel offsets, pads), and
    layers whose read and write channel ranges overlap inside one launch, must be refused, not turned into out-of-bounds device
    writes or races."""
    def build(mut):
        m = E.Model("lw_openpose_vggtiny", 64, 48)
        w = m.init_weights(1)
        layers = list(m.layers)
        mut(layers)
        return E.Engine(layers, m.outputs, w, 64, 48, max_batch=1)

    build(lambda L: None).close()

    def neg_out_coff(L):
        L[3].out_coff = -8
    def huge_out_coff(L):
        L[3].out_coff = 1 << 20
    def neg_in_coff(L):
        L[3].in_coff = -8
    def huge_pad(L):
        L[3].pad_explicit = 1
        L[3].pad[:] = [1 << 30, 1, 1, 1]
    def neg_pad(L):
        L[3].pad_explicit = 1
        L[3].pad[:] = [1, -3, 1, 1]
    def in_place(L):       # reads and writes the same channels of one tensor
        L[3].out = L[3].in_
    def res_in_place(L):   # residual = the tensor being written, same channels
        k = next(i for i, l in enumerate(L) if l.res >= 0)
        L[k].res = L[k].out
    def bad_res_flag(L):
        L[3].res_before_act = 7
    for mut in (neg_out_coff, huge_out_coff, neg_in_coff, huge_pad, neg_pad, in_place, res_in_place, bad_res_flag):
        with pytest.raises(hp.HpError):
            build(mut)

    # the same through the serialized-engine loader: every 32-bit word of the first layer records set to hostile values
    import struct as S
    m = E.Model("lw_openpose_vggtiny", 64, 48)
    eng = E.Engine.from_model(m, m.init_weights(1), max_batch=1)
    import tempfile, os
    d = tempfile.mkdtemp()
    path = os.path.join(d, "e.hpeng")
    eng.save(path)
    raw = bytearray(open(path, "rb").read())
    rejected = 0
    for word in range(8, 8 + 4 * 26):  # the header and the first four hp_layer records (26 words each)
        for val in (-8, -1, 1 << 30, 0x7fffffff):
            mut = bytearray(raw)
            mut[word * 4:word * 4 + 4] = S.pack("<i", val)
            p2 = os.path.join(d, "m.hpeng")
            open(p2, "wb").write(bytes(mut))
            try:
                e2 = E.Engine.load(p2)
                fr = np.zeros((1, 48, 64, 3), np.uint8)
                e2.inference(fr)   # whatever still loads must also run without faulting
                e2.close()
            except (hp.HpError, ValueError):
                rejected += 1
    assert rejected > 100


def test_onnx_pads_block_the_fused_kernels(hp):
    """A 1x1 convolution with ONNX pads (its output is larger than its input) must not be folded into the fused two-layer head or
    the fused separable block, which assume an unpadded 1x1: result == the unfused evaluation by the torch oracle."""
    from oracle import ref_net
    L = [E.make_layer(E.OP_CONV, 0, 1, 3, 128, k=3, act=E.ACT_RELU, w_off=0, b_off=3456),
         E.make_layer(E.OP_CONV, 1, 2, 128, 512, k=1, act=E.ACT_RELU, w_off=3584, b_off=3584 + 65536, pads=(1, 1, 1, 1)),
         E.make_layer(E.OP_CONV, 2, 3, 512, 19, k=1, w_off=3584 + 65536 + 512, b_off=3584 + 65536 + 512 + 9728)]
    n_w = 3584 + 65536 + 512 + 9728 + 19
    rng = np.random.default_rng(3)
    w = (rng.standard_normal(n_w) * 0.05).astype(np.float32)
    o = E.OutputDesc()
    o.name, o.tensor, o.coff, o.channels = b"y", 3, 0, 19
    eng = E.Engine(L, [o], w, 32, 24, max_batch=1)
    fr = rng.integers(0, 256, (1, 24, 32, 3), dtype=np.uint8)
    got = eng.inference(fr)[0][0][1]
    ref = ref_net.run(L, [o], w, frames_u8=fr, match_fp16=True)["y"][0]
    assert got.shape == ref.shape == (19, 26, 34)
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 2e-3


@pytest.mark.parametrize("k,cin,cout,h,w", [
    (7, 128, 128, 23, 29), (7, 185, 128, 23, 29), (7, 128, 256, 37, 50), (3, 256, 256, 30, 35), (3, 512, 128, 31, 23),
    (3, 192, 384, 15, 47), (3, 256, 256, 23, 29), (5, 128, 128, 23, 29), (5, 320, 128, 17, 31),
])
def test_direct_conv_any_kernel_and_width(hp, k, cin, cout, h, w):
    """conv_direct_kernel (8 wavefronts, halo tile of a channel chunk in LDS for all k*k taps, fragment-ordered weights from L2,
    double-buffered 64-channel chunks) against the torch oracle: 7x7 / 5x5 / 3x3, one chunk and several, a 185-channel concat
    slice (padded to 192), PReLU, residual after and before the activation, tiles hanging over the right / bottom edge."""
    net = Net(k * 1000 + cin)
    cat = net.new_tensor()
    c0 = min(cin, 128)
    net.conv(0, 3, c0, 3, 1, out=cat, out_coff=0)
    if cin > c0:
        net.conv(0, 3, cin - c0, 3, 1, act=E.ACT_LEAKY, act_param=0.2, out=cat, out_coff=c0)
    u = net.conv(cat, cin, cout, k, act=E.ACT_PRELU)
    v = net.conv(u, cout, cout, k, act=E.ACT_RELU, res=u, res_before_act=0)
    r = net.conv(v, cout, cout, k, act=E.ACT_RELU, res=u, res_before_act=1)
    y = net.conv(r, cout, 19, 1, act=E.ACT_NONE)
    fr = _frames(3, h, w, seed=k + cin)
    eng, got, ref = _run_both(net, [Out("y", y, 0, 19), Out("z_mid", v, 0, cout)], fr, h, w)
    _check(got, ref, 3)
    prof = eng.profile(3, 1)
    # the k x k layers whose output stays fp16 NHWC really ran on conv_direct_kernel (z_mid is also a network output: generic epilogue)
    want = 2 if (k > 3 or cout > 128) else 1  # (3x3 layers with 128 input channels stay on conv3x3_direct_kernel)
    if k == 3 and h * w < 0.68 * (-(-h // 16) * 16) * (-(-w // 12) * 12):
        want = 0                              # (3x3 on a map the 16 x 12 tiles cover badly - 23 x 29: 4 x 3 tiles for 3.5 - goes to the generic kernel)
    assert sum(1 for p in prof if p["tile"] >= 6000000) == want, [p["tile"] for p in prof]


@pytest.mark.parametrize("cin,cout,h,w,batch", [(512, 256, 12, 12, 3), (256, 128, 16, 24, 4), (2048, 128, 12, 12, 2)])
def test_direct_conv_split_k_on_small_maps(hp, monkeypatch, cin, cout, h, w, batch):
    """3x3 convolutions whose 16x12 tiles are fewer than the CUs (12 x 12 maps of the ResNet heads) split their channel chunks over 2 or
    4 blocks per tile (conv_splitk); the fp32 partial sums meet in a second launch that runs the epilogue (bias, PReLU, residual).
    Against the oracle and against the unsplit launch (HP_NO_SPLITK=1: same kernel, other fp32 summation order)."""
    net = Net(cin + cout)
    cat = net.new_tensor()
    for c0 in range(0, cin, 128):
        net.conv(0, 3, 128, 3, 1, out=cat, out_coff=c0, act=E.ACT_LEAKY, act_param=0.1 + c0 / 4096)
    u = net.conv(cat, cin, cout, 3, act=E.ACT_PRELU)
    v = net.conv(u, cout, cout, 1, act=E.ACT_NONE)
    fr = _frames(batch, h, w, seed=cin)
    outs = [Out("v", v, 0, cout)]
    eng, got, ref = _run_both(net, outs, fr, h, w)
    _check(got, ref, batch, rel=3e-3)
    mid = eng.debug_tensor(u, batch)
    refu = ref_net.run(net.layers, [Out("u", u, 0, cout)], net.blob(), frames_u8=fr, match_fp16=True)["u"]
    _close(mid, refu, rel=2e-3, abs_=1e-3)      # the split convolution's own output against the oracle
    monkeypatch.setenv("HP_NO_SPLITK", "1")
    eng2 = E.Engine(net.layers, [o.c() for o in outs], net.blob(), w, h, batch)
    got2 = eng2.inference(fr)
    mid2 = eng2.debug_tensor(u, batch)
    _close(mid, mid2, rel=2e-3, abs_=1e-3)
    assert (mid != mid2).mean() < 0.25          # mostly the same fp16 values ...
    assert (mid != mid2).any()                   # ... but the split really happened (another summation order somewhere)


def test_vgg19_small_through_the_direct_kernels(hp):
    """OpenPose-VGG19 at 64 x 96: the 5x5-free 3x3 / 7x7 layers run on conv_direct_kernel / conv3x3_direct_kernel where their shapes
    allow and on the implicit GEMM elsewhere (maps smaller than two tiles); all of it against the oracle."""
    m = E.Model("openpose_vgg19", 96, 64)
    w = m.init_weights(9)
    fr = _frames(2, 64, 96, seed=3)
    eng = E.Engine.from_model(m, w, max_batch=2)
    got = eng.inference(fr)
    tiles = [p["tile"] for p in eng.profile(2, 1)]
    assert sum(1 for t in tiles if 6000000 <= t < 7000000) >= 3, tiles
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=True, mean=m.mean, inv_std=m.inv_std)
    _check(got, ref, 2, rel=4e-3, abs_=2e-3)


def test_vgg_64_channel_layers_on_the_direct_kernel(hp):
    """3x3 64 -> 64 (VGG19's second layer: 64 output channels = ONE 64-row block column) and 128 -> 64 through conv3x3_direct_kernel,
    feeding further layers (fp16 NHWC epilogue), against the oracle."""
    net = Net(77)
    a = net.conv(0, 3, 64, 3, 1)
    b = net.conv(a, 64, 64, 3, 1)
    c = net.conv(b, 64, 128, 3, 1)
    d = net.conv(c, 128, 64, 3, 1, act=E.ACT_PRELU)
    y = net.conv(d, 64, 19, 1, act=E.ACT_NONE)
    fr = _frames(2, 37, 41, seed=4)
    eng, got, ref = _run_both(net, [Out("y", y, 0, 19)], fr, 37, 41)
    _check(got, ref, 2)
    tiles = [p["tile"] for p in eng.profile(2, 1)]
    assert tiles[1] // 1000000 == 5 and tiles[3] // 1000000 == 5, tiles


def test_c_abi_rccl_communicator_single_rank(hp):
    """hp_dist_* (RCCL bound at run time): unique id, communicator on this device, broadcast of a weight blob.  One GPU here, so
    world = 1 (the N-rank path is the same three calls; the driver's multi-GPU bench exercises RCCL through torch.distributed)."""
    import ctypes as C
    L = hp.lib()
    uid = (C.c_char * 128)()
    hp.check(L.hp_dist_unique_id(uid))
    assert any(b != 0 for b in bytes(uid))
    comm = C.c_void_p()
    hp.check(L.hp_dist_init(C.byref(comm), 0, 1, uid))
    w = np.arange(1000, dtype=np.float32)
    hp.check(L.hp_dist_broadcast_weights(comm, w.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(w.size), 0))
    assert np.array_equal(w, np.arange(1000, dtype=np.float32))
    with pytest.raises(hp.HpError):
        hp.check(L.hp_dist_broadcast_weights(comm, w.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(w.size), 3))
    L.hp_dist_destroy(comm)
