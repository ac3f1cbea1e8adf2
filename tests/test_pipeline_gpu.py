"""GPU: hp_pipeline_* (hyperpose::stream on the GPU) = the same stages run by hand: oracle resize / letterbox on the host ->
engine -> PAF parser -> oracle resume_ratio; batches come back in submission order, several in flight."""
import numpy as np
import pytest

from hyperpose_amd import engine as E
from hyperpose_amd.parser import Paf
from hyperpose_amd.pipeline import Pipeline
from oracle import loader

pytestmark = pytest.mark.gpu

SIZES = [(640, 480), (432, 368), (300, 500), (864, 736), (1280, 720), (97, 61)]


def _frames(rng, n, k0):
    return [rng.integers(0, 256, (SIZES[(k0 + i) % len(SIZES)][1], SIZES[(k0 + i) % len(SIZES)][0], 3), dtype=np.uint8) for i in range(n)]


def _by_hand(eng, paf, frames, keep_ratio, in_w, in_h):
    net = np.stack([loader.letterbox_u8(f, in_w, in_h) if keep_ratio else loader.resize_linear_u8(f, in_w, in_h) for f in frames])
    maps = eng.inference(net)
    humans = paf.process_batch(np.stack([m[0][1] for m in maps]), np.stack([m[1][1] for m in maps]))
    if keep_ratio:
        for f, hs in zip(frames, humans):
            for h in hs:
                xy = np.stack([h["parts"]["x"], h["parts"]["y"]], 1)
                r = loader.resume_ratio(xy, (f.shape[1], f.shape[0]), (in_w, in_h))
                h["parts"]["x"], h["parts"]["y"] = r[:, 0], r[:, 1]
    return humans


@pytest.mark.parametrize("keep_ratio", [False, True])
def test_pipeline_equals_stages_by_hand(hp, keep_ratio):
    in_w, in_h = 160, 128
    m = E.Model("lw_openpose_mobilenet", in_w, in_h)
    w = m.init_weights(11)
    for L in m.layers:  # blow up the two output convolutions: random weights then give O(1) maps, peaks, limbs and humans
        if L.op == E.OP_CONV and L.cout in (19, 38) and L.out in [o.tensor for o in m.outputs]:
            w[L.w_off:L.w_off + L.cout * L.cin] *= 400.0
    pl = Pipeline(m, w, max_batch=4, n_pipes=3, keep_ratio=keep_ratio, conf_thresh=0.05, paf_thresh=-1e9, max_frame_wh=(1280, 720))
    eng = E.Engine.from_model(m, w, max_batch=4)
    paf = Paf(conf_thresh=0.05, paf_thresh=-1e9, max_batch=4)
    rng = np.random.default_rng(5)
    batches = [_frames(rng, n, k) for n, k in ((4, 0), (3, 2), (1, 5), (4, 1), (2, 3))]
    got = []
    for b in batches:                      # three in flight, then steady state, then drain
        if pl.in_flight == pl.n_pipes:
            got.append(pl.collect())
        pl.submit(b)
    with pytest.raises(Exception):
        if pl.in_flight == pl.n_pipes:
            pl.submit(batches[0])          # all pipes busy -> HP_ERR_STATE
        else:
            raise RuntimeError("not full")
    while pl.in_flight:
        got.append(pl.collect())
    assert len(got) == len(batches)
    total = 0
    for b, g in zip(batches, got):
        ref = _by_hand(eng, paf, b, keep_ratio, in_w, in_h)
        assert len(g) == len(b)
        for hg, hr in zip(g, ref):
            assert hg.tobytes() == hr.tobytes()
            total += len(hg)
    assert total > 0  # the loose thresholds make the random-weight maps produce humans: the comparison is not vacuous


@pytest.mark.parametrize("kind", ["ppn", "pifpaf"])
def test_pipeline_other_parsers_equal_stages_by_hand(hp, kind):
    """hp_pipeline_create_ex with the PoseProposal / PifPaf parser == resize on the host -> engine -> the blocking parser call."""
    from hyperpose_amd.parser import PifPaf, PoseProposal
    if kind == "ppn":
        in_w = in_h = 192
        m = E.Model("pose_proposal_resnet50", in_w, in_h)
    else:
        in_w = in_h = 129
        m = E.Model("pifpaf_resnet50", in_w, in_h)
    w = m.init_weights(5)
    pl = Pipeline(m, w, max_batch=3, n_pipes=2, keep_ratio=False, max_frame_wh=(1280, 720), parser=kind,
                  thresholds=(0.02, 0.01, 0.3) if kind == "ppn" else (0.1,))
    eng = E.Engine.from_model(m, w, max_batch=3)
    rng = np.random.default_rng(9)
    batches = [_frames(rng, n, k) for n, k in ((3, 0), (2, 3), (3, 1))]
    batches[1][0] = rng.integers(0, 256, (in_h, in_w, 3), dtype=np.uint8)  # a network-sized frame: direct H2D, no resize kernel
    got = []
    for b in batches:
        if pl.in_flight == pl.n_pipes:
            got.append(pl.collect())
        pl.submit(b)
    while pl.in_flight:
        got.append(pl.collect())
    for b, g in zip(batches, got):
        net = np.stack([loader.resize_linear_u8(f, in_w, in_h) for f in b])
        maps = eng.inference(net)
        if kind == "ppn":
            par = PoseProposal((in_w, in_h), 0.02, 0.01, 0.3, max_batch=3)
            g6 = in_w // 32
            tens = [np.stack([fm[i][1] for fm in maps]) for i in range(6)] + [np.stack([fm[6][1] for fm in maps]).reshape(len(b), 17, 9, 9, g6, g6)]
            ref = par.process_batch(tens)
        else:
            par = PifPaf(in_h, in_w, 0.1, max_batch=3)
            fh = maps[0][1][1].shape[-1]
            ref = par.process_batch(np.stack([fm[0][1] for fm in maps]).reshape(len(b), 19, 9, fh, fh),
                                    np.stack([fm[1][1] for fm in maps]).reshape(len(b), 17, 5, fh, fh))
        assert len(g) == len(b)
        for hg, hr in zip(g, ref):
            assert hg.tobytes() == hr.tobytes()
