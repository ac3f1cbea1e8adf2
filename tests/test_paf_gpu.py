"""GPU: hyperpose::parser::paf on gfx950 (libhp_hip.so via the C ABI) vs the CPU oracle, bit for bit.

Integer outputs (peak coordinates, ids, connection indices, has_value) must be identical; float outputs
(scores, normalised x/y) must be identical BITS as well, because every float expression in the kernels keeps
the CPU code's operand order with contraction off.  That is stricter than the 1e-3 px bar of BASELINE.json.
"""
import json
import os

import numpy as np
import pytest

from hyperpose_amd import synth
from oracle import loader

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def _check_frame(parser, f, conf, paf, humans):
    oh, op, oc = loader.paf_process(conf, paf)
    gp = parser.debug_peaks(f)
    assert _same(gp, op), f"peaks differ: gpu {len(gp)} vs oracle {len(op)}"
    gc = parser.debug_conns(f)
    assert _same(gc, oc), f"connections differ: gpu {len(gc)} vs oracle {len(oc)}"
    assert _same(humans, oh), f"humans differ: gpu {len(humans)} vs oracle {len(oh)}"
    return len(oh)


def test_upsample_and_blur_maps_bit_exact(hp):
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=11)
    conf, _, _ = synth.paf_maps(rng, 1, people=(4,))
    p = Paf(max_batch=1)
    up, sm = p.debug_maps(conf[0], 216, 184)
    ref_up = loader.resize_area(conf[0], 216, 184)
    assert _same(up, ref_up)
    assert _same(sm, loader.smooth(ref_up))


def test_golden_vectors(hp):
    from hyperpose_amd.parser import Paf
    g = np.load(os.path.join(GOLD, "paf_golden.npz"))
    meta = json.loads(str(g["meta"]))
    for i, m in enumerate(meta):
        p = Paf(max_batch=1)
        humans = p.process(g[f"conf_{i}"], g[f"paf_{i}"])
        assert _same(humans, g[f"humans_{i}"]), m
        assert _same(p.debug_peaks(0), g[f"peaks_{i}"]), m
        assert _same(p.debug_conns(0), g[f"conns_{i}"]), m


@pytest.mark.parametrize("rows,cols", [(46, 54), (46, 46), (54, 96)])
def test_batch_parity_with_oracle(hp, rows, cols):
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=rows * 100 + cols)
    B = 8
    conf, paf, _ = synth.paf_maps(rng, B, rows, cols, people=(1, 2, 4, 8, 16, 3, 0, 6))
    p = Paf(max_batch=B)
    humans = p.process_batch(conf, paf)
    total = 0
    for f in range(B):
        total += _check_frame(p, f, conf[f], paf[f], humans[f])
    assert total >= 20  # the synthetic frames really contain parsable people


def test_explicit_resolution_and_thresholds(hp):
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=5)
    conf, paf, _ = synth.paf_maps(rng, 2, people=(3, 5))
    p = Paf(conf_thresh=0.1, paf_thresh=0.08, resolution_size=(216, 184), max_batch=2)  # cv::Size(w, h): un-swapped 4x
    humans = p.process_batch(conf, paf)
    for f in range(2):
        oh, op, oc = loader.paf_process(conf[f], paf[f], 0.1, 0.08, 216, 184)
        assert _same(p.debug_peaks(f), op) and _same(p.debug_conns(f), oc) and _same(humans[f], oh)
        assert len(oh) >= 1


def test_device_resident_and_async(hp):
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=9)
    conf, paf, _ = synth.paf_maps(rng, 4, people=(2, 3, 4, 5))
    dc, dp = hp.DevBuf.from_numpy(conf), hp.DevBuf.from_numpy(paf)
    p = Paf(max_batch=4)
    a = p.process_batch_device(dc, dp, 4, conf.shape[1:], paf.shape[1:])
    p.enqueue(dc, dp, 4, conf.shape[1:], paf.shape[1:])
    b = p.collect()
    c = p.process_batch(conf, paf)
    for f in range(4):
        assert _same(a[f], b[f]) and _same(a[f], c[f])
        assert _same(a[f], loader.paf_process(conf[f], paf[f])[0])


def test_errors_are_codes_not_crashes(hp):
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=2)
    conf, paf, _ = synth.paf_maps(rng, 1, people=(1,))
    p = Paf(max_batch=1)
    p.process(conf[0], paf[0])
    with pytest.raises(hp.HpError) as e:  # shape change after first call: reference = UB, here HP_ERR_STATE
        p.process(conf[0][:, :40], paf[0][:, :40])
    assert e.value.code == hp.HP_ERR_STATE
    with pytest.raises(hp.HpError) as e:  # batch larger than max_batch
        p.process_batch(np.repeat(conf, 2, 0), np.repeat(paf, 2, 0))
    assert e.value.code == hp.HP_ERR_CAPACITY
    with pytest.raises(hp.HpError):
        p.collect()


def test_noise_only_and_dense_frames(hp):
    """Edge cases: no people at all; heavy clutter (many peaks/candidates) still matches the oracle."""
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=21)
    conf, paf, _ = synth.paf_maps(rng, 2, people=(0, 24), noise=0.02)
    p = Paf(max_batch=2)
    humans = p.process_batch(conf, paf)
    assert len(humans[0]) == 0
    for f in range(2):
        _check_frame(p, f, conf[f], paf[f], humans[f])


def test_preproc_matches_oracle(hp):
    from hyperpose_amd.parser import preproc_u8hwc_to_f32nchw
    img = synth.images_u8(synth.rng_for(0), 3, 37, 53)
    for flip in (True, False):
        out = preproc_u8hwc_to_f32nchw(img, 1 / 255, flip)
        assert _same(out, loader.nhwc_u8_to_nchw_f32(img, 1 / 255, flip))


def test_state_between_batches(hp):
    """The assemble kernel leaves the peak counters and overflow flags zeroed for the next batch (no memsets between
    batches): batches of different sizes through ONE parser, an overflowing batch in between, each clean batch still bit-exact."""
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=33)
    conf, paf, _ = synth.paf_maps(rng, 8, people=(5, 1, 0, 9, 2, 7, 3, 12))
    p = Paf(max_batch=8)
    for lo, hi in ((0, 8), (3, 5), (5, 6), (0, 8), (6, 8)):
        humans = p.process_batch(conf[lo:hi], paf[lo:hi])
        for f in range(hi - lo):
            _check_frame(p, f, conf[lo + f], paf[lo + f], humans[f])
    # a frame that overflows the per-part peak list: reported as a capacity error ...
    noisy = np.zeros_like(conf[:1])
    noisy[:, :, ::2, ::2] = 1.0  # a bright cell every 2 x 2 feature cells: 23 x 27 = 621 maxima per part > 512
    with pytest.raises(hp.HpError) as e:
        p.process_batch(noisy, paf[:1])
    assert e.value.code == hp.HP_ERR_CAPACITY
    # ... and the parser is clean again afterwards
    humans = p.process_batch(conf[:4], paf[:4])
    for f in range(4):
        _check_frame(p, f, conf[f], paf[f], humans[f])
