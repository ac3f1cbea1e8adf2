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
    oh, op, oc = loader.ref_paf_process(conf, paf)
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


@pytest.mark.parametrize("rows,cols", [(8, 12), (20, 30), (60, 33), (13, 70)])
def test_peaks_kernel_odd_geometries(hp, rows, cols):
    """paf_peaks_kernel's wavefront-local column layout (46 columns + 9 halo lanes per wavefront, 184 per block) on maps that are
    narrower than one wavefront (4 x 8 = 32 columns), that end inside a wavefront (80), that need two strips with a remainder (240),
    and that are short (52 rows: two bands) - up-sampled / smoothed planes, peaks, connections and humans against the oracle."""
    from hyperpose_amd.parser import Paf
    rng = synth.rng_for(1, salt=7000 + rows * 100 + cols)
    B = 3
    conf, paf, _ = synth.paf_maps(rng, B, rows, cols, people=(1, 2, 3))
    p = Paf(max_batch=B)
    up, sm = p.debug_maps(conf[0], 4 * cols, 4 * rows)
    ref_up = loader.resize_area(conf[0], 4 * cols, 4 * rows)
    assert _same(up, ref_up) and _same(sm, loader.smooth(ref_up))
    humans = p.process_batch(conf, paf)
    for f in range(B):
        _check_frame(p, f, conf[f], paf[f], humans[f])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_crowds_against_oracle(hp, seed):
    """Random crowds (0 .. 30 people per frame, noisy maps, dropped joints): the assembly kernel's parallel limbs, its sequential loops
    for limbs that revisit a part, merges, and - above 64 skeleton fragments - the restart on the LDS tables, all against the oracle's
    peaks / connections / humans bit for bit."""
    from hyperpose_amd.parser import Paf
    rng = np.random.default_rng(4200 + seed)
    B = 8
    people = tuple(int(v) for v in rng.integers(0, 31, B))
    conf, paf, _ = synth.paf_maps(np.random.default_rng(77 + seed), B, people=people, noise=float(rng.choice([0.01, 0.03, 0.06])),
                                  drop_joint_prob=float(rng.choice([0.0, 0.05, 0.2])))
    p = Paf(max_batch=B, cap_per_frame=256)
    humans = p.process_batch(conf, paf)
    total = 0
    for f in range(B):
        total += _check_frame(p, f, conf[f], paf[f], humans[f])
    assert total >= 10


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
        oh, op, oc = loader.ref_paf_process(conf[f], paf[f], 0.1, 0.08, 216, 184)
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
        assert _same(a[f], loader.ref_paf_process(conf[f], paf[f])[0])


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
    # a frame that overflows the initial per-part peak list (512): the reference's vectors grow (src/post_process.hpp:171-193), so the
    # parser re-parses the batch with doubled lists inside collect and must match the oracle instead of reporting a capacity error
    noisy = np.zeros_like(conf[:2])
    noisy[0, :, ::2, ::2] = 1.0  # a bright cell every 2 x 2 feature cells: 23 x 27 = 621 maxima per part > 512
    noisy[1] = conf[3]
    npaf = np.stack([paf[0] * 0, paf[3]])
    humans = p.process_batch(noisy, npaf)
    oh, op, oc = loader.ref_paf_process(noisy[0], npaf[0], cap_peaks=32768, cap_conns=32768)
    assert len(op) > 18 * 600  # (the blurred grid also peaks between the bright cells at the border)
    assert _same(p.debug_peaks(0, cap=32768), op) and _same(humans[0], oh)
    _check_frame(p, 1, noisy[1], npaf[1], humans[1])   # the other frame of the grown batch
    # ... and the parser is clean (and still exact) afterwards, now with the larger lists
    humans = p.process_batch(conf[:4], paf[:4])
    for f in range(4):
        _check_frame(p, f, conf[f], paf[f], humans[f])


def test_lists_grow_like_the_references_vectors(hp):
    """More candidates on one limb than the initial list holds (2048): 66 necks left, 66 right shoulders right, a constant PAF pointing
    right -> thousands of the 4356 pairs pass both criteria.  The reference's vector grows (src/paf.cpp:108-141); the parser re-parses
    with doubled lists and must equal the oracle, connections and all."""
    from hyperpose_amd.parser import Paf
    rows, cols = 46, 54
    conf = np.zeros((2, 19, rows, cols), np.float32)
    paf = np.zeros((2, 38, rows, cols), np.float32)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    for gy in range(2, rows - 1, 4):
        for gx in range(1, cols // 2 - 2, 4):
            conf[0, 1] += np.exp(-((xx - gx) ** 2 + (yy - gy) ** 2) / 0.5).astype(np.float32)
            conf[0, 2] += np.exp(-((xx - (gx + cols // 2)) ** 2 + (yy - gy) ** 2) / 0.5).astype(np.float32)
    paf[0, 12] = 1.0
    conf[0, 18] = 1 - conf[0, :18].max(0)
    c1, p1, _ = synth.paf_maps(synth.rng_for(1, salt=72), 1, rows, cols, people=(6,))
    conf[1], paf[1] = c1[0], p1[0]
    p = Paf(max_batch=2)
    humans = p.process_batch(conf, paf)
    for f in range(2):
        oh, op, oc = loader.ref_paf_process(conf[f], paf[f], cap_peaks=32768, cap_conns=32768)
        assert _same(p.debug_peaks(f, cap=32768), op), f
        assert _same(p.debug_conns(f, cap=32768), oc), f
        assert _same(humans[f], oh), f
    assert len(loader.ref_paf_process(conf[0], paf[0])[2]) >= 66


def test_many_skeleton_fragments(hp):
    """More skeleton fragments in one frame than the assembly kernel held until round 5 (512; now 1024): three limbs that share no part
    (elbow-wrist right and left, knee-ankle right) as 230 isolated pairs each - the PAF is non-zero only on the two cells between a pair's
    parts, so no candidate survives but the true ones.  None of the 690 fragments reaches four parts: the reference builds them all
    (src/paf.cpp:146-231) and removes them; the frame next to it in the batch has ordinary people."""
    from hyperpose_amd.parser import Paf
    rows, cols = 46, 54
    conf = np.zeros((2, 19, rows, cols), np.float32)
    paf = np.zeros((2, 38, rows, cols), np.float32)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    limbs = ((3, 4, 16), (6, 7, 24), (9, 10, 4))   # (part a, part b, x channel of the limb's PAF): COCOPAIRS 3, 5, 8
    n = 0
    for gy in range(1, rows - 1, 2):
        for gx in range(1, cols - 3, 5):
            for a, b, ch in limbs:
                conf[0, a] += np.exp(-((xx - gx) ** 2 + (yy - gy) ** 2) / 0.5).astype(np.float32)
                conf[0, b] += np.exp(-((xx - (gx + 2)) ** 2 + (yy - gy) ** 2) / 0.5).astype(np.float32)
                paf[0, ch, gy, gx:gx + 3] = 1.0
            n += 1
    conf[0, 18] = 1 - conf[0, :18].max(0)
    c1, p1, _ = synth.paf_maps(synth.rng_for(1, salt=73), 1, rows, cols, people=(5,))
    conf[1], paf[1] = c1[0], p1[0]
    p = Paf(max_batch=2)
    humans = p.process_batch(conf, paf)
    for f in range(2):
        oh, op, oc = loader.ref_paf_process(conf[f], paf[f], cap_peaks=32768, cap_conns=32768)
        assert _same(p.debug_peaks(f, cap=32768), op), f
        assert _same(p.debug_conns(f, cap=32768), oc), f
        assert _same(humans[f], oh), f
        if f == 0:
            assert len(oc) > 512, len(oc)   # every connection of these limbs opens a fragment of its own


def _tie_maps(n_necks, rows=46, cols=54, gap=4):
    """Heat-maps in which candidate connections of limb 0 (neck -> right shoulder, COCOPAIRS[0] = (1, 2), PAF channels 12 / 13) TIE
    exactly: every neck has a shoulder `gap` cells to its right and one `gap` cells to its left, the PAF x-field is exactly +1 right of
    the neck column and -1 left of it (constant, so all ten samples of a candidate are the same value) and both candidates of a neck have
    the same length -> bit-equal criterion2.  The two tied candidates SHARE the neck: only one survives get_connections' greedy pass
    (src/paf.cpp:252-270), and which one is decided by the sort order of equal scores."""
    conf = np.zeros((19, rows, cols), np.float32)
    paf = np.zeros((38, rows, cols), np.float32)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    def blob(k, y, x, a=1.0):
        conf[k] += a * np.exp(-((xx - x) ** 2 + (yy - y) ** 2) / 2.0).astype(np.float32)
    cx = cols // 2
    ys = np.linspace(4, rows - 5, n_necks).astype(int)
    for y in ys:
        blob(1, y, cx)
        blob(2, y, cx + gap)
        blob(2, y, cx - gap)
    paf[12][:, cx + 1:] = 1.0   # x component of limb 0
    paf[12][:, :cx] = -1.0
    conf[18] = 1 - conf[:18].max(0)
    return conf, paf


@pytest.mark.parametrize("n_necks,gap", [(1, 4), (3, 4), (12, 4), (12, 3), (12, 5), (10, 6)])
def test_forced_ties_in_get_connections(hp, n_necks, gap):
    """Equal candidate scores (src/paf.cpp:249 `std::sort(..., std::greater)` leaves their order to the implementation; the
    reference's result is whatever libstdc++ leaves).  Up to 16 candidates per limb libstdc++'s std::sort is a pure insertion sort
    (stable = generation order; n_necks = 1, 3); with more (10 / 12 necks: 20 / 24-way ties among > 16 candidates) introsort's
    median-of-three partitioning decides, which paf_limbs_kernel reproduces step by step (`libstdcxx_sort_greater`; larger and
    adversarial sequences: test_restated_std_sort_against_libstdcxx).
    Connections and humans must equal the reference-compiled parser bit for bit in every case."""
    from hyperpose_amd.parser import Paf
    conf, paf = _tie_maps(n_necks, gap=gap)
    oh, op, oc = loader.ref_paf_process(conf, paf)
    limb0 = oc[oc["pair_id"] == 0]
    assert len(limb0) == n_necks, (len(limb0), n_necks)      # one survivor per neck: the ties really conflicted
    scores = limb0["score"]
    assert np.all(scores == scores[0])                        # ... and really were exact ties
    p = Paf(max_batch=1)
    gh = p.process(conf, paf)
    gc = p.debug_conns(0)
    assert _same(p.debug_peaks(0), op)
    assert _same(gc, oc), f"{n_necks} necks: connections differ from the libstdc++ order"
    assert _same(gh, oh)


def test_restated_std_sort_against_libstdcxx(hp):
    """`libstdcxx_sort_greater` alone (hp_paf_debug_sort) against the host's real std::sort (oracle_std_sort_greater, the call of
    src/paf.cpp:249) on score sequences in generation order: random scores drawn from few distinct values (mass ties), all-equal,
    sorted / reversed / organ-pipe runs, and McIlroy-adversary sequences built against this very libstdc++ that exhaust introsort's
    depth limit (2 floor(log2 n)) and run its heap-sort fall-back - which the device must report (`used_heap`) and reproduce."""
    from hyperpose_amd.parser import paf_debug_sort
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 16, 17, 18, 33, 100, 257, 1000, 4097):
        cases.append((f"ties{n}", rng.integers(0, max(2, n // 8), n).astype(np.float32), None))
        cases.append((f"rand{n}", rng.normal(size=n).astype(np.float32), None))
    cases.append(("equal", np.full(300, 0.25, np.float32), False))
    cases.append(("ascending", np.arange(500, dtype=np.float32), False))
    cases.append(("descending", -np.arange(500, dtype=np.float32), False))
    cases.append(("organ", np.concatenate([np.arange(200), np.arange(200)[::-1]]).astype(np.float32), None))
    for n in (40, 200, 1000, 5000):
        cases.append((f"killer{n}", loader.sort_killer(n), True))
        k = loader.sort_killer(n)
        cases.append((f"killer_ties{n}", np.floor(k / 3).astype(np.float32), None))
    heaps = 0
    for name, scores, want_heap in cases:
        order, used_heap = paf_debug_sort(scores)
        ref = loader.std_sort_greater(scores)
        assert np.array_equal(order, ref), name
        if want_heap is not None:
            assert used_heap == want_heap, (name, used_heap)
        heaps += used_heap
    assert heaps >= 4
