"""PoseProposal parser: golden vectors come from the REFERENCE'S OWN code (oracle/_ref, src/pose_proposal.cpp
compiled where it lies); the GPU path (threshold / box / NMS / edge-gather kernel + the assembly kernel: restated std::sort,
root rule, 64 x 64 hash merge, one wavefront per frame) must reproduce them bit for bit, with NO frame handed to the host
statements (hp_ppn_decode_flags)."""
import json
import os

import numpy as np
import pytest

from hyperpose_amd import synth
from oracle import loader

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ppn_golden.npz")


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def _cases():
    g = np.load(GOLD)
    meta = json.loads(str(g["meta"]))
    for i, m in enumerate(meta):
        yield m, [g[f"t{k}_{i}"].astype(np.float32) for k in range(7)], g[f"humans_{i}"]


def test_golden_matches_reference_build_when_present():
    """CPU: in a container that mounts /root/reference, the committed fixtures are what the reference computes."""
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    n = 0
    for m, t, humans in _cases():
        assert _same(loader.ref_ppn_process(t), humans), m
        n += len(humans)
    assert n >= 10


@pytest.mark.gpu
def test_gpu_matches_golden(hp):
    from hyperpose_amd.parser import PoseProposal
    p = PoseProposal((384, 384), max_batch=4)
    for m, t, humans in _cases():
        got = p.process(t)
        assert _same(got, humans), (m, len(got), len(humans))
        assert p.decode_flags(1)[0] == 0   # assembled on the device


@pytest.mark.gpu
def test_gpu_batch_matches_reference_live(hp):
    """Batch of 32 (BASELINE config 3 geometry: 384x384, 12x12 grid), clutter included, device-resident inputs."""
    from hyperpose_amd.parser import PoseProposal
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    B = 32
    t = synth.ppn_maps(synth.rng_for(3, salt=77), B, people=(1, 2, 3, 4, 6, 8, 0, 5), spurious=0.03)
    p = PoseProposal((384, 384), max_batch=B)
    got = p.process_batch(t)
    dev = [hp.DevBuf.from_numpy(a) for a in t]
    got_dev = p.process_batch(dev, on_device=True, n=B, conf_shape=t[0].shape[1:], edge_shape=t[6].shape[1:])
    assert not p.decode_flags(B).any()   # every frame assembled by ppn_assemble_kernel
    total = 0
    for b in range(B):
        ref = loader.ref_ppn_process([a[b] for a in t])
        assert _same(got[b], ref), f"frame {b}: {len(got[b])} vs {len(ref)}"
        assert _same(got_dev[b], ref)
        total += len(ref)
    assert total >= 60


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_device_tail_crowds_ties_and_host_tail(hp, monkeypatch, seed):
    """The device tail on what stresses it: crowds (up to 14 people on the 12 x 12 grid -> shared cells, merges through the hash with
    stale indices), heavy clutter (hundreds of limb candidates per limb: beyond std::sort's 16-element insertion-sort regime) and
    edge confidences quantised to a few values (mass ties: the order of equal candidates is libstdc++'s).  Against the reference's
    own code, frame by frame, with the decode flags asserted; and the host statements (HP_PPN_HOST_TAIL=1) give the same bytes."""
    from hyperpose_amd.parser import PoseProposal
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(100 + seed)
    B = 16
    people = tuple(int(v) for v in rng.integers(0, 15, B))
    t = synth.ppn_maps(synth.rng_for(3, salt=900 + seed), B, people=people, spurious=0.03 if seed == 3 else float(rng.choice([0.1, 0.25])))
    t = [np.ascontiguousarray(a) for a in t]
    if seed >= 2:   # quantise the edge confidences: exact ties among the limb candidates (seed 3: the clutter, U(0, 0.04), falls onto
        q_ = 8 if seed == 2 else 200   # four values above the 0.02 threshold -> hundreds of candidates per limb in a few tie classes)
        t[6] = (np.round(t[6] * q_) / q_).astype(np.float32)
    p = PoseProposal((384, 384), 0.10, 0.02 if seed == 3 else 0.05, 0.3, max_batch=B, cap_per_frame=256)
    got = p.process_batch(t)
    flags = p.decode_flags(B)
    n_ref = 0
    for b in range(B):
        ref = loader.ref_ppn_process([a[b] for a in t], 384, 384, 0.10, 0.02 if seed == 3 else 0.05, 0.3, cap=256)
        assert _same(got[b], ref), f"frame {b} ({people[b]} people, flags {flags[b]}): {len(got[b])} vs {len(ref)}"
        n_ref += len(ref)
    assert n_ref >= 20
    # (seed 3 lowers the limb threshold into the clutter: tens to hundreds of candidates per limb and several hundred skeleton fragments per
    # frame.  Until round 6 a 257th fragment sent the frame to the host statements - flag 4 - and half of these frames went there; the fragments
    # beyond the 256 kept in LDS now live in an HBM scratch list and every frame is assembled on the device)
    assert (flags == 0).all(), flags
    monkeypatch.setenv("HP_PPN_HOST_TAIL", "1")
    q = PoseProposal((384, 384), 0.10, 0.02 if seed == 3 else 0.05, 0.3, max_batch=B, cap_per_frame=256)
    host = q.process_batch(t)
    assert (q.decode_flags(B) == -1).all()
    for b in range(B):
        assert _same(host[b], got[b]), b


@pytest.mark.gpu
def test_gpu_thresholds_and_other_resolution(hp):
    from hyperpose_amd.parser import PoseProposal
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    t = synth.ppn_maps(synth.rng_for(3, salt=5), 4, people=(2, 4, 6, 3), spurious=0.05)
    p = PoseProposal((384, 384), 0.3, 0.1, 0.5, max_batch=4)
    got = p.process_batch(t)
    for b in range(4):
        assert _same(got[b], loader.ref_ppn_process([a[b] for a in t], 384, 384, 0.3, 0.1, 0.5))
    p.set_thresholds(0.05, 0.036, 0.1)
    got = p.process_batch(t)
    for b in range(4):
        assert _same(got[b], loader.ref_ppn_process([a[b] for a in t], 384, 384, 0.05, 0.036, 0.1))


@pytest.mark.gpu
def test_gpu_async_enqueue_collect(hp):
    """hp_ppn_enqueue / hp_ppn_collect (lists written straight to pinned memory, tails on the worker pool) == the blocking call ==
    the reference; a second enqueue before collect is refused; repeated batches do not leak state between frames."""
    from hyperpose_amd.parser import PoseProposal
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    B = 16
    p = PoseProposal((384, 384), max_batch=B)
    for salt in (1, 2):
        t = synth.ppn_maps(synth.rng_for(3, salt=salt), B, people=(4, 0, 2, 7), spurious=0.03)
        dev = [hp.DevBuf.from_numpy(a) for a in t]
        p.enqueue(dev, B, t[0].shape[1:], t[6].shape[1:])
        with pytest.raises(Exception):
            p.enqueue(dev, B, t[0].shape[1:], t[6].shape[1:])
        got = p.collect()
        for b in range(B):
            assert _same(got[b], loader.ref_ppn_process([a[b] for a in t])), (salt, b)
    with pytest.raises(Exception):
        p.collect()  # nothing in flight
