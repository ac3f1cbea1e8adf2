"""PifPaf parser: golden vectors come from the REFERENCE'S OWN code (oracle/_ref: src/pifpaf.cpp +
src/pifpaf_decoder/*.cpp compiled where they lie); the GPU path (cell compaction, on-demand hi-res look-ups, seed /
CAF scoring kernels + host grow/NMS tail) must reproduce them bit for bit (PifPaf key-points are integers after
the reference's truncation, so 1e-3 px parity means identical values — SURVEY.md A.3)."""
import json
import os

import numpy as np
import pytest

from hyperpose_amd import synth
from oracle import loader

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pifpaf_golden.npz")


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def _cases():
    g = np.load(GOLD)
    for i, m in enumerate(json.loads(str(g["meta"]))):
        yield m, g[f"paf_{i}"].astype(np.float32), g[f"pif_{i}"].astype(np.float32), g[f"humans_{i}"]


def test_golden_matches_reference_build_when_present():
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    n = 0
    for m, paf, pif, humans in _cases():
        assert _same(loader.ref_pifpaf_process(paf, pif, m["net_h"], m["net_w"]), humans), m
        n += len(humans)
    assert n >= 10


@pytest.mark.gpu
def test_gpu_matches_golden(hp):
    from hyperpose_amd.parser import PifPaf
    for m, paf, pif, humans in _cases():
        p = PifPaf(m["net_h"], m["net_w"], max_batch=1)
        got = p.process(paf, pif)
        assert _same(got, humans), (m, len(got), len(humans))


@pytest.mark.gpu
def test_gpu_batch_matches_reference_live(hp):
    """Batch of 16 at BASELINE config 4 geometry (385x385 -> 49x49 fields), host and device-resident inputs."""
    from hyperpose_amd.parser import PifPaf
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    B = 16
    paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=9), B, people=(1, 2, 3, 4, 6, 8, 0, 5))
    p = PifPaf(385, 385, max_batch=B)
    got = p.process_batch(paf, pif)
    dpaf, dpif = hp.DevBuf.from_numpy(paf), hp.DevBuf.from_numpy(pif)
    got_dev = p.process_batch(dpaf, dpif, on_device=True, n=B, fh=49, fw=49)
    total = 0
    for b in range(B):
        ref = loader.ref_pifpaf_process(paf[b], pif[b])
        assert _same(got[b], ref), f"frame {b}: {len(got[b])} vs {len(ref)}"
        assert _same(got_dev[b], ref)
        total += len(ref)
    assert total >= 40


@pytest.mark.gpu
def test_gpu_threshold_variants(hp):
    from hyperpose_amd.parser import PifPaf
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=3), 2, people=(4, 7), noise=0.15)
    for thr in (0.05, 0.3, 0.6):
        p = PifPaf(385, 385, thr, max_batch=2)
        got = p.process_batch(paf, pif)
        for b in range(2):
            assert _same(got[b], loader.ref_pifpaf_process(paf[b], pif[b], 385, 385, thr))


@pytest.mark.gpu
def test_gpu_async_enqueue_collect(hp):
    """hp_pifpaf_enqueue / hp_pifpaf_collect == the blocking call == the reference's own decoder, batch of 24 decoded on the host
    worker pool (frames in any order across the threads, results in frame order)."""
    from hyperpose_amd.parser import PifPaf
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    B = 24
    p = PifPaf(385, 385, max_batch=B)
    for salt in (3, 4):
        paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=salt), B, people=(3, 0, 1, 5, 2))
        dp, di = hp.DevBuf.from_numpy(paf), hp.DevBuf.from_numpy(pif)
        p.enqueue(dp, di, B, 49, 49)
        with pytest.raises(Exception):
            p.enqueue(dp, di, B, 49, 49)
        got = p.collect()
        total = 0
        for b in range(B):
            ref = loader.ref_pifpaf_process(paf[b], pif[b])
            assert got[b].tobytes() == ref.tobytes(), (salt, b, len(got[b]), len(ref))
            total += len(ref)
        assert total >= 20
    with pytest.raises(Exception):
        p.collect()
