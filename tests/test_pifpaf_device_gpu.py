"""PifPaf device decoder (pp_decode_kernel: seed order, grow with the reference's frontier queue, occupancy, soft-NMS, sort, remap as one
wavefront per frame; reference src/pifpaf_decoder/openpifpaf_postprocessor.cpp:382-635, 764-851) against the reference's own decoder
(oracle/_ref) and against the host tail (HP_PIFPAF_HOST_TAIL=1), byte for byte.  `decode_flags` says which path produced each frame, so a
device bug cannot hide behind the fall-back."""
import os

import numpy as np
import pytest

from hyperpose_amd import synth
from oracle import loader

pytestmark = pytest.mark.gpu


def _parser(host_tail, *a, **k):
    from hyperpose_amd.parser import PifPaf
    old = os.environ.get("HP_PIFPAF_HOST_TAIL")
    os.environ["HP_PIFPAF_HOST_TAIL"] = "1" if host_tail else "0"
    try:
        return PifPaf(*a, **k)
    finally:
        if old is None:
            del os.environ["HP_PIFPAF_HOST_TAIL"]
        else:
            os.environ["HP_PIFPAF_HOST_TAIL"] = old


def test_device_decoder_matches_reference_and_host_tail(hp):
    if loader.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    B = 24
    dev, host = _parser(False, 385, 385, max_batch=B), _parser(True, 385, 385, max_batch=B)
    on_device = humans = 0
    for salt, people, noise in ((3, (3, 0, 1, 5, 2), 0.02), (11, (1, 2, 3, 4, 6, 8, 0, 5), 0.02), (12, (4, 7, 9, 12), 0.1)):
        paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=salt), B, people=people, noise=noise)
        got, ref_host = dev.process_batch(paf, pif), host.process_batch(paf, pif)
        flags = dev.decode_flags(B)
        assert host.decode_flags(B) == [-1] * B
        for b in range(B):
            ref = loader.ref_pifpaf_process(paf[b], pif[b])
            assert got[b].tobytes() == ref.tobytes(), (salt, b, flags[b], len(got[b]), len(ref))
            assert ref_host[b].tobytes() == ref.tobytes()
            humans += len(ref)
        on_device += sum(f == 0 for f in flags)
        assert all(f in (0, 32) for f in flags), flags  # only the rounding guard may hand a frame of this kind to the host
    assert on_device >= 3 * B - 3, on_device
    assert humans >= 150


def test_device_decoder_stress_against_host_tail(hp):
    """Many noisy frames (dense seeds, long CAF lists, crowded occupancy): device == host tail on every frame; the frames the device
    handed back are counted and must stay the exception."""
    B = 64
    dev, host = _parser(False, 385, 385, 0.05, max_batch=B, cap_per_frame=256), _parser(True, 385, 385, 0.05, max_batch=B, cap_per_frame=256)
    fell_back = total = 0
    for salt, noise in ((21, 0.05), (22, 0.2), (23, 0.3)):
        paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=salt), B, people=(2, 5, 9, 14, 20, 1, 0, 30), noise=noise)
        got, ref = dev.process_batch(paf, pif), host.process_batch(paf, pif)
        flags = dev.decode_flags(B)
        for b in range(B):
            assert got[b].tobytes() == ref[b].tobytes(), (salt, b, flags[b], len(got[b]), len(ref[b]))
            total += len(ref[b])
        fell_back += sum(f != 0 for f in flags)
    assert total >= 500
    assert fell_back <= 10, fell_back


def test_device_decoder_other_geometry_and_async(hp):
    """Non-square fields (321x481 network), asynchronous halves, device-resident inputs."""
    B = 8
    fh, fw = 41, 61
    dev, host = _parser(False, 321, 481, max_batch=B), _parser(True, 321, 481, max_batch=B)
    paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=31), B, fh=fh, fw=fw, people=(2, 3, 5, 0))
    dp, di = hp.DevBuf.from_numpy(paf), hp.DevBuf.from_numpy(pif)
    dev.enqueue(dp, di, B, fh, fw)
    got = dev.collect()
    ref = host.process_batch(paf, pif)
    assert sum(f == 0 for f in dev.decode_flags(B)) >= B - 1
    for b in range(B):
        assert got[b].tobytes() == ref[b].tobytes(), b
        if loader.ref_lib() is not None:
            assert got[b].tobytes() == loader.ref_pifpaf_process(paf[b], pif[b], 321, 481).tobytes()
    assert sum(len(r) for r in ref) >= 10


@pytest.mark.parametrize("fh,fw", [(9, 9), (13, 33), (70, 25)])
def test_device_decoder_small_and_elongated_fields(hp, fh, fw):
    """Fields far from the 49 x 49 of configs[4]: a 65 x 65 network (9 x 9 cells, occupancy window of a few dozen cells), a wide and a tall
    one (70 x 25 = 1750 cells per list)."""
    B = 4
    net_h, net_w = (fh - 1) * 8 + 1, (fw - 1) * 8 + 1
    dev, host = _parser(False, net_h, net_w, max_batch=B), _parser(True, net_h, net_w, max_batch=B)
    paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=100 * fh + fw), B, fh=fh, fw=fw, people=(1, 2, 0, 3), noise=0.05)
    got, ref = dev.process_batch(paf, pif), host.process_batch(paf, pif)
    flags = dev.decode_flags(B)
    assert all(f in (0, 32) for f in flags), flags
    for b in range(B):
        assert got[b].tobytes() == ref[b].tobytes(), (b, flags[b], len(got[b]), len(ref[b]))
        if loader.ref_lib() is not None:
            assert got[b].tobytes() == loader.ref_pifpaf_process(paf[b], pif[b], net_h, net_w).tobytes(), b


def test_mixed_batch_device_and_host_tail(hp, monkeypatch):
    """A batch in which the device decoder hands every second frame back (test hook HP_PIFPAF_DECLINE_ODD): only those frames are packed
    and decoded by the host tail, the others come from the kernel; the batch equals the all-host result frame by frame."""
    B = 9
    monkeypatch.setenv("HP_PIFPAF_DECLINE_ODD", "1")
    dev = _parser(False, 385, 385, max_batch=B)
    monkeypatch.delenv("HP_PIFPAF_DECLINE_ODD")
    host = _parser(True, 385, 385, max_batch=B)
    for salt in (41, 42):
        paf, pif = synth.pifpaf_maps(synth.rng_for(4, salt=salt), B, people=(2, 3, 1, 4, 0, 5), noise=0.05)
        dp, di = hp.DevBuf.from_numpy(paf), hp.DevBuf.from_numpy(pif)
        dev.enqueue(dp, di, B, 49, 49)
        got = dev.collect()
        ref = host.process_batch(paf, pif)
        flags = dev.decode_flags(B)
        assert [f & 64 for f in flags] == [64 if b & 1 else 0 for b in range(B)], flags
        assert all((f & ~(64 | 32)) == 0 for f in flags), flags
        for b in range(B):
            assert got[b].tobytes() == ref[b].tobytes(), (salt, b, flags[b])
        assert sum(len(r) for r in ref) >= 15


def _lattice_maps(B, fh=49, fw=49, step=4, types=(0, 5, 11), scale=0.3):
    """A frame nobody would film: isolated PIF clusters on a lattice (3 x 3 voting cells every `step` cells, three joint types, small scales so
    that the occupancy of one annotation does not cover its neighbours) and empty CAF fields - every cluster seeds an annotation of its own:
    several hundred annotations per frame."""
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:fh, 0:fw].astype(np.float64)
    pif = np.zeros((B, 17, 5, fh, fw))
    paf = np.zeros((B, 19, 9, fh, fw))
    pif[:, :, 1], pif[:, :, 2] = xx, yy
    pif[:, :, 4] = scale
    for c in (1, 3):
        paf[:, :, c], paf[:, :, c + 1] = xx, yy
    paf[:, :, 7:9] = 1.0
    for b in range(B):
        for k, t in enumerate(types):
            for cy in range(2 + k % 2, fh - 2, step):
                for cx in range(2 + (k + b) % 2, fw - 2, step):
                    jx, jy = cx + rng.normal(0, 0.05), cy + rng.normal(0, 0.05)
                    conf = rng.uniform(0.7, 0.95)
                    for y in range(cy - 1, cy + 2):
                        for x in range(cx - 1, cx + 2):
                            pif[b, t, 0, y, x] = conf
                            pif[b, t, 1, y, x] = jx + rng.normal(0, 0.01)
                            pif[b, t, 2, y, x] = jy + rng.normal(0, 0.01)
    return paf.astype(np.float32), pif.astype(np.float32)


def test_hundreds_of_annotations_stay_on_the_device(hp):
    """Round 6: PD_MAXA and PD_Q went from 256 to 1024.  A lattice frame yields 300 - 700 annotations (each seed its own,
    most of them dropped by the final score filter: the ANNOTATION list is what PD_MAXA bounds); the device decoder keeps every frame (flag 1 =
    "more than PD_MAXA annotations" is not raised) and equals the host tail and the reference byte for byte."""
    B = 4
    dev, host = _parser(False, 385, 385, 0.05, max_batch=B, cap_per_frame=1024), _parser(True, 385, 385, 0.05, max_batch=B, cap_per_frame=1024)
    paf, pif = _lattice_maps(B)
    # ... and a few real skeletons among the clusters, so that the frames also return humans (a cluster alone is dropped by the final filter)
    ppaf, ppif = synth.pifpaf_maps(synth.rng_for(4, salt=55), B, people=(5, 6, 4, 7), noise=0.0)
    person = ppif[:, :, 0] > 0.05
    for c in range(5):
        pif[:, :, c][person] = ppif[:, :, c][person]
    paf = ppaf.astype(np.float32)
    got, ref = dev.process_batch(paf, pif), host.process_batch(paf, pif)
    flags = dev.decode_flags(B)
    assert all(f in (0, 32) for f in flags), flags
    assert sum(len(r) for r in ref) >= 10
    for b in range(B):
        assert got[b].tobytes() == ref[b].tobytes(), (b, flags[b], len(got[b]), len(ref[b]))
        if loader.ref_lib() is not None:
            assert got[b].tobytes() == loader.ref_pifpaf_process(paf[b], pif[b], 385, 385, 0.05, cap=1024).tobytes(), b
    # (that these frames are beyond the old capacity was checked by building with PD_MAXA = 256: every frame then carries flag 1 and goes to the
    # host tail - tools/r6_count_ann.py)
    assert all(int((pif[b, :, 0] >= 0.5).sum()) > 9 * 256 for b in range(B))
