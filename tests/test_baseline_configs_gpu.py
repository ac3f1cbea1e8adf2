"""GPU: the BASELINE.json configurations at their FULL sizes: oracle parity of probed frames against oracle/ref_net.py evaluated by
PyTorch's own fp32 GPU kernels (the CPU evaluation needs minutes per frame; small-size parity against it is in test_engine_gpu.py),
plus size-independent properties:
  * batch invariance — frame i of a full batch == the same frame run alone (bit for bit: every output pixel
    accumulates in the same order wherever its tile falls),
  * duplicated frames in one batch give identical maps, different frames do not,
  * the parser consuming the device-resident maps agrees bit for bit with the CPU oracle / the reference's own parser
    fed with the same maps copied to the host.
"""
import numpy as np
import pytest

from hyperpose_amd import engine as E
from hyperpose_amd import synth
from oracle import loader, ref_net

pytestmark = pytest.mark.gpu


def _maps(eng, frames):
    got = eng.inference(frames)
    return [[a for _, a in g] for g in got]


def _check_full_size_against_torch_gpu(m, w, frames, full, probe, rel, abs_):
    """Oracle parity AT FULL SIZE: the probed frames of the batch against oracle/ref_net.py evaluated in fp32 by PyTorch's own GPU
    kernels (MIOpen / rocBLAS: an implementation independent of libhp_hip.so; the CPU evaluation would take minutes per frame).
    fp16 storage points matched, so what remains is fp32 summation order and the fp16 rounding flips it causes downstream:
    |err| <= rel * max|ref| + abs.  Measured on MI355X in round 4 (this function prints it): 2.3e-3 (configs[0]), 2.5e-3 ([1]), 2.0e-3 ([2]),
    3.1e-3 ([3]: 53 layers, the deepest path) and 9.6e-4 ([4]) - against <= 2e-3 at the small sizes of test_engine_gpu.py.  The difference
    is statistics, not a layer: an output map of 54 x 96 x 57 values (or 144 cells x 1 485 channels) samples the tail of the same
    per-value error distribution 100 x more often than a 6 x 8 map, and every extra layer adds its own 2^-11 rounding flips.  The bound
    is therefore 6e-3 for every configuration (2 x the worst observed), not the 1e-2 / 2e-2 of round 3."""
    import torch
    assert torch.cuda.is_available()
    worst = 0.0
    for i in probe:
        ref = ref_net.run(m.layers, m.outputs, w, frames_u8=frames[i:i + 1], match_fp16=True, mean=m.mean, inv_std=m.inv_std, device="cuda")
        names = sorted(ref)
        assert len(names) == len(full[i])
        for k, nm in enumerate(names):
            r, g = ref[nm][0], full[i][k]
            assert r.shape == g.shape, (nm, r.shape, g.shape)
            err, scale = float(np.abs(g - r).max()), float(np.abs(r).max())
            worst = max(worst, err / max(scale, 1e-30))
            assert err <= rel * scale + abs_, f"frame {i} output {nm}: max err {err:.4g} vs scale {scale:.4g}"
    torch.cuda.empty_cache()
    print(f"\n{m.arch} @ {m.in_h}x{m.in_w}: worst |err| / max|ref| over the probed frames {worst:.2e} (bound {rel:.0e})")


def _check_invariance(eng, frames, probe=(0, 5)):
    full = _maps(eng, frames)
    for i in probe:
        alone = _maps(eng, frames[i:i + 1])[0]
        for a, b in zip(alone, full[i]):
            assert np.array_equal(a, b), f"frame {i} differs between batch and single run"
    assert all(np.isfinite(a).all() for m in full for a in m)
    return full


def test_config0_tinyvgg_single_image_368x432(hp):
    """BASELINE.json configs[0] at its full size: TinyVGG-V2 + PAF parser on ONE 368 x 432 image (the only GPU test of this
    topology ran at 64 x 48)."""
    from hyperpose_amd.parser import Paf
    m = E.Model("lw_openpose_vggtiny", 432, 368)
    w = m.init_weights(20240)
    eng = E.Engine.from_model(m, w, max_batch=1)
    fr = synth.images_u8(synth.rng_for(0), 1, 368, 432)
    full = _maps(eng, fr)
    assert full[0][0].shape == (19, 46, 54) and full[0][1].shape == (38, 46, 54) and all(np.isfinite(a).all() for a in full[0])
    _check_full_size_against_torch_gpu(m, w, fr, full, (0,), 6e-3, 2e-3)
    again = _maps(eng, fr)
    assert all(np.array_equal(a, b) for a, b in zip(full[0], again[0]))   # deterministic
    # the parser on the device-resident maps == the reference's own parser on the same maps
    eng.inference(fr)
    p = Paf(max_batch=1)
    (_, cs, cp), (_, ps, pp) = eng.outputs
    humans = p.process_batch_device(cp, pp, 1, cs, ps)
    oh, _, _ = loader.ref_paf_process(full[0][0], full[0][1])
    assert humans[0].tobytes() == oh.tobytes()
    # and the fp32-faithful engine of the same topology at this size
    e32 = E.Engine.from_model(m, w, max_batch=1, dtype="f32")
    ref = ref_net.run(m.layers, m.outputs, w, frames_u8=fr, match_fp16=False, mean=m.mean, inv_std=m.inv_std, device="cuda")
    for nm, arr in e32.inference(fr)[0]:
        assert float(np.abs(arr - ref[nm][0]).max()) <= 1e-4 * float(np.abs(ref[nm]).max()) + 1e-6, nm


def test_config1_lw_openpose_b8_368x432(hp):
    from hyperpose_amd.parser import Paf
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    eng = E.Engine.from_model(m, m.init_weights(20241), max_batch=8)
    fr = synth.images_u8(synth.rng_for(1), 8, 368, 432)
    fr[3] = fr[1]
    full = _check_invariance(eng, fr)
    _check_full_size_against_torch_gpu(m, m.init_weights(20241), fr, full, (0, 7), 6e-3, 2e-3)
    assert np.array_equal(full[3][0], full[1][0]) and not np.array_equal(full[2][0], full[1][0])
    # parser on the device-resident DNN output == oracle on the same maps
    eng.inference(fr)
    p = Paf(max_batch=8)
    (_, cs, cp), (_, ps, pp) = eng.outputs
    humans = p.process_batch_device(cp, pp, 8, cs, ps)
    for b in range(8):
        oh, _, _ = loader.ref_paf_process(full[b][0], full[b][1])
        assert humans[b].tobytes() == oh.tobytes()
    # asynchronous hand-over: engine on its stream, parser on its OWN stream behind it (hp_stream_wait_stream)
    dev = __import__("hyperpose_amd")._lib.DevBuf.from_numpy(fr)
    eng.enqueue_u8(dev, 8)
    p.after(eng.stream)
    p.enqueue(cp, pp, 8, cs, ps)
    again = p.collect()
    for b in range(8):
        assert again[b].tobytes() == humans[b].tobytes()


def test_config2_openpose_vgg19_b16_432x768(hp):
    from hyperpose_amd.parser import Paf
    m = E.Model("openpose_vgg19", 768, 432)
    eng = E.Engine.from_model(m, m.init_weights(20242), max_batch=16)
    fr = synth.images_u8(synth.rng_for(2), 16, 432, 768)
    full = _check_invariance(eng, fr, probe=(0, 9))
    assert full[0][0].shape == (19, 54, 96) and full[0][1].shape == (38, 54, 96)
    _check_full_size_against_torch_gpu(m, m.init_weights(20242), fr, full, (0, 15), 6e-3, 2e-3)
    eng.inference(fr)
    p = Paf(max_batch=16)
    (_, cs, cp), (_, ps, pp) = eng.outputs
    humans = p.process_batch_device(cp, pp, 16, cs, ps)
    for b in (0, 7, 15):
        oh, _, _ = loader.ref_paf_process(full[b][0], full[b][1])
        assert humans[b].tobytes() == oh.tobytes()
    # injected realistic maps at this geometry (54x96 -> 384 rows x 216 cols up-sampled)
    conf, paf, _ = synth.paf_maps(synth.rng_for(2, salt=1), 16, 54, 96, people=(2, 4, 8, 16))
    hs = p.process_batch(conf, paf)
    total = 0
    for b in (0, 3, 11):
        oh, _, _ = loader.ref_paf_process(conf[b], paf[b])
        assert hs[b].tobytes() == oh.tobytes()
        total += len(oh)
    assert total >= 10


def test_config3_pose_proposal_resnet50_b32_384(hp):
    from hyperpose_amd.parser import PoseProposal
    m = E.Model("pose_proposal_resnet50", 384, 384)
    eng = E.Engine.from_model(m, m.init_weights(20243), max_batch=32)
    fr = synth.images_u8(synth.rng_for(3), 32, 384, 384)
    full = _check_invariance(eng, fr, probe=(0, 20))
    assert [a.shape for a in full[0]] == [(18, 12, 12)] * 6 + [(17 * 81, 12, 12)]
    _check_full_size_against_torch_gpu(m, m.init_weights(20243), fr, full, (0, 31), 6e-3, 2e-3)
    eng.inference(fr)
    parser = PoseProposal((384, 384), max_batch=32)
    humans = parser.process_batch([p for _, _, p in eng.outputs], on_device=True, n=32, conf_shape=(18, 12, 12), edge_shape=(17, 9, 9, 12, 12))
    if loader.ref_lib() is not None:
        for b in (0, 13, 31):
            refh = loader.ref_ppn_process(full[b][:6] + [full[b][6].reshape(17, 9, 9, 12, 12)])
            assert humans[b].tobytes() == refh.tobytes()


def test_config4_pifpaf_resnet50_b64_385(hp):
    from hyperpose_amd.parser import PifPaf
    m = E.Model("pifpaf_resnet50", 385, 385)
    eng = E.Engine.from_model(m, m.init_weights(20244), max_batch=64)
    fr = synth.images_u8(synth.rng_for(4), 64, 385, 385)
    full = _check_invariance(eng, fr, probe=(0, 40))
    assert full[0][0].shape == (171, 49, 49) and full[0][1].shape == (85, 49, 49)
    _check_full_size_against_torch_gpu(m, m.init_weights(20244), fr, full, (0, 63), 6e-3, 2e-3)
    eng.inference(fr)
    parser = PifPaf(385, 385, max_batch=64)
    humans = parser.process_batch(eng.outputs[0][2], eng.outputs[1][2], on_device=True, n=64, fh=49, fw=49)
    if loader.ref_lib() is not None:
        for b in (0, 33, 63):
            refh = loader.ref_pifpaf_process(full[b][0].reshape(19, 9, 49, 49), full[b][1].reshape(17, 5, 49, 49))
            assert humans[b].tobytes() == refh.tobytes()
