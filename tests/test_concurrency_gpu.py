"""GPU: engines of DIFFERENT precisions side by side on two streams give the bits they give alone.

Round 6 found that they did not: isolated values of the fp32 first-layer kernel's output (`first_conv32_kernel`) differed between runs whenever
fp16-MFMA kernels of another stream - an fp16 engine, or the split engine `HP_DTYPE_F32S` - shared its CUs.  The kernel's LDS inputs were intact
and the same FMA chain evaluated twice in one thread gave two results (`HP_FIRST_CONV_VERIFY`); compiled without the packed FMAs
(`v_pk_fma_f32`) hipcc's SLP pass forms there it is exact (hyperpose_amd/build.py).  These tests are the regression:
before the fix 30 - 60 % of the delayed engine's runs differed."""
import time

import numpy as np
import pytest

from hyperpose_amd import engine as E
from hyperpose_amd import synth

pytestmark = pytest.mark.gpu


def _outs(e, n):
    return [e.output_to_host(i, n) for i in range(len(e.outputs))]


@pytest.mark.parametrize("pair,delay_us", [(("f16", "f32"), 50), (("f32s", "f32"), 150), (("f32s", "f32s"), 150), (("f16", "f32"), 250)])
def test_mixed_precision_engines_side_by_side(hp, pair, delay_us):
    n = 8
    m = E.Model("lw_openpose_mobilenet", 432, 368)
    w = m.init_weights(5)
    engs = [E.Engine.from_model(m, w, max_batch=n, dtype=d) for d in pair]
    dev = hp.DevBuf.from_numpy(synth.images_u8(synth.rng_for(12), n, 368, 432))
    alone = []
    for e in engs:
        e.enqueue_u8(dev, n)
        e.synchronize()
        alone.append(_outs(e, n))
    bad = 0
    for rep in range(60):
        for k, e in enumerate(engs):
            if k:  # the second engine's first layers meet the first engine's middle layers
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e6 < delay_us:
                    pass
            e.enqueue_u8(dev, n)
        for e in engs:
            e.synchronize()
        for k, e in enumerate(engs):
            bad += sum(not np.array_equal(a, b) for a, b in zip(alone[k], _outs(e, n)))
    assert bad == 0, f"{bad} output tensors of {pair} differed from the engines' own solo results"
    assert all(e.split_fallbacks == 0 for e in engs)
