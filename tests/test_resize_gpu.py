"""GPU: hp_resize_u8c3 / hp_letterbox_u8c3 (resize.hip) against the restated OpenCV arithmetic (oracle/resize_oracle.cpp),
bit for bit, over up-scales, down-scales, the 2x2 and identity special cases and both letterbox branches."""
import numpy as np
import pytest

from hyperpose_amd import frontend
from oracle import loader

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 432, 368), (1280, 720, 432, 368), (100, 80, 432, 368), (864, 736, 432, 368),
                                         (432, 368, 432, 368), (33, 57, 64, 64), (7, 5, 20, 3), (1920, 1080, 385, 385), (3, 2, 1, 1)])
def test_resize_bit_exact(hp, sw, sh, dw, dh):
    src = np.random.default_rng(sw * 31 + dh).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    assert np.array_equal(frontend.resize_host(src, dw, dh), loader.resize_linear_u8(src, dw, dh))


@pytest.mark.parametrize("sw,sh", [(640, 480), (480, 640), (1280, 720), (432, 368), (500, 500), (33, 900), (864, 736)])
def test_letterbox_bit_exact(hp, sw, sh):
    src = np.random.default_rng(sw + sh).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    got = frontend.resize_host(src, 432, 368, keep_ratio=True, bgcolor=(3, 250, 77))
    assert np.array_equal(got, loader.letterbox_u8(src, 432, 368, bgcolor=(3, 250, 77)))
    assert frontend.letterbox_inner(sw, sh, 432, 368) == loader.letterbox_inner(sw, sh, 432, 368)
