"""CPU: the restated cv::resize (INTER_LINEAR, 8UC3) / non_scaling_resize / resume_ratio of the stream front-end
(oracle/resize_oracle.cpp).  PARITY UNPINNED (no OpenCV here): checked against an independent float bilinear model
(+-1 LSB: the fixed-point rounding), exact special cases, and the reference's own arithmetic for the letterbox size."""
import numpy as np
import pytest

from oracle import loader


def _float_bilinear(src, dw, dh):
    sh, sw, _ = src.shape
    fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    ax = np.where((x0 < 0) | (x0 >= sw - 1), 0.0, ax)
    x0c, x1c = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    y0c, y1c = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    s = src.astype(np.float64)
    top = s[y0c][:, x0c] * (1 - ax)[None, :, None] + s[y0c][:, x1c] * ax[None, :, None]
    bot = s[y1c][:, x0c] * (1 - ax)[None, :, None] + s[y1c][:, x1c] * ax[None, :, None]
    return top * (1 - ay)[:, None, None] + bot * ay[:, None, None]


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 432, 368), (100, 80, 432, 368), (1280, 720, 432, 243), (33, 57, 64, 64),
                                         (432, 368, 431, 367), (7, 5, 20, 3)])
def test_resize_matches_float_bilinear(sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    src = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    got = loader.resize_linear_u8(src, dw, dh).astype(np.float64)
    ref = _float_bilinear(src, dw, dh)
    assert np.abs(got - ref).max() <= 1.0 + 1e-9


def test_resize_special_cases():
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert np.array_equal(loader.resize_linear_u8(src, 60, 40), src)                      # same size: copy
    half = loader.resize_linear_u8(src, 30, 20)                                           # exact 2x2: INTER_AREA
    s = src.astype(np.int32)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, box.astype(np.uint8))
    flat = np.full((17, 23, 3), 200, np.uint8)                                            # constants stay constant
    assert np.all(loader.resize_linear_u8(flat, 91, 64) == 200)
    up = loader.resize_linear_u8(src, 240, 160)                                           # x4: centre samples of every
    assert np.abs(up[2::4, 2::4].astype(int) - _float_bilinear(src, 240, 160)[2::4, 2::4]).max() <= 1


@pytest.mark.parametrize("sw,sh", [(640, 480), (480, 640), (1280, 720), (432, 368), (500, 500), (33, 900)])
def test_letterbox(sw, sh):
    dw, dh = 432, 368
    iw, ih = loader.letterbox_inner(sw, sh, dw, dh)
    h1, w2 = dw * (sh / sw), dh * (sw / sh)                                               # src/data.cpp:57-58
    assert (iw, ih) == ((dw, int(h1)) if h1 <= dh else (int(w2), dh))
    rng = np.random.default_rng(sw)
    src = rng.integers(1, 256, (sh, sw, 3), dtype=np.uint8)
    out = loader.letterbox_u8(src, dw, dh, bgcolor=(7, 8, 9))
    assert np.array_equal(out[:ih, :iw], loader.resize_linear_u8(src, iw, ih))
    assert np.all(out[ih:] == (7, 8, 9)) and np.all(out[:, iw:] == (7, 8, 9))


def test_resume_ratio():
    xy = np.array([[0.25, 0.5], [1.0, 1.0]], np.float32)
    wide = loader.resume_ratio(xy, (1280, 720), (432, 368))     # wide frame: the letterbox pads below -> y is stretched back
    assert np.allclose(wide[:, 0], xy[:, 0]) and np.allclose(wide[:, 1], xy[:, 1] * (368 * 1280) / (432 * 720))
    tall = loader.resume_ratio(xy, (480, 640), (432, 368))      # tall frame: x is stretched back
    assert np.allclose(tall[:, 1], xy[:, 1]) and np.allclose(tall[:, 0], xy[:, 0] * (432 * 640) / (368 * 480))
