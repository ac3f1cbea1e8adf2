"""Tolerance envelope for the two UNPINNED third-party calls behind the PAF parser (cv::resize INTER_AREA when up-scaling and
cv::GaussianBlur, OpenCV 4.4.0 - not in this image).

oracle/paf_oracle.cpp restates their published algorithm in scalar, non-fused fp32 (variant 0 = the parity oracle the GPU parser
matches bit for bit).  A real OpenCV build may round the same algorithm differently: its AVX2 kernels evaluate a*b + c*d and the
filter taps with fused multiply-adds (variant 1: v_muladd forms of VResizeLinearVec_32f / RowVec_32f / SymmColumnVec_32f), and a
compiler contracting the plain C++ paths may fuse the other product (variant 2).  Peak detection is an exact `smoothed == pooled`
test (reference src/post_process.hpp:177), so a last-bit difference COULD create or delete a peak.  This test bounds that: over
the golden frames, crowded frames and clutter frames, every variant must give the same humans at the same key-point positions
(north star: <= 1e-3 px; positions are integer pixels / resolution, so equal means bit-equal) with scores within 1e-5; any peak
whose `==` test flips between variants is reported."""
import numpy as np
import pytest

from hyperpose_amd import synth
from oracle import loader


def _frames():
    rng = synth.rng_for(1, salt=41)
    conf, paf, _ = synth.paf_maps(rng, 12, 46, 54, people=(1, 2, 4, 8, 16, 3, 5, 6, 12, 10, 7, 9))
    yield "config1 46x54", conf, paf
    conf, paf, _ = synth.paf_maps(synth.rng_for(2, salt=41), 4, 54, 96, people=(2, 6, 11, 16))
    yield "config2 54x96", conf, paf
    # clutter: strong noise -> hundreds of marginal peaks whose plateau ties are the most rounding-sensitive thing there is
    conf, paf, _ = synth.paf_maps(synth.rng_for(1, salt=42), 4, 46, 54, people=(3, 5, 2, 8), noise=0.08)
    yield "clutter", conf, paf
    # plateaus: quantised maps make exact ties between neighbouring up-sampled values common
    conf, paf, _ = synth.paf_maps(synth.rng_for(1, salt=43), 4, 46, 54, people=(4, 6, 2, 9))
    yield "quantised", np.round(conf * 64) / 64, np.round(paf * 64) / 64


def test_fma_variants_of_the_opencv_calls_leave_the_humans_unchanged():
    flips, frames, humans = [], 0, 0
    worst_score = 0.0
    for name, conf, paf in _frames():
        for f in range(conf.shape[0]):
            h0, p0, c0 = loader.paf_process(conf[f], paf[f])
            frames += 1
            humans += len(h0)
            for v in (1, 2):
                hv, pv, cv = loader.paf_process(conf[f], paf[f], variant=v)
                if len(pv) != len(p0) or not np.array_equal(pv[["part_id", "x", "y"]], p0[["part_id", "x", "y"]]):
                    a = {(int(k["part_id"]), int(k["x"]), int(k["y"])) for k in p0}
                    b = {(int(k["part_id"]), int(k["x"]), int(k["y"])) for k in pv}
                    flips.append((name, f, v, sorted(a ^ b)))
                assert len(hv) == len(h0), f"{name} frame {f} variant {v}: {len(hv)} humans vs {len(h0)}"
                # key-point positions: identical (<= 1e-3 px asks for less); presence identical
                assert np.array_equal(hv["parts"]["has_value"], h0["parts"]["has_value"]), (name, f, v)
                dx = np.abs(hv["parts"]["x"] - h0["parts"]["x"]).max(initial=0) * 184 if len(h0) else 0.0
                dy = np.abs(hv["parts"]["y"] - h0["parts"]["y"]).max(initial=0) * 216 if len(h0) else 0.0
                assert max(dx, dy) <= 1e-3, f"{name} frame {f} variant {v}: key-point moved by {max(dx, dy)} px"
                if len(h0):
                    worst_score = max(worst_score, float(np.abs(hv["parts"]["score"] - h0["parts"]["score"]).max()),
                                      float(np.abs(hv["score"] - h0["score"]).max()))
    assert frames == 24 and humans >= 100
    assert worst_score <= 1e-5, worst_score
    # peaks whose exact-equality test flipped: none on these frames; if a future generator produces one it is listed here and has
    # to be shown harmless (it was not part of any human, or the human survived) by the assertions above
    print(f"envelope: {frames} frames, {humans} humans, worst score delta {worst_score:.3g}, peak flips: {flips}")
    assert len(flips) <= 2, flips


def test_variants_really_differ_in_the_last_bits():
    """The envelope is not vacuous: the FMA variants do change low-order bits of the smoothed maps."""
    conf, _, _ = synth.paf_maps(synth.rng_for(1, salt=44), 1, 46, 54, people=(5,))
    up0 = loader.resize_area(conf[0], 216, 184)
    import ctypes as C
    outs = []
    for v in (0, 1, 2):
        L = loader.lib(variant=v)
        up = np.zeros((19, 216, 184), np.float32)
        L.oracle_resize_area(conf[0].ctypes.data_as(C.POINTER(C.c_float)), 19, 46, 54, up.ctypes.data_as(C.POINTER(C.c_float)), 216, 184)
        sm = np.zeros_like(up)
        L.oracle_smooth(up.ctypes.data_as(C.POINTER(C.c_float)), 19, 216, 184, 17, sm.ctypes.data_as(C.POINTER(C.c_float)))
        outs.append((up, sm))
    assert np.array_equal(outs[0][0], up0)
    assert not np.array_equal(outs[0][1], outs[1][1]) and np.abs(outs[0][1] - outs[1][1]).max() < 1e-6
    assert np.abs(outs[0][1] - outs[2][1]).max() < 1e-6
