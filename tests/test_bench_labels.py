"""CPU: bench.py's roofline bookkeeping - every profile tile code of the kernels the engine can schedule maps to a kernel-name key that is
FOUND - as exactly one row - in the newest committed rocprofv3 summaries (profiles/r*_kernel_stats*.csv, r*_pmc_traffic*.json), so
`committed_profile` / `traffic` cannot silently come back null (or name another template instance) because a kernel was renamed; and the
committed bench line carries the keys VERDICT r3 item 4 asks for."""
import csv
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _names(tag):
    out = set()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_stats{tag}.csv")), reverse=True)[:1]:
        with open(path, newline="") as f:
            out |= {row["Name"].replace(" ", "") for row in csv.DictReader(f)}
    return out


# (tile code as hp_engine_profile reports it, configuration whose committed summary must contain the kernel)
CASES = [
    (4000005, ""), (4000006, ""), (4000004, ""), (4000003, ""), (4000020, ""), (4000001, ""), (7000013, ""), (7000001, ""), (7000002, ""), (7000003, ""),
    (5064192, ""), (6000128, ""), (5201002, ""), (5100192, ""),
    (6128049, "_config2"), (6256009, "_config2"), (6192049, "_config2"),
    (5202002, "_config3"), (9001311, "_config3"), (9001011, "_config3"), (9001021, "_config3"), (9002020, "_config3"), (9002021, "_config3"),
    (9002041, "_config3"), (6512009, "_config3"),
    (5202002, "_config4"), (9001311, "_config4"), (9002021, "_config4"), (128128, "_config4"),
]


@pytest.mark.parametrize("tile,tag", CASES)
def test_tile_code_maps_to_a_profiled_kernel(tile, tag):
    names = _names(tag)
    if not names:
        pytest.skip("no committed kernel statistics for this configuration")
    key, label = bench.kernel_label(tile)
    hits = [n for n in names if key.replace(" ", "") in n]
    if tile < 4000000:  # the generic implicit GEMM: the tile code does not carry its K-step / epilogue template arguments, several
        assert label and hits, (tile, key)  # instances may be in a trace - and bench.py then (rightly) quotes no committed duration
        return
    assert label and len(hits) == 1, (tile, key, hits, sorted(names)[:5])
    us, src = bench.rocprof_avg_us(key, tag)
    assert us is not None and us > 0 and src


def test_ambiguous_or_unknown_kernel_names_give_no_number():
    us, _ = bench.rocprof_avg_us("conv", "")               # names many kernels
    assert us is None
    us, _ = bench.rocprof_avg_us("no_such_kernel<1,2>", "")
    assert us is None


def _bench_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_final.json")))
    if not paths:
        pytest.skip("no committed bench line")
    return json.loads(open(paths[-1]).read().strip().splitlines()[-1])


def test_dominant_kernel_of_the_headline_has_both_clocks_and_traffic():
    """The committed bench line carries the live per-launch time, the committed rocprofv3 average of the same kernel (under its own
    object, with its source file) and its PMC traffic, and the two clocks agree within the tracer's overhead."""
    d = _bench_line()
    r = d["roofline"]
    cp = r["committed_profile"]
    assert cp["source"] and cp["avg_launch_us"] is not None and cp["frac_mfma"] is not None and r["traffic"] is not None
    assert "frac_rocprof" not in r and "avg_launch_us_rocprof" not in r
    assert 0.8 < cp["avg_launch_us"] / r["avg_launch_us"] < 1.35
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert set(d["workloads"]) == {"configs[0]", "configs[2]", "configs[3]", "configs[4]", "configs[1]/fp32"}
    f32 = d["workloads"]["configs[1]/fp32"]
    assert f32["dtype"].startswith("f32") and f32["roofline"]["mfma_peak_tflops"] == 157.3 and f32["value"] > 0


def _check_roofline(r):
    ridge, peak = r["ridge_flop_per_byte"], r["mfma_peak_tflops"]
    assert peak in (2500.0, 157.3) and abs(ridge - peak * 1e12 / 8e12) < 0.01
    x = r["flops_per_launch"] / r["algorithmic_bytes_per_launch"]
    assert abs(x - r["intensity_flop_per_byte"]) <= 0.01 * x + 0.1
    assert r["bound"] == ("mfma" if r["intensity_flop_per_byte"] >= ridge else "hbm")
    assert r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s") and r["peak"] == (peak if r["bound"] == "mfma" else 8000.0)
    assert abs(r["frac"] - (r["frac_mfma"] if r["bound"] == "mfma" else r["frac_hbm"])) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    t = r["avg_launch_us"] * 1e-6
    assert abs(r["frac_mfma"] - r["flops_per_launch"] / t / (peak * 1e12)) < 5e-3
    assert abs(r["frac_hbm"] - r["algorithmic_bytes_per_launch"] / t / 8e12) < 5e-3
    assert 0 < r["mfma_frac_ceiling_at_hbm_peak"] <= 1.0


def test_every_roofline_names_its_binding_roof():
    d = _bench_line()
    _check_roofline(d["roofline"])
    for w in d["workloads"].values():
        _check_roofline(w["roofline"])


def test_pcie_inclusive_rate_and_fallback_counts_are_top_level():
    d = _bench_line()
    assert d["value_h2d_inclusive"] == d["h2d_inclusive"]["value"] and 0 < d["value_h2d_inclusive"] <= d["value"] * 1.05
    assert d["device_declined_frames"] == 0 and d["capacity_truncations"] == 0
    for w in d["workloads"].values():
        assert w["device_declined_frames"] >= 0 and w["capacity_truncations"] == 0 and w["frames_parsed_for_these_counts"] > 0
        assert w["device_declined_frames"] <= 0.01 * w["frames_parsed_for_these_counts"]


def test_headline_names_the_runner_up_kernel():
    """On configs[1] the 512-output separable block and the 128-channel chain are within a few percent of each other and swap places
    between runs: the line carries both, each with its own roof."""
    r = _bench_line()["roofline"]
    ru = r["runner_up"]
    assert ru["kernel"] and ru["kernel"] != r["kernel"] and ru["launches_per_step"] >= 1 and ru["avg_launch_us"] > 0
    assert ru["bound"] == ("mfma" if ru["intensity_flop_per_byte"] >= r["ridge_flop_per_byte"] else "hbm")
    assert 0 < ru["frac_mfma"] < 1 and 0 < ru["frac_hbm"] < 1
    assert 0 < ru["share_of_serial_step"] <= r["share_of_serial_step"] < 1
    both = (r["kernel"] + ru["kernel"])
    assert "separable block" in both and "conv_chain_kernel" in both
