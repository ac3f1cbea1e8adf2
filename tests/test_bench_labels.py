"""CPU: bench.py's roofline bookkeeping - every profile tile code of the kernels the engine can schedule maps to a kernel-name key that is
FOUND in the committed rocprofv3 summaries (profiles/r*_kernel_stats*.csv, r*_pmc_traffic*.json), so `frac_rocprof` / `traffic` cannot
silently come back null because a kernel was renamed."""
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
    (4000005, ""), (4000006, ""), (4000004, ""), (4000003, ""), (4000007, ""), (7000013, ""), (7000001, ""), (7000002, ""), (7000003, ""),
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
    assert label and any(key.replace(" ", "") in n for n in names), (tile, key, sorted(names)[:5])


def test_dominant_kernel_of_the_headline_has_both_clocks_and_traffic():
    """The committed bench line carries the live per-launch time, the rocprofv3 average of the same kernel and its PMC traffic, and the
    two clocks agree within the tracer's overhead."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_final.json")))
    if not paths:
        pytest.skip("no committed bench line")
    d = json.loads(open(paths[-1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    assert r["frac_rocprof"] is not None and r["traffic"] is not None and r["avg_launch_us_rocprof"] is not None
    assert 0.8 < r["avg_launch_us_rocprof"] / r["avg_launch_us"] < 1.35
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert set(d["workloads"]) == {"configs[0]", "configs[2]", "configs[3]", "configs[4]"}
