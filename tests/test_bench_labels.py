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
    # the fp32 engine (round 5): Winograd, the fused heads, both epilogue forms of conv32_kernel, the direct kernel of the narrow heads' fall-back
    (35003004, "_config1_fp32"), (37051219, "_config1_fp32"), (37051238, "_config1_fp32"), (39032064, "_config1_fp32"), (32464064, "_config1_fp32"),
    (35003004, "_config2_fp32"), (32064128, "_config2_fp32"), (32128128, "_config3_fp32"), (32128128, "_config4_fp32"), (39064064, "_config3_fp32"), (39064064, "_config4_fp32"),  # (round 6: conv32_wk_kernel)
    (33003002, "_config1_fp32s"), (33001008, "_config1_fp32s"), (33101008, "_config1_fp32s"),
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


def _detail():
    """the newest committed full record of a default `python bench.py` run (profiles/r*_bench_detail.json, written by bench.py next to
    the ONE line it prints)"""
    paths = [p for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_detail.json"))) if os.path.basename(p) >= "r06"]
    if not paths:
        pytest.skip("no committed bench detail record of the round-6 form (value = the host-fed leg)")
    return json.load(open(paths[-1]))


def _bench_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_final.json")))
    paths = [p for p in paths if os.path.basename(p) >= "r06"]
    if not paths:
        pytest.skip("no committed bench line of the compact form")
    return open(paths[-1]).read().strip().splitlines()[-1]


def test_line_is_small_enough_for_the_driver_and_regenerates_from_the_detail_record():
    """BENCH_r04.json had parsed = null because the line was 20.9 KB.  The printed line is a pure function of the
    detail record, <= 4096 bytes, and carries the contract's keys."""
    d = _detail()
    line = bench.compact_line(d)
    assert len(line) <= bench.LINE_LIMIT, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "detail"):
        assert k in out, k
    assert "dropped_for_size" not in out
    assert out["config"]["workload"] and "model" not in out["config"]
    # round 6: `value` IS the SURVEY 8(d) number - frames from pinned host memory, one H2D per batch, the parser on the
    # network's own maps - and round 5's headline is kept beside it under its own name
    assert "frames from pinned host memory" in out["config"]["workload"] and out["config"]["parser_input"] == "the network's own heat-maps"
    assert out["value"] == d["headline"]["h2d_inclusive"]["value"] == out["value_h2d_inclusive"]
    assert out["value_resident_injected"] == d["headline"]["value_resident_injected"] > 0 and out["fps_dnn_output"] > 0
    assert "mfma_busy" in out["roofline"] and "operator_api_fps" in out and "single_pipe_fps" in out
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["steps_timed"] % out["steps"] == 0 and out["timed_region_s"] >= 0.45
    committed = json.loads(_bench_line())
    assert committed["value"] == out["value"] and committed["roofline"] == out["roofline"], "profiles/r*_bench_final.json is not the line of r*_bench_detail.json"


def test_worst_case_line_still_fits():
    """every optional field at its longest: all ten workloads present, long kernel symbols and samples"""
    d = _detail()
    d = json.loads(json.dumps(d))
    long_sym = "some_kernel_name_with_template_arguments<128,128,2,2,true,false,17>" * 3
    for w in [d["headline"]] + list(d["workloads"].values()):
        if w.get("roofline"):
            w["roofline"]["kernel_symbol"] = long_sym
            w["roofline"]["committed_profile"]["source"] = "r05_kernel_stats_config4_fp32.csv"
        if w.get("cpu_baseline"):
            w["cpu_baseline"]["sample"] = "x" * 1000
    for i in (0, 2, 3, 4):
        for dt in ("f32", "f16"):
            d["workloads"].setdefault(f"configs[{i}]/{dt}", dict(next(iter(d["workloads"].values()))))
    line = bench.compact_line(d)
    assert len(line) <= bench.LINE_LIMIT
    assert json.loads(line)["value"] == d["headline"]["value"]


@pytest.mark.parametrize("argv,dtype,peak", [([], "f32", 157.3), (["--dtype", "f16"], "f16", 2500.0), (["--config", "5"], "f32", 157.3),
                                             (["--config", "3", "--dtype", "f16"], "f16", 2500.0), (["--config", "4"], "f32", 157.3)])
def test_dtype_label_follows_the_configuration(argv, dtype, peak):
    """the label was a constant.  The headline defaults to the mirror's default data_type (kFLOAT) and the label,
    the profile tag and the roofline's peak are derived from the configuration."""
    a = bench.parse_args(argv)
    c = bench.config(a.config, a.dtype)
    assert c["dtype"] == dtype and bench.DTYPE_LABEL[c["dtype"]] == dtype and c["key"].endswith("/" + dtype)
    assert ("_fp32" in bench.profile_tag(c)) == (dtype == "f32")
    assert (bench.PEAK_F32_TFLOPS if c["dtype"] == "f32" else bench.PEAK_F16_TFLOPS) == peak
    d = _detail()
    for w in [d["headline"]] + list(d["workloads"].values()):
        assert w["dtype"] == w["key"].split("/")[1]
        if w.get("roofline"):
            assert abs(w["roofline"]["mfma_peak_tflops"] - bench.PEAKS[w["dtype"]]) < 1e-9


def test_committed_profile_quotes_the_csv_row_it_names():
    """the committed line quoted 131.57 us for a kernel whose committed CSV row said 125.26.  For every workload of
    the committed record: committed_profile.avg_launch_us IS AverageNs / 1000 of the one row its source file has for that kernel."""
    d = _detail()
    n = 0
    for w in [d["headline"]] + list(d["workloads"].values()):
        r = w.get("roofline")
        if not r or not r["committed_profile"]["source"] or r["committed_profile"]["avg_launch_us"] is None:
            continue
        path = os.path.join(ROOT, "profiles", r["committed_profile"]["source"])
        want = r["kernel_symbol"].replace(" ", "")
        with open(path, newline="") as f:
            rows = [row for row in csv.DictReader(f) if want in row["Name"].replace(" ", "")]
        assert len(rows) == 1, (w["key"], want)
        assert abs(float(rows[0]["AverageNs"]) / 1e3 - r["committed_profile"]["avg_launch_us"]) < 0.006, (w["key"], rows[0]["AverageNs"])
        n += 1
    assert n >= 1


def test_dominant_kernel_of_the_headline_has_both_clocks_and_traffic():
    """The committed record carries the live per-launch time, the committed rocprofv3 average of the same kernel (under its own
    object, with its source file) and its PMC traffic, and the two clocks agree within the tracer's overhead."""
    d = _detail()
    h = d["headline"]
    r = h["roofline"]
    cp = r["committed_profile"]
    assert h["key"] == "configs[1]/f32" and r["mfma_peak_tflops"] == 157.3
    assert cp["source"] and cp["avg_launch_us"] is not None and cp["frac_mfma"] is not None and r["traffic"] is not None
    assert 0.8 < cp["avg_launch_us"] / r["avg_launch_us"] < 1.35
    assert h["cpu_baseline"]["kind"] in ("reference", "port") and h["cpu_baseline"]["value"] > 0
    assert set(d["workloads"]) == ({f"configs[{i}]/{dt}" for i in range(5) for dt in ("f32", "f16")} | {"configs[1]/f32s"}) - {"configs[1]/f32"}
    f16 = d["workloads"]["configs[1]/f16"]
    assert f16["roofline"]["mfma_peak_tflops"] == 2500.0 and f16["value"] > h["value"] > 0


def _check_roofline(r):
    ridge, peak = r["ridge_flop_per_byte"], r["mfma_peak_tflops"]
    assert (peak in (2500.0, 157.3) or abs(peak - 2500.0 / 3) < 1e-6) and abs(ridge - peak * 1e12 / 8e12) < 0.01
    x = r["flops_per_launch"] / r["algorithmic_bytes_per_launch"]
    assert abs(x - r["intensity_flop_per_byte"]) <= 0.01 * x + 0.1
    assert r["bound"] == ("mfma" if r["intensity_flop_per_byte"] >= ridge else "hbm")
    assert r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s") and r["peak"] == (peak if r["bound"] == "mfma" else 8000.0)
    assert abs(r["frac"] - (r["frac_mfma"] if r["bound"] == "mfma" else r["frac_hbm"])) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    t = r["avg_launch_us"] * 1e-6
    assert abs(r["frac_mfma"] - r["flops_per_launch"] / t / (peak * 1e12)) < 5e-3
    assert abs(r["frac_hbm"] - r["algorithmic_bytes_per_launch"] / t / 8e12) < 5e-3
    # (a Winograd kernel's `frac` is ALGORITHMIC, direct-form flops over time: it may pass 1 - that is the point of the transform; what the
    # pipe executes, 16 / 36 of them, may not)
    assert 0 < r["mfma_frac_ceiling_at_hbm_peak"] <= 1.0 and 0 < r["frac"] < (2.25 if "frac_mfma_issued" in r else 1)
    assert r.get("frac_mfma_issued", 0.5) < 1


def test_every_roofline_names_its_binding_roof():
    d = _detail()
    _check_roofline(d["headline"]["roofline"])
    for w in d["workloads"].values():
        _check_roofline(w["roofline"])


def test_pcie_inclusive_rate_and_fallback_counts_are_in_the_line():
    d = _detail()
    out = json.loads(bench.compact_line(d))
    assert out["value_h2d_inclusive"] == d["headline"]["h2d_inclusive"]["value"] and 0 < out["value_h2d_inclusive"] <= out["value"] * 1.05
    assert out["device_declined_frames"] == 0 and out["capacity_truncations"] == 0
    for key, w in d["workloads"].items():
        assert w["device_declined_frames"] >= 0 and w["capacity_truncations"] == 0 and w["frames_parsed_for_these_counts"] > 0
        assert w["device_declined_frames"] <= 0.01 * w["frames_parsed_for_these_counts"]
        assert out["workloads"][key]["value"] == w["value"]
    assert out["device_declined_frames_all"] == sum(w["device_declined_frames"] for w in d["workloads"].values()) + d["headline"]["device_declined_frames"]


def test_clock_samples_are_recorded_next_to_the_fractions():
    """the sustained shader clock of every timed region is a measured number in the record."""
    d = _detail()
    ck = d["headline"].get("clocks")
    assert ck and ck["samples"] >= 5 and ck["sclk_mhz_mean"] and 500 < ck["sclk_mhz_min"] <= ck["sclk_mhz_mean"] <= 2600


def test_clock_sampler_reads_its_sources(tmp_path):
    """ClockSampler against a fake sysfs tree: hwmon freq1_input (Hz) first, pp_dpm_sclk's starred level otherwise."""
    import time
    s = bench.ClockSampler(device_index=-1)
    assert s.freq_file is None and s._read_mhz() is None
    f = tmp_path / "freq1_input"
    f.write_text("2100000000\n")
    dpm = tmp_path / "pp_dpm_sclk"
    dpm.write_text("0: 132Mhz\n1: 1650Mhz *\n2: 2400Mhz\n")
    s.freq_file, s.dpm_file = str(f), str(dpm)
    assert s._read_mhz() == 2100.0
    s.freq_file = None
    assert s._read_mhz() == 1650.0
    s.freq_file = str(f)
    pw = tmp_path / "power1_average"
    pw.write_text("750000000\n")
    s.power_file = str(pw)
    with s:
        time.sleep(0.1)
    out = s.summary()
    assert out["samples"] >= 2 and out["sclk_mhz_mean"] == 2100.0 and out["power_w_mean"] == 750.0 and out["source"] == "hwmon freq1_input"
