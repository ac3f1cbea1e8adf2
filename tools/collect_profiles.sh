#!/bin/bash
# tools/collect_profiles.sh <tag> [config] — run ON THE GPU BOX (gpurun): rocprofv3 kernel stats + PMC traffic of the bench, one pipe so
# that kernels do not overlap (the per-launch durations then compare with hp_engine_profile / bench.py's roofline).
# Outputs (merged back by gpurun): gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.json
set -u
tag=${1:-r01}
cfg=${2:-1}
dt=${3:-f16}   # engine precision: f16 (data_type::kHALF) or f32 (data_type::kFLOAT); "5" is the old spelling of "1 f32"
[ "$cfg" == "5" ] && cfg=1 && dt=f32
sfx=""
[ "$cfg" != "1" ] && sfx="_config$cfg"
[ "$dt" == "f32" ] && sfx="_config${cfg}_fp32"
[ "$dt" == "f32s" ] && sfx="_config${cfg}_fp32s"
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps=12
[ "$cfg" != "1" ] && [ "$cfg" != "0" ] && steps=3
[ "$dt" != "f16" ] && [ "$cfg" != "1" ] && [ "$cfg" != "0" ] && steps=2
cmd="python $repo/bench.py --config $cfg --dtype $dt --no-clocks --min-seconds 0 --extra= --steps $steps --warmup 2 --pipes 1 --no-cpu-baseline --no-roofline --no-from-host --no-dnn-output --no-operator-api"
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- $cmd > /dev/null 2>&1
cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats${sfx}.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- $cmd > /dev/null 2>&1
python - "$out/${tag}_pmc_traffic${sfx}.json" $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) <<'PY'
import csv, json, sys
out, ff, fw = sys.argv[1:4]
agg = {}
for path, ctr in ((ff, "FETCH_SIZE"), (fw, "WRITE_SIZE")):
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != ctr:
            continue
        d = agg.setdefault(r["Kernel_Name"], {})
        d[ctr] = d.get(ctr, 0.0) + float(r["Counter_Value"])
        d["launches_" + ctr] = d.get("launches_" + ctr, 0) + 1
res = {}
for k, d in agg.items():
    e = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        if ctr in d:
            e[ctr] = d[ctr] / d["launches_" + ctr]
            e["launches_" + ctr] = d["launches_" + ctr]
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    res[k] = e
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 12 --warmup 3 --pipes 1`; "
                   "values are KiB per launch averaged over all launches of the kernel; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                   "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 16-B/lane streams at 1/2)", "kernels": res}, open(out, "w"), indent=1)
print("pmc kernels:", len(res))
PY
# third counter pass: where the wave cycles go (SQ) and how busy the matrix pipe is
rm -rf /tmp/prof_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -- $cmd > /dev/null 2>&1
python - "$out/${tag}_pmc_sq${sfx}.json" $(find /tmp/prof_sq -name "*counter_collection.csv" | head -1) "$out/${tag}_kernel_stats${sfx}.csv" <<'PY'
import csv, json, sys
out, path, stats = sys.argv[1:4]
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats))}
agg = {}
for r in csv.DictReader(open(path)):
    d = agg.setdefault(r["Kernel_Name"], {})
    c = r["Counter_Name"]
    d[c] = d.get(c, 0.0) + float(r["Counter_Value"])
    d["n_" + c] = d.get("n_" + c, 0) + 1
res = {}
for k, d in agg.items():
    e = {c: d[c] / d["n_" + c] for c in d if not c.startswith("n_")}
    e["launches"] = max(d[c] for c in d if c.startswith("n_"))
    # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the 1024 SIMDs (32 per 32x32x16 f16 MFMA); the duration is the
    # kernel's average in the un-instrumented kernel-trace pass (counter passes serialise and slow the kernels), at the 2.4 GHz peak clock
    if k in dur:
        e["avg_duration_us"] = dur[k] / 1e3
        e["mfma_busy_frac"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * dur[k] * 2.4)
    if e.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_share_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
    res[k] = e
json.dump({"note": "rocprofv3 --pmc (one pass, SQ + GRBM) over `python bench.py --steps 12 --warmup 3 --pipes 1 ...`; per-launch averages; "
                   "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * avg_duration * 2.4 GHz), duration from the kernel-trace pass (derived here: ROCm 7.2 has no gfx950 derived-metric section); "
                   "SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing (quad-cycle units, MI355X_MICROARCH.md)",
           "kernels": res}, open(out, "w"), indent=1)
print("sq kernels:", len(res))
PY
head -12 $out/${tag}_kernel_stats${sfx}.csv | cut -c1-150
