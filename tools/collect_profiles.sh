#!/bin/bash
# tools/collect_profiles.sh <tag> — run ON THE GPU BOX (gpurun): rocprofv3 kernel stats + PMC traffic of the bench, one pipe so
# that kernels do not overlap (the per-launch durations then compare with hp_engine_profile / bench.py's roofline).
# Outputs (merged back by gpurun): gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.json
set -u
tag=${1:-r01}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 12 --warmup 3 --pipes 1 --no-cpu-baseline --no-roofline --no-from-host --no-dnn-output"
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- $cmd > /dev/null 2>&1
cp $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- $cmd > /dev/null 2>&1
python - "$out/${tag}_pmc_traffic.json" $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) <<'PY'
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
head -12 $out/${tag}_kernel_stats.csv | cut -c1-150
