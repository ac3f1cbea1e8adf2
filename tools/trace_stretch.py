import csv, glob, os, subprocess, sys, collections
repo = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(pipes):
    out = f"/tmp/prof_st_{pipes}"
    subprocess.run(f"rm -rf {out}; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d {out} -- python {repo}/bench.py --steps 60 --warmup 10 --pipes {pipes} --no-cpu-baseline --no-roofline --no-from-host --no-dnn-output > /dev/null 2>&1", shell=True)
    rows = list(csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0])))
    rows = rows[len(rows) // 3:]
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: (sum(v) / len(v) / 1e3, len(v)) for k, v in d.items()}
a, b = run(1), run(4)
tot1 = tot4 = 0
print(f"{'kernel':60} {'1 pipe us':>10} {'4 pipes us':>10} {'stretch':>8} {'share of 4-pipe busy':>10}")
busy = sum(v[0] * v[1] for v in b.values())
for k, (t1, n1) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k in b:
        t4, n4 = b[k]
        print(f"{k[:60]:60} {t1:10.1f} {t4:10.1f} {t4 / t1:8.2f} {t4 * n4 / busy * 100:9.1f}%")
