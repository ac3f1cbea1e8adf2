"""Dispatch-gap / overlap analysis of a rocprofv3 kernel trace (run on the GPU box).

    python tools/trace_gaps.py <pipes>     # runs bench.py under rocprofv3 --kernel-trace and prints the summary

For every hardware queue: launches, busy time, and the distribution of the idle gap between the end of one kernel and the start
of the next on the same queue; for the device: the share of the traced interval with 0 / 1 / 2 / 3+ kernels in flight.
"""
import csv
import glob
import os
import subprocess
import sys

pipes = sys.argv[1] if len(sys.argv) > 1 else "4"
repo = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/prof_gap_" + pipes
subprocess.run(f"rm -rf {out}; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d {out} -- python {repo}/bench.py "
               f"--steps 40 --warmup 10 --pipes {pipes} --no-cpu-baseline --no-roofline --no-from-host > /dev/null 2>&1", shell=True)
path = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
# keep the steady-state tail: the last 60 % of the launches
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"pipes={pipes}: {len(ev)} launches over {(t1 - t0) / 1e3:.0f} us")
by_q = {}
for s, e, q, n in ev:
    by_q.setdefault(q, []).append((s, e, n))
for q, lst in sorted(by_q.items()):
    busy = sum(e - s for s, e, _ in lst)
    gaps = sorted(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
    if not gaps:
        continue
    pct = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3
    print(f"  queue {q}: {len(lst)} launches, busy {busy / (t1 - t0) * 100:.0f} %, gap us p10 {pct(.1):.1f} p50 {pct(.5):.1f} p90 {pct(.9):.1f} "
          f"mean {sum(gaps) / len(gaps) / 1e3:.1f}")
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
depth, last, hist = 0, t0, {}
for t, d in pts:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += d
    last = t
tot = sum(hist.values())
print("  kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / tot * 100:.0f} %" for k, v in sorted(hist.items())))
