cd /tmp && export TMPDIR=/tmp
cat > /tmp/vggrun.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from hyperpose_amd import _lib
from hyperpose_amd.engine import Engine, Model
_lib.init(0)
m = Model("openpose_vgg19", 768, 432)
eng = Engine.from_model(m, m.init_weights(1), max_batch=16)
eng.profile(16, 2)
PY
rm -rf /tmp/vp /tmp/vs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs -- python /tmp/vggrun.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/vp -- python /tmp/vggrun.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
dur = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(glob.glob("/tmp/vs/**/*kernel_stats.csv", recursive=True)[0]))}
agg = {}
for r in csv.DictReader(open(glob.glob("/tmp/vp/**/*counter_collection.csv", recursive=True)[0])):
    d = agg.setdefault(r["Kernel_Name"], {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["n_" + r["Counter_Name"]] = d.get("n_" + r["Counter_Name"], 0) + 1
for k, d in sorted(agg.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[0] * dur.get(kv[0], (0, 0))[1])[:6]:
    e = {c: d[c] / d["n_" + c] for c in d if not c.startswith("n_")}
    t, n = dur[k]
    print(k[:60], f"avg {t/1e3:.1f} us x{n}", "mfma_busy %.3f" % (e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * t * 2.4)),
          "wait_any %.2f wait_inst %.2f active %.2f lds_stall %.2f" % tuple(e[c] / e["SQ_WAVE_CYCLES"] for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")))
PY
