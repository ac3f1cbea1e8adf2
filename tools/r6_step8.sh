#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
python tools/arena_probe.py 2>&1 | tee $out/s8_arena.txt
timeout 1500 python -m pytest tests/test_engine_fp32_gpu.py tests/test_pipeline_gpu.py tests/test_onnx_import_gpu.py -q -m gpu -x 2>&1 | tail -12 > $out/s8_pytest.txt
cut -c1-300 $out/s8_pytest.txt
HP_NO_ARENA=1 timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s8_layers_f32_noarena.txt 2>&1
timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s8_layers_f32_arena.txt 2>&1
tail -n 3 $out/s8_layers_f32_noarena.txt $out/s8_layers_f32_arena.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/parser_dnn_probe.py 2>&1 | tail -6
rocprofv3 --kernel-trace --stats -d $out/prof_parser -o p -- python tools/parser_dnn_probe.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_parser/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "paf_" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
