#!/bin/bash
# tools/gpu_step.sh <name> - one gpurun call of this round's development loop (run ON the GPU box); everything lands in gpurun_out/<name>_*
name=${1:-s1}
cd $GRAFT_REPO_ROOT
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests/test_engine_fp32_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -25 > $out/${name}_pytest_fp32.txt
for v in base "nofuse HP_NO_FUSE32=1" "mw4 HP_DIRECT_MW_MAX=4"; do
  set -- $v
  env $2 timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32s > $out/${name}_layers_f32s_$1.txt 2>&1
  env $2 timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/${name}_layers_f32_$1.txt 2>&1
done
timeout 200 python tools/direct_timeline.py f32s 2> $out/${name}_timeline_f32s.txt >/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --extra 1/f16,1/f32s > $out/${name}_bench.json 2> $out/${name}_bench.err
cp bench_detail.json $out/${name}_bench_detail.json 2>/dev/null
tail -n 8 $out/${name}_pytest_fp32.txt | cut -c1-300
for f in $out/${name}_layers_*.txt; do echo $f; tail -n 3 $f; done
grep -E "layer (13|15) " $out/${name}_timeline_f32s.txt | tail -n 2 | cut -c1-400
python - <<'PY'
import json
d=json.load(open('bench_detail.json'))
for w in [d['headline']]+list(d['workloads'].values()): print(w['key'], w['value'], 'eng', w.get('engine_only_ms_per_step'), w['roofline']['kernel_symbol'], w['roofline']['avg_launch_us'], w['roofline']['frac'], w['clocks']['sclk_mhz_mean'])
PY
