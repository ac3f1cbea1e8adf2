cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_engine_gpu.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25) > gpurun_out/t_engine.log 2>&1
(timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu -k "config2" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8) > gpurun_out/t_base.log 2>&1
timeout 300 python tools/profile_layers.py openpose_vgg19 768 432 16 > gpurun_out/layers_pair_config2.txt 2>&1
timeout 300 python bench.py --config 2 --extra= --no-cpu-baseline --no-from-host > gpurun_out/bench_c2_pair.json 2>/dev/null
HP_NO_PAIR_CONV=1 timeout 300 python bench.py --config 2 --extra= --no-cpu-baseline --no-from-host > gpurun_out/bench_c2_nopair.json 2>/dev/null
tail -n 6 gpurun_out/t_engine.log gpurun_out/t_base.log; grep "total\|two streams" gpurun_out/layers_pair_config2.txt
python -c "
import json
for f in ('pair','nopair'):
    d=json.loads(open('gpurun_out/bench_c2_%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('engine_only_ms_per_step'))
"
