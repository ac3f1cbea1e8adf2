cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driverflags.json 2>> gpurun_out/bench.err
bash tools/collect_profiles.sh r04 5 > gpurun_out/collect_5.log 2>&1
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > gpurun_out/r04_layer_times_config1_fp32.txt 2>/dev/null
(time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6) > gpurun_out/t_all.log 2>&1
python -c "
import json
for f in ('r04_bench_final','r04_bench_driverflags'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['single_pipe_fps'], d['engine_only_ms_per_step'], d['value_h2d_inclusive'], {k:v['value'] for k,v in d['workloads'].items()})
"
tail -n 4 gpurun_out/t_all.log
