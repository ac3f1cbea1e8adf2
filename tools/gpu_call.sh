cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driverflags.json 2>> gpurun_out/bench.err
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30 > gpurun_out/smi.txt
python -c "
import json
for f in ('r04_bench_final','r04_bench_driverflags'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['single_pipe_fps'], d['engine_only_ms_per_step'], {k:v['value'] for k,v in d['workloads'].items()})
"
