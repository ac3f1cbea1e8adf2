#!/bin/bash
# tools/gpu_step.sh <name> - one gpurun call of this round's development loop (run ON the GPU box); everything lands in gpurun_out/<name>_*
name=${1:-s1}
cd $GRAFT_REPO_ROOT
out=gpurun_out
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_probe.hip -o /tmp/clock_probe && timeout 120 /tmp/clock_probe > $out/${name}_clock_probe.txt 2>&1
timeout 900 python -m pytest tests/test_engine_fp32_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -15 > $out/${name}_pytest_fp32.txt
timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/${name}_layers_f32.txt 2>&1
PYTHONPATH=. timeout 300 python tools/pifpaf_stress.py 4 64 > $out/${name}_pifpaf_stress64.txt 2>&1
timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32s > $out/${name}_layers_f32s.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --extra 1/f16,1/f32s > $out/${name}_bench.json 2> $out/${name}_bench.err
cp bench_detail.json $out/${name}_bench_detail.json 2>/dev/null
tail -c 300 $out/${name}_bench.json
tail -5 $out/${name}_pytest_fp32.txt
for f in $out/${name}_layers_f32.txt $out/${name}_layers_f32_nodirect.txt $out/${name}_layers_f32s.txt; do tail -n 3 $f; done; tail -n 3 $out/${name}_pifpaf_stress64.txt
