cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_engine_fp32_gpu.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6) > gpurun_out/t_fp32.log 2>&1
timeout 200 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > gpurun_out/layers_fp32_b.txt 2>&1
HP_F32_HALF_TILES=1 timeout 200 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > gpurun_out/layers_fp32_c.txt 2>&1
timeout 200 python bench.py --config 5 --extra= --no-cpu-baseline --no-from-host --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 bench', d['value'], d['ms_per_step'])"
tail -n 3 gpurun_out/t_fp32.log; grep " dw \|total\|two streams" gpurun_out/layers_fp32_b.txt | cut -c1-100; grep "512  512 1\|128  512 1\|total\|two streams" gpurun_out/layers_fp32_c.txt | cut -c1-100
