cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5) > gpurun_out/t_engine.log 2>&1
timeout 120 python tools/profile_layers.py 2>&1 | grep "sep \|total\|two streams" > gpurun_out/layers_pipe3.txt
HP_SEP_PIPE2=1 timeout 120 python tools/profile_layers.py 2>&1 | grep "sep \|total\|two streams" > gpurun_out/layers_pipe2c.txt
timeout 100 python tools/sep_timeline.py > gpurun_out/sep_timeline_pipe3.txt 2>&1
timeout 200 python tools/pipe_sweep.py 1 > gpurun_out/sweep_pipe3.txt 2>&1
HP_SEP_PIPE2=1 timeout 200 python tools/pipe_sweep.py 1 > gpurun_out/sweep_pipe2c.txt 2>&1
cat gpurun_out/t_engine.log
