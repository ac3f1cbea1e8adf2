cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(HP_SEP_PIPE1=1 timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "separable or lw_openpose or sep" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4) > gpurun_out/t_pipe1.log 2>&1
(HP_SEP_SLOT=1 timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "separable or lw_openpose or sep" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4) > gpurun_out/t_slot.log 2>&1
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
cat gpurun_out/t_pipe1.log gpurun_out/t_slot.log; tail -2 gpurun_out/smoke.log
