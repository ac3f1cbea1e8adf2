cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "weights_stationary" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12) > gpurun_out/t_ws.log 2>&1
timeout 100 python tools/ws_probe.py 2>/dev/null | grep mode > gpurun_out/ws_probe3.txt
HP_NO_WS1X1=1 timeout 100 python tools/ws_probe.py 2>/dev/null | grep mode >> gpurun_out/ws_probe3.txt
cat gpurun_out/t_ws.log gpurun_out/ws_probe3.txt
