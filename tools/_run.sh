cd hyperpose_amd && cp libhp_hip.so libhp_hip_tw8.so && cd ..
run() { python bench.py --extra= --no-cpu-baseline --no-from-host --no-roofline --no-dnn-output 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1',d['value'],d['ms_per_step'])"; }
for i in 1 2 3; do
  cp hyperpose_amd/libhp_hip_tw12.so hyperpose_amd/libhp_hip.so; run tw12
  cp hyperpose_amd/libhp_hip_tw8.so hyperpose_amd/libhp_hip.so; run tw8
done > gpurun_out/ab_chain_tile.txt
cat gpurun_out/ab_chain_tile.txt
