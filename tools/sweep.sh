HP_CONV_BIGTILE=1 timeout 120 tools/microbench.bin 2>&1 | grep -A2 "^conv 1x1\|timeline" | grep -B1 -A2 "conv 1x1" | head -30
python tools/queue_probe.py --pipes 4 --modes injected,engine
HP_CONV_BIGTILE=1 python tools/queue_probe.py --pipes 4 --modes injected,engine
HP_CONV_BIGTILE=1 timeout 100 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
