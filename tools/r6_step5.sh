#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_engine_fp32_gpu.py -q -m gpu -x -k "one_round or half_batches or vggtiny" 2>&1 | tail -15 > $out/s5_pytest.txt
cat $out/s5_pytest.txt | cut -c1-400
