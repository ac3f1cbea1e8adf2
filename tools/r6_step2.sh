#!/bin/bash
# round 6, GPU step 2: conv32 loads pinned + conv32_t16_kernel<160>
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
HP_C32_BN160=1 timeout 900 python -m pytest tests/test_engine_fp32_gpu.py -q -m gpu -x -k "not full_size" 2>&1 | tail -5 > $out/s2_pytest_bn160_forced.txt
timeout 900 python -m pytest tests/test_engine_fp32_gpu.py -q -m gpu -x 2>&1 | tail -5 > $out/s2_pytest_fp32.txt
HP_C32_BN160=0 timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s2_layers_f32_bn128.txt 2>&1
timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s2_layers_f32_bn160.txt 2>&1
cat $out/s2_pytest_bn160_forced.txt $out/s2_pytest_fp32.txt | cut -c1-300
tail -n 3 $out/s2_layers_f32_bn128.txt; tail -n 3 $out/s2_layers_f32_bn160.txt
grep -E "^ *(2|4|6|8|10|12|14|16|22|23|35|38) " $out/s2_layers_f32_bn128.txt $out/s2_layers_f32_bn160.txt
