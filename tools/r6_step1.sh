#!/bin/bash
# round 6, GPU step 1: conv32 64x160 tile A/B, half-batch probe, the fp32 tests with the five-wavefront tile forced
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
HP_C32_BN160=1 timeout 900 python -m pytest tests/test_engine_fp32_gpu.py -q -m gpu -x -k "not full_size" 2>&1 | tail -5 > $out/s1_pytest_bn160_forced.txt
HP_C32_BN160=0 timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s1_layers_f32_bn128.txt 2>&1
timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s1_layers_f32_bn160.txt 2>&1
timeout 200 python tools/half_batch_probe.py f32 > $out/s1_half_batch.txt 2>&1
timeout 200 python tools/half_batch_probe.py f16 >> $out/s1_half_batch.txt 2>&1
cat $out/s1_pytest_bn160_forced.txt | cut -c1-300
tail -n 3 $out/s1_layers_f32_bn128.txt; tail -n 3 $out/s1_layers_f32_bn160.txt
grep -E "^ *(12|14|16|22) " $out/s1_layers_f32_bn128.txt $out/s1_layers_f32_bn160.txt
cat $out/s1_half_batch.txt
