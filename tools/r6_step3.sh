#!/bin/bash
# round 6, GPU step 3: prefetch placement (HP_C32_PF) x tile (HP_C32_BN160) on the fp32 GEMM kernels
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
for pf in 3 0 1 2; do for bn in 0 -1; do
  HP_C32_PF=$pf HP_C32_BN160=$bn timeout 300 python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > $out/s3_layers_pf${pf}_bn${bn}.txt 2>&1
  echo "== PF=$pf BN160=$bn"; grep -E "^ *(4|6|12|14|22) " $out/s3_layers_pf${pf}_bn${bn}.txt; tail -n 3 $out/s3_layers_pf${pf}_bn${bn}.txt
done; done
