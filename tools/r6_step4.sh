#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
for pf in 1 0; do
HP_C32_PF=$pf HP_C32_BN160=0 timeout 200 python tools/direct_timeline.py f32 2>&1 >/dev/null | grep -A1 "^conv32 layer" > $out/s4_timeline_bn128_pf$pf.txt
echo "PF=$pf"; grep -A1 "layer 1[4]" $out/s4_timeline_bn128_pf$pf.txt | cut -c1-1400
done
