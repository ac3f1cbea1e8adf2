#!/bin/bash
# tools/gpu_step.sh <name> - one gpurun call of this round's development loop (run ON the GPU box); everything lands in gpurun_out/<name>_*
name=${1:-s1}
cd $GRAFT_REPO_ROOT
out=gpurun_out
mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu -x ) 2>&1 | tail -25 > $out/${name}_pytest_all.txt
( time bash tools/collect_profiles.sh t05 1 f32 ) > $out/${name}_collect_trial.txt 2>&1
tail -n 12 $out/${name}_pytest_all.txt
tail -n 12 $out/${name}_collect_trial.txt
