#!/bin/bash
# tools/final_collect.sh [tag] - run ON the GPU box (gpurun): the round's evidence in one call.
#   1. the whole GPU test suite;  2. rocprofv3 kernel stats + PMC passes of every workload and precision (tools/collect_all.sh), copied into
#   profiles/ of the box's copy so that 3. the bench line quotes the summaries of THIS tree;  4. `python bench.py` with the driver's flags.
# Everything lands in gpurun_out/ (merged back by gpurun); copy <tag>_* into profiles/ afterwards.
set -u
cd $GRAFT_REPO_ROOT
tag=${1:-r06}
o=gpurun_out
mkdir -p $o
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $o/${tag}_pytest_gpu.txt
echo "pytest: $(( $(date +%s) - t0 )) s" > $o/${tag}_times.txt
bash tools/collect_all.sh $tag all > $o/collect_all.log 2>&1
echo "profiles: $(( $(date +%s) - t0 )) s" >> $o/${tag}_times.txt
cp $o/${tag}_kernel_stats*.csv $o/${tag}_pmc_*.json $o/noarena_${tag}_*.json $o/noarena_${tag}_*.csv profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/${tag}_bench_final.json 2> $o/${tag}_bench_final.err
cp bench_detail.json $o/${tag}_bench_detail.json
echo "bench: $(( $(date +%s) - t0 )) s" >> $o/${tag}_times.txt
PYTHONPATH=. timeout 300 python tools/pifpaf_stress.py 4 64 > $o/${tag}_pifpaf_stress.txt 2>&1
hyperpose_amd/operator_api_bench.bin lw_openpose_mobilenet 432 368 8 2 f32 > $o/${tag}_operator_api.txt 2>&1
HP_MIRROR_HOST_MAPS=1 hyperpose_amd/operator_api_bench.bin lw_openpose_mobilenet 432 368 8 2 f32 >> $o/${tag}_operator_api.txt 2>&1
hyperpose_amd/operator_api_bench.bin lw_openpose_mobilenet 432 368 8 2 f16 >> $o/${tag}_operator_api.txt 2>&1
python tools/half_batch_probe.py f32 > $o/${tag}_half_batch_probe.txt 2>&1
python tools/parser_dnn_probe.py > $o/${tag}_parser_dnn_probe.txt 2>&1
python __graft_entry__.py --smoke 2>&1 | tail -n 2 > $o/${tag}_smoke.txt
tail -n 6 $o/${tag}_pytest_gpu.txt | cut -c1-200
cat $o/${tag}_times.txt $o/${tag}_smoke.txt
tail -n 1 $o/${tag}_pifpaf_stress.txt
wc -c $o/${tag}_bench_final.json
cut -c1-300 $o/${tag}_bench_final.json
