cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/fillbench.bin > gpurun_out/fillbench.txt 2>&1
(time timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40) > gpurun_out/t_all.log 2>&1
timeout 300 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -n 5 gpurun_out/t_all.log
