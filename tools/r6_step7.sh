#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
( time timeout 900 python bench.py --steps 20 --warmup 5 --extra 1/f16,3/f32 ) > $out/s7_bench.json 2> $out/s7_bench.err
tail -c 1500 $out/s7_bench.err; echo; cat $out/s7_bench.json | tail -n 1 | cut -c1-3900
cp bench_detail.json $out/s7_bench_detail.json
