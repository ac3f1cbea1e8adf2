#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_cpp_mirror.py -q -m gpu -x 2>&1 | tail -8 > $out/s6_pytest.txt
cut -c1-300 $out/s6_pytest.txt
for v in "" "HP_MIRROR_ONE_STREAM=1" "HP_MIRROR_HOST_MAPS=1" "HP_MIRROR_EAGER_HOST_COPY=1"; do echo "== $v"; env $v hyperpose_amd/operator_api_bench.bin lw_openpose_mobilenet 432 368 8 2 f32; done
hyperpose_amd/operator_api_bench.bin lw_openpose_mobilenet 432 368 8 2 f16
