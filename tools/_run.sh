python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -25 > gpurun_out/t12.log
tail -12 gpurun_out/t12.log
