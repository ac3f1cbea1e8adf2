python -m pytest tests/test_cpp_mirror.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -15
