python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -s -k drift 2>&1 | grep -E "^fp16 engine|passed|failed|Error" > gpurun_out/t11.log
cat gpurun_out/t11.log
