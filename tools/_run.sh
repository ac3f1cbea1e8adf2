python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -6 > gpurun_out/t_final.log
tail -3 gpurun_out/t_final.log
bash tools/collect_profiles.sh r03 1 > gpurun_out/collect1.log 2>&1
python tools/profile_layers.py > gpurun_out/r03_layer_times_config1.txt 2>&1
python tools/profile_layers.py openpose_vgg19 768 432 16 > gpurun_out/r03_layer_times_config2.txt 2>&1
python tools/profile_layers.py pose_proposal_resnet50 384 384 32 > gpurun_out/r03_layer_times_config3.txt 2>&1
python tools/profile_layers.py pifpaf_resnet50 385 385 64 > gpurun_out/r03_layer_times_config4.txt 2>&1
python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/b_final.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driverflags.json 2>> gpurun_out/b_final.err
tail -2 gpurun_out/r03_layer_times_config1.txt
