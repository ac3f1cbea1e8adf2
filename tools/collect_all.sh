set -u
cd $GRAFT_REPO_ROOT
tag=${1:-r04}
for c in 1 4 3 2 5; do
  t0=$(date +%s)
  bash tools/collect_profiles.sh $tag $c > gpurun_out/collect_$c.log 2>&1
  echo "config $c: $(( $(date +%s) - t0 )) s" >> gpurun_out/collect_times.log
done
cd $GRAFT_REPO_ROOT
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 > gpurun_out/${tag}_layer_times_config1.txt 2>/dev/null
python tools/profile_layers.py openpose_vgg19 768 432 16 > gpurun_out/${tag}_layer_times_config2.txt 2>/dev/null
python tools/profile_layers.py pose_proposal_resnet50 384 384 32 > gpurun_out/${tag}_layer_times_config3.txt 2>/dev/null
python tools/profile_layers.py pifpaf_resnet50 385 385 64 > gpurun_out/${tag}_layer_times_config4.txt 2>/dev/null
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > gpurun_out/${tag}_layer_times_config1_fp32.txt 2>/dev/null
python tools/pipe_sweep.py 1 > gpurun_out/${tag}_pipe_sweep_config1.txt 2>/dev/null
