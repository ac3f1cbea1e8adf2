# tools/collect_all.sh <tag> [what] - run ON the GPU box: rocprofv3 kernel stats + PMC passes of every workload (tools/collect_profiles.sh), the
# per-layer tables, the pipe sweep and the clock probe.  what = "all" (default) | "fp32" (configs[1] at the three precisions only)
set -u
cd $GRAFT_REPO_ROOT
tag=${1:-r06}
what=${2:-all}
mkdir -p gpurun_out
list="1:f32 1:f16 1:f32s"
[ "$what" == "all" ] && list="1:f32 1:f16 1:f32s 4:f32 3:f32 2:f32 0:f32 4:f16 3:f16 2:f16 0:f16"
for cd in $list; do
  c=${cd%%:*}; d=${cd##*:}
  t0=$(date +%s)
  bash tools/collect_profiles.sh $tag $c $d > gpurun_out/collect_${c}_${d}.log 2>&1
  echo "config $c $d: $(( $(date +%s) - t0 )) s" >> gpurun_out/collect_times.log
done
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32 > gpurun_out/${tag}_layer_times_config1_fp32.txt 2>/dev/null
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 f32s > gpurun_out/${tag}_layer_times_config1_fp32s.txt 2>/dev/null
python tools/profile_layers.py lw_openpose_mobilenet 432 368 8 > gpurun_out/${tag}_layer_times_config1.txt 2>/dev/null
if [ "$what" == "all" ]; then
  python tools/profile_layers.py openpose_vgg19 768 432 16 > gpurun_out/${tag}_layer_times_config2.txt 2>/dev/null
  python tools/profile_layers.py pose_proposal_resnet50 384 384 32 > gpurun_out/${tag}_layer_times_config3.txt 2>/dev/null
  python tools/profile_layers.py pifpaf_resnet50 385 385 64 > gpurun_out/${tag}_layer_times_config4.txt 2>/dev/null
  python tools/profile_layers.py openpose_vgg19 768 432 16 f32 > gpurun_out/${tag}_layer_times_config2_fp32.txt 2>/dev/null
  python tools/profile_layers.py pose_proposal_resnet50 384 384 32 f32 > gpurun_out/${tag}_layer_times_config3_fp32.txt 2>/dev/null
  python tools/profile_layers.py pifpaf_resnet50 385 385 64 f32 > gpurun_out/${tag}_layer_times_config4_fp32.txt 2>/dev/null
  python tools/pipe_sweep.py 1 > gpurun_out/${tag}_pipe_sweep_config1.txt 2>/dev/null
fi
# the activation arena's effect on HBM traffic: the same passes with one allocation per tensor; the file names do not start with
# "r" so that bench.py's "newest committed profile" look-up never takes them
HP_NO_ARENA=1 bash tools/collect_profiles.sh noarena_${tag} 1 f32 > gpurun_out/collect_noarena.log 2>&1
python tools/arena_probe.py > gpurun_out/${tag}_arena_hbm.txt 2>&1
PYTHONPATH=. python tools/direct_timeline.py f32 2> gpurun_out/${tag}_block_timelines_f32.txt > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe > gpurun_out/${tag}_clock_probe.txt 2>&1
cat gpurun_out/collect_times.log
