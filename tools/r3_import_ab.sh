set -e
mkdir -p gpurun_out/imp
S=$PWD/optixpathtracer_amd/variants/libptamd_stats.so
for sc in terrain stadium; do
  W=""; [ $sc = stadium ] && W="--workload stadium1M_1080p_4spp_d8"
  for mode in auto import; do
    E=""; [ $mode = import ] && E="PT_BVH_IMPORT=$PWD/gpurun_in/${sc}_sah.tree"
    env $E PT_DEBUG_BVH=1 PT_DEBUG_COUNTS=1 PT_LIB=$S python bench.py $W --no-extra-schedules --steps 1 --warmup 1 > gpurun_out/imp/stats_${sc}_$mode.json 2> gpurun_out/imp/stats_${sc}_$mode.err
    echo "== $sc $mode"; grep -E "calibration|SAH cost|traversal:|per loop" gpurun_out/imp/stats_${sc}_$mode.err | tail -4
    env $E python bench.py $W --no-extra-schedules --steps 20 --warmup 5 > gpurun_out/imp/b_${sc}_$mode.json 2> gpurun_out/imp/b_${sc}_$mode.err
    python -c "
import json;d=json.loads(open('gpurun_out/imp/b_${sc}_$mode.json').read().strip().splitlines()[-1]);print('ms',d['ms_per_step'],'Mrays',d['value'],'trace_iso',d['roofline']['dominant_kernel']['avg_launch_ms'])"
  done
done
