for pad in 0 1400 3000 5400; do
  MQE_PHYS_LDS_PAD=$pad MQE_VERBOSE=1 python bench.py --task go1football-defender --num_envs 4096 --steps 150 --warmup 20 --no_cpu_baseline --no_strict_f32 2> gpurun_out/r05f/err_$pad.txt | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('pad $pad', round(d['value']/1e6,2), d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel'] if 'substeps' in r['kernel']])"
  grep "physics LDS" gpurun_out/r05f/err_$pad.txt | head -1
done
MQE_VERBOSE=1 MQE_COLLISION_MODEL=exact python bench.py --steps 50 --warmup 10 --no_cpu_baseline --no_strict_f32 2>&1 | grep "physics LDS\|k_substeps runs" | head -3
