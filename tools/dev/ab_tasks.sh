#!/bin/bash
# A/B of two engine libraries over several tasks on one box: usage ab_tasks.sh libA.so libB.so "task N" ...
A=$1; B=$2; shift 2
R=$PWD
for spec in "$@"; do set -- $spec; for i in 1 2; do for lib in $A $B; do
  MQE_HIP_LIB=$R/multiagent-quadruped-environment_amd/csrc/$lib python bench.py --task $1 --num_envs $2 --steps 150 --warmup 20 --no_cpu_baseline --no_strict_f32 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', '$lib', round(d['value']/1e6,2), d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel'] if 'substeps' in r['kernel']])"
done; done; done
