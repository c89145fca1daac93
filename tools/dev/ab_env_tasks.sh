#!/bin/bash
# A/B of an env switch over several tasks on one box: usage ab_env_tasks.sh VAR A B "task N" ...   (A / B = the values; "-" = unset)
V=$1; A=$2; B=$3; shift 3
for spec in "$@"; do set -- $spec; for i in 1 2; do for x in $A $B; do
  if [ "$x" = "-" ]; then E="env -u $V"; else E="env $V=$x"; fi
  $E python bench.py --task $1 --num_envs $2 --steps 150 --warmup 20 --no_cpu_baseline --no_strict_f32 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', '$V=$x', round(d['value']/1e6,2), d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel']])"
done; done; done
