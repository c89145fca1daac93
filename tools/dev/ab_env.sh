#!/bin/bash
# A/B of an env switch on one box: usage ab_env.sh VAR A B [bench args]
V=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do for x in $A $B; do
  env $V=$x python bench.py --steps 150 --warmup 20 --no_cpu_baseline --no_strict_f32 "$@" 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$V=$x', d['value'], d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel']])"
done; done
