#!/bin/bash
# A/B of two builds of the engine on ONE box (box-to-box variance is +-5 %): alternates the two libraries, three runs each.
#   usage (GPU box): tools/ab_bench.sh libA.so libB.so [bench args...]   -> gpurun_out/ab.txt
A=$1; B=$2; shift 2
R=$PWD; mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab.txt
for i in 1 2 3; do for lib in $A $B; do
  MQE_HIP_LIB=$R/multiagent-quadruped-environment_amd/csrc/$lib python bench.py --steps 150 --warmup 20 --no_cpu_baseline --no_strict_f32 "$@" 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel']])" >> $R/gpurun_out/ab.txt
done; done
cat $R/gpurun_out/ab.txt
