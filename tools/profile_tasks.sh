#!/bin/bash
# per-task bench line (WITH cpu_baseline / strict_f32) + kernel-trace stats (no PMC) for the BASELINE.json configs 3-5
#   usage: tools/profile_tasks.sh r02
TAG=${1:-r02}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "go1sheep-hard 2048" "go1seesaw 4096" "go1football-defender 4096"; do
  set -- $spec; t=$1; n=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}t_${t}_trace -- python $R/bench.py --task $t --num_envs $n --steps 60 --warmup 10 --no_cpu_baseline --no_strict_f32 > $OUT/${TAG}t_${t}_trace.log 2>&1
  (cd $R && python bench.py --task $t --num_envs $n --steps 300 --warmup 30 --cpu_sample_envs 512 --cpu_sample_steps 10 2>/dev/null | grep "^{" > $OUT/${TAG}t_${t}_bench.json)
  tail -c 400 $OUT/${TAG}t_${t}_bench.json; echo
done
