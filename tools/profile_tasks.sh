#!/bin/bash
# per-task bench line + kernel-trace stats (no PMC) for the BASELINE.json parity configs
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "go1seesaw 4096" "go1sheep-hard 2048" "go1football-defender 4096"; do
  set -- $spec; t=$1; n=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r01t_${t}_trace -- python $R/bench.py --task $t --num_envs $n --steps 60 --warmup 10 --no_cpu_baseline > $OUT/r01t_${t}_trace.log 2>&1
  (cd $R && python bench.py --task $t --num_envs $n --steps 300 --warmup 30 --no_cpu_baseline 2>/dev/null | grep "^{" > $OUT/r01t_${t}_bench.json)
  tail -c 300 $OUT/r01t_${t}_bench.json; echo
done
