#!/bin/bash
# The part of tools/dev/refresh_profiles.sh that depends on kernel SCHEDULING only (kernel stats, PMC, bench lines, phase counters / wall
# times): what a round's last commits change when they leave the arithmetic alone.  Run from the repo root on the GPU box:
#   usage: tools/dev/refresh_profiles_short.sh r06f
set -u
T=${1:-r06f}
R=$PWD; mkdir -p gpurun_out
tools/profile_round.sh $T > gpurun_out/${T}_round.log 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{"; done > gpurun_out/${T}_bench_driver_style.json
tools/profile_tasks.sh $T > gpurun_out/${T}_tasks.log 2>&1
: > gpurun_out/${T}_other_tasks.txt
for spec in "go1football-1vs1 4096" "go1pushbox 4096" "go1bridge 4096" "go1wrestling 4096" "go1sheep-easy 4096" "go1revolvingdoor 4096" "go1tug 4096" "go1football-2vs2 2048" "go1plane 4096"; do
  set -- $spec
  python bench.py --task $1 --num_envs $2 --steps 200 --warmup 30 --no_cpu_baseline --no_strict_f32 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel']])" >> gpurun_out/${T}_other_tasks.txt
done
cd /tmp; export TMPDIR=/tmp
MQE_SOLVER=tgs rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/${T}_phase_tgs -- python $R/tools/phase_counters.py run go1gate 4096 > $R/gpurun_out/${T}_phase_tgs.log 2>&1
(cd $R; python tools/phase_counters.py report gpurun_out/${T}_phase_tgs > gpurun_out/${T}_phase_counters_go1gate_tgs.txt 2>&1)
cd $R
python tests/parity_sweep.py 256 gpurun_out/${T}_parity_sweep.json > gpurun_out/${T}_parity.log 2>&1
python tools/dev/phase_walltimes.py 4096 60 go1gate > gpurun_out/${T}_phase_walltimes_go1gate.txt 2>&1
python tools/dev/phase_walltimes.py 2048 60 go1sheep-hard > gpurun_out/${T}_phase_walltimes_go1sheep-hard.txt 2>&1
python tools/dev/phase_walltimes.py 4096 60 go1football-defender > gpurun_out/${T}_phase_walltimes_go1football-defender.txt 2>&1
python tools/dev/wave_times.py go1gate 4096 120 > gpurun_out/${T}_wave_times_go1gate.txt 2>&1
# keep what travels back small: the raw traces stay on the box
find gpurun_out -name "*_agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out
