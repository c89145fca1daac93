#!/bin/bash
# One gpurun call that re-measures everything profiles/<tag>_* holds (run from the repo root on the GPU box):
#   usage: tools/dev/refresh_profiles.sh r04
# then, back in the container: python tools/summarize_pmc.py <tag>, and copy the gpurun_out/<tag>_* summaries into profiles/.
set -u
T=${1:-r04}
R=$PWD; mkdir -p gpurun_out
tools/profile_round.sh $T > gpurun_out/${T}_round.log 2>&1
# the line the driver takes (--steps 20 --warmup 5, defaults otherwise), three times
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_strict_f32 2>/dev/null | grep "^{"; done > gpurun_out/${T}_bench_driver_style.json
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
for s in tgs pgs; do
  MQE_SOLVER=$s rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/${T}_phase_$s -- python $R/tools/phase_counters.py run go1gate 4096 > $R/gpurun_out/${T}_phase_$s.log 2>&1
  (cd $R; python tools/phase_counters.py report gpurun_out/${T}_phase_$s > gpurun_out/${T}_phase_counters_go1gate_$s.txt 2>&1)
done
# lane utilisation per phase: active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${T}_phase_lanes -- python $R/tools/phase_counters.py run go1gate 4096 > $R/gpurun_out/${T}_phase_lanes.log 2>&1
(cd $R; python tools/phase_counters.py report gpurun_out/${T}_phase_lanes > gpurun_out/${T}_phase_lanes_go1gate.txt 2>&1)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc_lds -- python $R/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_strict_f32 > $R/gpurun_out/${T}_pmc_lds.log 2>&1
cd $R
python tests/parity_sweep.py 256 gpurun_out/${T}_parity_sweep.json > gpurun_out/${T}_parity.log 2>&1
MQE_SOLVER=pgs python tests/parity_sweep.py 256 gpurun_out/${T}_parity_sweep_pgs.json > gpurun_out/${T}_parity_pgs.log 2>&1
# where a full launch's time goes, phase by phase (every wavefront stamps the wall clock at its taps), for the headline and the two large scenes
python tools/dev/phase_walltimes.py 4096 60 go1gate > gpurun_out/${T}_phase_walltimes_go1gate.txt 2>&1
python tools/dev/phase_walltimes.py 2048 60 go1sheep-hard > gpurun_out/${T}_phase_walltimes_go1sheep-hard.txt 2>&1
python tools/dev/phase_walltimes.py 4096 60 go1football-defender > gpurun_out/${T}_phase_walltimes_go1football-defender.txt 2>&1
python tools/dev/wave_times.py go1gate 4096 120 > gpurun_out/${T}_wave_times_go1gate.txt 2>&1
python tools/dt_convergence.py gpurun_out/${T}_dt_convergence.json 32 > gpurun_out/${T}_dt_convergence.txt 2>&1
python tools/dev/graph_probe.py go1gate 4096 400 > gpurun_out/${T}_graph_probe.txt 2>&1
# the long form of the parity sweep last (11 minutes of CPU oracle): 1024 envs x 200 fused steps, nothing re-synchronised
python tests/parity_sweep.py 1024 gpurun_out/${T}_parity_sweep_long.json 200 > gpurun_out/${T}_parity_long.log 2>&1
