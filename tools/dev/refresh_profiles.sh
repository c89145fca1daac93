set -u
R=$PWD; mkdir -p gpurun_out
tools/profile_round.sh r03f > gpurun_out/r03f_round.log 2>&1
tools/profile_tasks.sh r03f > gpurun_out/r03f_tasks.log 2>&1
: > gpurun_out/r03f_other_tasks.txt
for spec in "go1football-1vs1 4096" "go1pushbox 4096" "go1bridge 4096" "go1wrestling 4096" "go1sheep-easy 4096" "go1revolvingdoor 4096" "go1tug 4096" "go1football-2vs2 2048" "go1plane 4096"; do
  set -- $spec
  python bench.py --task $1 --num_envs $2 --steps 200 --warmup 30 --no_cpu_baseline --no_strict_f32 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], [(r['kernel'][:8], r['avg_launch_ms']) for r in d['roofline_per_kernel']])" >> gpurun_out/r03f_other_tasks.txt
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/r03f_phase -- python $R/tools/phase_counters.py run go1gate 4096 > $R/gpurun_out/r03f_phase.log 2>&1
cd $R
python tools/phase_counters.py report gpurun_out/r03f_phase > gpurun_out/r03f_phase_counters.txt 2>&1
python tests/parity_sweep.py 256 gpurun_out/r03f_parity_sweep.json > gpurun_out/r03f_parity.log 2>&1
