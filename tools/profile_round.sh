#!/bin/bash
# Run on the GPU box from the repo root (gpurun): kernel-trace stats, three PMC passes (counters never share a run with a
# trace domain other than --kernel-trace), and the bench line.  Results land in gpurun_out/<tag>_*; tools/summarize_pmc.py
# condenses them into profiles/.
#   usage: tools/profile_round.sh r01 [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_strict_f32 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- $BENCH > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- $BENCH > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1
cd $R
python bench.py --steps 500 --warmup 50 $* 2> $OUT/${TAG}_bench.err | grep "^{" > $OUT/${TAG}_bench.json
tail -1 $OUT/${TAG}_bench.json
