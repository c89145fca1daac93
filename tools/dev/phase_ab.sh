#!/bin/bash
# per-phase instruction counters of the physics substep (tools/phase_counters.py) for both contact solvers on one box -> gpurun_out/r04_phase_{tgs,pgs}.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in tgs pgs; do
  rm -rf gpurun_out/phase_pmc_$s
  MQE_SOLVER=$s rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/phase_pmc_$s -- python tools/phase_counters.py run go1gate 4096 > /dev/null 2>&1
  python tools/phase_counters.py report gpurun_out/phase_pmc_$s > gpurun_out/r04_phase_$s.txt 2>&1
  rm -rf gpurun_out/phase_pmc_$s
done
