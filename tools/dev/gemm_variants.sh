#!/bin/bash
# k_gemm_h2: which of the loop's streams bounds it.  `build` (here, cross-compiles) writes build/gemm_<variant>; `run` (GPU box) times each.
#   variants: base | nomfma (fragment reads kept alive by one VALU op each) | noload (stagers store register garbage) | nost (loads kept alive, no LDS stores)
#             | nold_nost (multipliers alone: fragment reads + MFMAs) | mfma_only (no loads, no stores, and the reads hoisted: the matrix pipe alone)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-uninitialized -w"
NOMFMA='-DH2_MFMA(a_,b_,c_)=([&]{auto c2_=c_;c2_[0]+=__builtin_bit_cast(float,(a_).x^(b_).x);return c2_;}())'
NOLDA='-DH2_LDA(dst,i_)=dst.x=aoff;'
NOLDW='-DH2_LDW(dst,i_)=dst.x=woff;'
NOST='-DH2_ST(buf_,src_,r_)=asm volatile(""::"v"(src_));'
if [ "${1:-build}" = build ]; then
  mkdir -p $R/build
  hipcc $FLAGS $R/tools/dev/gemm_bench.hip -o $R/build/gemm_base &
  hipcc $FLAGS "$NOMFMA" $R/tools/dev/gemm_bench.hip -o $R/build/gemm_nomfma &
  hipcc $FLAGS "$NOLDA" "$NOLDW" $R/tools/dev/gemm_bench.hip -o $R/build/gemm_noload &
  hipcc $FLAGS "$NOST" $R/tools/dev/gemm_bench.hip -o $R/build/gemm_nost &
  hipcc $FLAGS "$NOLDA" "$NOLDW" "$NOST" $R/tools/dev/gemm_bench.hip -o $R/build/gemm_nold_nost &
  hipcc $FLAGS -DH2_STAGER_TIMES $R/tools/dev/gemm_bench.hip -o $R/build/gemm_tstager &
  hipcc $FLAGS -DH2_MULT_TIMES $R/tools/dev/gemm_bench.hip -o $R/build/gemm_tmult &
  hipcc $FLAGS -DH2_STAGER_TIMES "$NOMFMA" $R/tools/dev/gemm_bench.hip -o $R/build/gemm_tstager_nomfma &
  wait
  ls -la $R/build
else
  for v in ${VARIANTS:-base nomfma noload nost nold_nost tstager tmult tstager_nomfma}; do
    echo "== $v"; $R/build/gemm_$v 8192 300 | grep -v "^mode"
  done
fi
