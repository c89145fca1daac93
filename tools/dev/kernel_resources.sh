#!/bin/bash
# Register / LDS / spill table of every kernel in a built engine library (default: the in-tree one):
#   tools/dev/kernel_resources.sh [lib.so] [name filter]
LIB=${1:-$(cd $(dirname $0)/../.. && pwd)/multiagent-quadruped-environment_amd/csrc/libmqe_hip.so}
T=$(mktemp -d); cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB >/dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $LIB 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=dev.co 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
flt = '${2:-}'
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if flt and flt not in name: continue
    print(f\"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}\")
"
rm -rf $T
