#!/bin/bash
# instruction-cache counters of the step's kernels (run from the repo root on the GPU box): gpurun_out/r06_icache/summary.txt
R=$PWD; O=$R/gpurun_out/r06_icache; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc -- python $R/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_strict_f32 > $O/pmc.log 2>&1
cd $R
python - > $O/summary.txt <<PY
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/r06_icache/pmc/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if k.startswith("k_"):
        print(k[:44], {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, "launches", len(list(v.values())[0]))
PY
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
