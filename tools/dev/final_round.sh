set -u
T=r06f
tools/profile_round.sh $T > gpurun_out/${T}_round.log 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{"; done > gpurun_out/${T}_bench_driver_style.json
python bench.py 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_default.json
tail -c 300 gpurun_out/${T}_bench.json
