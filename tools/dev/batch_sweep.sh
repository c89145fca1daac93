#!/bin/bash
# go1gate throughput by batch size on one box (run on the GPU box from the repo root): gpurun_out/<tag>_batch_sweep.json
#   usage: tools/dev/batch_sweep.sh r04
T=${1:-r04}
R=$PWD; mkdir -p gpurun_out
python - <<PY > gpurun_out/${T}_batch_sweep.json
import json, subprocess, sys
rows = []
for n in (1, 256, 1000, 2048, 3072, 4096, 4097, 4352, 4608, 5120, 6144, 8192, 8193, 10240, 12288):
    out = subprocess.run([sys.executable, "bench.py", "--num_envs", str(n), "--steps", "100", "--warmup", "20", "--no_cpu_baseline", "--no_strict_f32"],
                         capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    rows.append({"num_envs": n, "env_steps_per_s": d["value"], "ms_per_step": d["ms_per_step"], "us_per_env_step": 1e3 * d["ms_per_step"] / n,
                 "kernels_ms": {r["kernel"].split("(")[0]: r["avg_launch_ms"] for r in d["roofline_per_kernel"]}})
ref = next(r for r in rows if r["num_envs"] == 4096)
for r in rows:
    r["per_env_throughput_vs_4096"] = round(ref["us_per_env_step"] / r["us_per_env_step"], 3)
print(json.dumps({"what": "go1gate x 2 agents, bench.py --steps 100 --warmup 20 per batch size, one MI355X, one box; per_env_throughput_vs_4096 = (time per env-step at 4096) / (time per env-step at this size)",
                  "rows": rows}, indent=1))
PY
tail -5 gpurun_out/${T}_batch_sweep.json
