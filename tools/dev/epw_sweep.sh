#!/bin/bash
# go1gate by batch size with one / two envs per wavefront of k_substeps (run on the GPU box from the repo root): gpurun_out/<tag>_epw_sweep.json
#   usage: tools/dev/epw_sweep.sh r06
T=${1:-r06}
mkdir -p gpurun_out
python - <<PY > gpurun_out/${T}_epw_sweep.json
import json, subprocess, sys, os
rows = []
for n in (4096, 4608, 5120, 6144, 8192, 12288, 16384):
    for epw in (1, 2):
        env = dict(os.environ, MQE_ENVS_PER_WAVE=str(epw))
        out = subprocess.run([sys.executable, "bench.py", "--num_envs", str(n), "--steps", "100", "--warmup", "20", "--no_cpu_baseline", "--no_strict_f32"],
                             capture_output=True, text=True, env=env).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        rows.append({"num_envs": n, "envs_per_wavefront": epw, "env_steps_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                     "kernels_ms": {r["kernel"].split("(")[0]: r["avg_launch_ms"] for r in d["roofline_per_kernel"]}})
print(json.dumps({"what": "go1gate x 2 agents, bench.py --steps 100 --warmup 20, MQE_ENVS_PER_WAVE = 1 / 2 (k_substeps<2,0,1> / <2,0,2>, both with the fused post-physics epilogue), one MI355X, one box", "rows": rows}, indent=1))
PY
python - <<PY
import json
for r in json.load(open("gpurun_out/${T}_epw_sweep.json"))["rows"]: print(r["num_envs"], r["envs_per_wavefront"], r["env_steps_per_s"], r["ms_per_step"], r["kernels_ms"].get("substeps"))
PY
