#!/usr/bin/env python3
"""Where do the driver's 20 timed steps spend their time?  Reads a rocprofv3 --kernel-trace csv of `bench.py --steps 20 --warmup 5` and prints, for the LAST
20 steps before the long run (k_substeps launches 6..25 of the headline engine), each step's span (start of k_pre_policy .. end of k_substeps), the
GPU-idle gap before it, and the kernel durations.  Usage: python tools/dev/timeline_20.py <kernel_trace.csv>"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
steps, cur = [], None
prev_end = None
for name, s, e in ev:
    if name == "k_pre_policy":
        cur = {"start": s, "gap": (s - prev_end) if prev_end else 0, "k": {}}
    if cur is not None and name in ("k_pre_policy", "k_gemm_h2", "k_policy_tail", "k_substeps"):
        cur["k"][name] = (s, e)
        if name == "k_substeps":
            cur["end"] = e
            steps.append(cur); cur = None
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"{len(steps)} steps in the trace")
for i, st in enumerate(steps[:60]):
    k = st["k"]
    d = {n: (k[n][1] - k[n][0]) / 1e3 for n in k}
    inner = sum((k[b][0] - k[a][1]) / 1e3 for a, b in (("k_pre_policy", "k_gemm_h2"), ("k_gemm_h2", "k_policy_tail"), ("k_policy_tail", "k_substeps")) if a in k and b in k)
    nxt = (steps[i + 1]["start"] - st["end"]) / 1e3 if i + 1 < len(steps) else 0.0
    print(f"step {i:3d}: span {(st['end'] - st['start']) / 1e3:7.1f} us  idle before {st['gap'] / 1e3:8.1f}  gaps inside {inner:5.1f}  to next {nxt:7.1f}   " +
          "  ".join(f"{n[2:]} {d[n]:6.1f}" for n in ("k_pre_policy", "k_gemm_h2", "k_policy_tail", "k_substeps") if n in d))
