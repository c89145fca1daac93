#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc_summary.json (per kernel: launches, mean duration, FETCH/WRITE bytes per launch with the gfx950
correction of MI355X_MICROARCH.md, SQ occupancy/issue counters)."""
import csv, glob, json, os, sys, collections, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern), recursive=True), key=os.path.getmtime)     # newest run wins
    return f[-1] if f else None


def short(name):
    """kernel name without signature, return type and template arguments: `void k_substeps<2, 0>(...)` -> `k_substeps`"""
    n = name.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    if n.startswith("k_"):
        n = n.split("<")[0]
    return n


summary = {"_note": ("rocprofv3 on `bench.py --steps 60 --warmup 10` (go1gate 4096 envs x 2 agents, 1 MI355X), separate runs: "
                     "--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_*/GRBM.  FETCH_SIZE/WRITE_SIZE are KB per "
                     "dispatch.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) "
                     "coalesced reads -> `fetch_bytes_x2` doubles it (the upper estimate for kernels that read 16 B/lane: k_gemm_*); "
                     "4 B/lane kernels (k_substeps, k_post_physics, ...) are uncalibrated and should be read with the raw value. "
                     "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES cycles.")}
stats = one(f"{tag}_trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(P, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        summary.setdefault(k, {})
        summary[k].update({"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "pct_gpu_time": float(r["Percentage"])})
for key, pat in (("FETCH_SIZE", f"{tag}_pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", f"{tag}_pmc_write/**/*counter_collection.csv"),
                 (None, f"{tag}_pmc_sq/**/*counter_collection.csv")):
    f = one(pat)
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summary.setdefault(short(r["Kernel_Name"]), {})["vgpr"] = int(r["VGPR_Count"]); summary[short(r["Kernel_Name"])]["agpr"] = int(r["Accum_VGPR_Count"])
        summary[short(r["Kernel_Name"])]["lds_bytes"] = int(r["LDS_Block_Size"])
    for k, cs in acc.items():
        for c, v in cs.items():
            summary[k][c + "_mean"] = round(sum(v) / len(v), 1)
for k, e in summary.items():
    if not isinstance(e, dict):
        continue
    if "FETCH_SIZE_mean" in e:
        e["fetch_bytes_raw"] = int(e["FETCH_SIZE_mean"] * 1024); e["fetch_bytes_x2"] = 2 * e["fetch_bytes_raw"]
    if "WRITE_SIZE_mean" in e:
        e["write_bytes_raw"] = int(e["WRITE_SIZE_mean"] * 1024)
    if "fetch_bytes_raw" in e and "write_bytes_raw" in e:
        e["hbm_bytes_raw"] = e["fetch_bytes_raw"] + e["write_bytes_raw"]
    if "SQ_WAVE_CYCLES_mean" in e and e["SQ_WAVE_CYCLES_mean"] > 0:
        wc = e["SQ_WAVE_CYCLES_mean"]
        e["frac_wave_time_waiting_on_waitcnt_or_barrier"] = round(e.get("SQ_WAIT_ANY_mean", 0) / wc, 3)
        e["frac_wave_time_issue_stalled"] = round(e.get("SQ_WAIT_INST_ANY_mean", 0) / wc, 3)
        e["frac_wave_time_issuing"] = round(e.get("SQ_ACTIVE_INST_ANY_mean", 0) / wc, 3)
json.dump(summary, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
b = os.path.join(G, f"{tag}_bench.json")
if os.path.isfile(b):
    line = open(b).read().strip().split("\n")[-1]
    json.loads(line)
    open(os.path.join(P, f"{tag}_bench_1gpu.json"), "w").write(line + "\n")
print("wrote", os.path.join(P, f"{tag}_pmc_summary.json"))
