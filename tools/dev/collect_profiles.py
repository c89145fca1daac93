#!/usr/bin/env python3
"""Copies what tools/dev/refresh_profiles.sh left in gpurun_out/ into profiles/ under the round tag (after tools/summarize_pmc.py <tag>):
    python tools/dev/collect_profiles.py r04"""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.isfile(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst)); print("  ", dst)


for f in (f"{T}_bench_driver_style.json", f"{T}_other_tasks.txt", f"{T}_parity_sweep.json", f"{T}_parity_sweep_pgs.json", f"{T}_phase_counters_go1gate_tgs.txt",
          f"{T}_phase_counters_go1gate_pgs.txt", f"{T}_phase_walltimes_go1gate.txt", f"{T}_phase_walltimes_go1sheep-hard.txt", f"{T}_phase_walltimes_go1football-defender.txt",
          f"{T}_wave_times_go1gate.txt", f"{T}_batch_sweep.json", f"{T}_phase_lanes_go1gate.txt", f"{T}_dt_convergence.json", f"{T}_graph_probe.txt",
          f"{T}_parity_sweep_long.json"):
    cp(f, f)
for task in ("go1sheep-hard", "go1seesaw", "go1football-defender"):
    cp(f"{T}t_{task}_bench.json", f"{T}_bench_{task}.json")
    st = sorted(glob.glob(os.path.join(G, f"{T}t_{task}_trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(P, f"{T}_kernel_stats_{task}.csv")); print("  ", f"{T}_kernel_stats_{task}.csv")
# the LDS counter pass
f = sorted(glob.glob(os.path.join(G, f"{T}_pmc_lds", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        n = r["Kernel_Name"].split("(")[0].strip()
        n = n[5:] if n.startswith("void ") else n
        acc[n.split("<")[0] if n.startswith("k_") else n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_note": "rocprofv3 --pmc SQ_LDS_* pass on bench.py --steps 30 (go1gate 4096 x 2): per-launch means, summed over the chip; cycles", "kernels": {}}
    for k, v in acc.items():
        if not k.startswith("k_"):
            continue
        d = {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["bank_conflict_share_of_lds_active"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 3)
        out["kernels"][k] = d
    json.dump(out, open(os.path.join(P, f"{T}_pmc_lds.json"), "w"), indent=1); print("  ", f"{T}_pmc_lds.json")
