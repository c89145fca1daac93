"""Per-phase dynamic instruction counts of the physics substep.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
        --kernel-trace --output-format csv -d gpurun_out/phase_pmc -- python tools/phase_counters.py run go1gate 4096
    python tools/phase_counters.py report gpurun_out/phase_pmc

`run` steps the scene to a walking state, then launches the one-substep debug kernel once per phase tap with
mqe_debug_stop_phase(handle, i) (the wavefront returns at tap i, nothing is written back), so the counters of launch i minus
those of launch i-1 are phase i's.  `report` prints the differences per wavefront."""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["load", "FK", "inertia+shuffles+Mcols", "leg blocks", "schur 6x6", "Minv rows", "v*", "spheres", "terrain contacts",
         "pair contacts", "per-contact records", "(K build)", "GS", "lambda->v, limits", "store"]
TAPS = list(range(0, 15))

if sys.argv[1] == "run":
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
    import torch
    from helpers import make_desc, hip_engine
    from mqe.engine import abi
    task, n = sys.argv[2], int(sys.argv[3])
    d, k, _ = make_desc(task, n)
    e = hip_engine(d, k)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for t in range(60):
        e.step((torch.rand(n, Aw, 3, device="cuda", generator=g) * 2 - 1) * 0.5)
    torch.cuda.synchronize()
    for rep in range(2):
        for tap in TAPS + [100]:
            e._call("debug_stop_phase", tap)
            e.simulate()
            torch.cuda.synchronize()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    per = collections.OrderedDict()          # dispatch id -> {counter: value}
    grid = {}
    for r in csv.DictReader(open(f)):
        if "k_simulate" not in r["Kernel_Name"]:
            continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        grid[int(r["Dispatch_Id"])] = int(r["Grid_Size"]) // 64
    ids = sorted(per)
    ids = ids[len(ids) // 2:]                # second repetition (warm)
    assert len(ids) == len(TAPS) + 1, len(ids)
    cols = sorted(per[ids[0]])
    lanes = "SQ_THREAD_CYCLES_VALU" in cols and "SQ_ACTIVE_INST_VALU" in cols      # active lanes per VALU instruction = thread-cycles / instruction-cycles (of 64)

    def tail(delta):
        return f"{delta['SQ_THREAD_CYCLES_VALU'] / max(delta['SQ_ACTIVE_INST_VALU'], 1e-9):14.1f}" if lanes else ""
    print(f"{'phase':26s}" + "".join(f"{c.replace('SQ_', ''):>20s}" for c in cols) + (f"{'active lanes':>14s}" if lanes else ""))
    prev = {c: 0.0 for c in cols}
    for i, did in enumerate(ids):
        w = grid[did]
        cur = per[did]
        if i < len(TAPS):
            name = "(prologue)" if i == 0 else NAMES[i - 1]
            print(f"{name:26s}" + "".join(f"{(cur[c] - prev[c]) / w:20.1f}" for c in cols) + tail({c: cur[c] - prev[c] for c in cols}))
            prev = cur
        else:
            print(f"{'store (full - tap 14)':26s}" + "".join(f"{(cur[c] - prev[c]) / w:20.1f}" for c in cols) + tail({c: cur[c] - prev[c] for c in cols}))
            print(f"{'whole substep':26s}" + "".join(f"{cur[c] / w:20.1f}" for c in cols) + tail(cur))
