#!/usr/bin/env python3
"""Static VALU instruction count of one k_substeps instantiation, per phase of phys_substep (no GPU needed).

    python tools/dev/static_valu.py [kernel-symbol-substring] [--lines LO HI]

Builds the engine with -gline-tables-only into /tmp, disassembles the kernel, symbolizes every instruction with its inline chain
(llvm-symbolizer --inlines) and attributes each VALU instruction to the source line of phys_substep that (transitively) issued it.
The phases are the TSTAMP taps of csrc/kernels_physics.hpp, found by their line numbers.  The physics body is nearly straight-line
code (only the contact sweep loops), so the static count of a phase tracks its dynamic one (tools/phase_counters.py) closely."""
import collections, os, re, subprocess, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "multiagent-quadruped-environment_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
T = "/tmp/static_valu"
os.makedirs(T, exist_ok=True)
sym_sub = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "k_substepsILi2ELi0ELi1ELb0ELb0EE"
lines_rng = None
if "--lines" in sys.argv:
    i = sys.argv.index("--lines"); lines_rng = (int(sys.argv[i + 1]), int(sys.argv[i + 2]))
fp = ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros", "-fno-math-errno",
      "-freciprocal-math", "-fgpu-flush-denormals-to-zero"]
extra = os.environ.get("MQE_EXTRA_FLAGS", "").split()
if not os.environ.get("MQE_SKIP_BUILD"): subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-comment",
                       "-fno-slp-vectorize", *fp, *extra, "-gline-tables-only", os.path.join(CSRC, "mqe_engine.hip"), "-o", f"{T}/dbg.so"])
subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={T}/fat.bin", f"{T}/dbg.so"])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={T}/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={T}/dev.co"])
syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", f"{T}/dev.co"], capture_output=True, text=True).stdout
name = [l.split()[-1] for l in syms.splitlines() if sym_sub in l and " FUNC " in l][0]
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={name}", f"{T}/dev.co"], capture_output=True, text=True).stdout
addrs, ops = [], []
for l in dis.splitlines():
    m = re.match(r"\s+(\S+)\s.*// ([0-9A-F]{12}):", l)
    if m:
        addrs.append(int(m.group(2), 16)); ops.append(m.group(1))
out = subprocess.run([f"{LLVM}/llvm-symbolizer", f"--obj={T}/dev.co", "--inlines", "--functions=short", "--output-style=LLVM"],
                     input="\n".join(hex(a) for a in addrs) + "\n", capture_output=True, text=True).stdout
blocks = out.strip().split("\n\n")
src = open(os.path.join(CSRC, "kernels_physics.hpp")).read().split("\n")
taps = [(i + 1, int(re.search(r"TSTAMP\((\d+)\)", l).group(1))) for i, l in enumerate(src) if re.match(r"\s*TSTAMP\(\d+\);", l)]
NAMES = {0: "prologue", 1: "load", 2: "FK", 3: "inertia+Mcols", 4: "-", 5: "schur", 6: "Minv rows", 7: "v*", 8: "spheres/prims", 9: "terrain", 10: "pairs/self",
         11: "records", 12: "-", 13: "GS", 14: "lambda->v, limits"}
end_line = next(i + 1 for i, l in enumerate(src) if l.startswith("__global__") and "k_simulate(" in l)


def phase(ln):
    for line, tap in taps:
        if ln < line:
            return f"{tap:2d} {NAMES.get(tap, '?')}"
    return "15 store/integrate"


byphase, byline, other, bypost = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
src_step = open(os.path.join(CSRC, "kernels_step.hpp")).read().split("\n")
nv = 0
for op, b in zip(ops, blocks):
    if not op.startswith("v_"):
        continue
    nv += 1
    ls = b.strip().split("\n")
    frames = [(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)]
    ln = None
    for fn, loc in frames:
        if fn.startswith("phys_substep") and "kernels_physics.hpp" in loc:
            ln = int(loc.split(":")[-2])
    if ln is None:
        for fn, loc in frames:                  # --post: the epilogue's instructions by the line of post_body that (transitively) issued them
            if fn.startswith("post_body") and "kernels_step.hpp" in loc:
                bypost[int(loc.split(":")[-2])] += 1
        other["post_body (epilogue)" if any("post_body" in f[0] for f in frames) else "k_substeps (actuator net, loads, logs)"] += 1
    else:
        byphase[phase(ln)] += 1; byline[ln] += 1
print(name, "static VALU", nv)
for k, v in sorted(byphase.items()):
    print(f"  {k:28s} {v}")
print("  phys_substep total          ", sum(byphase.values()))
for k, v in other.items():
    print(f"  {k:28s} {v}")
if lines_rng:
    for ln in sorted(byline):
        if lines_rng[0] <= ln < lines_rng[1] and byline[ln] >= 4:
            print(f"    {ln:5d} {byline[ln]:4d}  {src[ln - 1].strip()[:120]}")
if "--post" in sys.argv:
    acc = 0
    for ln in sorted(bypost):
        acc += bypost[ln]
        if bypost[ln] >= 6:
            print(f"    {ln:5d} {bypost[ln]:4d}  {src_step[ln - 1].strip()[:140]}")
