#!/usr/bin/env python3
"""LDS footprint of the physics kernel per scene shape, from phys_lds_layout() of csrc/kernels_physics.hpp itself (the function is
extracted from the header and compiled on the host): `python tools/dev/lds_layout.py`.  16 envs fit a CU below 10240 B."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "multiagent-quadruped-environment_amd", "csrc", "kernels_physics.hpp")).read()
fn = src[src.index("__host__ __device__ inline PhysLds phys_lds_layout"):]
fn = fn[:fn.index("\n}\n") + 3]
struct = re.search(r"struct PhysLds \{.*?\n\};", src, re.S).group(0)
defs = "\n".join(l for l in src.splitlines() if re.match(r"#define (SIDE_STRIDE|RS_STRIDE|RSB_STRIDE|LEG_STRIDE|LEGC_STRIDE|FCOL_STRIDE|BODY_STRIDE_OF|CON_STRIDE_OF)\b", l))
CASES = [  # name, A, P, ND, nbody, ndof, nsph, nprim, maxc, rowgs, pad   (32 feature points / 18 primitives per robot)
    ("go1gate", 2, 0, 24, 26, 36, 64, 36, 16, 1, 1), ("go1plane", 1, 0, 12, 13, 18, 32, 18, 8, 1, 1),
    ("go1seesaw", 2, 1, 25, 26, 37, 64, 36, 18, 1, 0), ("go1football-defender", 3, 1, 36, 40, 60, 97, 54, 26, 1, 0),
    ("go1sheep-hard", 2, 9, 24, 35, 63, 82, 36, 42, 1, 0), ("go1pushbox", 2, 1, 24, 27, 42, 72, 36, 20, 1, 0),
]
prog = "#include <cstdio>\n#define __host__\n#define __device__\n" + defs + "\n" + struct + "\ninline int mqe_maxpair(int maxc) { return maxc / 2; }\n" + fn + "\nint main() {\n"
for c in CASES:
    prog += '  { PhysLds L = phys_lds_layout(%s); printf("%-22s %%6d B  (16 per CU: %%s)\\n", L.total * 4, L.total * 4 <= 10240 ? "yes" : "no"); }\n' % (", ".join(map(str, c[1:])), c[0])
prog += "}\n"
with tempfile.TemporaryDirectory() as t:
    open(os.path.join(t, "l.cpp"), "w").write(prog)
    subprocess.check_call(["g++", "-O0", "-o", os.path.join(t, "l"), os.path.join(t, "l.cpp")])
    print(subprocess.check_output([os.path.join(t, "l")]).decode(), end="")
