import os, sys
ROOT = "/root/repo"
for p in ("multiagent-quadruped-environment_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from mqe.engine import abi
from test_gpu_parity import _pair, _randomize
eh, eo, d = _pair("go1gate", 256)
for seed, drop in ((1, 0.0), (2, 0.11), (3, 0.2)):
    _randomize(eh, eo, seed, drop=drop)
    print("MARK", seed, flush=True); sys.stderr.write("MARK %d\n" % seed); sys.stderr.flush()
    eh.simulate(); torch.cuda.synchronize()
    sys.stdout.flush()
    eo.simulate()
