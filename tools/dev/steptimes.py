import sys, os, time, types
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT, "multiagent-quadruped-environment_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import make_args
from mqe.envs.utils import make_mqe_env, custom_cfg
dev="cuda:0"
margs = make_args("go1gate", 4096, 0, dev)
env,_ = make_mqe_env("go1gate", margs, custom_cfg(margs))
env.reset()
g = torch.Generator(device=dev).manual_seed(1234)
acts=[torch.rand(4096,2,3,device=dev,generator=g)*2-1 for _ in range(45)]
for t in range(5): env.step(acts[t])
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv)>1 else 0.0)
ev=[torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for t in range(40):
    env.step(acts[5+t]); ev[t+1].record()
torch.cuda.synchronize()
print([round(ev[i].elapsed_time(ev[i+1]),4) for i in range(40)])
