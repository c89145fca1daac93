"""Host launch time per env.step() and the GPU cost of the sharded path (all-gather of the returned batch issued between policy\nand physics) with a world-size-1 RCCL group on one GPU: prints host ms/step and total ms/step for both.  Run on the GPU box."""
import os, sys, time, types, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import bench
from mqe.envs.utils import make_mqe_env, custom_cfg
from mqe.envs.go1.go1 import Go1
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), pg_options=opts)
N = 4096; dev = "cuda:0"
margs = bench.make_args("go1gate", N, 0, dev)
env, cfg = make_mqe_env("go1gate", margs, custom_cfg(margs))
A = env.num_agents
obs = env.reset()
D = obs.shape[1] * obs.shape[2] + A + 1
from mqe.engine import abi
L = env.env.engine.tensor(abi.T_WRAPPER_PACKED).numel()
for mode in ("plain", "dist"):
    gather = [torch.empty(1, L, device=dev) for _ in range(2)]
    pending = [None, None]; sent = [None, None]; ready = [None]
    def issue():
        if ready[0] is not None:
            b, snap = ready[0]
            if pending[b] is not None: pending[b].wait()
            sent[b] = snap
            pending[b] = dist.all_gather_into_tensor(gather[b], snap, async_op=True); ready[0] = None
    env.env.between_policy_and_physics = issue if mode == "dist" else None
    acts = [torch.rand(N, A, 3, device=dev) * 2 - 1 for _ in range(300)]
    torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(300):
            o, r, d, info = env.step(acts[t])
            if mode == "dist":
                ready[0] = (t & 1, env.returned_batch)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(mode, "host ms/step %.3f   total ms/step %.3f" % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
dist.destroy_process_group()
