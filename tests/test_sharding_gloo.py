"""Env sharding: world_size-2 gloo run on CPU.  Each rank owns a contiguous range of GLOBAL env ids;
results must not depend on the number of ranks, and the only collective is the all-gather of the returned batch."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "multiagent-quadruped-environment_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run(task, n_local, g0, steps, seed=5):
    _setup_paths()
    from helpers import make_desc
    from oracle_engine import OracleEngine
    from mqe.engine import abi
    d, k, _ = make_desc(task, n_local, env_id_offset=g0)
    e = OracleEngine(d, k)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    gN = 8
    g = torch.Generator().manual_seed(seed)
    outs = []
    for t in range(steps):
        a = (torch.rand(gN, Aw, 3, generator=g) * 2 - 1)[g0:g0 + n_local].contiguous()   # actions by global env id
        e.step(a)
        obs = e.tensor(abi.T_WRAPPER_OBS).reshape(n_local, -1)
        rew = e.tensor(abi.T_WRAPPER_REWARD).reshape(n_local, -1)
        done = e.tensor(abi.T_RESET_BUF).reshape(n_local, 1).float()
        outs.append(torch.cat([obs, rew, done], 1).clone())
    return torch.stack(outs)


def _worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = 8 // world
    packed = _run("go1gate", n_local, rank * n_local, 6)
    gathered = torch.empty(6, 8, packed.shape[-1])
    for t in range(6):
        dist.all_gather_into_tensor(gathered[t], packed[t].contiguous())
    if rank == 0:
        np.save(path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "g.npy")
        mp.spawn(_worker, args=(2, port, path), nprocs=2, join=True)
        sharded = np.load(path)
    single = _run("go1gate", 8, 0, 6).numpy()
    assert sharded.shape == single.shape
    np.testing.assert_allclose(sharded, single, atol=1e-6)
