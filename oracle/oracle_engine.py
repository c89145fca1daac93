"""TEST INFRASTRUCTURE: ctypes loader of the CPU oracle (oracle/libmqe_oracle*.so) with the same Python surface
as mqe.engine.hip_engine.HipEngine, on host memory.  Imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the package."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "multiagent-quadruped-environment_amd"))
from mqe.engine import abi  # noqa: E402
from mqe.engine.base import EngineBase, _NP_DT  # noqa: E402

_LIBS = {}


def build():
    subprocess.check_call(["make", "-C", HERE, "-s"])


def usable_cpus():
    """CPUs this process may use: the scheduler affinity, capped by the cgroup's CPU quota (cgroup v2 cpu.max / v1 cfs_quota_us).  libgomp sizes its
    team by the hardware threads it sees (256 on the GPU box, whose container is granted 16 CPUs: the checker then runs 18 x slower than on 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def load_library(f64=False):
    name = "libmqe_oracle_f64.so" if f64 else "libmqe_oracle.so"
    if os.environ.get("MQE_ORACLE_LIB"):          # the sanitizer build (oracle/Makefile: asan), tests/test_oracle_sanitized.py
        if f64:
            raise RuntimeError("MQE_ORACLE_LIB overrides the f32 oracle only; unset it to load the f64 build")
        name = os.environ["MQE_ORACLE_LIB"]
    if name not in _LIBS:
        path = os.path.join(HERE, name)
        if not os.path.isfile(path):
            build()
        _LIBS[name] = C.CDLL(path)
        if not os.environ.get("OMP_NUM_THREADS"):       # one OpenMP runtime per process: the setting holds for every build of the checker loaded afterwards
            _LIBS[name].mqo_set_num_threads(usable_cpus())
    return _LIBS[name]


class OracleEngine(EngineBase):
    prefix = "mqo_"
    device = "cpu"

    def __init__(self, desc, keepalive, f64=False, device="cpu"):
        self.torch_device = torch.device("cpu")
        super().__init__(load_library(f64), desc, keepalive)
        vp = C.c_void_p
        for name, args in (("policy_step", [vp, vp]), ("compute_torques", [vp]), ("simulate", [vp]),
                           ("post_decimation_step", [vp, C.c_int]), ("post_physics_step", [vp]), ("post_physics_stage", [vp, C.c_int]),
                           ("reset_all", [vp]), ("step", [vp, vp]), ("step_joint", [vp, vp]), ("hist_pos", [vp]), ("wrapper_eval", [vp, C.c_int])):
            f = getattr(self.lib, "mqo_" + name)
            f.argtypes, f.restype = args, C.c_int

    def _wrap(self, ptr, shape, dtype):
        n = int(np.prod(shape)) if len(shape) else 1
        ct = {0: C.c_float, 1: C.c_int32, 2: C.c_uint8}[dtype]
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(n, 1),))[:n].reshape(shape)
        return torch.from_numpy(arr)

    def policy_step(self, command):
        c = np.ascontiguousarray(command.detach().cpu().numpy(), np.float32)
        self._call("policy_step", C.c_void_p(c.ctypes.data))

    def compute_torques(self):
        self._call("compute_torques")

    def simulate(self):
        self._call("simulate")

    def post_decimation_step(self, i):
        self._call("post_decimation_step", int(i))

    def post_physics_step(self):
        self._call("post_physics_step")

    def post_physics_stage(self, stages):
        self._call("post_physics_stage", int(stages))

    def reset_all(self):
        self._call("reset_all")

    def step_command(self, command):
        """what mqe_step_command fuses, call by call (go1.py:35-62): policy, decimation x (torques, substep, logs), post-physics step"""
        self.policy_step(command)
        for k in range(self.desc.decimation):
            self.compute_torques()
            self.simulate()
            self.post_decimation_step(k)
        self.post_physics_step()

    def step_joint(self, actions12):
        a = np.ascontiguousarray(actions12.detach().cpu().numpy() if hasattr(actions12, "detach") else actions12, np.float32)
        self._call("step_joint", C.c_void_p(a.ctypes.data))

    def step(self, actions, between=None):
        a = np.ascontiguousarray(actions.detach().cpu().numpy(), np.float32)
        if between is not None:      # the HIP engine runs this between its policy and physics launches; here order is all there is
            between()
        self._call("step", C.c_void_p(a.ctypes.data))

    def wrapper_eval(self, is_reset):
        self._call("wrapper_eval", int(is_reset))

    def defender_command(self, out):
        o = np.zeros((self.desc.num_envs, 3), np.float32)
        self.lib.mqo_defender_command.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.mqo_defender_command(self.h, C.c_void_p(o.ctypes.data))
        out.copy_(torch.from_numpy(o))

    def debug_dynamics(self, env, robot):
        M = np.zeros((18, 18), np.float32)
        minv = np.zeros((18, 18), np.float32)
        nc = C.c_int(0)
        con = np.zeros((64, 8), np.float32)
        f = self.lib.mqo_debug_dynamics
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        f(self.h, int(env), int(robot), C.c_void_p(M.ctypes.data), C.c_void_p(minv.ctypes.data), C.byref(nc), C.c_void_p(con.ctypes.data))
        return M, minv, con[:nc.value]

    def history_sync(self):
        pass      # the oracle's layer 0 reads the f32 ring itself

    def render_depth(self, height, width, hfov_deg, pos, rpy, far=20.0, out=None):
        """mqo_render_depth: the scalar ray caster that DEFINES the forward depth image (the reference's rasteriser is closed): (R, H, W),
        negative depth along the optical axis, -inf = nothing within `far` -- what the HIP kernel is compared with (tests/test_camera_gpu.py)"""
        R = self.desc.num_envs * self.desc.num_agents
        img = np.zeros((R, int(height), int(width)), np.float32)
        f = self.lib.mqo_render_depth
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
        f.restype = C.c_int
        p3, r3 = (C.c_float * 3)(*[float(x) for x in pos]), (C.c_float * 3)(*[float(x) for x in rpy])
        rc = f(self.h, C.c_void_p(img.ctypes.data), int(height), int(width), float(hfov_deg), p3, r3, float(far))
        if rc != 0:
            raise RuntimeError(f"mqo_render_depth failed ({rc}): {self.lib.mqo_last_error().decode()}")
        return torch.from_numpy(img)

    def history(self):
        R = self.desc.num_envs * self.desc.num_agents
        out = np.zeros((R, 2100), np.float32)
        self.lib.mqo_history.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.mqo_history(self.h, C.c_void_p(out.ctypes.data))
        return torch.from_numpy(out)
