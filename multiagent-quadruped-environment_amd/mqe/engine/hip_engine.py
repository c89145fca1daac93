"""Loader of the HIP engine (csrc/ -> libmqe_hip.so).  There is NO CPU fallback: without the built library or
without a GPU this raises -- the product path never silently runs anything else."""
import ctypes as C
import os

import numpy as np
import torch

from . import abi
from .base import EngineBase, _DevArray, _TORCH_DT

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "csrc", "libmqe_hip.so")


def load_library():
    global _LIB
    if _LIB is None:
        path = os.environ.get("MQE_HIP_LIB", LIB_PATH)       # experiments: an alternative build of the same ABI
        if path != LIB_PATH:
            import sys
            print(f"mqe: override in effect: MQE_HIP_LIB={path} (not the in-tree build)", file=sys.stderr)      # never silent
            _LIB = C.CDLL(path)
            return _LIB
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"HIP engine not built: {LIB_PATH} missing (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        _LIB = C.CDLL(LIB_PATH)
    return _LIB


class HipEngine(EngineBase):
    prefix = "mqe_"
    device = "cuda"

    def __init__(self, desc, keepalive, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("mqe HIP engine needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.torch_device = torch.device(device)
        torch.cuda.set_device(self.torch_device)
        super().__init__(load_library(), desc, keepalive)
        lib = self.lib
        vp = C.c_void_p
        for name, args in (("policy_step", [vp, vp, vp]), ("compute_torques", [vp, vp]), ("simulate", [vp, vp]),
                           ("post_decimation_step", [vp, C.c_int, vp]), ("post_physics_step", [vp, vp]), ("post_physics_stage", [vp, C.c_int, vp]),
                           ("reset_all", [vp, vp]), ("step", [vp, vp, vp]), ("step_begin", [vp, vp, vp]), ("step_end", [vp, vp]), ("step_head", [vp, vp, vp]), ("step_tail", [vp, vp]), ("set_return_buffer", [vp, vp]), ("step_joint", [vp, vp, vp]), ("step_command", [vp, vp, vp]), ("defender_command", [vp, vp, vp]),
                           ("wrapper_eval", [vp, C.c_int, vp]),
                           ("debug_dynamics", [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int), vp]),
                           ("debug_stop_phase", [vp, C.c_int]),
                           ("debug_wave_times", [vp, vp]), ("debug_tail_times", [vp, vp]),
                           ("debug_phase_times", [vp, vp]), ("debug_epilogue_times", [vp, vp]),
                           ("history_sync", [vp, vp]),
                           ("state_save", [vp, vp, vp]), ("state_load", [vp, vp, vp]),
                           ("profile_enable", [vp, C.c_int]),
                           ("profile_read", [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)])):
            f = getattr(lib, "mqe_" + name)
            f.argtypes, f.restype = args, C.c_int

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.torch_device).cuda_stream)

    def _wrap(self, ptr, shape, dtype):
        typestr = {0: "<f4", 1: "<i4", 2: "|u1"}[dtype]
        return torch.as_tensor(_DevArray(ptr, shape, typestr), device=self.torch_device)

    def policy_step(self, command):
        assert command.is_cuda and command.dtype == torch.float32 and command.is_contiguous()
        self._call("policy_step", C.c_void_p(command.data_ptr()), self._stream())
        self._n_policy = getattr(self, "_n_policy", 0) + 1

    def compute_torques(self):
        self._call("compute_torques", self._stream())

    def simulate(self):
        self._call("simulate", self._stream())

    def post_decimation_step(self, i):
        self._call("post_decimation_step", int(i), self._stream())

    def post_physics_step(self):
        self._call("post_physics_step", self._stream())

    def post_physics_stage(self, stages):
        """mqe_post_physics_stage: an OR of abi.POST_* (post_physics_step in the stages the reference's method has)"""
        self._call("post_physics_stage", int(stages), self._stream())

    def reset_all(self):
        self._call("reset_all", self._stream())

    def step_command(self, command):
        """Fused Go1.step for control type C (mqe_step_command): (R, num_command_dims) per-robot command rows on the device."""
        assert command.is_cuda and command.dtype == torch.float32 and command.is_contiguous()
        self._call("step_command", C.c_void_p(command.data_ptr()), self._stream())
        self._n_policy = getattr(self, "_n_policy", 0) + 1

    def step_joint(self, actions12):
        """Fused step for control types P / V / T: (R, 12) joint-space actions on the device."""
        assert actions12.is_cuda and actions12.dtype == torch.float32 and actions12.is_contiguous()
        self._call("step_joint", C.c_void_p(actions12.data_ptr()), self._stream())

    def step(self, actions, between=None, before_tail=None):
        """mqe_step; with `between` (a callable) the two halves mqe_step_begin / mqe_step_end with the callable's own
        launches placed after the policy kernels and before the physics kernel; with `before_tail` as well the policy itself in two
        parts, mqe_step_head / mqe_step_tail, with that callable's launches after layer 0 (see include/mqe_hip.h)."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        if between is None and before_tail is None:
            self._call("step", C.c_void_p(actions.data_ptr()), self._stream())
        elif before_tail is not None:
            self._call("step_head", C.c_void_p(actions.data_ptr()), self._stream())
            try:
                before_tail()
            finally:
                self._call("step_tail", self._stream())
                try:
                    if between is not None:
                        between()
                finally:
                    self._call("step_end", self._stream())
        else:
            self._call("step_begin", C.c_void_p(actions.data_ptr()), self._stream())
            try:
                between()
            finally:
                self._call("step_end", self._stream())
        self._n_policy = getattr(self, "_n_policy", 0) + 1

    def set_return_buffer(self, packed):
        """mqe_set_return_buffer: the following launches write obs | reward | done into `packed` (None: the engine's own buffer)"""
        if packed is not None:
            assert packed.is_cuda and packed.dtype == torch.float32 and packed.is_contiguous()
            assert packed.numel() >= self.tensor(abi.T_WRAPPER_PACKED).numel()
        self._call("set_return_buffer", C.c_void_p(packed.data_ptr() if packed is not None else None))

    def defender_command(self, out):
        self._call("defender_command", C.c_void_p(out.data_ptr()), self._stream())

    def wrapper_eval(self, is_reset):
        self._call("wrapper_eval", int(is_reset), self._stream())

    def save_state(self):
        """Checkpoint: every state buffer of the handle + its ring positions as one host blob (numpy uint8); see mqe_state_save.
        Not part of it: a caller-owned return buffer (mqe_set_return_buffer) and the wrappers' host-side counters."""
        self.lib.mqe_state_size.argtypes, self.lib.mqe_state_size.restype = [C.c_void_p], C.c_longlong
        blob = np.empty(int(self.lib.mqe_state_size(self.h)) + 8, np.uint8)
        self._call("state_save", C.c_void_p(blob.ctypes.data), self._stream())
        blob[-8:] = np.frombuffer(np.int64(getattr(self, "_n_policy", 0)).tobytes(), np.uint8)      # host-side frame counter of history()
        return blob

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        self.lib.mqe_state_size.argtypes, self.lib.mqe_state_size.restype = [C.c_void_p], C.c_longlong
        want = int(self.lib.mqe_state_size(self.h)) + 8
        if blob.nbytes != want:     # the C side checks the header against the handle, not the length of the caller's buffer
            raise ValueError(f"checkpoint blob has {blob.nbytes} bytes, this handle's state takes {want} (truncated file or another scene shape)")
        self._call("state_load", C.c_void_p(blob.ctypes.data), self._stream())
        self._n_policy = int(np.frombuffer(blob[-8:].tobytes(), np.int64)[0])

    def render_depth(self, height, width, hfov_deg, pos, rpy, far=20.0, out=None):
        """forward depth images of every robot from the current state: (R, height, width) device tensor, negative depth along the optical
        axis, -inf = nothing within `far` (mqe_render_depth)"""
        R = self.desc.num_envs * self.desc.num_agents
        if out is None:
            out = torch.empty(R, int(height), int(width), dtype=torch.float32, device=self.torch_device)
        f = self.lib.mqe_render_depth
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_void_p]
        f.restype = C.c_int
        p3, r3 = (C.c_float * 3)(*[float(x) for x in pos]), (C.c_float * 3)(*[float(x) for x in rpy])
        rc = f(self.h, C.c_void_p(out.data_ptr()), int(height), int(width), float(hfov_deg), p3, r3, float(far), self._stream())
        if rc != 0:
            raise RuntimeError(f"mqe_render_depth failed ({rc}): {self.lib.mqe_last_error().decode()}")
        return out

    def history_sync(self):
        """after writing tensor(T_HISTORY): the compact layer-0 operand is rebuilt from the ring (mqe_history_sync)"""
        self._call("history_sync", self._stream())

    def history(self):
        """(R, 2100) time-ordered locomotion history gathered from the ring (host-side bookkeeping of the slot)."""
        h = self.tensor(abi.T_HISTORY)
        pos = getattr(self, "_n_policy", 0) % abi.HIST
        idx = (torch.arange(abi.HIST, device=h.device) + pos) % abi.HIST
        return h[:, idx, :70].reshape(h.shape[0], -1)

    def debug_dynamics(self, env, robot):
        minv = np.zeros((18, 18), np.float32)
        nc = C.c_int(0)
        con = np.zeros((64, 8), np.float32)
        torch.cuda.synchronize()
        self._call("debug_dynamics", int(env), int(robot), C.c_void_p(minv.ctypes.data), C.byref(nc), C.c_void_p(con.ctypes.data))
        return minv, con[:nc.value]

    def profile_enable(self, on=True):
        """on: False/0 = off, True/1 = bracket every fused step with HIP events, k > 1 = one fused step per period of k (the THIRD of each
        period, so that a run's first steps -- allocator and clock warm-up -- are never the sample; a run shorter than three steps with k > 2
        records nothing: profile_read() then returns cnt == 0 and callers must handle that)."""
        self._call("profile_enable", int(on))

    def profile_read(self, n=16):
        buf = (C.c_float * n)()
        cnt = C.c_int(0)
        self._call("profile_read", buf, n, C.byref(cnt))
        return list(buf), cnt.value
