"""Thin object over a library exporting the include/mqe_hip.h entry points with some symbol prefix."""
import ctypes as C

import numpy as np
import torch

from . import abi

_TORCH_DT = {0: torch.float32, 1: torch.int32, 2: torch.uint8}
_NP_DT = {0: np.float32, 1: np.int32, 2: np.uint8}


class _DevArray:
    """Non-owning device buffer exposed through __cuda_array_interface__ (zero-copy into torch)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class EngineBase:
    prefix = "mqe_"
    device = "cuda"

    def __init__(self, lib, desc, keepalive):
        self.lib, self.desc, self._keep = lib, desc, keepalive
        self.api = abi.bind(lib, self.prefix)
        h = C.c_void_p()
        self._check(self.api["sim_create"](C.byref(desc), C.byref(h)))
        self.h = h
        self._views = {}

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"{self.prefix}engine error {rc}: {self.api['last_error']().decode()}")

    def _stream(self):
        return None

    def _call(self, name, *args):
        f = getattr(self.lib, self.prefix + name)
        self._check(f(self.h, *args))

    def tensor(self, kind):
        if kind in self._views:
            return self._views[kind]
        v = abi.TensorView()
        self._check(self.api["sim_tensor"](self.h, kind, C.byref(v)))
        shape = [int(v.shape[i]) for i in range(v.ndim)]
        t = self._wrap(v.ptr, shape, v.dtype)
        self._views[kind] = t
        return t

    def _wrap(self, ptr, shape, dtype):
        raise NotImplementedError

    def close(self):
        if getattr(self, "h", None):
            self.api["sim_destroy"](self.h)
            self.h = None
