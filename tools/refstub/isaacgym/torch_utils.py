"""Quaternion helpers (xyzw) with the semantics the reference relies on.

Restated from the standard identities; the same functions live in the product package
(mqe/utils/torch_utils.py) and are pinned there by analytic tests.  We import the product
copy so that the golden vectors and the product share one definition of these external
(Isaac Gym) helpers.
"""
import os
import sys

_PKG = os.path.join(os.path.dirname(__file__), "..", "..", "..", "multiagent-quadruped-environment_amd", "mqe", "utils")
import importlib.util as _ilu

_spec = _ilu.spec_from_file_location("_mqe_hip_torch_utils", os.path.join(_PKG, "torch_utils.py"))
_m = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_m)
for _k in dir(_m):
    if not _k.startswith("_"):
        globals()[_k] = getattr(_m, _k)
