"""Turns an entry of `_tables.SPEC` into the nested class tree the environments read.

The reference spells its task configurations as Python class bodies; here they are data (one table, `_tables.py`) and this
module is the only code: `cfg(name)` builds -- once, cached -- the class `name` derived from its base entry, with one nested
class per section.  Semantics that the class-body form has and the environments rely on:
  * a section marked `inherit` derives from the base entry's section of the same name, so unlisted attributes fall through;
    a section not marked so replaces the base's section wholesale (e.g. `rewards.scales` of a task);
  * `init_state` sections are also constructors of start states (`init_state(pos=..., rot=...)`), `obs` / `privileged_obs`
    list their switched-on components with `keys()`;
  * `BaseConfig()` instantiates the tree recursively (mqe/envs/base/base_config.py).
"""
from mqe.envs.base.base_config import BaseConfig


class SEC:
    """nested section: attrs = {name: value | SEC}"""
    def __init__(self, inherit, attrs):
        self.inherit, self.attrs = bool(inherit), attrs


class S:
    """start state of an actor; materialised as an instance of the Go1 init_state section"""
    def __init__(self, pos, rot, lin_vel, ang_vel):
        self.kw = dict(pos=pos, rot=rot, lin_vel=lin_vel, ang_vel=ang_vel)


class REF:
    """reference to the section class `section` of entry `entry`"""
    def __init__(self, entry, section):
        self.entry, self.section = entry, section


class _StateCtor:
    def __init__(self, pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]):
        self.pos, self.rot, self.lin_vel, self.ang_vel = pos, rot, lin_vel, ang_vel


class _Keys:
    def keys(self):
        return [k for k in dir(self.cfgs) if getattr(self.cfgs, k) == True and k]  # noqa: E712


_MIXIN = {"init_state": _StateCtor, "obs": _Keys, "privileged_obs": _Keys}
_CACHE = {"BaseConfig": BaseConfig}


def _value(v):
    if isinstance(v, S):
        return cfg("Go1Cfg").init_state(**v.kw)
    if isinstance(v, REF):
        return getattr(cfg(v.entry), v.section)
    if isinstance(v, list):
        return [_value(x) for x in v]
    return v


def _section(name, spec, base, qual, top):
    parent = getattr(base, name, None) if (spec.inherit and base is not None) else None
    bases = (parent,) if isinstance(parent, type) else ((_MIXIN[name],) if top and name in _MIXIN else ())
    ns = {"__qualname__": f"{qual}.{name}", "__module__": __name__}
    for k, v in spec.attrs.items():
        ns[k] = _section(k, v, parent, ns["__qualname__"], False) if isinstance(v, SEC) else _value(v)
    return type(name, bases, ns)


def cfg(name):
    if name in _CACHE:
        return _CACHE[name]
    from ._tables import SPEC
    base_name, sections = SPEC[name]
    base = cfg(base_name)
    ns = {"__module__": __name__}
    cls = type(name, (base,), ns)
    _CACHE[name] = cls                      # registered before the sections: S(...) inside Go1Cfg's own tasks refer back to it
    for k, v in sections.items():
        setattr(cls, k, _section(k, v, base, name, True) if isinstance(v, SEC) else _value(v))
    return cls
