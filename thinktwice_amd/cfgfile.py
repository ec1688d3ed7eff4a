"""Minimal `Config.fromfile` for the reference's python config files (train.py:116 `Config.fromfile(args.config)`,
configs/thinktwice.py:1-3 `_base_ = ['./_base_/default_runtime.py']`): executes the file, merges its `_base_` files
first (child keys override, dicts merge recursively, `_delete_=True` replaces), and returns an attribute dict, so
`build_model(cfg.model)` takes the reference config verbatim.  No mmcv dependency."""
import os
import types


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
    return out


def _load(path):
    path = os.path.abspath(path)
    ns = {"__file__": path, "__name__": "_tt_config_"}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config:
    @staticmethod
    def fromfile(path):
        return _wrap(_load(path))
