"""Minimal reader for the `conf/` tree with the reference's Hydra grammar (reference:
conf/config.yaml:1-11, compress.py:53): a `defaults` list of config groups, `# @package <group>`
files, `${a.b}` interpolation, and command-line overrides `key.sub=value`, `group=option`,
`+group=option`, comma sweeps `key=a,b` (expanded to the cartesian product like `-m`).
Hydra / omegaconf are not installed offline; this covers the subset `make fit` uses."""
import copy
import itertools
import os
import re
from typing import Any, Dict, List

import yaml


class Cfg(dict):
    """dict with attribute access and .get(), like the DictConfig the reference code expects."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _parse_scalar(s: str) -> Any:
    try:
        v = yaml.safe_load(s)
    except yaml.YAMLError:
        return s
    if isinstance(v, str):
        try:                      # YAML 1.1 reads "3e-4" as a string
            return float(v)
        except ValueError:
            return v
    return v


def _fix_floats(x):
    if isinstance(x, dict):
        return {k: _fix_floats(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_fix_floats(v) for v in x]
    if isinstance(x, str) and re.fullmatch(r"[-+]?\d+(\.\d*)?[eE][-+]?\d+", x):
        return float(x)
    return x


def _load_yaml(path: str):
    with open(path) as f:
        return _fix_floats(yaml.safe_load(f) or {})


def _set_path(cfg: dict, dotted: str, value):
    keys = dotted.split(".")
    d = cfg
    for k in keys[:-1]:
        if not isinstance(d.get(k), dict):
            d[k] = {}
        d = d[k]
    d[keys[-1]] = value


def _get_path(cfg: dict, dotted: str):
    d = cfg
    for k in dotted.split("."):
        d = d[k]
    return d


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve(cfg: dict, root: dict, cwd: str):
    def sub(val):
        for _ in range(8):
            if not isinstance(val, str) or "${" not in val:
                return val
            m = _INTERP.fullmatch(val)
            if m:                                     # whole-value interpolation keeps the type
                val = lookup(m.group(1))
                continue
            val = _INTERP.sub(lambda mm: str(lookup(mm.group(1))), val)
        return val

    def lookup(key):
        if key in ("cwd", "hydra:runtime.cwd"):
            return cwd
        return sub(_get_path(root, key))

    for k, v in list(cfg.items()):
        if isinstance(v, dict):
            _resolve(v, root, cwd)
        elif isinstance(v, list):
            cfg[k] = [sub(x) for x in v]
        else:
            cfg[k] = sub(v)


def expand_sweeps(overrides: List[str]) -> List[List[str]]:
    """`a=1,2 b=x` -> [[a=1,b=x],[a=2,b=x]] (Hydra multirun, reference Makefile:6 `-m`)."""
    axes = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        vals = val.split(",") if ("," in val and not val.startswith("[")) else [val]
        axes.append([f"{key}={v}" for v in vals])
    return [list(c) for c in itertools.product(*axes)] if axes else [[]]


def load_config(conf_dir: str, overrides: List[str] = (), cwd: str = None) -> Cfg:
    cwd = cwd or os.getcwd()
    base = _load_yaml(os.path.join(conf_dir, "config.yaml"))
    defaults = base.pop("defaults", [])
    groups: Dict[str, str] = {}
    for d in defaults:
        if isinstance(d, dict):
            (g, opt), = d.items()
            if not g.startswith("override"):
                groups[g] = opt
    value_overrides = []
    for ov in overrides:
        key, eq, val = ov.partition("=")
        if not eq:
            raise ValueError(f"override '{ov}' is not key=value")
        key = key.lstrip("+")
        if "." not in key and os.path.isdir(os.path.join(conf_dir, key)):
            groups[key] = val                          # config-group selection
        else:
            value_overrides.append((key, _parse_scalar(val)))
    cfg = dict(base)
    for g, opt in groups.items():
        path = os.path.join(conf_dir, g, f"{opt}.yaml")
        if not os.path.exists(path):
            have = sorted(f[:-5] for f in os.listdir(os.path.join(conf_dir, g)) if f.endswith(".yaml"))
            raise ValueError(f"config group '{g}' has no option '{opt}' (available: {have})")
        cfg[g] = _load_yaml(path)
    for key, val in value_overrides:
        _set_path(cfg, key, val)
    _resolve(cfg, cfg, cwd)
    return _wrap(cfg)
