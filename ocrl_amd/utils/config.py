"""A small resolver for the reference's Hydra 1.x command-line grammar (hydra / omegaconf are not installed):

    python train_ocr.py ocr=slate ocr.slotattr.num_slots=6 dataset=random-N5C4S4S2 device=cuda:0 tags="slate"

Supported: `defaults:` lists with `_self_`, relative group files (`- _base`), group selection (`- ocr: ???` is
mandatory, chosen on the command line as `ocr=slate`), dotted `key=value` overrides with YAML-typed values,
`+key=value` additions, and `${a.b}` interpolation of scalar values.  The result is an attribute namespace that
supports `hasattr` probes the way the reference uses its DictConfig (ocrs/base.py:21-22,65-66)."""
import copy
import os
import re

import yaml


class Config:
    """attribute + item access; missing keys raise AttributeError (so `hasattr` works)"""

    def __init__(self, d=None):
        object.__setattr__(self, "_d", {})
        for k, v in (d or {}).items():
            self._d[k] = Config(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_d")[k]
        except KeyError:
            raise AttributeError(f"Missing key {k}") from None

    def __setattr__(self, k, v):
        self._d[k] = Config(v) if isinstance(v, dict) else v

    __getitem__ = __getattr__
    __setitem__ = __setattr__

    def __contains__(self, k):
        return k in self._d

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def get(self, k, default=None):
        return self._d.get(k, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self._d.items()}

    def __repr__(self):
        return f"Config({self.to_dict()})"


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _coerce(v):
    """PyYAML (YAML 1.1) reads `3e-4` as a string; OmegaConf reads it as a float — follow OmegaConf"""
    if isinstance(v, dict):
        return {k: _coerce(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_coerce(x) for x in v]
    if isinstance(v, str) and _FLOAT.match(v):
        return float(v)
    return v


def _load_file(path):
    if not os.path.isfile(path):
        raise FileNotFoundError(f"config file not found: {path}")
    with open(path) as f:
        return _coerce(yaml.safe_load(f) or {})


def _compose_file(config_dir, rel, choices, top):
    """returns the dict of one config file with its `defaults:` resolved.  rel: path without .yaml relative to config_dir"""
    raw = _load_file(os.path.join(config_dir, rel + ".yaml"))
    defaults = raw.pop("defaults", None)
    if defaults is None:
        return raw
    out = {}
    group_dir = os.path.dirname(rel)
    self_done = False
    for item in defaults:
        if item == "_self_":
            _merge(out, raw)
            self_done = True
        elif isinstance(item, str):
            _merge(out, _compose_file(config_dir, os.path.join(group_dir, item), choices, False))
        elif isinstance(item, dict):
            (group, choice), = item.items()
            choice = choices.get(group, choice) if top else choice
            if choice == "???" or choice is None:
                raise ValueError(f"You must specify '{group}', e.g. {group}=<option> (available: {available(config_dir, group)})")
            sub = _compose_file(config_dir, os.path.join(group_dir, group, str(choice)), choices, False)
            _merge(out.setdefault(group, {}), sub)
        else:
            raise ValueError(f"unsupported defaults entry {item!r} in {rel}.yaml")
    if not self_done:
        _merge(out, raw)
    return out


def available(config_dir, group):
    d = os.path.join(config_dir, group)
    if not os.path.isdir(d):
        return []
    return sorted(f[:-5] for f in os.listdir(d) if f.endswith(".yaml") and not f.startswith("_"))


def _set_dotted(d, key, value, create):
    parts = key.split(".")
    node = d
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            if not create:
                raise KeyError(f"Could not override '{key}': key '{p}' is not in the config (use +{key}=... to add it)")
            node[p] = {}
        node = node[p]
    if parts[-1] not in node and not create:
        raise KeyError(f"Could not override '{key}': no such key (use +{key}=... to add it)")
    node[parts[-1]] = value


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _interpolate(d, root):
    for k, v in list(d.items()):
        if isinstance(v, dict):
            _interpolate(v, root)
        elif isinstance(v, str) and "${" in v:
            def rep(m):
                node = root
                for p in m.group(1).split("."):
                    node = node[p]
                return str(node)
            d[k] = _INTERP.sub(rep, v)


def compose(config_dir, config_name, overrides=()):
    """-> Config.  overrides: iterable of 'group=choice', 'a.b=value', '+a.b=value'"""
    groups = {g for g in os.listdir(config_dir) if os.path.isdir(os.path.join(config_dir, g))}
    choices, sets = {}, []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not of the form key=value")
        key, val = ov.split("=", 1)
        add = key.startswith("+")
        key = key.lstrip("+")
        if key in groups and "." not in key:
            choices[key] = val
        else:
            sets.append((key, _coerce(yaml.safe_load(val)) if val != "" else "", add))
    cfg = _compose_file(config_dir, config_name, choices, True)
    for key, val, add in sets:
        _set_dotted(cfg, key, val, add)
    _interpolate(cfg, cfg)
    return Config(cfg)
