"""Dotted-path plugin glue kept from the reference (ldm/util.py:71-86): configs and
checkpoints name classes as `target: pkg.mod.Class` + `params: {...}`."""
import importlib
from inspect import isfunction


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if isfunction(d) else d


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{type(model).__name__} has {n * 1e-6:.2f} M params.")
    return n


def get_obj_from_str(string, reload=False):
    module_name, _, attr = string.rpartition(".")
    module = importlib.import_module(module_name)
    if reload:
        module = importlib.reload(module)
    return getattr(module, attr)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))
