"""Signature <-> config-dict helpers (reference ``src/segmantic/utils/cli.py:6-44``): the keyword
signature of ``train()`` is the config schema; unknown keys are rejected, Path-annotated values
are cast."""
import inspect
from pathlib import Path
from typing import Any, Dict


def is_path(param: inspect.Parameter) -> bool:
    ann = param.annotation
    return ann is not inspect.Parameter.empty and inspect.isclass(ann) and issubclass(ann, Path)


def cast_from_path(v: Any, param: inspect.Parameter) -> Any:
    return str(v) if v and is_path(param) else v


def cast_to_path(v: Any, param: inspect.Parameter) -> Any:
    return Path(v) if v and is_path(param) else v


def get_default_args(signature: inspect.Signature) -> Dict[str, Any]:
    out = {}
    for k, v in signature.parameters.items():
        if v.default is not inspect.Parameter.empty:
            d = cast_from_path(v.default, v)
            out[k] = list(d) if isinstance(d, tuple) else d
        else:
            out[k] = f"<required option: {v.annotation.__name__}>"
    return out


def validate_args(args: Dict[str, Any], signature: inspect.Signature) -> Dict[str, Any]:
    valid = {}
    for k in args:
        if k not in signature.parameters:
            raise ValueError(f"Unexpected argument {k}")
        valid[k] = cast_to_path(args[k], signature.parameters[k])
    return valid


__all__ = ("get_default_args", "validate_args", "is_path")
