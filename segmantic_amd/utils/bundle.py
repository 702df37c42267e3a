"""Resolver for the MONAI-bundle style ``preprocessing`` / ``augmentation`` dictionaries of the
``train()`` config schema (reference ``src/segmantic/seg/monai_unet.py:232-262``), restricted to
the transforms this build implements on the device.

The reference hands these dictionaries to ``monai.bundle.ConfigParser``; its semantics, as pinned by
the reference's own tests (``tests/seg/test_unet.py:30-96``), are restated here:

* ``"_target_": "Name"`` + keyword entries  -> an instance of the named component;
* ``"_disabled_": true``                     -> the component resolves to ``None``;
* ``"@key"`` / ``"@key#sub#0"``              -> the *resolved* content of another entry;
* a dictionary without ``_target_``          -> a dictionary of resolved entries (``{}`` stays ``{}``);
* a missing key                              -> ``KeyError`` (and ``key not in parser``);
* ``"$expression"`` strings (Python ``eval``) are refused: nothing in this build evaluates code
  taken from a configuration file.

Components are not MONAI objects but ``TransformSpec`` records (name + resolved keyword arguments)
and ``Compose`` (a list of them): ``plan_preprocessing`` / ``plan_augmentation`` translate such a
composition into the settings of the fixed on-device pipeline (``seg/pipeline.py``,
``seg/trainer.py``) and refuse, by name, whatever that pipeline cannot express.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

# MONAI transform names (with and without the dictionary-transform suffix spellings "d" / "D" /
# "Dict") that map onto stages of the device pipeline.
PREPROCESSING = ("LoadImage", "EnsureChannelFirst", "Orientation", "NormalizeIntensity", "CropForeground",
                 "EnsureType", "Spacing", "DataStats")
AUGMENTATION = ("SpatialPad", "RandCropByLabelClasses", "RandFlip", "RandRotate", "RandZoom",
                "RandAdjustContrast", "RandHistogramShift", "RandBiasField", "RandGibbsNoise",
                "RandKSpaceSpikeNoise", "EnsureType", "DataStats")
POSTPROCESSING = ("EnsureType", "Invert", "AsDiscrete", "SaveImage", "Activations")


def _base_name(name: str) -> str:
    name = name.rsplit(".", 1)[-1]
    for suf in ("Dict", "d", "D"):
        if name.endswith(suf) and name[:-len(suf)] in PREPROCESSING + AUGMENTATION + POSTPROCESSING:
            return name[:-len(suf)]
    return name


class TransformSpec:
    """A configured transform: MONAI name + resolved keyword arguments."""

    def __init__(self, target: str, kwargs: Dict[str, Any]):
        self.target, self.name, self.kwargs = target, _base_name(target), dict(kwargs)

    def __repr__(self):
        return f"{self.target}({', '.join(f'{k}={v!r}' for k, v in self.kwargs.items())})"


class Compose:
    def __init__(self, transforms: Optional[List[Any]] = None, **_unused):
        self.transforms = [t for t in (transforms or []) if t is not None]

    def flatten(self) -> "Compose":
        out: List[Any] = []
        for t in self.transforms:
            out.extend(t.flatten().transforms if isinstance(t, Compose) else [t])
        return Compose(out)

    def __len__(self):
        return len(self.transforms)

    def __iter__(self):
        return iter(self.transforms)


class ConfigParser:
    """``monai.bundle.ConfigParser`` surface used by the reference: ``parse(True)``,
    ``get_parsed_content(key)``, ``key in parser``."""

    def __init__(self, config: Optional[dict] = None):
        self.config = dict(config or {})
        self._cache: Dict[str, Any] = {}
        self._busy: set = set()

    def __contains__(self, key: str) -> bool:
        return key in self.config

    def parse(self, reset: bool = True) -> None:
        if reset:
            self._cache.clear()

    def get_parsed_content(self, key: str) -> Any:
        if key in self._cache:
            return self._cache[key]
        node: Any = self.config
        for part in key.split("#"):
            if isinstance(node, dict):
                if part not in node:
                    raise KeyError(f"config has no entry '{key}'")
                node = node[part]
            elif isinstance(node, (list, tuple)):
                node = node[int(part)]
            else:
                raise KeyError(f"config has no entry '{key}'")
        if key in self._busy:
            raise ValueError(f"circular reference through '@{key}'")
        self._busy.add(key)
        try:
            val = self._resolve(node)
        finally:
            self._busy.discard(key)
        self._cache[key] = val
        return val

    def _resolve(self, node: Any) -> Any:
        if isinstance(node, str):
            if node.startswith("@"):
                return self.get_parsed_content(node[1:])
            if node.startswith("$"):
                raise NotImplementedError(
                    f"segmantic_amd does not evaluate '$' expressions from configuration files: {node!r}")
            return node
        if isinstance(node, (list, tuple)):
            return [self._resolve(v) for v in node]
        if isinstance(node, dict):
            if node.get("_disabled_") is True or (isinstance(node.get("_disabled_"), str)
                                                   and node["_disabled_"].lower() == "true"):
                return None
            if "_target_" not in node:
                return {k: self._resolve(v) for k, v in node.items()}
            target = str(node["_target_"])
            kwargs = {k: self._resolve(v) for k, v in node.items() if not (k.startswith("_") and k.endswith("_"))}
            if _base_name(target) == "Compose":
                return Compose(**kwargs)
            return TransformSpec(target, kwargs)
        return node


def _check(comp, allowed, what):
    if comp is None or (isinstance(comp, dict) and not comp):
        return None
    if isinstance(comp, TransformSpec):
        comp = Compose([comp])
    if not isinstance(comp, Compose):
        raise ValueError(f"'{what}' must resolve to a Compose of transforms, got {type(comp).__name__}")
    flat = comp.flatten()
    for t in flat:
        if not isinstance(t, TransformSpec) or t.name not in allowed:
            raise ValueError(
                f"'{what}': transform {getattr(t, 'target', t)!r} is not available in segmantic_amd's on-device "
                f"pipeline; supported: {', '.join(a + 'd' for a in allowed)}")
    return flat


def plan_preprocessing(comp) -> Optional[dict]:
    """Compose -> settings of ``PredictPipeline`` (load -> RAS -> normalise -> crop foreground ->
    float32 -> optional Spacing); ``None`` = use the default pipeline.  The stage ORDER is fixed on
    the device, so a composition that orders the stages differently is refused."""
    flat = _check(comp, PREPROCESSING, "preprocessing")
    if flat is None:
        return None
    order = {n: i for i, n in enumerate(PREPROCESSING)}
    seq = [t for t in flat if t.name != "DataStats"]
    if [order[t.name] for t in seq] != sorted(order[t.name] for t in seq):
        raise ValueError("'preprocessing': the on-device pipeline runs LoadImage, EnsureChannelFirst, Orientation, "
                         "NormalizeIntensity, CropForeground, EnsureType, Spacing in this order")
    plan = {"orientation": False, "normalize": False, "crop_foreground": False, "spacing": []}
    for t in seq:
        kw = t.kwargs
        if t.name == "Orientation":
            if str(kw.get("axcodes", "RAS")).upper() != "RAS":
                raise ValueError("'preprocessing': Orientationd is implemented for axcodes='RAS'")
            plan["orientation"] = True
        elif t.name == "NormalizeIntensity":
            if kw.get("nonzero", False) or not kw.get("channel_wise", True) or kw.get("subtrahend") is not None \
                    or kw.get("divisor") is not None:
                raise ValueError("'preprocessing': NormalizeIntensityd is implemented for nonzero=False, "
                                 "channel_wise=True without fixed subtrahend / divisor (the reference's default)")
            plan["normalize"] = True
        elif t.name == "CropForeground":
            if kw.get("margin", 0) not in (0, [0, 0, 0]) or kw.get("k_divisible", 1) != 1:
                raise ValueError("'preprocessing': CropForegroundd is implemented without margin / k_divisible")
            plan["crop_foreground"] = True
        elif t.name == "Spacing":
            plan["spacing"] = [float(v) for v in kw["pixdim"]]
    return plan


def plan_augmentation(comp) -> Optional[dict]:
    """Compose -> {num_samples, flip_prob, augment_spatial, augment_intensity}; ``None`` = defaults."""
    flat = _check(comp, AUGMENTATION, "augmentation")
    if flat is None:
        return None
    names = [t.name for t in flat]
    if "RandCropByLabelClasses" not in names:
        raise ValueError("'augmentation': the training sampler needs RandCropByLabelClassesd (patch extraction)")
    plan = {"num_samples": 4, "flip_prob": 0.0, "augment_spatial": False, "augment_intensity": False,
            "spatial_size": None}
    for t in flat:
        kw = t.kwargs
        if t.name == "RandCropByLabelClasses":
            plan["num_samples"] = int(kw.get("num_samples", 1))
            plan["spatial_size"] = kw.get("spatial_size")
        elif t.name == "RandFlip":
            plan["flip_prob"] = float(kw.get("prob", 0.1))
        elif t.name in ("RandRotate", "RandZoom"):
            plan["augment_spatial"] = True
        elif t.name.startswith("Rand") and t.name not in ("RandFlip",):
            plan["augment_intensity"] = True
    return plan
