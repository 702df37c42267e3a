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


# ---------------------------------------------------------------------------------------------
# Keyword whitelists.  The device pipeline is FIXED (seg/pipeline.py, seg/trainer.py, seg/augment.py):
# a configured transform is accepted only if every keyword it carries is known AND has the value the
# device pipeline implements; anything else raises, naming the transform and the keyword -- a config
# must never silently train with a different pipeline than the one it spells out.
# ---------------------------------------------------------------------------------------------
def _any(_v) -> bool:
    return True


def _is(*vals):
    return lambda v: any(v == x or (isinstance(x, float) and isinstance(v, (int, float)) and abs(v - x) < 1e-12)
                         for x in vals)


def _seq_is(*vals):
    """value given as a list / tuple equal to one of ``vals`` (compared as float tuples)"""
    def ok(v):
        try:
            t = tuple(float(x) for x in v)
        except TypeError:
            return False
        return any(t == tuple(float(x) for x in w) for w in vals)
    return ok


def _keys_in(*allowed):
    def ok(v):
        ks = [v] if isinstance(v, str) else list(v)
        return all(k in allowed for k in ks) and len(ks) > 0
    return ok


_IMG_LAB = _keys_in("image", "label")
_IMG = _keys_in("image")
_COMMON = {"allow_missing_keys": _is(False)}


def _validate(t: "TransformSpec", what: str, table: Dict[str, Any], required=()) -> None:
    table = dict(_COMMON, **table)
    for k, v in t.kwargs.items():
        if k not in table:
            raise ValueError(f"'{what}': {t.target}({k}=...) -- this argument is not implemented by segmantic_amd's "
                             f"on-device pipeline (accepted: {', '.join(sorted(table))})")
        if not table[k](v):
            raise ValueError(f"'{what}': {t.target}({k}={v!r}) differs from what segmantic_amd's on-device pipeline "
                             f"does; it would silently train / predict with another value, so it is refused")
    for k in required:
        if k not in t.kwargs:
            raise ValueError(f"'{what}': {t.target} needs an explicit '{k}' (its MONAI default differs from the "
                             f"on-device pipeline)")


def _spacing_modes(kw, what) -> bool:
    """-> label_nearest.  ``mode`` of Spacingd: one mode for all keys or one per key (image, label)."""
    keys = kw.get("keys", ["image", "label"])
    keys = [keys] if isinstance(keys, str) else list(keys)
    mode = kw.get("mode", "bilinear")
    modes = [mode] * len(keys) if isinstance(mode, str) else list(mode)
    if len(modes) != len(keys):
        raise ValueError(f"'{what}': Spacingd mode {mode!r} does not match keys {keys!r}")
    label_nearest = False
    for k, m in zip(keys, modes):
        m = str(m).lower()
        if k == "image" and m != "bilinear":
            raise ValueError(f"'{what}': Spacingd resamples the image with mode='bilinear' on the device, not {m!r}")
        if k == "label":
            if m not in ("bilinear", "nearest"):
                raise ValueError(f"'{what}': Spacingd label mode must be 'bilinear' or 'nearest', not {m!r}")
            label_nearest = m == "nearest"
    return label_nearest


def plan_preprocessing(comp) -> Optional[dict]:
    """Compose -> settings of ``PredictPipeline`` (load -> RAS -> normalise -> crop foreground ->
    float32 -> optional Spacing); ``None`` = use the default pipeline.  The stage ORDER is fixed on
    the device, so a composition that orders the stages differently is refused; so is any keyword
    the device stages do not implement (see ``_validate``)."""
    what = "preprocessing"
    flat = _check(comp, PREPROCESSING, what)
    if flat is None:
        return None
    order = {n: i for i, n in enumerate(PREPROCESSING)}
    seq = [t for t in flat if t.name != "DataStats"]
    if [order[t.name] for t in seq] != sorted(order[t.name] for t in seq):
        raise ValueError("'preprocessing': the on-device pipeline runs LoadImage, EnsureChannelFirst, Orientation, "
                         "NormalizeIntensity, CropForeground, EnsureType, Spacing in this order")
    plan = {"orientation": False, "normalize": False, "crop_foreground": False, "spacing": [],
            "spacing_label_nearest": False}
    for t in seq:
        kw = t.kwargs
        if t.name == "LoadImage":
            _validate(t, what, {"keys": _IMG_LAB, "reader": _is(None, "ITKReader", "itkreader", "NibabelReader"),
                                "image_only": _any, "ensure_channel_first": _any, "dtype": _any,
                                "meta_keys": _any, "meta_key_postfix": _any, "overwriting": _any,
                                "simple_keys": _any})
        elif t.name == "EnsureChannelFirst":
            _validate(t, what, {"keys": _IMG_LAB, "strict_check": _any, "channel_dim": _is(None, "no_channel", 0)})
        elif t.name == "Orientation":
            _validate(t, what, {"keys": _IMG_LAB, "axcodes": lambda v: str(v).upper() == "RAS",
                                "as_closest_canonical": _is(False), "labels": _any})
            plan["orientation"] = True
        elif t.name == "NormalizeIntensity":
            _validate(t, what, {"keys": _IMG, "nonzero": _is(False), "channel_wise": _is(True),
                                "subtrahend": _is(None), "divisor": _is(None), "dtype": _any},
                      required=("channel_wise",))
            plan["normalize"] = True
        elif t.name == "CropForeground":
            _validate(t, what, {"keys": _IMG_LAB, "source_key": _is("label", "image"), "allow_smaller": _any,
                                "margin": lambda v: v in (0, [0, 0, 0], (0, 0, 0), [0, 0], (0, 0)),
                                "k_divisible": _is(1), "select_fn": _is(None), "channel_indices": _is(None),
                                "mode": _any, "start_coord_key": _any, "end_coord_key": _any, "lazy": _is(False)},
                      required=("source_key",))
            plan["crop_foreground"] = True
        elif t.name == "EnsureType":
            _validate(t, what, {"keys": _IMG_LAB, "dtype": _any, "device": _any, "data_type": _is("tensor"),
                                "wrap_sequence": _any, "track_meta": _any})
        elif t.name == "Spacing":
            _validate(t, what, {"keys": _IMG_LAB, "pixdim": _any, "mode": _any, "padding_mode": _is("border"),
                                "diagonal": _is(False), "align_corners": _any, "dtype": _any,
                                "scale_extent": _is(False), "recompute_affine": _is(False),
                                "min_pixdim": _is(None), "max_pixdim": _is(None), "ensure_same_shape": _any,
                                "lazy": _is(False)}, required=("pixdim",))
            plan["spacing"] = [float(v) for v in kw["pixdim"]]
            plan["spacing_label_nearest"] = _spacing_modes(kw, what)
    return plan


_SPATIAL_MODES = lambda v: all(str(m).lower() in ("nearest", "bilinear", "area", "trilinear")           # noqa: E731
                               for m in ([v] if isinstance(v, str) else v))
_INTENSITY = {
    "RandAdjustContrast": {"keys": _IMG, "prob": _is(0.2), "gamma": _seq_is((0.5, 4.5)),
                           "invert_image": _is(False), "retain_stats": _is(False)},
    "RandHistogramShift": {"keys": _IMG, "prob": _is(0.2), "num_control_points": _is(10)},
    "RandBiasField": {"keys": _IMG, "prob": _is(0.2), "degree": _is(3), "coeff_range": _seq_is((0.0, 0.1)),
                      "dtype": _any},
    "RandGibbsNoise": {"keys": _IMG, "prob": _is(0.2), "alpha": _seq_is((0.0, 1.0))},
    "RandKSpaceSpikeNoise": {"keys": _IMG, "prob": _is(0.2), "intensity_range": _is(None),
                             "channel_wise": _is(True)},
}


def plan_augmentation(comp) -> Optional[dict]:
    """Compose -> {num_samples, flip_prob, flip_axes, augment_spatial, augment_intensity, spatial_size,
    num_classes}; ``None`` = defaults.  The device sampler implements the reference's
    ``default_augmentation`` (``monai_unet.py:178-219``) and nothing else: the spatial block is the
    three RandRotated + RandZoomd with the reference's arguments or absent, the intensity block the
    five transforms with the reference's arguments or absent, RandFlipd one entry per spatial axis
    with one probability; partial blocks and other argument values are refused."""
    what = "augmentation"
    flat = _check(comp, AUGMENTATION, what)
    if flat is None:
        return None
    names = [t.name for t in flat]
    if "RandCropByLabelClasses" not in names:
        raise ValueError("'augmentation': the training sampler needs RandCropByLabelClassesd (patch extraction)")
    plan = {"num_samples": 4, "flip_prob": 0.0, "flip_axes": [], "augment_spatial": False,
            "augment_intensity": False, "spatial_size": None, "num_classes": None}
    flips, rots, zooms, intens = [], [], [], {}
    for t in flat:
        kw = t.kwargs
        if t.name == "SpatialPad":
            _validate(t, what, {"keys": _IMG_LAB, "spatial_size": _any, "method": _is("symmetric"),
                                "mode": _is("constant"), "lazy": _is(False)})
            plan["pad_size"] = kw.get("spatial_size")
        elif t.name == "RandCropByLabelClasses":
            _validate(t, what, {"keys": _IMG_LAB, "label_key": _is("label"), "spatial_size": _any,
                                "num_samples": _any, "num_classes": _any, "ratios": _any,
                                "image_key": _is(None), "image_threshold": _any, "indices_key": _is(None),
                                "allow_smaller": _is(False), "warn": _any, "max_samples_per_class": _is(None),
                                "lazy": _is(False)}, required=("ratios",))
            r = [float(v) for v in kw["ratios"]]
            if len(r) < 2 or r[0] != 0.0 or any(v != r[1] or v <= 0 for v in r[1:]):
                raise ValueError("'augmentation': RandCropByLabelClassesd ratios must be [0, 1, 1, ...] (no background "
                                 f"centres, all other classes alike) -- the sampler the device implements; got {kw['ratios']!r}")
            if kw.get("num_classes") is not None and int(kw["num_classes"]) != len(r):
                raise ValueError("'augmentation': RandCropByLabelClassesd num_classes does not match len(ratios)")
            plan["num_classes"] = len(r)
            plan["num_samples"] = int(kw.get("num_samples", 1))
            plan["spatial_size"] = kw.get("spatial_size")
        elif t.name == "RandFlip":
            _validate(t, what, {"keys": _IMG_LAB, "prob": _any, "spatial_axis": lambda v: v in (0, 1, 2),
                                "lazy": _is(False)}, required=("spatial_axis",))
            flips.append((int(kw["spatial_axis"]), float(kw.get("prob", 0.1))))
        elif t.name == "RandRotate":
            _validate(t, what, {"keys": _IMG_LAB, "prob": _is(0.2), "range_x": _is(0.0, 0.4), "range_y": _is(0.0, 0.4),
                                "range_z": _is(0.0, 0.4), "keep_size": _is(True), "mode": _SPATIAL_MODES,
                                "padding_mode": _is("border"), "align_corners": _is(False), "dtype": _any,
                                "lazy": _is(False)}, required=("prob",))
            rots.append(tuple(ax for ax in "xyz" if float(kw.get(f"range_{ax}", 0.0)) != 0.0))
        elif t.name == "RandZoom":
            _validate(t, what, {"keys": _IMG_LAB, "prob": _is(0.2), "min_zoom": _is(0.8), "max_zoom": _is(1.3),
                                "mode": _SPATIAL_MODES, "padding_mode": _is("edge"), "align_corners": _is(None),
                                "keep_size": _is(True), "dtype": _any, "lazy": _is(False)},
                      required=("prob", "min_zoom", "max_zoom"))
            zooms.append(t)
        elif t.name in _INTENSITY:
            _validate(t, what, _INTENSITY[t.name], required=("prob",))
            if t.name in intens:
                raise ValueError(f"'augmentation': {t.target} appears twice")
            intens[t.name] = t
        elif t.name == "EnsureType":
            _validate(t, what, {"keys": _IMG_LAB, "dtype": _any, "device": _any, "data_type": _is("tensor"),
                                "wrap_sequence": _any, "track_meta": _any})
        # DataStats: prints only
    if flips:
        axes = sorted(a for a, _ in flips)
        probs = {p for _, p in flips}
        if axes not in ([0, 1], [0, 1, 2]) or len(probs) != 1:
            raise ValueError("'augmentation': RandFlipd must be configured once per spatial axis (spatial_axis 0, 1[, 2]) "
                             f"with one probability -- the device flips every axis independently; got {flips!r}")
        plan["flip_prob"], plan["flip_axes"] = probs.pop(), axes
    if rots or zooms:
        if sorted(rots) != [("x",), ("y",), ("z",)] or len(zooms) != 1:
            raise ValueError("'augmentation': the device implements the reference's spatial block as a whole -- "
                             "RandRotated(prob=0.2, range_z=0.4), RandRotated(range_x=0.4), RandRotated(range_y=0.4) and "
                             "RandZoomd(prob=0.2, min_zoom=0.8, max_zoom=1.3) -- or none of it")
        plan["augment_spatial"] = True
    if intens:
        if set(intens) != set(_INTENSITY):
            raise ValueError("'augmentation': the device implements the reference's intensity block as a whole ("
                             + ", ".join(n + "d" for n in _INTENSITY) + f") or none of it; got only {sorted(intens)}")
        plan["augment_intensity"] = True
    return plan
