"""MONAI-bundle ``preprocessing`` / ``augmentation`` dictionaries (reference
``src/segmantic/seg/monai_unet.py:232-262``): the resolver semantics pinned by the reference's own
tests ``tests/seg/test_unet.py:30-96``, on the reference's own fixture
(``tests/testing_data/config.json`` -> ``tests/golden/reference_bundle_config.json``)."""
import json

import pytest

from segmantic_amd.utils.bundle import (Compose, ConfigParser, TransformSpec, plan_augmentation,
                                        plan_preprocessing)


@pytest.fixture()
def options(golden_dir):
    return json.loads((golden_dir / "reference_bundle_config.json").read_text())


def test_load_preprocessing(options):          # reference test_load_preprocessing
    parser = ConfigParser({"image_key": "image", "preprocessing": options["preprocessing"]})
    parser.parse(True)
    transforms = parser.get_parsed_content("preprocessing")
    assert isinstance(transforms, Compose) and len(transforms) == 5
    assert [t.target for t in transforms] == ["LoadImaged", "EnsureChannelFirstd", "NormalizeIntensityd",
                                              "RandRotated", "EnsureTyped"]
    assert all(t.kwargs["keys"] == "image" for t in transforms)        # "@image_key" resolved
    assert transforms.transforms[2].kwargs["nonzero"] is True


def test_load_empty_preprocessing():            # reference test_load_empty_preprocessing
    parser = ConfigParser({"image_key": "image", "preprocessing": {}})
    parser.parse(True)
    transforms = parser.get_parsed_content("preprocessing")
    assert isinstance(transforms, dict) and len(transforms) == 0


def test_load_not_existing_preprocessing():     # reference test_load_not_existing_preprocessing
    parser = ConfigParser({"image_key": "image"})
    parser.parse(True)
    with pytest.raises(KeyError):
        parser.get_parsed_content("preprocessing")
    assert "preprocessing" not in parser


def test_load_disabled_preprocessing():         # reference test_load_disabled_preprocessing
    parser = ConfigParser({"image_key": "image",
                           "preprocessing": {"_target_": "DataStatsD", "_disabled_": True, "keys": "image"}})
    parser.parse(True)
    assert parser.get_parsed_content("preprocessing") is None


def test_load_postprocessing_reference_to_preprocessing(options):    # reference test_load_postprocessing
    parser = ConfigParser({k: options[k] for k in ("image_key", "preprocessing", "postprocessing")})
    parser.parse(True)
    post = parser.get_parsed_content("postprocessing")
    assert isinstance(post, Compose) and [t.name for t in post] == ["EnsureType", "Invert", "AsDiscrete"]
    inv = post.transforms[1]
    assert isinstance(inv.kwargs["transform"], Compose)              # "@preprocessing" -> the resolved Compose
    assert inv.kwargs["transform"] is parser.get_parsed_content("preprocessing")
    assert inv.kwargs["nearest_interp"] is False and post.transforms[2].kwargs["argmax"] is True


def test_trainer_entry_resolves_to_a_spec(options):                  # reference test_load_trainer (no Lightning here)
    parser = ConfigParser({"trainer": options["trainer"]})
    parser.parse(True)
    tr = parser.get_parsed_content("trainer")
    assert isinstance(tr, TransformSpec) and tr.target == "pytorch_lightning.Trainer"
    assert tr.kwargs["max_epochs"] == options["trainer"]["max_epochs"]


def test_expressions_and_cycles_are_refused():
    p = ConfigParser({"a": "$__import__('os').system('true')", "b": "@c", "c": "@b", "d": {"x": "@e#1"}, "e": [1, 2]})
    with pytest.raises(NotImplementedError):
        p.get_parsed_content("a")
    with pytest.raises(ValueError, match="circular"):
        p.get_parsed_content("b")
    assert p.get_parsed_content("d") == {"x": 2}


def test_plans_map_onto_the_device_pipeline(options):
    default_like = {"_target_": "Compose", "transforms": [
        {"_target_": "LoadImaged", "keys": ["image", "label"], "reader": "ITKReader", "ensure_channel_first": True},
        {"_target_": "Orientationd", "keys": ["image", "label"], "axcodes": "RAS"},
        {"_target_": "NormalizeIntensityd", "keys": "image", "nonzero": False, "channel_wise": True},
        {"_target_": "CropForegroundd", "keys": ["image", "label"], "source_key": "label"},
        {"_target_": "EnsureTyped", "keys": ["image", "label"]},
        {"_target_": "Spacingd", "keys": ["image", "label"], "pixdim": [1.0, 1.0, 2.5]}]}
    p = ConfigParser({"preprocessing": default_like})
    plan = plan_preprocessing(p.get_parsed_content("preprocessing"))
    assert plan == {"orientation": True, "normalize": True, "crop_foreground": True, "spacing": [1.0, 1.0, 2.5]}
    assert plan_preprocessing({}) is None and plan_preprocessing(None) is None
    # the reference's fixture asks for nonzero=True and a random rotation inside pre-processing: refused by name
    ref = ConfigParser({"image_key": "image", "preprocessing": options["preprocessing"]})
    with pytest.raises(ValueError, match="RandRotated"):
        plan_preprocessing(ref.get_parsed_content("preprocessing"))
    aug = ConfigParser({"augmentation": {"_target_": "Compose", "transforms": [
        {"_target_": "SpatialPadd", "keys": ["image", "label"], "spatial_size": [96, 96, 96]},
        {"_target_": "RandCropByLabelClassesd", "keys": ["image", "label"], "label_key": "label",
         "spatial_size": [96, 96, 96], "num_classes": 3, "num_samples": 2},
        {"_target_": "RandRotated", "keys": ["image", "label"], "prob": 0.2, "_disabled_": True},
        {"_target_": "RandFlipd", "keys": ["image", "label"], "prob": 0.2, "spatial_axis": 0}]}})
    plan = plan_augmentation(aug.get_parsed_content("augmentation"))
    assert plan["num_samples"] == 2 and plan["flip_prob"] == 0.2 and not plan["augment_spatial"]
    with pytest.raises(ValueError, match="RandGaussianNoised"):
        plan_augmentation(ConfigParser({"a": {"_target_": "Compose", "transforms": [
            {"_target_": "RandCropByLabelClassesd", "keys": "image"},
            {"_target_": "RandGaussianNoised", "keys": "image"}]}}).get_parsed_content("a"))
