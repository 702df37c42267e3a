"""``segmantic-unet`` command line (reference ``src/segmantic/commands/monai_unet_cli.py``):
``train-config -c FILE [--print-defaults]``, ``train -d DATALIST ...``, ``predict -d DATALIST -m CKPT ...``.
The config file's keys are the keyword arguments of ``train()`` (unknown key -> ValueError)."""
from __future__ import annotations

import inspect
import json
from pathlib import Path
from typing import List, Optional

import typer

from ..image.labels import load_decathlon_tissuelist, load_tissue_list
from ..utils import config
from ..utils.cli import get_default_args, validate_args

app = typer.Typer()


def _monai_unet():
    from ..seg import monai_unet
    return monai_unet


def load_decathlon_datalist(datalist_file: Path, data_list_key: str = "training") -> List[dict]:
    """Decathlon datalist section with paths made absolute relative to the file's directory
    (what ``monai.data.load_decathlon_datalist`` returns for this use)."""
    datalist_file = Path(datalist_file)
    js = json.loads(datalist_file.read_text())
    if data_list_key not in js:
        raise ValueError(f'Data list {data_list_key} not specified in "{datalist_file}".')
    base = datalist_file.parent
    out = []
    for item in js[data_list_key]:
        if isinstance(item, str):
            item = {"image": item}
        d = {}
        for k, v in item.items():
            d[k] = str((base / v).resolve()) if isinstance(v, str) and not Path(v).is_absolute() else v
        out.append(d)
    return out


@app.command()
def train_config(
    config_file: Path = typer.Option(None, "--config-file", "-c", help="config file in json format"),
    print_defaults: bool = False,
) -> None:
    """Train UNet with configuration provided as json/yaml file

    To generate a default config:  --config-file my_config.json --print-defaults
    """
    mu = _monai_unet()
    sig = inspect.signature(mu.train)
    if print_defaults:
        config.dump(get_default_args(signature=sig), config_file=config_file)
        return
    if not config_file:
        raise ValueError("Invalid '--config-file' argument")
    args: dict = validate_args(config.load(config_file), signature=sig)
    mu.train(**args)


@app.command()
def train(
    datalist_file: Path = typer.Option(..., "--datalist", "-d", help="decathlon style datalist json file"),
    tissue_list: Optional[Path] = typer.Option(None, "--tissue-list", "-t", help="label descriptors in iSEG format"),
    output_dir: Path = typer.Option(Path("results"), "--output-dir", "-r",
                                    help="output directory where model checkpoints and logs are saved"),
    num_channels: int = 1,
    max_epochs: int = 600,
    gpu_ids: List[int] = [0],
) -> None:
    """Train UNet"""
    _monai_unet().train(datalist=datalist_file, tissue_list=tissue_list, num_channels=num_channels,
                        max_epochs=max_epochs, output_dir=output_dir, gpu_ids=gpu_ids)


@app.command()
def predict(
    datalist_file: Path = typer.Option(..., "--datalist", "-d", help="decathlon style datalist json file"),
    model_file: Path = typer.Option(..., "--model-file", "-m", help="saved model checkpoint"),
    tissue_list: Path = typer.Option(None, "--tissue-list", "-t", help="label descriptors in iSEG format"),
    results_dir: Path = typer.Option(None, "--results-dir", "-r", help="output directory"),
    spacing: List[float] = typer.Option([], "--spacing", help="if specified, the image is first resampled"),
    gpu_ids: List[int] = [0],
    datalist_key: str = "test",
) -> None:
    """Predict segmentations"""
    datalist = load_decathlon_datalist(datalist_file, data_list_key=datalist_key)
    test_images = [Path(d["image"]) for d in datalist]
    test_labels = [Path(d["label"]) for d in datalist if "label" in d]
    if tissue_list is not None:
        tissue_dict = load_tissue_list(tissue_list)
    else:
        tissue_dict = load_decathlon_tissuelist(datalist_file)
    _monai_unet().predict(model_file=model_file, test_images=test_images, test_labels=test_labels,
                          tissue_dict=tissue_dict, output_dir=results_dir, spacing=spacing,
                          gpu_ids=gpu_ids)


def main():
    app()


if __name__ == "__main__":
    main()
