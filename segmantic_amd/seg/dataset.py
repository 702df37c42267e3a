"""Image / label file lists (reference ``src/segmantic/seg/dataset.py``): directory pairing,
train / validation split, decathlon-style JSON with globs.  Host-side bookkeeping only."""
from __future__ import annotations

import json
import random
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

from ..utils.file_iterators import find_matching_files
from ..utils.json import PathEncoder


def create_data_dict(list_to_convert, data_dir: Path, data_dicts: list) -> list:
    """Expand (possibly glob) image / label entries relative to ``data_dir`` (``:14-37``)."""
    for element in list_to_convert:
        if Path(element["image"]).is_absolute():
            image_files = [Path(element["image"])]
            label_files = [Path(element["label"])]
        else:
            image_files = list(data_dir.glob(element["image"]))
            label_files = list(data_dir.glob(element["label"]))
        assert len(image_files) == len(label_files)
        for i, o in zip(sorted(image_files), sorted(label_files)):
            data_dicts.append({"image": i, "label": o})
    return data_dicts


class PairedDataSet:
    def __init__(self, image_dir: Optional[Path] = None, image_glob: str = "*.nii.gz",
                 labels_dir: Optional[Path] = None, labels_glob: str = "*.nii.gz", *,
                 valid_split: float = 0.2, shuffle: bool = True, random_seed: int = None,
                 max_files: int = 0):
        dd = self.create_data_dict(image_dir, image_glob, labels_dir, labels_glob)
        self._create_split(dd, valid_split, shuffle, random_seed, max_files)

    def training_files(self) -> Sequence[Dict[str, Path]]:
        return self._train_files

    def validation_files(self) -> Sequence[Dict[str, Path]]:
        return self._val_files

    def test_files(self) -> Sequence[Dict[str, Path]]:
        return self._test_files

    def _create_split(self, data_dicts, valid_split, shuffle, random_seed=None, max_files=0,
                      test_data_dicts=None):
        self._test_files = test_data_dicts or []
        if shuffle:
            random.Random(random_seed).shuffle(data_dicts)
        num_total = len(data_dicts)
        if max_files > 0:
            num_total = min(num_total, max_files)
        num_valid = int(valid_split * num_total)
        if num_total > 1 and valid_split > 0:
            num_valid = max(num_valid, 1)
        self._train_files = data_dicts[num_valid:num_total]
        self._val_files = data_dicts[:num_valid]

    def check_matching_filenames(self):
        for d in self._train_files + self._val_files:
            a = d["image"].stem.replace(".nii", "").lower()
            b = d["label"].stem.replace(".nii", "").lower()
            if not ((a in b) or (b in a)):
                raise RuntimeError(
                    f"The pair image/label pair {d['image']} : {d['label']} doesn't correspond.")

    def dump_dataset(self) -> str:
        return json.dumps({"training": self._train_files, "validation": self._val_files,
                           "test": [t["image"] for t in self._test_files]}, cls=PathEncoder)

    @staticmethod
    def create_data_dict(image_dir=None, image_glob="*.nii.gz", labels_dir=None,
                         labels_glob="*.nii.gz") -> List[Dict[str, Path]]:
        out: List[Dict[str, Path]] = []
        if image_dir is None or labels_dir is None:
            return out
        image_dir, labels_dir = Path(image_dir), Path(labels_dir)
        assert image_dir.is_dir() and labels_dir.is_dir()
        if Path(image_glob).is_absolute():
            image_glob = str(Path(image_glob).relative_to(image_dir))
        if Path(labels_glob).is_absolute():
            labels_glob = str(Path(labels_glob).relative_to(labels_dir))
        for p in find_matching_files([image_dir / image_glob, labels_dir / labels_glob]):
            out.append({"image": p[0], "label": p[1]})
        return out

    @staticmethod
    def kfold_crossval(num_splits: int, data_dicts, output_dir: Path, test_data_dicts=None,
                       shuffle: bool = True, random_seed: int = None) -> list:
        """Contiguous k folds (sklearn ``KFold(n_splits)`` without shuffling: the first
        ``n % k`` folds get one extra sample)."""
        if shuffle:
            random.Random(random_seed).shuffle(data_dicts)
        output_dir = Path(output_dir)
        output_dir.mkdir(exist_ok=True, parents=True)
        n = len(data_dicts)
        sizes = [n // num_splits + (1 if i < n % num_splits else 0) for i in range(num_splits)]
        paths, start = [], 0
        for count, sz in enumerate(sizes):
            val_idx = set(range(start, start + sz))
            ds = PairedDataSet()
            ds._train_files = [data_dicts[i] for i in range(n) if i not in val_idx]
            ds._val_files = [data_dicts[i] for i in sorted(val_idx)]
            ds._test_files = test_data_dicts or []
            path = output_dir / f"fold_{count}.json"
            path.write_text(ds.dump_dataset())
            paths.append(path)
            start += sz
        return paths

    @staticmethod
    def load_from_json(datalist_paths: Union[Path, List[Path]]):
        if isinstance(datalist_paths, (Path, str)):
            datalist_paths = [datalist_paths]
        train: list = []
        val: list = []
        test: list = []
        for json_path in [Path(f) for f in datalist_paths]:
            ds = json.loads(json_path.read_text())
            train = create_data_dict(ds["training"], json_path.parent, train)
            val = create_data_dict(ds["validation"], json_path.parent, val)
            test = [{"image": Path(f)} for f in ds.get("test", [])]
        out = PairedDataSet()
        out._train_files, out._val_files, out._test_files = train, val, test
        return out
