"""Tissue-list IO used to derive ``num_classes`` (reference ``src/segmantic/image/labels.py:89-117``)."""
import json
from pathlib import Path
from typing import Dict


def load_tissue_list(file_name: Path) -> Dict[str, int]:
    """iSEG format: every line starting with 'C' names the next label id; Background = 0."""
    tissue_label_map = {"Background": 0}
    next_id = 1
    with open(file_name) as f:
        for line in f.readlines():
            if line.startswith("C"):
                tissue = line.strip().rsplit(" ", 1)[-1].rstrip()
                if tissue in tissue_label_map:
                    raise KeyError(f"duplicate label '{tissue}' found in '{file_name}'")
                tissue_label_map[tissue] = next_id
                next_id += 1
    return tissue_label_map


def save_tissue_list(tissue_label_map: Dict[str, int], tissue_list_file_name: Path) -> None:
    """Write an iSEG tissue list with deterministic colours."""
    names = [n for n, i in sorted(tissue_label_map.items(), key=lambda kv: kv[1]) if i != 0]
    with open(tissue_list_file_name, "w") as f:
        print("V7", file=f)
        print(f"N{len(names)}", file=f)
        for k, n in enumerate(names):
            r, g, b = ((k * 53) % 256) / 255.0, ((k * 97 + 80) % 256) / 255.0, ((k * 193 + 160) % 256) / 255.0
            print(f"C{r:.2f} {g:.2f} {b:.2f} 0.50 {n}", file=f)


def load_decathlon_tissuelist(file_name: Path) -> Dict[str, int]:
    """Decathlon datalist 'labels': {"1": "name", ...} plus Background = 0."""
    file_name = Path(file_name)
    print(f"Reading {file_name}")
    labels = json.loads(file_name.read_text())["labels"]
    labels["0"] = "Background"
    return {n: int(i) for i, n in labels.items()}
