"""File discovery helpers (reference ``src/segmantic/utils/file_iterators.py:9-36``)."""
from __future__ import annotations

from pathlib import Path
from typing import List


def find_matching_files(input_globs: List[Path], verbose: bool = True) -> List[List[Path]]:
    """Tuples of files whose names agree once the glob's suffix (text after the last '*') is
    removed; the first glob defines the candidate keys."""
    input_globs = [Path(g) for g in input_globs]
    dir_0 = Path(input_globs[0].anchor)
    glob_0 = str(input_globs[0].relative_to(dir_0))
    ext_0 = input_globs[0].name.rsplit("*")[-1]
    candidates = {p.name.replace(ext_0, ""): [p] for p in dir_0.glob(glob_0)}
    for other in input_globs[1:]:
        dir_i = Path(other.anchor)
        glob_i = str(other.relative_to(dir_i))
        ext_i = other.name.rsplit("*")[-1]
        for p in dir_i.glob(glob_i):
            key = p.name.replace(ext_i, "")
            if key in candidates:
                candidates[key].append(p)
            elif verbose:
                print(f"No match found for {key} : {p}")
    out = [v for v in candidates.values() if len(v) == len(input_globs)]
    if verbose:
        print(f"Number of files in {input_globs[0]}: {len(candidates)}")
        print(f"Number of tuples: {len(out)}\n")
    return out
