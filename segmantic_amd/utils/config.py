"""JSON-or-YAML config files chosen by suffix (reference ``src/segmantic/utils/config.py:9-32``)."""
import json
import sys
from pathlib import Path
from typing import Any, Optional

import yaml


def load(config_file: Path) -> Any:
    config_file = Path(config_file)
    return loads(config_file.read_text(), config_file.suffix.lower() == ".json")


def loads(text: str, is_json: bool) -> Any:
    return json.loads(text) if is_json else yaml.safe_load(text)


def dump(obj: Any, config_file: Optional[Path] = None) -> None:
    if config_file:
        config_file = Path(config_file)
        config_file.write_text(dumps(obj, config_file.suffix.lower() == ".json"))
    else:
        yaml.safe_dump(obj, stream=sys.stdout, sort_keys=False)


def dumps(obj: Any, is_json: bool) -> str:
    if is_json:
        return json.dumps(obj, indent=4)
    return yaml.safe_dump(obj, stream=None, sort_keys=False)
