"""The five helpers the inference scripts import from vorta.utils (scripts/hunyuan/inference.py:33-39)."""
import argparse
import json
import logging
import re
import sys
from enum import Enum
from pathlib import Path
from typing import Optional, Tuple

import torch


def setup_logging(output_file: Optional[str] = None, log_level: str = "INFO",
                  log_format: str = "%(asctime)s [%(levelname)s] %(message)s (%(filename)s:%(lineno)d)",
                  date_format: str = "%Y-%m-%d %H:%M:%S") -> None:
    """stdout (+ optional file) logging (vorta/utils/log.py:12-29)."""
    handlers = [logging.StreamHandler(stream=sys.stdout)] + ([logging.FileHandler(output_file)] if output_file else [])
    logging.basicConfig(level=log_level, format=log_format, datefmt=date_format, handlers=handlers)


def arg_to_json(arg: argparse.Namespace) -> str:
    """vorta/utils/log.py:32-42: Paths and Enums serialised by value."""
    def enc(o):
        if isinstance(o, Path):
            return str(o)
        if isinstance(o, Enum):
            return o.value
        raise TypeError(f"Object of type {type(o).__name__} is not JSON serializable")
    return json.dumps(vars(arg), default=enc, indent=4, sort_keys=True)


def prompt_to_file_name(text_prompt: str, prefix=None, suffix=None, max_str_len=20) -> str:
    """vorta/utils/misc.py:26-37."""
    name = re.sub(r"\s+", "-", re.sub(r"[^\w\s]", "", text_prompt).strip()).strip().lower()[:max_str_len]
    if prefix is not None:
        name = f"{prefix:03d}-{name}"
    if suffix is not None:
        name += f"-{suffix:02d}"
    return name


def _step_of(name: str) -> int:
    return int(name.split("-")[1].split(".")[0])


def parent_to_ckpt_dir(resume: Optional[str], ckpt_dir: Path) -> Tuple[Optional[Path], int]:
    """None | 'latest' | 'step-N' -> (checkpoint dir, step) (vorta/utils/misc.py:52-65)."""
    if resume is None:
        return None, 0
    if resume == "latest":
        found = sorted(ckpt_dir.glob("step-*"), key=lambda p: _step_of(p.name), reverse=True)
        if not found:
            logging.getLogger(__name__).warning(f"No checkpoint found in {ckpt_dir}")
            return None, 0
        return found[0], _step_of(found[0].name)
    ckpt = ckpt_dir / resume
    if not ckpt.exists():
        raise FileNotFoundError(f"Checkpoint {ckpt} does not exist")
    return ckpt, _step_of(resume)


_DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def str_to_dtype(dtype: str) -> torch.dtype:
    """vorta/utils/misc.py:68-77."""
    try:
        return _DTYPES[dtype.lower()]
    except KeyError:
        raise ValueError(f"Unsupported dtype {dtype}") from None


def dtype_to_str(dtype: torch.dtype) -> str:
    for k, v in _DTYPES.items():
        if v == dtype:
            return k
    raise ValueError(f"Unsupported dtype {dtype}")


# ---- the remaining small helpers of vorta/utils (log.py:45-46, misc.py:11-23,40-49,91-92); video file I/O
# (video_io.py: decord / torchvision) is not part of this build -------------------------------------------------
def arg_to_yaml(arg: argparse.Namespace) -> str:
    import yaml
    return yaml.dump(vars(arg), indent=4, sort_keys=True)


def isinstance_str(x: object, cls_name: str) -> bool:
    """does any class in x's ancestry carry this NAME (no import of the class needed; used when patching)"""
    return any(c.__name__ == cls_name for c in type(x).__mro__)


def get_cuda_memory_usage(device: torch.device) -> float:
    free, total = torch.cuda.mem_get_info(device)
    used = (total - free) / 1024 ** 3
    logging.getLogger(__name__).debug(f"CUDA memory used: {used:.2f} GB")
    return used


def format_metrics_to_gb(item) -> float:
    return round(item / 1024 ** 3, ndigits=4)


def accumulate_loss(current_loss, new_loss):
    return new_loss if current_loss is None else current_loss + new_loss
