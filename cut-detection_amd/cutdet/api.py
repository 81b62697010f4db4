"""Process-wide access to the bundled prod_net weights and one Engine per GPU."""
import os

import torch

from .engine import Engine

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_BLOB = os.path.join(_PKG, "frameID", "prod_net", "prod_net_f32.bin")
DEFAULT_MANIFEST = os.path.join(_PKG, "frameID", "prod_net", "prod_net_manifest.json")

_engines = {}


def default_blob():
    with open(DEFAULT_BLOB, "rb") as f:
        return f.read()


def default_engine(device="cuda:0", dtype="f32"):
    """The Engine holding the bundled prod_net on `device` in `dtype` arithmetic (created on first use)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"cutdet has no CPU path; asked for device {device!r}")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if (idx, dtype) not in _engines:
        _engines[(idx, dtype)] = Engine(default_blob(), torch.device("cuda", idx), dtype)
    return _engines[(idx, dtype)]
