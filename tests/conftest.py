import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding with libsdv_hip.so loaded (built in-tree if missing)."""
    from stable_diffusion_videos_amd import hip as _hip
    _hip.load(build_if_missing=True)
    return _hip


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda", 0)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def psnr(a: torch.Tensor, b: torch.Tensor, peak=None) -> float:
    a, b = a.double(), b.double()
    mse = float(((a - b) ** 2).mean())
    peak = float(b.abs().max()) if peak is None else peak
    if mse == 0:
        return float("inf")
    import math
    return 10.0 * math.log10(peak * peak / mse)


def report(msg: str):
    """Print a measured parity number and append it to gpurun_out/parity_report.txt (kept as evidence)."""
    print(msg)
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "parity_report.txt", "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass
