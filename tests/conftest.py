import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("VIDSEG_KEEP_LAST", "1")      # analysis.LAST_KMEANS / LAST_CENTER_IDS (restart-level parity tests); off in the product
os.environ.setdefault("VIDSEG_X_POISON_PLANE3", "1")   # exact mode: NaN in the third plane of every operand image (no kernel may read it)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def act_mode():
    """("f16" | "bf16", whole-network tolerance) of the library under test: the default fp16 build must stay within 2.5e-3
    normalised rms of the fp32 reference (taps, UNet outputs, sampler trajectories; measured 1.2e-3 ... 2.0e-3), a
    -DVIDSEG_ACT_BF16 build within 2.5e-2 (measured 1.2e-2 ... 1.6e-2)."""
    import torch
    from vidseg_diffusion_amd import ops
    return ("f16", 2.5e-3) if ops.act_dtype() == torch.float16 else ("bf16", 2.5e-2)
