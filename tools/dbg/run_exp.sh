VIDSEG_LIB=libvidseg_hip.so VIDSEG_GEMM_BIG=2 timeout 300 python tools/dbg/ph_bench.py base 2>&1 | grep "^base"
VIDSEG_LIB=libvidseg_hip_soft.so VIDSEG_GEMM_BIG=2 timeout 300 python tools/dbg/ph_bench.py ph 2>&1 | grep "^ph"
python - <<PY
import torch
from tools.dbg.ph_bench import SHAPES
a, b = torch.load("/tmp/ph_base.pt"), torch.load("/tmp/ph_ph.pt")
for sh, x, y in zip(SHAPES, a, b):
    print(sh, "bit-equal" if torch.equal(x, y) else f"DIFF max {(x.float()-y.float()).abs().max().item():.4g} frac {(x != y).float().mean().item():.4g}")
PY
