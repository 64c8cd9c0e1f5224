"""Experiment builds of the same C ABI with extra -D flags: python tools/build_exp.py NAME -DFOO=1 ...  writes
vidseg_diffusion_amd/libvidseg_exp_NAME.so; load it with VIDSEG_LIB=libvidseg_exp_NAME.so (see _lib.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name, defines = sys.argv[1], sys.argv[2:]
# libvidseg_exp_*.so stays out of gpurun pushes (.gpurunignore): build those on the GPU box; a name starting with "ab_" gives
# libvidseg_ab_*.so, which travels with the snapshot for a same-box A/B (delete it afterwards)
prefix = "libvidseg_" if name.startswith("ab_") else "libvidseg_exp_"
G._build_variant(os.path.join(ROOT, "vidseg_diffusion_amd", f"{prefix}{name}.so"), f".exp_{name}", defines, True)
print("built", name, defines)
