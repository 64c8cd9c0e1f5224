"""Import the read-only reference (/root/reference) inside the BUILD container only.

Used exclusively by tools/gen_golden_*.py to produce tests/golden/*.npz.  Nothing is copied:
the reference's modules are imported in place, with namespace stubs so that the
pytorch_lightning / open_clip / omegaconf / cv2 / torchvision imports that the path never
uses are not executed (recipe: SURVEY.md Appendix C).
"""
import functools
import os
import sys
import types

REF = "/root/reference"


def import_reference():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present (this tool only runs in the build container)")
    for name, path in [("sgm", f"{REF}/sgm"), ("sgm.modules", f"{REF}/sgm/modules"),
                       ("scripts", f"{REF}/scripts"), ("scripts.sampling", f"{REF}/scripts/sampling")]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    for name in ["omegaconf", "cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["omegaconf"].ListConfig = list
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    import scripts.sampling.feature_extraction as fe
    fe.dense_tracking = functools.partial(fe.dense_tracking, device="cpu")   # FE:328 hard-codes "cuda"
    return fe
