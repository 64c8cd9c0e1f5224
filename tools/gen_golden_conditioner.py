"""tests/golden/conditioner.npz: the reference's ConcatTimestepEmbedderND and GeneralConditioner (sgm/modules/encoders/modules.py)
on SVD-shaped inputs.  Build-container only; open_clip / kornia (imported at the top of that module, unused by these classes) are
stubbed like the other absent packages."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402


def main():
    import transformers  # noqa: F401  (before the torchvision stub exists: its availability probe needs a real module spec)
    import_reference()
    for name in ("open_clip", "kornia"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from sgm.modules.encoders import modules as M
    torch.set_grad_enabled(False)
    F = 5
    fps_id = torch.full((F,), 6.0)
    motion = torch.full((F,), 127.0)
    cond_aug = torch.full((F,), 0.02)
    two = torch.stack([fps_id, motion], 1)
    e1 = M.ConcatTimestepEmbedderND(256)
    rec = dict(fps_id=fps_id.numpy(), motion=motion.numpy(), cond_aug=cond_aug.numpy(), emb_fps=e1(fps_id).numpy(),
               emb_two=e1(two).numpy(), emb_aug=e1(cond_aug).numpy())

    class Pre(M.AbstractEmbModel):
        def forward(self, x):
            return x
    M.Pre = Pre
    cfgs = [{"target": "sgm.modules.encoders.modules.Pre", "input_key": "cond_frames_without_noise"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "fps_id"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "motion_bucket_id"},
            {"target": "sgm.modules.encoders.modules.Pre", "input_key": "cond_frames"},
            {"target": "sgm.modules.encoders.modules.ConcatTimestepEmbedderND", "params": {"outdim": 256}, "input_key": "cond_aug"}]
    g = np.random.Generator(np.random.PCG64(5))
    batch = {"cond_frames_without_noise": torch.from_numpy(g.standard_normal((F, 1, 1024)).astype(np.float32)),
             "cond_frames": torch.from_numpy(g.standard_normal((F, 4, 9, 16)).astype(np.float32)),
             "fps_id": fps_id, "motion_bucket_id": motion, "cond_aug": cond_aug}
    cond = M.GeneralConditioner(cfgs)
    c, uc = cond.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    for k, v in batch.items():
        rec["batch_" + k] = v.numpy()
    for k in c:
        rec["c_" + k], rec["uc_" + k] = c[k].numpy(), uc[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "conditioner.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: tuple(v.shape) for k, v in c.items()})


if __name__ == "__main__":
    main()
