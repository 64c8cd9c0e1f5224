"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/vidseg_hip.h
declares; the ctypes table covers exactly those symbols.  No compute calls (CPU only)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from vidseg_diffusion_amd import _lib
    return _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "vidseg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vidseg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from vidseg_diffusion_amd import exact, openclip, ops, process_output  # noqa: F401  (register the UNet / exact-mode / conditioner / Step 5 signatures)
    l = ctypes.CDLL(lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/vidseg_hip.h but not exported"
    assert sorted(lib.exported_symbols()) == names, "ctypes table and header disagree"


def test_version_and_error_string(lib):
    l = lib.lib()
    assert l.vidseg_version() == 100
    assert l.vidseg_last_error() is not None


def test_product_path_refuses_cpu_tensors(lib):
    import torch
    from vidseg_diffusion_amd import analysis
    with pytest.raises(lib.VidsegError):
        analysis.kmeans_fit(torch.zeros(8, 8, dtype=torch.float16), 2)


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "vidseg_diffusion_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in re.sub(r'""".*?"""', "", open(os.path.join(pkg, f)).read(), flags=re.S).replace("oracle-backed", ""), f


def test_checkpoint_and_dump_file_formats(tmp_path):
    """upstream checkpoint keys load into the mirror UNet; the FeatureStore exports the reference's
    ``feature_maps/*.pt`` layout (host-side file handling only, no device needed)."""
    import torch
    from safetensors.torch import save_file
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd import synthetic
    from vidseg_diffusion_amd.unet import UNetModel
    from vidseg_diffusion_amd.util import load_checkpoint_state_dict
    net = UNetModel(**synthetic.SD21_NARROW)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {"model.diffusion_model." + k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, seed=7).items()}
    sd["first_stage_model.encoder.conv_in.weight"] = torch.zeros(2, 2)          # other modules of the checkpoint are ignored
    path = str(tmp_path / "ckpt.safetensors")
    save_file(sd, path)
    got = load_checkpoint_state_dict(path)
    assert set(got) == set(shapes)
    missing, unexpected = net.load_state_dict(got, strict=False)
    assert not missing and not unexpected
    FE.FeatureStore.clear()
    q = torch.arange(2 * 3 * 4, dtype=torch.float16).reshape(2, 3, 4)
    FE.FeatureStore.put(str(tmp_path), "exp", "output_block_7_spatial_self_attn_q_time_24", q)
    files = FE.FeatureStore.export_pt(str(tmp_path), "exp")
    assert files == [str(tmp_path / "exp" / "feature_maps" / "output_block_7_spatial_self_attn_q_time_24.pt")]
    assert torch.equal(torch.load(files[0]), q)
    FE.FeatureStore.clear()


def test_png_mask_folder_round_trip(tmp_path):
    """The Step 3 -> Step 4 hand-off on disk (feature_extraction.py:618-636 writes `kmeans_time_{t}_frame_{name}/mask_{l}.png`,
    sd_pipeline_vspw.py:64-101 reads them back): the PNG folder written with VIDSEG_WRITE_PNG gives `load_feature_masks` exactly
    what the in-HBM MaskStore entry gives it -- same resolution and resized to another decoder block's resolution."""
    import numpy as np
    import torch
    from PIL import Image
    from vidseg_diffusion_amd import feature_extraction as FE
    from vidseg_diffusion_amd.pipeline import load_feature_masks
    F, h, w, labels = 3, 8, 8, np.array([2, 5, 7])
    g = np.random.Generator(np.random.PCG64(11))
    lab = labels[g.integers(0, 3, (F, h, w))].astype(np.int32)
    names = ["00007", "00008", "00010"]
    folder = str(tmp_path / "exp" / "match_gt_mask" / "output_block_7_spatial_self_attn_q_masks_3")
    FE._write_png_masks(folder, torch.from_numpy(lab), names, 24, labels, True)
    for i, n in enumerate(names):                                        # the reference's layout and pixel values
        for l in labels:
            m = np.array(Image.open(os.path.join(folder, f"kmeans_time_24_frame_{n}", f"mask_{l}.png")))
            assert m.shape == (h, w) and m.dtype == np.uint8 and np.array_equal(m, np.where(lab[i] == l, 255, 0))
    cpu = torch.device("cpu")
    FE.MaskStore.clear()
    for block, base in ((7, 2), (10, 2)):                                # block 7: 8x8 = the stored resolution; block 10: PIL-resized to 16x16
        from_png = load_feature_masks(folder, 5, num_frames=F, feature_timestep="24", modulate_block_idx=block, base_height=base,
                                      base_width=base, frame_name_list=names, device=cpu)
        FE.MaskStore.put(folder, torch.from_numpy(lab.reshape(F, -1)), names, labels)
        from_store = load_feature_masks(folder, 5, num_frames=F, feature_timestep="24", modulate_block_idx=block, base_height=base,
                                        base_width=base, frame_name_list=names, device=cpu)
        FE.MaskStore.clear()
        assert len(from_png) == len(from_store) == F
        for a, b in zip(from_png, from_store):
            assert a.dtype == torch.float64 and torch.equal(a, b)
            assert a.numel() == (base * (4 if block == 7 else 8)) ** 2
    same_res = load_feature_masks(folder, 5, num_frames=F, modulate_block_idx=7, base_height=2, base_width=2, frame_name_list=names, device=cpu)
    assert torch.equal(same_res[0], torch.from_numpy((lab[0] == 5).astype(np.float64).reshape(-1)))


def test_hand_counted_kernels_have_no_scratch(tmp_path):
    """k_gemm_ws waits for its ring loads with hand-written `s_waitcnt vmcnt(N)` counts; a register spill would put scratch loads
    (VMEM operations the counts do not know about) into its loop and the ring would be read too early.  Cross-compile the GEMM
    file to assembly for both builds and require: no private segment, no scratch instruction in any k_gemm_ws instantiation."""
    import re
    import subprocess
    import __graft_entry__ as G
    src = os.path.join(G.CSRC, "gemm_conv.hip")
    for defines in ([], ["-DVIDSEG_ACT_BF16"]):
        out = str(tmp_path / ("gemm" + ("_bf16" if defines else "") + ".s"))
        subprocess.check_call([G.HIPCC, *[f for f in G.FLAGS if f != "-fPIC"], *defines, "-I", os.path.join(G.ROOT, "include"),
                               "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read()
        bodies = re.findall(r"^(_Z9k_gemm_wsILi\d+ELi\d+ELi\d+EEv\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
        assert len(bodies) >= 6, [b[0] for b in bodies]
        for name, body in bodies:
            assert "scratch_" not in body, (name, defines)
        for name, size in re.findall(r"\.amdhsa_kernel (_Z9k_gemm_ws\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", text, flags=re.S):
            assert int(size) == 0, (name, size, defines)
        # and, as a performance invariant, no MFMA GEMM kernel of the family may spill at all: every one of them sits at 200-256 VGPRs,
        # and an epilogue addition that tips one over costs every launch (round 3: a LayerNorm-folding epilogue put k_gemm_dma<2> /
        # <3> and the 128x320 tiles 92-340 bytes into scratch -- 148 -> 133 frames/s, folded or not; reverted)
        sizes = re.findall(r"\.amdhsa_kernel (_Z\d+k_gemm_\w+).*?\.amdhsa_private_segment_fixed_size (\d+)", text, flags=re.S)
        assert len(sizes) >= 15, len(sizes)
        for name, size in sizes:
            assert int(size) == 0, (name, size, defines)


def test_gemm_profiler_handle_is_bound_to_its_thread(lib):
    """The profiler handle is caller-owned and records the launches of ONE thread (a thread-local pointer to it while recording):
    ending or destroying it from another thread while it records is refused instead of leaving that pointer dangling (no HIP call is
    involved: host-side bookkeeping only)."""
    import threading
    from vidseg_diffusion_amd import ops
    p = ops.GemmProfiler()
    p.begin()
    res = {}

    def other():
        l = lib.lib()
        out = (ctypes.c_double * 3)()
        res["end"] = l.vidseg_gemm_profile_end(p.h, out)
        res["destroy"] = l.vidseg_gemm_profiler_destroy(p.h)
        res["msg"] = l.vidseg_last_error().decode()
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert res["end"] != 0 and res["destroy"] != 0 and "another thread" in res["msg"], res
    ms, flops, launches = p.end()                                    # the owner ends it: an empty region
    assert (ms, flops, launches) == (0.0, 0.0, 0)
    q = ops.GemmProfiler()                                           # an idle handle may be destroyed from any thread
    t = threading.Thread(target=lambda: res.__setitem__("idle", lib.lib().vidseg_gemm_profiler_destroy(q.h)))
    t.start()
    t.join()
    q.h = None
    assert res["idle"] == 0
