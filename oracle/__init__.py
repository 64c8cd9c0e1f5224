"""CPU oracle for the VidSeg hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the reference algorithm for the
path named by BASELINE.json:north_star.  It exists to CHECK the HIP product path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it;
  * nothing under ``vidseg_diffusion_amd/`` imports it, and the product path raises if the
    HIP extension is missing -- there is no CPU fallback.

Pinning: the reference ships no tests/golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, run in the build container by
``tools/gen_golden_*.py`` (which import /root/reference read-only) and committed under
``tests/golden/``.  ``tests/test_oracle_*.py`` replays those fixtures.

Third-party arithmetic restated here (absent from /root/reference, pinned in
requirements/pt2.txt): scikit-learn ``KMeans``/``KNeighborsClassifier`` (pinned 1.5.0 by the
reference, 1.7.2 in the build container -- fixtures were generated with 1.7.2), numpy
``argpartition`` tie-breaking (2.0.0 pinned, 2.2.6 here); ``oracle/openclip.py``: open_clip_torch
2.24.0's ViT-H text / image towers and kornia 0.7.2's antialiased resize (both absent from this
image) -- pinned against transformers' CLIP classes on the same weights instead of against the
reference run (tests/test_oracle_openclip.py); kornia's gaussian pass: parity unpinned.
"""
