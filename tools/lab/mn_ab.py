"""k_mean_normalize at the headline's sizes: time + hash of the result (run under VIDSEG_MN_ROWS16=0 / 1 for the A/B)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidseg_diffusion_amd import analysis as A, synthetic
dev = torch.device("cuda:0")
F, fh, fw, C = 14, 32, 32, 640
n = F * fh * fw
blocks, _ = synthetic.attention_q_dumps(F, fh, fw, C, num_blocks=3, seed=1)
dumps = [torch.from_numpy(b).to(dev) for b in blocks]
for _ in range(3): m, f = A.mean_normalize(dumps, n, n, want_mean=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): A.mean_normalize(dumps, n, n)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
by = 3 * n * C * 2 + n * C * 2
print(f"VIDSEG_MN_ROWS16={os.environ.get('VIDSEG_MN_ROWS16', '(default)')}: {us:.1f} us, {by / us / 1e6:.2f} TB/s, sha {hashlib.sha256(f.cpu().numpy().tobytes() + m.cpu().numpy().tobytes()).hexdigest()[:16]}")
# odd row counts (tail lanes)
m2, f2 = A.mean_normalize(dumps, n + 3, n - 7, want_mean=True)
print("tail sha", hashlib.sha256(f2.cpu().numpy().tobytes() + m2.cpu().numpy().tobytes()).hexdigest()[:16])
