"""Print the figures of one bench.py JSON line that a build record quotes: python tools/bench_digest.py <bench.json>"""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:                                                   # noqa: BLE001
    raise SystemExit(f"bench parse failed: {e}")
m = d.get("mask_iou_vs_reference") or {}
print("value", d["value"], d["unit"], d["ms_per_step"], "ms | masks: mean", m.get("mean_iou"), "at 0.99:", m.get("windows_at_0.99"), "of", m.get("n_windows"),
      "| lanes", (d.get("config") or {}).get("overlap", "")[:1])
for k in ("single_lane", "two_lanes", "chained_window", "full_schedule", "fast_mode"):
    v = d.get(k) or {}
    if v:
        print(k, v.get("value"), v.get("ms_per_step"), v.get("mask_iou_vs_reference"), v.get("error"))
r = d["roofline"]
print("roofline", r["kernel"][:24], "achieved", r["achieved"], "frac", r["frac"], "frac_algorithmic", r.get("frac_algorithmic"), "avg us", r["avg_launch_us"],
      "traffic", r.get("traffic"), "alg bytes", r.get("algorithmic_bytes"))
print("family", json.dumps(r.get("family", {}))[:700])
if d.get("roofline_post_unet"):
    print("post-UNet", json.dumps(d["roofline_post_unet"])[:900])
s = d.get("secondary") or {}
sm = s.get("mask_iou_vs_reference") or {}
print("secondary", s.get("value"), s.get("ms_per_step"), [w.get("iou") for w in sm.get("windows", [])], s.get("fast_mode"), s.get("error"), s.get("step4_latent_blending"))
s = d.get("secondary_fp8") or {}
print("secondary_fp8", s.get("value"), s.get("ms_per_step"), s.get("error"))
c = d.get("cpu_baseline") or {}
print("cpu", c.get("value"), c.get("cores"))
