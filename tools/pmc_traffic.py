"""HBM-side traffic of the GEMM kernel family from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_traffic.py A/.../p_results.db B/.../p_results.db profiles/r01_traffic.json

Correction (MI355X_MICROARCH.md, section HBM): rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 128-B requests at
64 B, i.e. exactly half of a wide (16 B/lane) coalesced read stream -- every operand load of these kernels is such a
stream (buffer_load ... lds, 16 B/lane) -- so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  The counters sit on the
L2's memory side: Infinity-Cache hits are included, so this is an upper bound on true HBM traffic.
"""
import json
import sqlite3
import sys

FAMILY = ("k_gemm_tile", "k_gemm_dma", "k_gemm_conv", "k_gemm_ph")


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        if any(f in name for f in FAMILY):
            key = name.split("(")[0]
            a = out.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(val)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    tot_n = 0
    tot_bytes = 0.0
    for k in sorted(fetch):
        n, f = fetch[k]
        n2, w = write.get(k, (0, 0.0))
        assert n == n2, (k, n, n2)
        b = (2.0 * f + w) * 1024.0
        kernels[k] = {"launches": n, "fetch_size_kb_raw": round(f, 1), "write_size_kb": round(w, 1), "traffic_bytes_per_launch": int(b / n)}
        tot_n += n
        tot_bytes += b
    res = {
        "_how": "two separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps 1 --warmup 0 "
                "--no-cpu-baseline; per-dispatch counters summed over the GEMM-family launches of one window, divided by the launch count. "
                "rocprofv3 reports KB; on gfx950 FETCH_SIZE counts exactly half of a wide (16 B/lane) coalesced read stream "
                "(MI355X_MICROARCH.md section HBM), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. Memory-side L2 counters: "
                "Infinity-Cache hits are included, so this is an upper bound on true HBM traffic.",
        "kernel": "bf16 MFMA implicit-GEMM family (all conv/linear launches of one window)",
        "launches": tot_n,
        "traffic_bytes_per_launch": int(tot_bytes / max(tot_n, 1)),
        "per_kernel": kernels,
    }
    with open(sys.argv[3], "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
