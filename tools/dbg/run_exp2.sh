echo "== base"; VIDSEG_GEMM_PH=0 VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/clock_probe.py 2>&1 | tail -4
echo "== ph"; VIDSEG_GEMM_PH=1 VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/clock_probe.py 2>&1 | tail -4
for n in 3 16; do echo "== exp$n"; VIDSEG_LIB=libvidseg_hip_exp$n.so VIDSEG_GEMM_PH=1 VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/clock_probe.py 2>&1 | tail -4; done
