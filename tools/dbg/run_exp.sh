for i in 1 2; do VIDSEG_GEMM_PH=1 VIDSEG_GEMM_BIG=2 timeout 300 python tools/dbg/ph_bench.py ph 2>&1 | grep "^ph" | tail -3; done
VIDSEG_GEMM_PH=0 VIDSEG_GEMM_BIG=2 timeout 300 python tools/dbg/ph_bench.py base 2>&1 | grep "^base" | tail -2
