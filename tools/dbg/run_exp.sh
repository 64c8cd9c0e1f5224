echo "== product"; timeout 120 python tools/dbg/geglu_bench.py 2>&1 | grep GEGLU
echo "== product big"; VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/geglu_bench.py 2>&1 | grep GEGLU
echo "== fast gelu"; VIDSEG_LIB=libvidseg_hip_exp512.so timeout 120 python tools/dbg/geglu_bench.py 2>&1 | grep GEGLU
echo "== fast gelu big"; VIDSEG_LIB=libvidseg_hip_exp512.so VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/geglu_bench.py 2>&1 | grep GEGLU
