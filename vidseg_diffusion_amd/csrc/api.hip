// Library-level entry points of libvidseg_hip.so.
#include "common.h"

thread_local char g_vs_err[512] = {0};

extern "C" {
int vidseg_version(void) { return 100; }                        // 0.1.0
const char* vidseg_last_error(void) { return g_vs_err; }
int vidseg_act_dtype(void) { return VIDSEG_ACT_IS_F16; }          // 1: fp16 activations/weights (default), 0: bf16 build
}
