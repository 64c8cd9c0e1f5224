// Library-level entry points of libvidseg_hip.so.
#include "common.h"

thread_local char g_vs_err[512] = {0};

extern "C" {
int vidseg_version(void) { return 100; }                        // 0.1.0
const char* vidseg_last_error(void) { return g_vs_err; }
}
