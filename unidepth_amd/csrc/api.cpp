// Error plumbing + library info for the C-ABI (include/unidepth_hip.h).
#include <string.h>
#include "ud_common.h"

static thread_local char g_err[256] = "";

void ud_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* ud_last_error(void) { return g_err; }
extern "C" int ud_version(void) { return 111; }

// struct sizes, so the Python binding can verify its ctypes mirror of include/unidepth_hip.h
extern "C" int ud_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(UdGemm);
    case 1: return (int)sizeof(UdLayerNorm);
    case 2: return (int)sizeof(UdAttention);
    case 3: return (int)sizeof(UdPreprocess);
    case 4: return (int)sizeof(UdRayEmbed);
    case 5: return (int)sizeof(UdUpsample2x);
    case 6: return (int)sizeof(UdResizeAC);
    case 7: return (int)sizeof(UdFinalize);
    case 8: return (int)sizeof(UdLinearF32);
    case 9: return (int)sizeof(UdDwConv7);
    case 10: return (int)sizeof(UdV1Op);
    case 11: return (int)sizeof(UdKnn);
    case 12: return (int)sizeof(UdExtractPatches);
    case 13: return (int)sizeof(UdCameraHead);
    default: return -1;
  }
}

// numerics bisect / ablation switches: compiled in ONLY for the instrumented tools builds (csrc/build.sh -DUD_TOOLS, used by tools/);
// the product library neither exports ud_set_debug_flags nor reads any switch (ud_debug_flags_host() is a constant 0 there).
#ifdef UD_TOOLS
static int g_debug_flags = 0;
int ud_debug_flags_host() { return g_debug_flags; }
extern "C" void ud_set_debug_flags(int f) { g_debug_flags = f; }
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is cached per (kernel, device): one process may drive several GPUs.
bool ud_attr_once(bool (&done)[UD_MAX_DEVICES]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UD_MAX_DEVICES) return false;   // unknown device: set the attribute again
  if (done[dev]) return true;
  done[dev] = true;
  return false;
}
