// Error reporting and build identification for libodvae_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

extern "C" {

void odvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// text of the last failing call on this thread ("" if none)
const char* odvae_last_error(void) { return g_err; }

// ABI version of include/odvae_hip.h this library was built against
int odvae_abi_version(void) { return 4; }

const char* odvae_target_arch(void) { return "gfx950"; }

}  // extern "C"
