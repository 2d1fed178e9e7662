// capi.cpp -- error reporting and version of libgsplat_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gsplat_hip.h"

static thread_local char g_error[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_error; }
extern "C" int gs_version(void) { return 1; }
