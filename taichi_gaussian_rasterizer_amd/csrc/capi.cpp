// capi.cpp -- error reporting and version of libgsplat_hip.so (host only).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

// see gs_common.h.  Measured crossovers (tools/exp_nb.py, tools/exp_shard_nb.py): the forward (lighter per overlap,
// more latency-bound) wants the extra waves up to larger grids than the backward.
int gs_raster_sub_blocks(int tile_size, int64_t num_tiles, int backward) {
  if (tile_size == 8) return 1;
  if (const char* e = getenv("GS_RASTER_NB")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) return v;
  }
  const int64_t regions = num_tiles * (tile_size == 32 ? 4 : 1);  // 16x16 regions
  const int64_t nb4 = backward ? 2048 : 3072, nb2 = backward ? 768 : 1536;
  if (regions >= nb4) return 4;
  if (regions >= nb2) return 2;
  return 1;
}
