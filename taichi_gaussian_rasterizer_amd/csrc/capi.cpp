// capi.cpp -- error reporting and version of libgsplat_hip.so (host only).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gs_common.h"

static thread_local char g_error[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_error; }
extern "C" int gs_version(void) { return 2; }

// see gs_common.h.  Measured crossovers (tools/exp_nb.py, tools/exp_shard_nb.py): the forward (lighter per overlap,
// more latency-bound) wants the extra waves up to larger grids than the backward.
int gs_raster_sub_blocks(const GsRasterConfig* cfg, int64_t num_tiles, int backward) {
  const int tile_size = cfg->tile_size;
  if (tile_size == 8) return 1;
  const int forced = cfg->tune_wave_sub_blocks;
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  const int64_t regions = num_tiles * (tile_size == 32 ? 4 : 1);  // 16x16 regions
  const int64_t nb4 = backward ? 2048 : 3072, nb2 = backward ? 768 : 1536;
  if (regions >= nb4) return 4;
  if (regions >= nb2) return 2;
  return 1;
}
