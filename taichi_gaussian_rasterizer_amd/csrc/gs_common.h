// gs_common.h -- shared by every translation unit of libgsplat_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsplat_hip.h"

#define GS_WAVE 64

void gs_set_error(const char* fmt, ...);

#define GS_REQUIRE(cond, status, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      gs_set_error(__VA_ARGS__);       \
      return (status);                 \
    }                                  \
  } while (0)

#define GS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      gs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return GS_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

static inline int64_t gs_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gs_align_up(int64_t a, int64_t b) { return gs_div_up(a, b) * b; }

static inline int gs_check_cfg(const GsRasterConfig* cfg) {
  GS_REQUIRE(cfg != nullptr, GS_ERR_INVALID_ARGUMENT, "config is NULL");
  GS_REQUIRE(cfg->tile_size == 8 || cfg->tile_size == 16 || cfg->tile_size == 32, GS_ERR_UNSUPPORTED,
             "tile_size %d not supported (8, 16 or 32)", cfg->tile_size);
  GS_REQUIRE(cfg->alpha_threshold > 0.f, GS_ERR_INVALID_ARGUMENT, "alpha_threshold must be > 0");
  return GS_OK;
}

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__

// XCD-aware block -> work-item remap: blocks b and b+8 share an XCD (round-robin dispatch), so
// giving XCD x the contiguous chunk [x*chunk, (x+1)*chunk) keeps neighbouring tiles -- which
// gather the same splat rows -- on one L2.  Speed only; any placement is correct.
// Launch with grid = 8 * ceil(n/8); returns -1 for the padding blocks.
__device__ __forceinline__ int gs_xcd_remap(int block, int n) {
  const int chunk = (n + 7) >> 3;
  const int item = (block & 7) * chunk + (block >> 3);
  return ((block >> 3) < chunk && item < n) ? item : -1;
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float gs_dpp_add(float x) {
  // x + dpp(x): lanes whose source is invalid or masked add 0 (old = 0, bound_ctrl = false)
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false);
  return x + __int_as_float(moved);
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63.
// row_ror 8/4/2/1 leaves every lane of a 16-lane row holding the row sum (4 fused v_add_f32_dpp),
// row_bcast:15 / row_bcast:31 then fold the rows (ISA: DPP_ROW_BCAST15 = 0x142, BCAST31 = 0x143).
__device__ __forceinline__ float gs_wave_sum_to_lane63(float x) {
  x = gs_dpp_add<0x128>(x);  // row_ror:8
  x = gs_dpp_add<0x124>(x);  // row_ror:4
  x = gs_dpp_add<0x122>(x);  // row_ror:2
  x = gs_dpp_add<0x121>(x);  // row_ror:1
  x = gs_dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x = gs_dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}

__device__ __forceinline__ float gs_exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float gs_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32

#endif  // __HIPCC__
