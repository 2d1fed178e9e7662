// Micro-benchmark (gfx950): can the per-(tile, splat) wave reduction of raster_bwd (9 f32 values summed over 64 lanes)
// leave the VALU issue port?  Three candidates beside the shipped transposed butterfly, each embedded in FILL plain
// v_fma_f32 per call (the backward's own per-overlap arithmetic, ~80 VALU) at 6 waves per SIMD, so that what is
// measured is the reduction's cost INSIDE a VALU-bound loop, matrix pipe and LDS running beside it:
//   mfma : the wave sum as a matrix product on the idle matrix pipe.  v_mfma_f32_16x16x4_f32 (exact f32): lane l
//          supplies A[l % 16][l / 16] and B[l / 16][l % 16], D[i][j] = sum_k A[i][k] B[k][j].  With A = the per-lane
//          partial sums of value c and B = the selector (j == c), one instruction folds the four 16-lane rows and
//          drops the 16 column sums of value c into column c of the accumulator; N values accumulate into ONE 4-VGPR
//          D.  Three adds sum a lane's 4 accumulator rows, a second MFMA (A = that sum, B = ones) folds the remaining
//          4 x 4 row groups: lane (g = l / 16) register r then holds the total of value 4 g + r.  N + 1 MFMAs of
//          32 cycles each (MI355X_MICROARCH.md) for N values.
//   hybrid: K values through the MFMA path, 9 - K through the butterfly.
//   lds  : every lane stores 8 partial sums to LDS (row = value), lane (c = l >> 3, s = l & 7) adds 8 columns of row
//          c, three DPP adds fold the 8 lanes; the ninth value takes six DPP adds.
//   hipcc --offload-arch=gfx950 -O3 -I../../taichi_gaussian_rasterizer_amd/csrc mfma_reduce.hip -o bin/mfma_reduce
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "gs_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

// totals of N <= 12 values through the matrix pipe; lane l returns the total of value 4 (l / 16) + (l % 16) when
// l % 16 < 4 (other lanes: unspecified)
template <int N>
__device__ __forceinline__ float mfma_reduce(const float (&v)[N], int lane) {
  const int col = lane & 15;
  v4f d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const float sel = col == c ? 1.0f : 0.0f;
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c], sel, d, 0, 0, 0);
  }
  const float s = (d[0] + d[1]) + (d[2] + d[3]);
  v4f z = {0.f, 0.f, 0.f, 0.f};
  const v4f t = __builtin_amdgcn_mfma_f32_16x16x4f32(s, 1.0f, z, 0, 0, 0);
  const int r = lane & 3;
  return r == 0 ? t[0] : r == 1 ? t[1] : r == 2 ? t[2] : t[3];
}

__device__ __forceinline__ float lds_reduce8(const float (&v)[9], int lane, float* tr, float& ninth) {
#pragma unroll
  for (int c = 0; c < 8; ++c) tr[c * 68 + lane] = v[c];
  float x = v[8];
  x = gs_dpp_add_full<0x128>(x); x = gs_dpp_add_full<0x124>(x); x = gs_dpp_add_full<0x122>(x);
  x = gs_dpp_add_full<0x121>(x); x = gs_dpp_add_full<0x142>(x); x = gs_dpp_add_full<0x143>(x);
  ninth = x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float4* rowp = reinterpret_cast<const float4*>(tr + (lane >> 3) * 68 + (lane & 7) * 8);
  const float4 u0 = rowp[0], u1 = rowp[1];
  float t = ((u0.x + u0.y) + (u0.z + u0.w)) + ((u1.x + u1.y) + (u1.z + u1.w));
  t = gs_dpp_add_full<0xB1>(t); t = gs_dpp_add_full<0x4E>(t); t = gs_dpp_add_full<0x141>(t);
  __builtin_amdgcn_wave_barrier();
  return t;
}

#define FILL 80
// MODE 0: filler only; 1: + butterfly<9>; 2: + butterfly<8> + 1 value by MFMA; 3: + butterfly<5> + 4 by MFMA;
// 4: + all 9 by MFMA; 5: + LDS transpose of 8 + DPP ninth
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float tr[8 * 68];
  const int lane = threadIdx.x;
  float x[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) x[c] = float(lane + c) * 1e-3f;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f % 9]) : "v"(a), "v"(b));
    if (MODE == 1) {
      acc += gs_wave_reduce_transposed<9>(x, lane);
    } else if (MODE == 2) {
      float w[8], m[1] = {x[8]};
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = x[c];
      acc += gs_wave_reduce_transposed<8>(w, lane) + mfma_reduce<1>(m, lane);
    } else if (MODE == 3) {
      float w[5], m[4];
#pragma unroll
      for (int c = 0; c < 5; ++c) w[c] = x[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = x[5 + c];
      acc += gs_wave_reduce_transposed<5>(w, lane) + mfma_reduce<4>(m, lane);
    } else if (MODE == 4) {
      acc += mfma_reduce<9>(x, lane);
    } else if (MODE == 5) {
      float ninth;
      acc += lds_reduce8(x, lane, tr, ninth) + ninth;
    }
  }
  float s = acc;
#pragma unroll
  for (int c = 0; c < 9; ++c) s += x[c];
  out[blockIdx.x * 64 + lane] = s;
}

template <int N>
__global__ void check_kernel(float* out) {
  const int lane = threadIdx.x;
  float v[N];
  for (int c = 0; c < N; ++c) v[c] = float((lane * 7 + c * 13) % 31) - 11.0f + 0.25f * c;
  out[lane] = mfma_reduce<N>(v, lane);
}

__global__ void check_lds_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) float tr[8 * 68];
  const int lane = threadIdx.x;
  float v[9];
  for (int c = 0; c < 9; ++c) v[c] = float((lane * 7 + c * 13) % 31) - 11.0f + 0.25f * c;
  float ninth;
  out[lane] = lds_reduce8(v, lane, tr, ninth);
  out[64 + lane] = ninth;
}

static float want(int c) {
  float w = 0.f;
  for (int k = 0; k < 64; ++k) w += float((k * 7 + c * 13) % 31) - 11.0f + 0.25f * c;
  return w;
}

template <int N>
void check() {
  float* out; hipMalloc(&out, 64 * 4);
  hipLaunchKernelGGL(check_kernel<N>, dim3(1), dim3(64), 0, 0, out);
  float h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int c = 0; c < N; ++c) {
    const int l = (c / 4) * 16 + (c % 4);
    if (fabsf(h[l] - want(c)) > 1e-3f * (1.f + fabsf(want(c)))) { ++bad; printf("  value %d lane %d got %f want %f\n", c, l, h[l], want(c)); }
  }
  printf("check mfma_reduce<%d>: %s\n", N, bad ? "FAIL" : "ok");
  hipFree(out);
}

void check_lds() {
  float* out; hipMalloc(&out, 128 * 4);
  hipLaunchKernelGGL(check_lds_kernel, dim3(1), dim3(64), 0, 0, out);
  float h[128]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int c = 0; c < 8; ++c)
    if (fabsf(h[c * 8] - want(c)) > 1e-3f * (1.f + fabsf(want(c)))) { ++bad; printf("  value %d got %f want %f\n", c, h[c * 8], want(c)); }
  if (fabsf(h[64 + 60] - want(8)) > 1e-3f * (1.f + fabsf(want(8)))) { ++bad; printf("  ninth got %f want %f\n", h[64 + 60], want(8)); }
  printf("check lds_reduce8: %s\n", bad ? "FAIL" : "ok");
  hipFree(out);
}

template <int MODE>
float run(const char* name, int waves_per_simd, float base) {
  const int iters = 2000, blocks = 256 * 4 * waves_per_simd;
  float* out; hipMalloc(&out, size_t(blocks) * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const float per_call = ms * 1e6f / (float(iters) * waves_per_simd);  // ns per call per SIMD
  printf("%-34s waves/SIMD=%d  %.3f ms  %.1f ns per call per SIMD  (reduction: %+.1f ns)\n", name, waves_per_simd, ms,
         per_call, per_call - base);
  hipFree(out);
  return per_call;
}

int main() {
  check<1>(); check<4>(); check<9>(); check_lds();
  for (int w : {6, 3}) {
    const float base = run<0>("80 v_fma only", w, 0.f);
    run<1>("+ butterfly<9> (shipped)", w, base);
    run<2>("+ butterfly<8> + 1 value by MFMA", w, base);
    run<3>("+ butterfly<5> + 4 values by MFMA", w, base);
    run<4>("+ 9 values by MFMA (10 MFMAs)", w, base);
    run<5>("+ LDS transpose of 8 + DPP ninth", w, base);
  }
  return 0;
}
