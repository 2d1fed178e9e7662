// Micro-benchmark (gfx950): what do the cross-lane building blocks of the backward's per-splat reduction cost, and
// does a wave64 VALU instruction whose EXEC mask covers only lanes 0-31 issue faster than a full one?
//   hipcc --offload-arch=gfx950 -O3 -I../../taichi_gaussian_rasterizer_amd/csrc xlane_rate.hip -o bin/xlane_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "gs_common.h"

// ---- EXPERIMENT (not in the library): transposed butterfly, cheap stages first.  Same speed as the shipped
// gs_wave_reduce_transposed in the micro-benchmark (76 vs 77 ns per 9-value call at 5 waves per SIMD) and in
// raster_bwd_kernel itself (0.653 - 0.663 vs 0.661 - 0.666 ms): the reduction is bound by its ~25 dependent cross-lane
// instructions, not by which of them are swaps.
// Measured on MI355X (tools/ubench/xlane_rate.hip, ns per wave instruction per SIMD): plain VALU 1.1, DPP add 1.8,
// v_permlane32_swap / v_permlane16_swap 3.4.  gs_wave_reduce_transposed above spends its swaps on ALL N values first
// (9 values: 67 ns per call).  This version first halves the register count twice INSIDE the 16-lane rows with DPP
// adds whose bank mask write-enables the even / the odd quads (a DPP bank = 4 consecutive lanes): two v_add_f32_dpp
// per register pair; only then does it cross the 16- and 32-lane boundaries with the swaps, on the 3 and 2 registers
// that are left.  The four lanes of each quad are summed last, on the single remaining register.
//   stage 1  lane bit 2   row_shl:4 on quads 0,2 (bank mask 0x5) / row_shr:4 on quads 1,3 (0xa)    N  -> h1 registers
//   stage 2  lane bit 3   row_shl:8 on quads 0,1 (0x3)           / row_shr:8 on quads 2,3 (0xc)    h1 -> h2
//   stage 3  lane bit 4   v_permlane16_swap                                                        h2 -> h3
//   stage 4  lane bit 5   v_permlane32_swap                                                        h3 -> h4 = 1
//   stage 5  lane bits 0,1  quad_perm xor 1, xor 2 (plain sums inside each quad)
// On return lane l holds the wave total of value gs_reduce2_slot<N>(l) (or garbage where that is -1).
template <int N>
struct GsReduce2Shape {
  static constexpr int h1 = (N + 1) / 2, h2 = (h1 + 1) / 2, h3 = (h2 + 1) / 2, h4 = (h3 + 1) / 2;
  static_assert(N >= 1 && N <= 16 && h4 == 1, "one register must remain after four halvings");
};

template <int N>
__device__ __forceinline__ int gs_reduce2_slot(int lane) {
  typedef GsReduce2Shape<N> S;
  const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, p = (lane >> 4) & 1, half = lane >> 5;
  const int ci = half;  // stage 4 picked c[half]
  if (ci >= S::h3) return -1;
  const int bi = ci + p * S::h3;  // stage 3: odd rows carry b[ci + h3]
  if (bi >= S::h2) return -1;
  const int ai = bi + b3 * S::h2;  // stage 2: quads 2,3 carry a[bi + h2]
  if (ai >= S::h1) return -1;
  const int vi = ai + b2 * S::h1;  // stage 1: quads 1,3 carry v[ai + h1]
  return vi < N ? vi : -1;
}

// out = x + (x shifted inside its 16-lane row) on the quads selected by BANK_MASK; the other lanes of `out` are kept
#define GS_DPP_ADD_ROW(out, x, SHIFT, BANK_MASK)                                                                    \
  __asm__ volatile("v_add_f32_dpp %0, %1, %1 " SHIFT " row_mask:0xf bank_mask:" BANK_MASK : "+v"(out) : "v"(x))

template <int N>
__device__ __forceinline__ float gs_wave_reduce2_transposed(float (&v)[N], int lane) {
  typedef GsReduce2Shape<N> S;
  float a[S::h1];  // deliberately not initialised: lanes a stage leaves unwritten carry no value (slot -1)
  // A VGPR written by a VALU instruction may be read as a DPP source only 2 wait states later, and the compiler's
  // hazard recognizer does not look inside inline assembly.  The empty volatile statements pin the producers of every
  // v[i] in front of the s_nop (volatile statements keep their order); nothing can be scheduled between them and the
  // DPP reads that would write a v[i].
#pragma unroll
  for (int i = 0; i < N; ++i) __asm__ volatile("" : "+v"(v[i]));
  __asm__ volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < S::h1; ++i) {
    GS_DPP_ADD_ROW(a[i], v[i], "row_shl:4", "0x5");
    if (i + S::h1 < N) GS_DPP_ADD_ROW(a[i], v[i + S::h1 < N ? i + S::h1 : 0], "row_shr:4", "0xa");
  }
  float b[S::h2];
  __asm__ volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < S::h2; ++i) {
    GS_DPP_ADD_ROW(b[i], a[i], "row_shl:8", "0x3");
    if (i + S::h2 < S::h1) GS_DPP_ADD_ROW(b[i], a[i + S::h2 < S::h1 ? i + S::h2 : 0], "row_shr:8", "0xc");
  }
  __asm__ volatile("s_nop 1");  // the swaps below are the compiler's own instructions, the writes above are not
  float c[S::h3];
#pragma unroll
  for (int i = 0; i < S::h3; ++i) {
    const float partner = (i + S::h3 < S::h2) ? b[i + S::h3 < S::h2 ? i + S::h3 : 0] : 0.0f;
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(b[i]), __float_as_uint(partner), false, false);
    c[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float d;
  {
    const float partner = S::h3 > 1 ? c[S::h3 > 1 ? 1 : 0] : 0.0f;
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[0]), __float_as_uint(partner), false, false);
    d = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  d = gs_dpp_add_full<0xB1>(d);  // quad_perm:[1,0,3,2]
  d = gs_dpp_add_full<0x4E>(d);  // quad_perm:[2,3,0,1]
  return d;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float acc = 0.f;
  if (MODE == 0 || MODE == 1 || MODE == 8) {
    // MODE 0: full EXEC; MODE 1: EXEC = low 32 lanes only; MODE 8: lanes 0-15 only
    const bool active = MODE == 0 ? true : (MODE == 1 ? lane < 32 : lane < 16);
    if (active) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
        }
      }
    }
  } else if (MODE == 2) {  // v_permlane32_swap
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x2), "+v"(x3));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x4), "+v"(x5));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x6), "+v"(x7));
      }
    }
  } else if (MODE == 3) {  // v_permlane16_swap
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x2), "+v"(x3));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x4), "+v"(x5));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x6), "+v"(x7));
      }
    }
  } else if (MODE == 4) {  // v_rcp_f32
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x1));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x2)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x3));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x4)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x5));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x6)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x7));
      }
    }
  } else if (MODE == 5) {  // the 9-value transposed butterfly of raster_bwd (one call per "instruction")
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[9] = {x0, x1, x2, x3, x4, x5, x6, x7, x0 + x1};
        const float t = gs_wave_reduce_transposed<9>(v, lane);
        x0 += t * 1e-9f; x1 += a; x2 += b; x3 += a; x4 += b; x5 += a; x6 += b; x7 += a;
      }
    }
  } else if (MODE == 6) {  // 8 values
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
        const float t = gs_wave_reduce_transposed<8>(v, lane);
        x0 += t * 1e-9f; x1 += a; x2 += b; x3 += a; x4 += b; x5 += a; x6 += b; x7 += a;
      }
    }
  } else if (MODE == 7) {  // 16 values
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[16] = {x0, x1, x2, x3, x4, x5, x6, x7, x0 + a, x1 + a, x2 + a, x3 + a, x4 + a, x5 + a, x6 + a, x7 + a};
        const float t = gs_wave_reduce16_transposed(v, lane);
        x0 += t * 1e-9f; x1 += a; x2 += b; x3 += a; x4 += b; x5 += a; x6 += b; x7 += a;
      }
    }
  } else if (MODE == 10 || MODE == 11) {  // cheap-stages-first butterfly, 9 / 11 values, checked against plain sums
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 10) {
          float v[9] = {x0, x1, x2, x3, x4, x5, x6, x7, x0 + x1};
          const float t = gs_wave_reduce2_transposed<9>(v, lane);
          x0 += t * 1e-9f;
        } else {
          float v[11] = {x0, x1, x2, x3, x4, x5, x6, x7, x0 + x1, x2 + x3, x4 + x5};
          const float t = gs_wave_reduce2_transposed<11>(v, lane);
          x0 += t * 1e-9f;
        }
        x1 += a; x2 += b; x3 += a; x4 += b; x5 += a; x6 += b; x7 += a;
      }
    }
  } else if (MODE == 9) {  // same update arithmetic as MODE 5-7 without any reduction (baseline to subtract)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x0 += x1 * 1e-9f; x1 += a; x2 += b; x3 += a; x4 += b; x5 += a; x6 += b; x7 += a;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc;
}

// correctness of the reduce variants against __shfl-free plain sums (one wave)
template <int N, int VER>
__global__ void check_kernel(float* out) {
  const int lane = threadIdx.x;
  float v[N];
  for (int c = 0; c < N; ++c) v[c] = float((lane * 7 + c * 13) % 31) - 11.0f + 0.25f * c;
  float t;
  int slot;
  if (VER == 1) { t = gs_wave_reduce_transposed<N>(v, lane); slot = gs_reduce_slot<N>(lane); if ((lane & 3) != 0) slot = -1; }
  else { t = gs_wave_reduce2_transposed<N>(v, lane); slot = gs_reduce2_slot<N>(lane); if (lane & 3) slot = -1; }
  out[lane * 2] = t;
  out[lane * 2 + 1] = float(slot);
}

template <int N, int VER>
void check(const char* name) {
  float* out;
  hipMalloc(&out, 128 * 4);
  hipLaunchKernelGGL((check_kernel<N, VER>), dim3(1), dim3(64), 0, 0, out);
  float h[128];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int found[16] = {0}, bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int slot = int(h[l * 2 + 1]);
    if (slot < 0) continue;
    float want = 0.f;
    for (int k = 0; k < 64; ++k) want += float((k * 7 + slot * 13) % 31) - 11.0f + 0.25f * slot;
    if (fabsf(want - h[l * 2]) > 1e-3f * (1.f + fabsf(want))) { ++bad; printf("  lane %d slot %d got %f want %f\n", l, slot, h[l * 2], want); }
    found[slot]++;
  }
  int missing = 0;
  for (int c = 0; c < N; ++c) missing += found[c] == 0;
  printf("check %-10s N=%2d: %s (%d wrong, %d values without an owner lane)\n", name, N, (bad || missing) ? "FAIL" : "ok", bad, missing);
  hipFree(out);
}

template <int MODE>
void run(const char* name, int waves_per_simd, int instr_per_iter) {
  const int cus = 256, iters = 4000;
  const int blocks = cus * waves_per_simd;
  float* out;
  hipMalloc(&out, size_t(blocks) * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_simd = double(iters) * instr_per_iter * waves_per_simd;
  printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per op per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
  hipFree(out);
}

int main() {
  check<9, 1>("reduce v1"); check<9, 2>("reduce v2"); check<11, 2>("reduce v2"); check<14, 2>("reduce v2");
  check<16, 2>("reduce v2"); check<8, 2>("reduce v2"); check<6, 2>("reduce v2"); check<1, 2>("reduce v2");
  for (int w : {2, 5}) {
    run<0>("v_fma full exec", w, 64);
    run<1>("v_fma lanes 0-31", w, 64);
    run<8>("v_fma lanes 0-15", w, 64);
    run<2>("v_permlane32_swap", w, 64);
    run<3>("v_permlane16_swap", w, 64);
    run<4>("v_rcp_f32", w, 64);
    run<9>("update only", w, 4);
    run<5>("reduce9 + update", w, 4);
    run<6>("reduce8 + update", w, 4);
    run<7>("reduce16 + update", w, 4);
    run<10>("reduce2<9> + update", w, 4);
    run<11>("reduce2<11> + update", w, 4);
  }
  return 0;
}
