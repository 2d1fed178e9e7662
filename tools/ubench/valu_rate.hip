// Micro-benchmark: VALU issue rate on gfx950 for v_fma_f32, v_pk_fma_f32, v_exp_f32, DPP adds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  float2_ av = {a, a}, bv = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
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
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(av), "v"(bv));
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x2)); asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x4)); asm volatile("v_exp_f32 %0, %0" : "+v"(x5));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x6)); asm volatile("v_exp_f32 %0, %0" : "+v"(x7));
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x0));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x1));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x2));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x3));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x4));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x5));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x6));
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x7));
      }
    } else if (MODE == 4) {  // mul (v_mul_f32)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x1) : "v"(a));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(a));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x4) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x5) : "v"(a));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x6) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x7) : "v"(a));
      }
    } else if (MODE == 5 || MODE == 6) {  // compare + add-with-carry pairs: 64-bit (5) or 32-bit (6) compare
      unsigned long k0 = threadIdx.x * 77u + i, k1 = k0 * 3u;
      unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      unsigned long m0 = threadIdx.x, m1 = m0 + 9, m2 = m0 + 17, m3 = m0 + 29;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE == 5) {
          asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r0) : "v"(k0), "v"(m0) : "vcc");
          asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r1) : "v"(k0), "v"(m1) : "vcc");
          asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r2) : "v"(k1), "v"(m2) : "vcc");
          asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r3) : "v"(k1), "v"(m3) : "vcc");
        } else {
          asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r0) : "v"(unsigned(k0)), "v"(unsigned(m0)) : "vcc");
          asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r1) : "v"(unsigned(k0)), "v"(unsigned(m1)) : "vcc");
          asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r2) : "v"(unsigned(k1)), "v"(unsigned(m2)) : "vcc");
          asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r3) : "v"(unsigned(k1)), "v"(unsigned(m3)) : "vcc");
        }
      }
      x0 += float(r0 + r1 + r2 + r3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int instr_per_iter) {
  const int cus = 256, iters = 4000;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
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
  double instr_per_simd = double(iters) * instr_per_iter * waves_per_simd;
  printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, 64);
    run<1>("v_pk_fma_f32", w, 64);
    run<2>("v_exp_f32", w, 64);
    run<3>("v_add_f32_dpp", w, 64);
    run<4>("v_mul_f32", w, 64);
    run<5>("cmp_u64+addc", w, 64);
    run<6>("cmp_u32+addc", w, 64);
  }
  return 0;
}
