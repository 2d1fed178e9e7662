// optim.hip -- per-row sparse / fractional Adam and LaProp steps (the step that follows
// the render path in the training loop).  Reference: optim/fractional_adam.py:7-85 (scalar_kernel,
// vector_kernel) and optim/fractional_laprop.py (same signatures).  For every visible row i
// (idx = indexes[i], fractional step weight w = weight[i]):
//   Adam   : m = lerp(b1^w, m, g); v = lerp(b2^w, v, g*g [or |g|^2]); step = m / max(sqrt(v), eps) * bias * lr
//            bias = sqrt(1 - b2^tw) / (1 - b1^tw),  tw = total_weight[idx]
//   LaProp : v = lerp(b2^w, v, g*g [or |g|^2]); m = lerp(b1^w, m, g / max(sqrt(v / bias2), eps)); step = m * lr / bias1
// with lerp(t, a, b) = a*t + b*(1-t) (taichi_lib/generic.py:489-491).  "scalar": v per element;
// "vector": one v per row from the squared norm of the row's gradient.
// Elementwise and HBM-bound: one lane per (row, element) for scalar groups, one lane per row for vector groups.

#include "gs_common.h"

namespace {

struct OptArgs {
  float* lr_step;
  const int64_t* indexes;
  const float* weight;
  float* m;
  float* v;
  const float* total_weight;
  const float* grad;
  int64_t rows;
  int dims;
  float lr, beta1, beta2, eps;
  int bias_correction;
};

// 1 - beta^t without the cancellation the literal form has for small t (beta2 = 0.999, t < 1 leaves ~3 digits in f32)
__device__ __forceinline__ float one_minus_pow(float beta, float t) { return -expm1f(t * logf(beta)); }

// lerp(beta^w, a, b) = a * beta^w + b * (1 - beta^w), with 1 - beta^w formed without cancellation
__device__ __forceinline__ float lerp_pow(float beta, float w, float a, float b) {
  const float e = w * logf(beta);
  return a * expf(e) - b * expm1f(e);
}

template <bool LAPROP>
__global__ __launch_bounds__(256) void optim_scalar_kernel(OptArgs a) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (e >= a.rows * a.dims) return;
  const int64_t i = e / a.dims;
  const int j = int(e - i * a.dims);
  const int64_t idx = a.indexes[i];
  const float w = a.weight[i], tw = a.total_weight[idx];
  const int64_t at = idx * a.dims + j;
  const float g = a.grad[at];
  float m, v, step;
  if (LAPROP) {
    const float bias1 = a.bias_correction ? one_minus_pow(a.beta1, tw) : 1.0f;
    const float bias2 = a.bias_correction ? one_minus_pow(a.beta2, tw) : 1.0f;
    v = lerp_pow(a.beta2, w, a.v[at], g * g);
    m = lerp_pow(a.beta1, w, a.m[at], g / fmaxf(sqrtf(v / bias2), a.eps));
    step = m * a.lr / bias1;
  } else {
    const float bias = a.bias_correction ? sqrtf(one_minus_pow(a.beta2, tw)) / (one_minus_pow(a.beta1, tw)) : 1.0f;
    m = lerp_pow(a.beta1, w, a.m[at], g);
    v = lerp_pow(a.beta2, w, a.v[at], g * g);
    step = m / fmaxf(sqrtf(v), a.eps) * bias * a.lr;
  }
  a.lr_step[e] = step;
  a.m[at] = m;
  a.v[at] = v;
}

template <bool LAPROP>
__global__ __launch_bounds__(256) void optim_vector_kernel(OptArgs a) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= a.rows) return;
  const int64_t idx = a.indexes[i];
  const float w = a.weight[i], tw = a.total_weight[idx];
  float norm = 0.0f;
  for (int j = 0; j < a.dims; ++j) { const float g = a.grad[idx * a.dims + j]; norm += g * g; }
  const float v = lerp_pow(a.beta2, w, a.v[idx], norm);
  if (LAPROP) {
    const float bias1 = a.bias_correction ? one_minus_pow(a.beta1, tw) : 1.0f;
    const float bias2 = a.bias_correction ? one_minus_pow(a.beta2, tw) : 1.0f;
    const float denom = fmaxf(sqrtf(v / bias2), a.eps);
    for (int j = 0; j < a.dims; ++j) {
      const float m = lerp_pow(a.beta1, w, a.m[idx * a.dims + j], a.grad[idx * a.dims + j] / denom);
      a.lr_step[i * a.dims + j] = m * a.lr / bias1;
      a.m[idx * a.dims + j] = m;
    }
  } else {
    const float bias = a.bias_correction ? sqrtf(one_minus_pow(a.beta2, tw)) / (one_minus_pow(a.beta1, tw)) : 1.0f;
    const float denom = fmaxf(sqrtf(v), a.eps);
    for (int j = 0; j < a.dims; ++j) {
      const float m = lerp_pow(a.beta1, w, a.m[idx * a.dims + j], a.grad[idx * a.dims + j]);
      a.lr_step[i * a.dims + j] = m / denom * bias * a.lr;
      a.m[idx * a.dims + j] = m;
    }
  }
  a.v[idx] = v;
}

}  // namespace

extern "C" int gs_optim_step(int32_t laprop, int32_t vector_group, int64_t rows, int32_t dims, const int64_t* indexes,
                             const float* weight, float* m, float* v, const float* total_weight, const float* grad,
                             float lr, float beta1, float beta2, float eps, int32_t bias_correction, float* lr_step,
                             void* stream) {
  GS_REQUIRE(dims >= 1, GS_ERR_INVALID_ARGUMENT, "gs_optim_step: dims %d", dims);
  if (rows == 0) return GS_OK;
  GS_REQUIRE(indexes && weight && m && v && total_weight && grad && lr_step, GS_ERR_INVALID_ARGUMENT,
             "gs_optim_step: NULL buffer");
  OptArgs a{lr_step, indexes, weight, m, v, total_weight, grad, rows, dims, lr, beta1, beta2, eps, bias_correction};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vector_group) {
    const dim3 grid(unsigned(gs_div_up(rows, 256)));
    if (laprop) hipLaunchKernelGGL(optim_vector_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(optim_vector_kernel<false>, grid, dim3(256), 0, s, a);
  } else {
    const dim3 grid(unsigned(gs_div_up(rows * dims, 256)));
    if (laprop) hipLaunchKernelGGL(optim_scalar_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(optim_scalar_kernel<false>, grid, dim3(256), 0, s, a);
  }
  GS_CHECK_LAUNCH("gs_optim_step");
  return GS_OK;
}
