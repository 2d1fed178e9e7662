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
  // optional fusions of what follows / precedes the moment update in FractionalOpt.step (optim/fractional.py:36-63)
  // and VisibilityOptimizer.step (optim/visibility_aware.py:85-108):
  const float* row_scale;  // (rows): the gradient of visible row i is multiplied by row_scale[i] first
  float* param;            // (N,dims): param[idx] -= step * saturate(weight) [* mask_lr[j]] [* point_lr[idx]]
  const float* mask_lr;    // (dims) or null
  const float* point_lr;   // (N) or null
};

// 1 - exp(-2 w) (optim/fractional.py:31-32), without cancellation for small w
__device__ __forceinline__ float saturate_weight(float w) { return -expm1f(-2.0f * w); }

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
  const float g = a.row_scale ? a.grad[at] * a.row_scale[i] : a.grad[at];
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
  if (a.lr_step) a.lr_step[e] = step;
  a.m[at] = m;
  a.v[at] = v;
  if (a.param) {
    float upd = step * saturate_weight(w);
    if (a.mask_lr) upd *= a.mask_lr[j];
    if (a.point_lr) upd *= a.point_lr[idx];
    a.param[at] -= upd;
  }
}

template <bool LAPROP>
__global__ __launch_bounds__(256) void optim_vector_kernel(OptArgs a) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= a.rows) return;
  const int64_t idx = a.indexes[i];
  const float w = a.weight[i], tw = a.total_weight[idx];
  const float gs = a.row_scale ? a.row_scale[i] : 1.0f;
  float norm = 0.0f;
  for (int j = 0; j < a.dims; ++j) { const float g = a.grad[idx * a.dims + j] * gs; norm += g * g; }
  float apply = 0.0f;
  if (a.param) {
    apply = saturate_weight(w);
    if (a.point_lr) apply *= a.point_lr[idx];
  }
  const float v = lerp_pow(a.beta2, w, a.v[idx], norm);
  if (LAPROP) {
    const float bias1 = a.bias_correction ? one_minus_pow(a.beta1, tw) : 1.0f;
    const float bias2 = a.bias_correction ? one_minus_pow(a.beta2, tw) : 1.0f;
    const float denom = fmaxf(sqrtf(v / bias2), a.eps);
    for (int j = 0; j < a.dims; ++j) {
      const float m = lerp_pow(a.beta1, w, a.m[idx * a.dims + j], a.grad[idx * a.dims + j] * gs / denom);
      const float step = m * a.lr / bias1;
      if (a.lr_step) a.lr_step[i * a.dims + j] = step;
      a.m[idx * a.dims + j] = m;
      if (a.param) a.param[idx * a.dims + j] -= step * apply * (a.mask_lr ? a.mask_lr[j] : 1.0f);
    }
  } else {
    const float bias = a.bias_correction ? sqrtf(one_minus_pow(a.beta2, tw)) / (one_minus_pow(a.beta1, tw)) : 1.0f;
    const float denom = fmaxf(sqrtf(v), a.eps);
    for (int j = 0; j < a.dims; ++j) {
      const float m = lerp_pow(a.beta1, w, a.m[idx * a.dims + j], a.grad[idx * a.dims + j] * gs);
      const float step = m / denom * bias * a.lr;
      if (a.lr_step) a.lr_step[i * a.dims + j] = step;
      a.m[idx * a.dims + j] = m;
      if (a.param) a.param[idx * a.dims + j] -= step * apply * (a.mask_lr ? a.mask_lr[j] : 1.0f);
    }
  }
  a.v[idx] = v;
}

// Per-view pacing of the visibility-aware optimizers (optim/visibility_aware.py:24-31, :93-103) for the visible rows:
// running visibility <- power mean (p = 4) of the view's visibility and the history; step weight = visibility /
// running; total_weight += weight; gradient scale = grad_scale / (visibility + vis_smooth).
__global__ __launch_bounds__(256) void visibility_weights_kernel(int64_t rows, const int64_t* indexes,
                                                                 const float* visibility, float* running_vis,
                                                                 float* total_weight, float vis_beta, float grad_scale,
                                                                 float vis_smooth, float* weight, float* row_scale) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= rows) return;
  const int64_t idx = indexes[i];
  const float vis = visibility[i], old = running_vis[idx];
  const float v2 = vis * vis, o2 = old * old;
  const float v4 = v2 * v2, o4 = o2 * o2;
  const float now = sqrtf(sqrtf(v4 + (o4 - v4) * vis_beta));
  running_vis[idx] = now;
  const float w = vis / fmaxf(now, 1e-12f);
  weight[i] = w;
  total_weight[idx] += w;  // indexes are unique (a visible set)
  row_scale[i] = grad_scale / (vis + vis_smooth);
}

}  // namespace

extern "C" int gs_optim_visibility_weights(int64_t rows, const int64_t* indexes, const float* visibility,
                                           float* running_vis, float* total_weight, float vis_beta, float grad_scale,
                                           float vis_smooth, float* weight, float* row_scale, void* stream) {
  if (rows == 0) return GS_OK;
  GS_REQUIRE(indexes && visibility && running_vis && total_weight && weight && row_scale, GS_ERR_INVALID_ARGUMENT,
             "gs_optim_visibility_weights: NULL buffer");
  hipLaunchKernelGGL(visibility_weights_kernel, dim3(unsigned(gs_div_up(rows, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rows, indexes, visibility, running_vis, total_weight, vis_beta,
                     grad_scale, vis_smooth, weight, row_scale);
  GS_CHECK_LAUNCH("gs_optim_visibility_weights");
  return GS_OK;
}

extern "C" int gs_optim_step(int32_t laprop, int32_t vector_group, int64_t rows, int32_t dims, const int64_t* indexes,
                             const float* weight, float* m, float* v, const float* total_weight, const float* grad,
                             float lr, float beta1, float beta2, float eps, int32_t bias_correction, float* lr_step,
                             const float* row_scale, float* param, const float* mask_lr, const float* point_lr,
                             void* stream) {
  GS_REQUIRE(dims >= 1, GS_ERR_INVALID_ARGUMENT, "gs_optim_step: dims %d", dims);
  if (rows == 0) return GS_OK;
  GS_REQUIRE(indexes && weight && m && v && total_weight && grad && (lr_step || param), GS_ERR_INVALID_ARGUMENT,
             "gs_optim_step: NULL buffer");
  OptArgs a{lr_step, indexes, weight, m, v, total_weight, grad, rows, dims, lr, beta1, beta2, eps, bias_correction,
            row_scale, param, mask_lr, point_lr};
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
