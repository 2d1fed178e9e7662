// gsplat_oracle.cpp -- CPU restatement of the taichi_splatting render path.
//
// TEST INFRASTRUCTURE.  This file is the parity oracle and the CPU baseline ("port") for
// bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
// library built from it.  The product path (taichi_gaussian_rasterizer_amd/) never does.
//
// Pinning: projection and spherical harmonics (forward values and gradients) are checked
// against golden vectors produced by the reference's own torch_lib (oracle/make_golden.py,
// tests/golden/*.npz).  The tile mapper and the rasterizer have no executable reference in this
// container (Taichi + CUDA only) and the reference holds no golden vectors for them:
// for those stages this oracle is "parity unpinned" against the reference itself and is
// instead validated by independent means in tests/ (a dense O(pixels x splats) torch autograd
// renderer, f64 gradcheck, the visibility identity of tests/test_visibility.py, brute-force
// mapper invariants).
//
// Each function cites the reference file:line it restates (paths relative to
// /root/reference/taichi_splatting/).  One deliberate deviation: the reference computes
// `remaining_points = tile_point_count - point_group_id` (rasterizer/forward.py:88,
// backward.py:144) which re-blends stale entries of the last partial group; this file blends
// each overlap exactly once (the evident intent).
//
// Build: see oracle/Makefile (g++ -O2 -fopenmp -ffp-contract=off).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../include/gs_detmath.h"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// forward-mode dual numbers: used to obtain the gradients of projection / SH mechanically from
// the forward restatement (the HIP kernels use hand-derived adjoints; two independent
// derivations that must agree).
// ---------------------------------------------------------------------------------------------
template <typename T, int N>
struct Dual {
  T v;
  T d[N];
  Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
  Dual(T x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0; }
  static Dual seed(T x, int k) { Dual r(x); r.d[k] = 1; return r; }
};

template <typename T, int N> Dual<T, N> operator+(const Dual<T, N>& a, const Dual<T, N>& b) {
  Dual<T, N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <typename T, int N> Dual<T, N> operator-(const Dual<T, N>& a, const Dual<T, N>& b) {
  Dual<T, N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <typename T, int N> Dual<T, N> operator-(const Dual<T, N>& a) {
  Dual<T, N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <typename T, int N> Dual<T, N> operator*(const Dual<T, N>& a, const Dual<T, N>& b) {
  Dual<T, N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <typename T, int N> Dual<T, N> operator/(const Dual<T, N>& a, const Dual<T, N>& b) {
  Dual<T, N> r; r.v = a.v / b.v; T inv = T(1) / b.v;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
template <typename T, int N> Dual<T, N> operator+(const Dual<T, N>& a, T b) { Dual<T, N> r = a; r.v += b; return r; }
template <typename T, int N> Dual<T, N> operator+(T b, const Dual<T, N>& a) { Dual<T, N> r = a; r.v += b; return r; }
template <typename T, int N> Dual<T, N> operator-(const Dual<T, N>& a, T b) { Dual<T, N> r = a; r.v -= b; return r; }
template <typename T, int N> Dual<T, N> operator-(T b, const Dual<T, N>& a) { Dual<T, N> r = -a; r.v += b; return r; }
template <typename T, int N> Dual<T, N> operator*(const Dual<T, N>& a, T b) {
  Dual<T, N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <typename T, int N> Dual<T, N> operator*(T b, const Dual<T, N>& a) { return a * b; }
template <typename T, int N> Dual<T, N> operator/(const Dual<T, N>& a, T b) {
  Dual<T, N> r; r.v = a.v / b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r; }
template <typename T, int N> Dual<T, N> operator/(T a, const Dual<T, N>& b) { return Dual<T, N>(a) / b; }

template <typename T> T value_of(T x) { return x; }
template <typename T, int N> T value_of(const Dual<T, N>& x) { return x.v; }

inline float m_sqrt(float x) { return std::sqrt(x); }
inline double m_sqrt(double x) { return std::sqrt(x); }
inline float m_exp(float x) { return std::exp(x); }
inline double m_exp(double x) { return std::exp(x); }
inline float m_log(float x) { return std::log(x); }
inline double m_log(double x) { return std::log(x); }
template <typename T, int N> Dual<T, N> m_sqrt(const Dual<T, N>& a) {
  Dual<T, N> r; r.v = std::sqrt(a.v); T g = T(0.5) / r.v;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] == T(0)) ? T(0) : a.d[i] * g; return r; }
template <typename T, int N> Dual<T, N> m_exp(const Dual<T, N>& a) {
  Dual<T, N> r; r.v = std::exp(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v; return r; }

// max(x, c) and clamp(x, lo, hi) with the sub-gradient conventions of torch.clamp
// (gradient 1 on the closed interval, torch_lib/projection.py:31,92).
template <typename T> T m_max0(T x) { return x > T(0) ? x : T(0); }
template <typename T, int N> Dual<T, N> m_max0(const Dual<T, N>& x) {
  if (x.v >= T(0)) return x;
  return Dual<T, N>(T(0));
}
template <typename T> T m_clamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
template <typename T, int N> Dual<T, N> m_clamp(const Dual<T, N>& x, T lo, T hi) {
  if (x.v < lo) return Dual<T, N>(lo);
  if (x.v > hi) return Dual<T, N>(hi);
  return x;
}

// ---------------------------------------------------------------------------------------------
// Projection.  perspective/projection.py:32-80 (project_kernel), taichi_lib/generic.py:96-158
// (project_with_jacobian, gaussian_covariance_in_image, project_gaussian), :217-230 (eig),
// :164-165 (sigmoid), :419-427 (scaled_quat_to_mat; quaternion unpacked x,y,z,w).
// S is T or Dual<T,N>; constants stay T.
// ---------------------------------------------------------------------------------------------
template <typename T, typename S>
struct Projected {
  S mean[2];
  S z;
  S axis[2];
  S sigma[2];
  S alpha;
};

template <typename T, typename S>
Projected<T, S> project_core(const S pos[3], const S log_scale[3], const S quat[4], S alpha_logit,
                             const S Tcw[12], const S proj[4], T width, T height, T clamp_margin, T blur_cov) {
  Projected<T, S> out;
  // ti.math.normalize(rotation): v / sqrt(v.v)   (projection.py:50)
  S qn2 = quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3];
  S qn = m_sqrt(qn2);
  S x = quat[0] / qn, y = quat[1] / qn, zq = quat[2] / qn, w = quat[3] / qn;
  S s0 = m_exp(log_scale[0]), s1 = m_exp(log_scale[1]), s2 = m_exp(log_scale[2]);

  // generic.py:107-110
  S cam[3];
  for (int r = 0; r < 3; ++r)
    cam[r] = Tcw[r * 4 + 0] * pos[0] + Tcw[r * 4 + 1] * pos[1] + Tcw[r * 4 + 2] * pos[2] + Tcw[r * 4 + 3];
  S z = cam[2];
  S u = (proj[0] * cam[0]) / z + proj[2];
  S v = (proj[1] * cam[1]) / z + proj[3];

  // generic.py:114  t = clamp(uv, -size*margin, (size-1)*(1+margin))
  S tx = m_clamp(u, -width * clamp_margin, (width - T(1)) * (T(1) + clamp_margin));
  S ty = m_clamp(v, -height * clamp_margin, (height - T(1)) * (T(1) + clamp_margin));

  // generic.py:116-119
  S J00 = proj[0] / z, J02 = -(tx - proj[2]) / z;
  S J11 = proj[1] / z, J12 = -(ty - proj[3]) / z;

  // generic.py:419-427
  S x2 = x * x, y2 = y * y, z2 = zq * zq;
  S RS[3][3] = {
      {s0 * (T(1) - T(2) * y2 - T(2) * z2), s1 * (T(2) * x * y - T(2) * w * zq), s2 * (T(2) * x * zq + T(2) * w * y)},
      {s0 * (T(2) * x * y + T(2) * w * zq), s1 * (T(1) - T(2) * x2 - T(2) * z2), s2 * (T(2) * y * zq - T(2) * w * x)},
      {s0 * (T(2) * x * zq - T(2) * w * y), s1 * (T(2) * y * zq + T(2) * w * x), s2 * (T(1) - T(2) * x2 - T(2) * y2)}};

  // m = J @ W @ RS ; cov = m m^T   (generic.py:134-143)
  S JW[2][3];
  for (int c = 0; c < 3; ++c) {
    JW[0][c] = J00 * Tcw[0 * 4 + c] + J02 * Tcw[2 * 4 + c];
    JW[1][c] = J11 * Tcw[1 * 4 + c] + J12 * Tcw[2 * 4 + c];
  }
  S m[2][3];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      m[r][c] = JW[r][0] * RS[0][c] + JW[r][1] * RS[1][c] + JW[r][2] * RS[2][c];
  S c00 = m[0][0] * m[0][0] + m[0][1] * m[0][1] + m[0][2] * m[0][2];
  S c01 = m[0][0] * m[1][0] + m[0][1] * m[1][1] + m[0][2] * m[1][2];
  S c11 = m[1][0] * m[1][0] + m[1][1] * m[1][1] + m[1][2] * m[1][2];
  // projection.py:55-56
  c00 = c00 + blur_cov;
  c11 = c11 + blur_cov;

  // generic.py:217-230
  S tr = c00 + c11;
  S det = c00 * c11 - c01 * c01;
  S gap = tr * tr - T(4) * det;
  S sqrt_gap = m_sqrt(m_max0(gap));
  S l1 = (tr + sqrt_gap) * T(0.5);
  S l2 = (tr - sqrt_gap) * T(0.5);
  S vx = c00 - l2, vy = c01;
  S vn = m_sqrt(vx * vx + vy * vy);

  out.mean[0] = u;
  out.mean[1] = v;
  out.z = z;
  out.axis[0] = vx / vn;
  out.axis[1] = vy / vn;
  out.sigma[0] = m_sqrt(l1);
  out.sigma[1] = m_sqrt(l2);
  out.alpha = T(1) / (T(1) + m_exp(-alpha_logit));
  return out;
}

inline float cull_log(float x) { return gs_det_logf(x); }
inline double cull_log(double x) { return std::log(x); }

template <typename T>
void project_fwd(int64_t n, const T* pos, const T* log_scale, const T* quat, const T* alpha_logit, const T* Tcw44,
                 const T* proj, int W, int H, double near_d, double far_d, double blur_cov, double clamp_margin,
                 double alpha_thr, T* points, T* depth, uint8_t* visible) {
  T Tcw[12];
  for (int i = 0; i < 12; ++i) Tcw[i] = Tcw44[i];
  const T width = T(W), height = T(H), near_p = T(near_d), far_p = T(far_d), thr = T(alpha_thr);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    Projected<T, T> g = project_core<T, T>(pos + 3 * i, log_scale + 3 * i, quat + 4 * i, alpha_logit[i], Tcw, proj,
                                           width, height, T(clamp_margin), T(blur_cov));
    // projection.py:60-67: opacity-aware extent and the view test.  NaN (alpha < thr) fails
    // every comparison, i.e. culls.
    T gs = m_sqrt(T(2) * cull_log(g.alpha / thr));
    T sx = g.sigma[0] * gs, sy = g.sigma[1] * gs;
    T v1x = g.axis[0] * sx, v1y = g.axis[1] * sx;
    T v2x = -g.axis[1] * sy, v2y = g.axis[0] * sy;
    T ex = m_sqrt(v1x * v1x + v2x * v2x), ey = m_sqrt(v1y * v1y + v2y * v2y);
    T lox = g.mean[0] - ex, loy = g.mean[1] - ey, hix = g.mean[0] + ex, hiy = g.mean[1] + ey;
    bool in_view = (g.z > near_p) && (g.z < far_p) && (hix > T(0)) && (hiy > T(0)) && (lox < width) && (loy < height);
    visible[i] = in_view ? 1 : 0;
    T* p = points + 7 * i;
    if (in_view) {
      depth[i] = g.z;
      p[0] = g.mean[0]; p[1] = g.mean[1]; p[2] = g.axis[0]; p[3] = g.axis[1];
      p[4] = g.sigma[0]; p[5] = g.sigma[1]; p[6] = g.alpha;
    } else {
      depth[i] = T(0);  // projection.py:69-70 (row left undefined there; zero here)
      for (int k = 0; k < 7; ++k) p[k] = T(0);
    }
  }
}

// Gradient of sum(gpoints*points + gdepth*depth) w.r.t. every input of the indexed projection
// (projection.py:84-118 differentiated by Taichi autodiff at :175-180; no cull inside).
// Camera gradients are summed over the visible set (what autograd's expand-backward does,
// projection.py:212-213).
template <typename T>
void project_bwd(int64_t n, int64_t v, const T* pos, const T* log_scale, const T* quat, const T* alpha_logit,
                 const T* Tcw44, const T* proj, int W, int H, double blur_cov, double clamp_margin,
                 const int64_t* indexes, const T* gpoints, const T* gdepth, T* dpos, T* dlog_scale, T* dquat,
                 T* dalpha_logit, T* dTcw44, T* dproj) {
  constexpr int ND = 27;  // 3 pos + 3 scale + 4 quat + 1 alpha + 12 T + 4 proj
  typedef Dual<T, ND> D;
  std::fill(dpos, dpos + 3 * n, T(0));
  std::fill(dlog_scale, dlog_scale + 3 * n, T(0));
  std::fill(dquat, dquat + 4 * n, T(0));
  std::fill(dalpha_logit, dalpha_logit + n, T(0));
  std::vector<double> cam_acc(16, 0.0);
#pragma omp parallel
  {
    std::vector<double> local(16, 0.0);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < v; ++i) {
      int64_t idx = indexes[i];
      D p[3], ls[3], q[4], al, Tm[12], pr[4];
      int k = 0;
      for (int j = 0; j < 3; ++j) p[j] = D::seed(pos[3 * idx + j], k++);
      for (int j = 0; j < 3; ++j) ls[j] = D::seed(log_scale[3 * idx + j], k++);
      for (int j = 0; j < 4; ++j) q[j] = D::seed(quat[4 * idx + j], k++);
      al = D::seed(alpha_logit[idx], k++);
      for (int j = 0; j < 12; ++j) Tm[j] = D::seed(Tcw44[j], k++);
      for (int j = 0; j < 4; ++j) pr[j] = D::seed(proj[j], k++);
      Projected<T, D> g = project_core<T, D>(p, ls, q, al, Tm, pr, T(W), T(H), T(clamp_margin), T(blur_cov));
      const T* gp = gpoints + 7 * i;
      D L = g.mean[0] * gp[0] + g.mean[1] * gp[1] + g.axis[0] * gp[2] + g.axis[1] * gp[3] + g.sigma[0] * gp[4] +
            g.sigma[1] * gp[5] + g.alpha * gp[6] + g.z * gdepth[i];
      for (int j = 0; j < 3; ++j) dpos[3 * idx + j] = L.d[j];
      for (int j = 0; j < 3; ++j) dlog_scale[3 * idx + j] = L.d[3 + j];
      for (int j = 0; j < 4; ++j) dquat[4 * idx + j] = L.d[6 + j];
      dalpha_logit[idx] = L.d[10];
      for (int j = 0; j < 16; ++j) local[j] += double(L.d[11 + j]);
    }
#pragma omp critical
    for (int j = 0; j < 16; ++j) cam_acc[j] += local[j];
  }
  for (int j = 0; j < 16; ++j) dTcw44[j] = T(0);
  for (int j = 0; j < 12; ++j) dTcw44[j] = T(cam_acc[j]);
  for (int j = 0; j < 4; ++j) dproj[j] = T(cam_acc[12 + j]);
}

// ---------------------------------------------------------------------------------------------
// Spherical harmonics.  spherical_harmonics.py:38-106 (rsh_cart_0..3), :118-134 (kernel).
// ---------------------------------------------------------------------------------------------
template <typename T, typename S>
void rsh_cart(int degree, S x, S y, S z, S* Y) {
  Y[0] = S(T(0.282094791773878));
  if (degree < 1) return;
  Y[1] = T(-0.48860251190292) * y;
  Y[2] = T(0.48860251190292) * z;
  Y[3] = T(-0.48860251190292) * x;
  if (degree < 2) return;
  S x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
  Y[4] = T(1.09254843059208) * xy;
  Y[5] = T(-1.09254843059208) * yz;
  Y[6] = T(0.94617469575756) * z2 - T(0.31539156525252);
  Y[7] = T(-1.09254843059208) * xz;
  Y[8] = T(0.54627421529604) * x2 - T(0.54627421529604) * y2;
  if (degree < 3) return;
  Y[9] = T(-0.590043589926644) * y * (T(3.0) * x2 - y2);
  Y[10] = T(2.89061144264055) * xy * z;
  Y[11] = T(0.304697199642977) * y * (T(1.5) - T(7.5) * z2);
  Y[12] = T(1.24392110863372) * z * (T(1.5) * z2 - T(0.5)) - T(0.497568443453487) * z;
  Y[13] = T(0.304697199642977) * x * (T(1.5) - T(7.5) * z2);
  Y[14] = T(1.44530572132028) * z * (x2 - y2);
  Y[15] = T(-0.590043589926644) * x * (x2 - T(3.0) * y2);
}

template <typename T>
void sh_fwd(int64_t v, int C, int degree, const T* params, const T* points, const int64_t* indexes, const T* cam,
            T* out) {
  const int D = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < v; ++i) {
    int64_t idx = indexes[i];
    T dx = points[3 * idx] - cam[0], dy = points[3 * idx + 1] - cam[1], dz = points[3 * idx + 2] - cam[2];
    T nrm = m_sqrt(dx * dx + dy * dy + dz * dz);
    T Y[16];
    rsh_cart<T, T>(degree, dx / nrm, dy / nrm, dz / nrm, Y);
    for (int c = 0; c < C; ++c) {
      const T* row = params + (idx * C + c) * D;
      T acc = T(0);
      for (int d = 0; d < D; ++d) acc += Y[d] * row[d];
      out[i * C + c] = m_clamp(acc + T(0.5), T(0), T(1));  // :133-134
    }
  }
}

// grads to params (dense, rows of non-indexed Gaussians zero; repeated indexes accumulate),
// points (same) and camera_pos.  spherical_harmonics.py:154-161.
template <typename T>
void sh_bwd(int64_t n, int64_t v, int C, int degree, const T* params, const T* points, const int64_t* indexes,
            const T* cam, const T* gout, T* dparams, T* dpoints, T* dcam) {
  const int D = (degree + 1) * (degree + 1);
  typedef Dual<T, 3> DD;
  std::fill(dparams, dparams + n * C * D, T(0));
  std::fill(dpoints, dpoints + 3 * n, T(0));
  double cam_acc[3] = {0, 0, 0};
  for (int64_t i = 0; i < v; ++i) {  // serial: repeated indexes are allowed (tests/test_spherical_harmonics.py:27)
    int64_t idx = indexes[i];
    DD dx = DD::seed(points[3 * idx] - cam[0], 0), dy = DD::seed(points[3 * idx + 1] - cam[1], 1),
       dz = DD::seed(points[3 * idx + 2] - cam[2], 2);
    DD nrm = m_sqrt(dx * dx + dy * dy + dz * dz);
    DD Y[16];
    rsh_cart<T, DD>(degree, dx / nrm, dy / nrm, dz / nrm, Y);
    T gdir[3] = {0, 0, 0};
    for (int c = 0; c < C; ++c) {
      const T* row = params + (idx * C + c) * D;
      T acc = T(0);
      for (int d = 0; d < D; ++d) acc += Y[d].v * row[d];
      T pre = acc + T(0.5);
      T g = gout[i * C + c];
      if (!(pre >= T(0) && pre <= T(1))) g = T(0);  // clamp sub-gradient (torch.clamp: 1 on the closed interval)
      T* drow = dparams + (idx * C + c) * D;
      for (int d = 0; d < D; ++d) {
        drow[d] += g * Y[d].v;
        for (int k = 0; k < 3; ++k) gdir[k] += g * row[d] * Y[d].d[k];
      }
    }
    for (int k = 0; k < 3; ++k) {
      dpoints[3 * idx + k] += gdir[k];
      cam_acc[k] -= double(gdir[k]);
    }
  }
  for (int k = 0; k < 3; ++k) dcam[k] = T(cam_acc[k]);
}

// ---------------------------------------------------------------------------------------------
// Tile mapper (f32 only, mapper/tile_mapper.py:12).  taichi_lib/grid_query.py:10-91,
// mapper/tile_mapper.py:74-84 (count), :113-144 (keys), :34-40 / :53-59 (key layout),
// :91-110 (ranges).  Compiled with -ffp-contract=off; every operation below is a single
// correctly rounded IEEE op, and the logarithm is gs_det_logf, so the HIP kernels reproduce the
// integer results bit for bit.
// ---------------------------------------------------------------------------------------------
struct GridQuery {
  float ib00, ib01, ib10, ib11;  // inv_basis rows: axis1/scale.x ; axis2/scale.y
  float rel_min_x, rel_min_y;    // min_tile*tile_size - mean
  int min_tx, min_ty, span_x, span_y;
};

inline GridQuery grid_query(const float* g, int Wp, int Hp, int tile_size, float alpha_thr) {
  GridQuery q;
  const float mx = g[0], my = g[1], ax = g[2], ay = g[3], sgx = g[4], sgy = g[5], alpha = g[6];
  // explicit cull for alpha <= threshold (SURVEY 8a': the reference produces NaN bounds there)
  if (!(alpha > alpha_thr)) {
    q.ib00 = q.ib01 = q.ib10 = q.ib11 = q.rel_min_x = q.rel_min_y = 0.f;
    q.min_tx = q.min_ty = q.span_x = q.span_y = 0;
    return q;
  }
  const float gscale = gs_det_sqrtf(2.0f * gs_det_logf(alpha / alpha_thr));  // grid_query.py:76
  const float sx = sgx * gscale, sy = sgy * gscale;
  const float a2x = -ay, a2y = ax;  // :79
  // ellipse_bounds(mean, axis1*sx, axis2*sy)  generic.py:235-237
  const float v1x = ax * sx, v1y = ay * sx, v2x = a2x * sy, v2y = a2y * sy;
  const float ex = gs_det_sqrtf(v1x * v1x + v2x * v2x), ey = gs_det_sqrtf(v1y * v1y + v2y * v2y);
  const float lox = mx - ex, loy = my - ey, hix = mx + ex, hiy = my + ey;
  q.ib00 = ax / sx; q.ib01 = ay / sx; q.ib10 = a2x / sy; q.ib11 = a2y / sy;  // :83
  // tile_ranges  grid_query.py:10-27
  const float ts = float(tile_size);
  const int max_tx = (Wp - 1) / tile_size, max_ty = (Hp - 1) / tile_size;
  int min_tx = int(std::floor(lox / ts)), min_ty = int(std::floor(loy / ts));
  min_tx = std::max(min_tx, 0); min_ty = std::max(min_ty, 0);
  int hi_tx = int(std::ceil(hix / ts)), hi_ty = int(std::ceil(hiy / ts));
  hi_tx = std::min(std::max(hi_tx, min_tx + 1), max_tx + 1);
  hi_ty = std::min(std::max(hi_ty, min_ty + 1), max_ty + 1);
  q.min_tx = min_tx; q.min_ty = min_ty;
  q.span_x = std::max(hi_tx - min_tx, 0); q.span_y = std::max(hi_ty - min_ty, 0);
  q.rel_min_x = float(min_tx * tile_size) - mx;  // :88
  q.rel_min_y = float(min_ty * tile_size) - my;
  return q;
}

// separates_bbox, grid_query.py:30-43: the tile is rejected iff on either ellipse axis all four
// corners lie beyond +1 or beyond -1.
inline bool test_tile(const GridQuery& q, int u, int v, int tile_size) {
  const float lx = q.rel_min_x + float(u * tile_size), ly = q.rel_min_y + float(v * tile_size);
  const float ux = lx + float(tile_size), uy = ly + float(tile_size);
  const float cx[4] = {lx, ux, ux, lx}, cy[4] = {ly, ly, uy, uy};
  for (int a = 0; a < 2; ++a) {
    const float b0 = a == 0 ? q.ib00 : q.ib10, b1 = a == 0 ? q.ib01 : q.ib11;
    float mn = std::numeric_limits<float>::infinity(), mxv = -mn;
    for (int c = 0; c < 4; ++c) {
      const float t = b0 * cx[c] + b1 * cy[c];
      mn = std::min(mn, t);
      mxv = std::max(mxv, t);
    }
    if (mn > 1.0f || mxv < -1.0f) return false;
  }
  return true;
}

inline uint64_t make_key(float depth, int tile_id, bool depth16) {
  if (!depth16) return uint64_t(gs_f32_bits(depth)) | (uint64_t(uint32_t(tile_id)) << 32);  // tile_mapper.py:34-40
  float d = depth < 0.f ? 0.f : (depth > 1.f ? 1.f : depth);                                    // :53-59
  return uint64_t(uint32_t(d * 65535.0f)) | (uint64_t(uint32_t(tile_id)) << 16);
}

void tile_counts(int64_t v, const float* g, int Wp, int Hp, int tile_size, float alpha_thr, int32_t* counts) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < v; ++i) {
    GridQuery q = grid_query(g + 7 * i, Wp, Hp, tile_size, alpha_thr);
    int c = 0;
    for (int ty = 0; ty < q.span_y; ++ty)
      for (int tx = 0; tx < q.span_x; ++tx)
        if (test_tile(q, tx, ty, tile_size)) ++c;
    counts[i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// Rasterizer.  rasterizer/forward.py:25-137, rasterizer/backward.py:53-228,
// taichi_lib/generic.py:311-336 (gaussian_pdf, _with_grad), :341-404 (antialias variants).
// ---------------------------------------------------------------------------------------------
struct RasterCfg {
  int tile_size;
  int antialias;
  int use_alpha_blending;
  int compute_visibility;
  int compute_point_heuristic;
  double clamp_max_alpha;
  double alpha_threshold;
  double saturate_threshold;
};

template <typename T>
inline T pdf_plain(T px, T py, const T* g) {
  T dx = px - g[0], dy = py - g[1];
  T tx = (dx * g[2] + dy * g[3]) / g[4];
  T ty = (dx * -g[3] + dy * g[2]) / g[5];
  return m_exp(T(-0.5) * (tx * tx + ty * ty));
}

template <typename T>
inline T pdf_plain_grad(T px, T py, const T* g, T dmean[2], T daxis[2], T dsigma[2]) {
  T dx = px - g[0], dy = py - g[1];
  T ax = g[2], ay = g[3], sx = g[4], sy = g[5];
  T tx = (dx * ax + dy * ay) / sx;
  T ty = (dx * -ay + dy * ax) / sy;
  T tx2 = tx * tx, ty2 = ty * ty;
  T p = m_exp(T(-0.5) * (tx2 + ty2));
  dsigma[0] = tx2 * p / sx;
  dsigma[1] = ty2 * p / sy;
  T txs = tx / sx, tys = ty / sy;
  // dp_daxis = p * (tx_s * -d + ty_s * perp(d)),  perp(d) = (-d.y, d.x)
  daxis[0] = p * (txs * -dx + tys * -dy);
  daxis[1] = p * (txs * -dy + tys * dx);
  // dp_dmean = p * (tx_s * axis + ty_s * perp(axis))
  dmean[0] = p * (txs * ax + tys * -ay);
  dmean[1] = p * (txs * ay + tys * ax);
  return p;
}

template <typename T>
inline T s_sig(T x, T sigma) {
  T z = x / sigma;
  return T(1) / (T(1) + m_exp(T(-1.6) * z - T(0.07) * z * z * z));
}
template <typename T>
inline void s_sig_grad(T x, T sigma, T& s, T& ds_dx, T& ds_dsig) {
  T z = x / sigma;
  s = T(1) / (T(1) + m_exp(T(-1.6) * z - T(0.07) * z * z * z));
  T d = (T(1.6) + T(0.21) * z * z) * s * (T(1) - s);
  ds_dx = d / sigma;
  ds_dsig = ds_dx * -z;
}

template <typename T>
inline T pdf_aa(T px, T py, const T* g) {
  const T tau = T(2.0 * 3.14159265358979323846);
  T dx = px - g[0], dy = py - g[1];
  T sx = g[4], sy = g[5];
  T tx = dx * g[2] + dy * g[3];
  T ty = dx * -g[3] + dy * g[2];
  T Sx1 = s_sig(tx + T(0.5), sx), Sx2 = s_sig(tx - T(0.5), sx);
  T Sy1 = s_sig(ty + T(0.5), sy), Sy2 = s_sig(ty - T(0.5), sy);
  return tau * sx * (Sx1 - Sx2) * sy * (Sy1 - Sy2);
}

template <typename T>
inline T pdf_aa_grad(T px, T py, const T* g, T dmean[2], T daxis[2], T dsigma[2]) {
  const T tau = T(2.0 * 3.14159265358979323846);
  T dx = px - g[0], dy = py - g[1];
  T ax = g[2], ay = g[3], sx = g[4], sy = g[5];
  T tx = dx * ax + dy * ay;
  T ty = dx * -ay + dy * ax;
  T Sx1, dSx1, dSx1s, Sx2, dSx2, dSx2s, Sy1, dSy1, dSy1s, Sy2, dSy2, dSy2s;
  s_sig_grad(tx + T(0.5), sx, Sx1, dSx1, dSx1s);
  s_sig_grad(tx - T(0.5), sx, Sx2, dSx2, dSx2s);
  s_sig_grad(ty + T(0.5), sy, Sy1, dSy1, dSy1s);
  s_sig_grad(ty - T(0.5), sy, Sy2, dSy2, dSy2s);
  T ix = sx * (Sx1 - Sx2), iy = sy * (Sy1 - Sy2);
  T i2d = tau * ix * iy;
  T dSx = iy * sx * (dSx1 - dSx2);
  T dSy = ix * sy * (dSy1 - dSy2);
  // di_dmean = tau * (dSx * -axis + dSy * -perp(axis))
  dmean[0] = tau * (dSx * -ax + dSy * ay);
  dmean[1] = tau * (dSx * -ay + dSy * -ax);
  dsigma[0] = tau * iy * (Sx1 - Sx2 + (dSx1s - dSx2s) * sx);
  dsigma[1] = tau * ix * (Sy1 - Sy2 + (dSy1s - dSy2s) * sy);
  // di_daxis = tau * (dSx * d + dSy * -perp(d)),  perp(d) = (-d.y, d.x)
  daxis[0] = tau * (dSx * dx + dSy * dy);
  daxis[1] = tau * (dSx * dy + dSy * -dx);
  return i2d;
}

template <typename T>
void raster_fwd(int64_t V, int F, const T* points, const T* features, const int32_t* ranges, const int32_t* o2p, int W,
                int H, const RasterCfg& cfg, T* image, T* alpha_img, T* visibility) {
  const int ts = cfg.tile_size;
  const int tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
  const T cmax = T(cfg.clamp_max_alpha), thr = T(cfg.alpha_threshold);
  const T sat_thr = T(1.0 - cfg.saturate_threshold);
  if (cfg.compute_visibility) std::fill(visibility, visibility + V, T(0));
  std::vector<std::vector<T>> vis_local;
#pragma omp parallel
  {
    std::vector<T> acc(F);
    std::vector<T> tile_vis;
#pragma omp for schedule(dynamic, 4)
    for (int tile = 0; tile < tw * th; ++tile) {
      const int tx0 = (tile % tw) * ts, ty0 = (tile / tw) * ts;
      const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
      if (cfg.compute_visibility) tile_vis.assign(std::max(end - start, 0), T(0));
      for (int py = ty0; py < std::min(ty0 + ts, H); ++py)
        for (int px = tx0; px < std::min(tx0 + ts, W); ++px) {
          const T pxf = T(px) + T(0.5), pyf = T(py) + T(0.5);
          std::fill(acc.begin(), acc.end(), T(0));
          T total = T(0);
          bool saturated = false;
          for (int k = start; k < end; ++k) {
            const int idx = o2p[k];
            const T* g = points + 7 * int64_t(idx);
            T ga = cfg.antialias ? pdf_aa(pxf, pyf, g) : pdf_plain(pxf, pyf, g);
            T a = g[6] * ga;
            a = std::min(a, cmax);  // forward.py:99
            if (a > thr) {
              T w = a * (T(1) - total);
              total += w;
              const T* f = features + int64_t(idx) * F;
              if (cfg.use_alpha_blending) {
                for (int c = 0; c < F; ++c) acc[c] += f[c] * w;
              } else {  // forward.py:109-114 (quantile mode)
                if (total >= sat_thr && !saturated)
                  for (int c = 0; c < F; ++c) acc[c] = f[c];
                saturated = total >= sat_thr;
              }
              if (cfg.compute_visibility) tile_vis[k - start] += w;
              if (saturated) break;  // layout-independent reading of forward.py:92-94
            }
          }
          T* out = image + (int64_t(py) * W + px) * F;
          for (int c = 0; c < F; ++c) out[c] = acc[c];
          alpha_img[int64_t(py) * W + px] = cfg.use_alpha_blending ? total : T(total > T(0));
        }
      if (cfg.compute_visibility)
        for (int k = start; k < end; ++k) {
          T val = tile_vis[k - start];
          if (val != T(0)) {
#pragma omp atomic
            visibility[o2p[k]] += val;
          }
        }
    }
  }
}

// Per-overlap gradient rows are written to a scratch table and summed in overlap order
// afterwards, so the result is deterministic for a given thread count or any other.
template <typename T>
void raster_bwd(int64_t V, int F, const T* points, const T* features, const int32_t* ranges, const int32_t* o2p,
                int64_t K, int W, int H, const RasterCfg& cfg, const T* image, const T* grad_image, T* grad_points,
                T* grad_features, T* heuristic) {
  const int ts = cfg.tile_size;
  const int tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
  const T cmax = T(cfg.clamp_max_alpha), thr = T(cfg.alpha_threshold), sat = T(cfg.saturate_threshold);
  const int R = 7 + F + 2;
  std::vector<T> rows(size_t(std::max<int64_t>(K, 1)) * R, T(0));
#pragma omp parallel
  {
    std::vector<T> rem(F), gpix(F);
#pragma omp for schedule(dynamic, 4)
    for (int tile = 0; tile < tw * th; ++tile) {
      const int tx0 = (tile % tw) * ts, ty0 = (tile / tw) * ts;
      const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
      for (int py = ty0; py < std::min(ty0 + ts, H); ++py)
        for (int px = tx0; px < std::min(tx0 + ts, W); ++px) {
          const T pxf = T(px) + T(0.5), pyf = T(py) + T(0.5);
          const T* im = image + (int64_t(py) * W + px) * F;
          const T* gi = grad_image + (int64_t(py) * W + px) * F;
          for (int c = 0; c < F; ++c) { rem[c] = im[c]; gpix[c] = gi[c]; }
          T total = T(0);
          for (int k = start; k < end; ++k) {
            if (total >= sat) break;  // backward.py:160
            const int idx = o2p[k];
            const T* g = points + 7 * int64_t(idx);
            T dmean[2], daxis[2], dsigma[2];
            T ga = cfg.antialias ? pdf_aa_grad(pxf, pyf, g, dmean, daxis, dsigma)
                                 : pdf_plain_grad(pxf, pyf, g, dmean, daxis, dsigma);
            T a = g[6] * ga;
            if (!(a > thr)) continue;  // backward.py:166 (unclamped alpha)
            a = std::min(a, cmax);     // :169
            const T* f = features + int64_t(idx) * F;
            T Ti = T(1) - total;
            T w = a * Ti;
            total += w;
            T alpha_grad = T(0);
            for (int c = 0; c < F; ++c) {
              rem[c] -= f[c] * w;
              T diff = f[c] * Ti - rem[c] / (T(1) - a);  // :180
              alpha_grad += diff * gpix[c];
            }
            T aag = g[6] * alpha_grad;  // :184
            T* row = rows.data() + size_t(k) * R;
            row[0] += aag * dmean[0]; row[1] += aag * dmean[1];
            row[2] += aag * daxis[0]; row[3] += aag * daxis[1];
            row[4] += aag * dsigma[0]; row[5] += aag * dsigma[1];
            row[6] += ga * alpha_grad;
            for (int c = 0; c < F; ++c) row[7 + c] += w * gpix[c];
            row[7 + F] += aag * aag;  // :194-198
            row[8 + F] += std::abs(aag * dmean[0]) + std::abs(aag * dmean[1]);
          }
        }
    }
  }
  std::fill(grad_points, grad_points + 7 * V, T(0));
  std::fill(grad_features, grad_features + int64_t(F) * V, T(0));
  if (cfg.compute_point_heuristic) std::fill(heuristic, heuristic + 2 * V, T(0));
  for (int64_t k = 0; k < K; ++k) {
    const int64_t idx = o2p[k];
    const T* row = rows.data() + size_t(k) * R;
    for (int c = 0; c < 7; ++c) grad_points[7 * idx + c] += row[c];
    for (int c = 0; c < F; ++c) grad_features[F * idx + c] += row[7 + c];
    if (cfg.compute_point_heuristic) {
      heuristic[2 * idx] += row[7 + F];
      heuristic[2 * idx + 1] += row[8 + F];
    }
  }
}

}  // namespace

// =============================================================================================
// C entry points (host pointers; loaded with ctypes by oracle/oracle.py)
// =============================================================================================
extern "C" {

#define ORC_PROJECT(SUF, T)                                                                                          \
  void orc_project_fwd_##SUF(int64_t n, const T* pos, const T* ls, const T* q, const T* al, const T* Tcw,            \
                             const T* proj, int W, int H, double near_p, double far_p, double blur, double margin,   \
                             double thr, T* points, T* depth, uint8_t* visible) {                                    \
    project_fwd<T>(n, pos, ls, q, al, Tcw, proj, W, H, near_p, far_p, blur, margin, thr, points, depth, visible);    \
  }                                                                                                                  \
  void orc_project_bwd_##SUF(int64_t n, int64_t v, const T* pos, const T* ls, const T* q, const T* al, const T* Tcw, \
                             const T* proj, int W, int H, double blur, double margin, const int64_t* indexes,        \
                             const T* gpoints, const T* gdepth, T* dpos, T* dls, T* dq, T* dal, T* dT, T* dproj) {   \
    project_bwd<T>(n, v, pos, ls, q, al, Tcw, proj, W, H, blur, margin, indexes, gpoints, gdepth, dpos, dls, dq,     \
                   dal, dT, dproj);                                                                                  \
  }                                                                                                                  \
  void orc_sh_fwd_##SUF(int64_t v, int C, int degree, const T* params, const T* points, const int64_t* indexes,      \
                        const T* cam, T* out) {                                                                      \
    sh_fwd<T>(v, C, degree, params, points, indexes, cam, out);                                                      \
  }                                                                                                                  \
  void orc_sh_bwd_##SUF(int64_t n, int64_t v, int C, int degree, const T* params, const T* points,                   \
                        const int64_t* indexes, const T* cam, const T* gout, T* dparams, T* dpoints, T* dcam) {      \
    sh_bwd<T>(n, v, C, degree, params, points, indexes, cam, gout, dparams, dpoints, dcam);                          \
  }                                                                                                                  \
  void orc_raster_fwd_##SUF(int64_t V, int F, const T* points, const T* features, const int32_t* ranges,             \
                            const int32_t* o2p, int W, int H, const RasterCfg* cfg, T* image, T* alpha,              \
                            T* visibility) {                                                                         \
    raster_fwd<T>(V, F, points, features, ranges, o2p, W, H, *cfg, image, alpha, visibility);                        \
  }                                                                                                                  \
  void orc_raster_bwd_##SUF(int64_t V, int F, const T* points, const T* features, const int32_t* ranges,             \
                            const int32_t* o2p, int64_t K, int W, int H, const RasterCfg* cfg, const T* image,       \
                            const T* grad_image, T* grad_points, T* grad_features, T* heuristic) {                   \
    raster_bwd<T>(V, F, points, features, ranges, o2p, K, W, H, *cfg, image, grad_image, grad_points,                \
                  grad_features, heuristic);                                                                         \
  }

ORC_PROJECT(f32, float)
ORC_PROJECT(f64, double)

// Diagnostic for the parity tests (not a reference function): per pixel, how close the forward walk of
// rasterizer/forward.py:84-128 comes to flipping one of its `alpha > alpha_threshold` decisions --
// margin[pixel] = min over the tile's splats of |alpha_p * pdf - thr| / thr, evaluated in f32 exactly as raster_fwd
// does.  A pixel on which two correct f32 implementations disagree by more than rounding must have a tiny margin:
// the only discontinuity of the forward is that comparison, and one flip moves the pixel by <= thr * |feature|.
// feat_max (optional, (H, W, F)): per pixel and channel, the largest |feature| among the splats it blends -- the scale
// of the absolute rounding error of a blend (the accumulated weight carries ~1e-6 of absolute error whatever the
// features are, so a z^2 channel reaching 1e4 cannot be held to the absolute tolerance of a colour in [0, 1]).
void orc_raster_flip_margin_f32(const float* points, const float* features, int F, const int32_t* ranges,
                                const int32_t* o2p, int W, int H, const RasterCfg* cfg, float* margin,
                                float* feat_max) {
  const int ts = cfg->tile_size;
  const int tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
  const float thr = float(cfg->alpha_threshold);
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < tw * th; ++tile) {
    const int tx0 = (tile % tw) * ts, ty0 = (tile / tw) * ts;
    const int start = ranges[2 * tile], end = ranges[2 * tile + 1];
    for (int py = ty0; py < std::min(ty0 + ts, H); ++py)
      for (int px = tx0; px < std::min(tx0 + ts, W); ++px) {
        const float pxf = float(px) + 0.5f, pyf = float(py) + 0.5f;
        float best = 1e30f;
        float* fm = feat_max ? feat_max + (int64_t(py) * W + px) * F : nullptr;
        for (int c = 0; fm && c < F; ++c) fm[c] = 0.0f;
        for (int k = start; k < end; ++k) {
          const float* g = points + int64_t(o2p[k]) * 7;
          const float a = g[6] * (cfg->antialias ? pdf_aa<float>(pxf, pyf, g) : pdf_plain<float>(pxf, pyf, g));
          best = std::min(best, std::fabs(a - thr) / thr);
          // largest feature magnitude among the splats that can reach this pixel's blend (margin included)
          if (fm && a > thr * 0.999f)
            for (int c = 0; c < F; ++c) fm[c] = std::max(fm[c], std::fabs(features[int64_t(o2p[k]) * F + c]));
        }
        margin[int64_t(py) * W + px] = best;
      }
  }
}

// ndc depth with the fixed f32 operation order of SURVEY 8a-3 (torch_lib/projection.py:120-123).
void orc_ndc_depth_f32(int64_t n, const float* depth, double near_p, double far_p, float* out) {
  const float inv_far = float(1.0 / far_p);
  const float denom = float(1.0 / near_p - 1.0 / far_p);
  for (int64_t i = 0; i < n; ++i) {
    float inv_d = 1.0f / depth[i];
    float a = inv_d - inv_far;
    out[i] = 1.0f - a / denom;
  }
}

// mapper: image_size is already padded to a tile multiple (tile_mapper.py:172).
void orc_tile_counts(int64_t v, const float* g, int Wp, int Hp, int tile_size, double alpha_thr, int32_t* counts) {
  tile_counts(v, g, Wp, Hp, tile_size, float(alpha_thr), counts);
}

// exclusive scan with total appended (cuda_lib/full_cumsum.cu:17-47); returns the total
int64_t orc_full_cumsum_i32(int64_t n, const int32_t* in, int32_t* out) {
  int64_t acc = 0;
  for (int64_t i = 0; i < n; ++i) { out[i] = int32_t(acc); acc += in[i]; }
  out[n] = int32_t(acc);
  return acc;
}

// tile_mapper.py:113-144: keys/values in generation order (Gaussian-major)
void orc_tile_emit(int64_t v, const float* g, const float* depth, const int32_t* offsets, int Wp, int Hp, int tile_size,
                   double alpha_thr, int depth16, uint64_t* keys, int32_t* values) {
  const int tiles_wide = Wp / tile_size;
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < v; ++i) {
    GridQuery q = grid_query(g + 7 * i, Wp, Hp, tile_size, float(alpha_thr));
    int64_t k = offsets[i];
    // ti.ndrange(span.x, span.y): x outer, y inner
    for (int tx = 0; tx < q.span_x; ++tx)
      for (int ty = 0; ty < q.span_y; ++ty)
        if (test_tile(q, tx, ty, tile_size)) {
          int tile_id = (tx + q.min_tx) + (ty + q.min_ty) * tiles_wide;
          keys[k] = make_key(depth[i], tile_id, depth16 != 0);
          values[k] = int32_t(i);
          ++k;
        }
  }
}

// stable ascending sort on key bits [begin_bit, end_bit)  (cuda_lib/radix_sort_pairs.cu:8-29)
void orc_sort_pairs_u64(int64_t k, const uint64_t* keys, const int32_t* values, int begin_bit, int end_bit,
                        uint64_t* keys_out, int32_t* values_out) {
  if (end_bit <= 0) end_bit = 64;
  const uint64_t mask = (end_bit - begin_bit >= 64) ? ~uint64_t(0) : ((uint64_t(1) << (end_bit - begin_bit)) - 1);
  std::vector<int64_t> order(k);
  std::iota(order.begin(), order.end(), int64_t(0));
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
    return ((keys[a] >> begin_bit) & mask) < ((keys[b] >> begin_bit) & mask);
  });
  for (int64_t i = 0; i < k; ++i) { keys_out[i] = keys[order[i]]; values_out[i] = values[order[i]]; }
}

// tile_mapper.py:91-110; ranges pre-zeroed (:186)
void orc_tile_ranges(int64_t k, const uint64_t* sorted_keys, int depth16, int64_t num_tiles, int32_t* ranges) {
  std::fill(ranges, ranges + 2 * num_tiles, 0);
  const int shift = depth16 ? 16 : 32;
  for (int64_t i = 0; i < k; ++i) {
    int64_t t = int64_t(sorted_keys[i] >> shift);
    if (i == 0 || int64_t(sorted_keys[i - 1] >> shift) != t) ranges[2 * t] = int32_t(i);
    if (i + 1 == k || int64_t(sorted_keys[i + 1] >> shift) != t) ranges[2 * t + 1] = int32_t(i + 1);
  }
}

int orc_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

float orc_det_logf(float x) { return gs_det_logf(x); }

}  // extern "C"
