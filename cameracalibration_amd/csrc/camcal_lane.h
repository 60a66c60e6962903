// Per-lane bodies of the calibration kernels: one lane == one view.
// Shared by camcal_kernels.hip (device) and the tests' CPU lane emulator.
// See camcal_math.h for conventions and the reference citations.
#pragma once
#include <cstdlib>
#include "camcal_math.h"

namespace camcal {

// One lane's window onto a packed tile.
struct LaneTile {
  const float* base;   // pts + tile_off + lane
  int n;               // corners of this lane's view (0 for padding lanes)
  int nmax;            // max corners over the tile's 64 lanes (loop bound)
};

struct Obs { float X, Y, Z, u, v; };

// PLANAR: every board point has Z == 0; the Z rows are never fetched.
template <bool PLANAR>
CC_HD Obs load_obs(const float* base, int k) {
  const float* p = base + (long long)k * (PT_COMP * LANES);
  Obs o;
  o.X = p[0 * LANES]; o.Y = p[1 * LANES]; o.Z = PLANAR ? 0.f : p[2 * LANES]; o.u = p[3 * LANES]; o.v = p[4 * LANES];
  return o;
}

// Walk the corners of the lane's view: f(obs) for k < n.  The trip count is the tile's
// nmax (uniform over the wavefront); two corners per trip, each loaded one step ahead of
// its use, so the loads overlap the arithmetic and no register rotation is needed.
template <bool PLANAR, class F>
CC_HD void for_each_obs(const LaneTile& tv, F&& f) {
  const int last = tv.nmax - 1;
  Obs a = load_obs<PLANAR>(tv.base, 0);
  int k = 0;
  for (; k + 1 < tv.nmax; k += 2) {
    const Obs b = load_obs<PLANAR>(tv.base, k + 1);
    if (k < tv.n) f(a);
    a = load_obs<PLANAR>(tv.base, k + 2 < last ? k + 2 : last);
    if (k + 1 < tv.n) f(b);
  }
  if (k < tv.nmax && k < tv.n) f(a);
}

// The same walk with every corner loaded THREE steps ahead of its use (four corners per trip,
// three in flight).  A corner of the pose refinement is ~800 issue cycles at two wavefronts per
// SIMD, less than a trip to HBM under load, so a single corner of look-ahead leaves the loop
// waiting for its points; the extra look-ahead costs 8 more float registers.
template <bool PLANAR, class F>
CC_HD void for_each_obs_deep(const LaneTile& tv, F&& f) {
  const int last = tv.nmax > 0 ? tv.nmax - 1 : 0;
  auto at = [&](int k) { return load_obs<PLANAR>(tv.base, k < last ? k : last); };
  Obs q0 = at(0), q1 = at(1), q2 = at(2);
  for (int k = 0; k < tv.nmax; k += 4) {
    const Obs q3 = at(k + 3);
    if (k < tv.n) f(q0);
    q0 = at(k + 4);
    if (k + 1 < tv.n) f(q1);
    q1 = at(k + 5);
    if (k + 2 < tv.n) f(q2);
    q2 = at(k + 6);
    if (k + 3 < tv.n) f(q3);
  }
}

// ---------------------------------------------------------------------------
// K1: per-view blocks V, W, g of the normal equations (Rodrigues frame).
template <int MODEL, bool PLANAR>
CC_HD void lane_view_blocks(const LaneTile& tv, const double* intr, const double pose[6],
                            ViewAcc<MODEL>& acc) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  acc.zero();
  for_each_obs<PLANAR>(tv, [&](const Obs& o) {
    double Yr[3], Yc[3];
    transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
    PointEval<MODEL> p;
    eval_point<true>(Yc, intr, p);
    accumulate_view(acc, p, Yr, p.pu - (double)o.u, p.pv - (double)o.v);
  });
  double Jl[9];
  left_jacobian(pose, Jl);
  to_rodrigues_frame<MT<MODEL>::NI>(Jl, acc.V, acc.W, acc.g);
}

// ---------------------------------------------------------------------------
// K1+K2 fused: one pass over the view's corners feeding both accumulator sets
// (projection and Jacobian evaluated once per observation).
// LOCAL: leave the blocks in the local tangent frame of the pose (rig solver).
template <int MODEL, bool PLANAR>
CC_HD void fused_obs(const double R[9], const double t[3], const double* intr, const Obs& o,
                     ViewAcc<MODEL>& acc, IntrAcc<MODEL>& gacc) {
  double Yr[3], Yc[3];
  transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
  PointEval<MODEL> p;
  eval_point<true>(Yc, intr, p);
  const double eu = p.pu - (double)o.u, ev = p.pv - (double)o.v;
  accumulate_intr(gacc, p, eu, ev);
  accumulate_view(acc, p, Yr, eu, ev);
}

template <int MODEL, bool PLANAR, bool LOCAL = false>
CC_HD void lane_normal_fused(const LaneTile& tv, const double* intr, const double pose[6],
                             ViewAcc<MODEL>& acc, IntrAcc<MODEL>& gacc) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  acc.zero();
  for_each_obs<PLANAR>(tv, [&](const Obs& o) { fused_obs<MODEL, PLANAR>(R, t, intr, o, acc, gacc); });
  gacc.add_count(tv.n);
  if (!LOCAL) {
    double Jl[9];
    left_jacobian(pose, Jl);
    to_rodrigues_frame<MT<MODEL>::NI>(Jl, acc.V, acc.W, acc.g);
  }
}

// Schur contribution of one view from blocks held in registers (consumes V, W, g: on
// return V holds the factor U of V' = U^T U, id its reciprocal diagonal, and W, g the
// forward-substituted U^-T W_a^T, U^-T g).
template <int NI>
CC_HD void lane_schur_regs(double V[21], double* W, double g[6], double lam, double* acc, double id[6]) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i) V[tri(6, i, i)] *= 1.0 + lam;
  chol6(V, id);
  chol6_fwd(V, id, g);
  CC_UNROLL
  for (int a = 0; a < NI; ++a) chol6_fwd(V, id, W + a * 6);
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    CC_UNROLL
    for (int b = a; b < NI; ++b) {
      double s = 0.0;
      CC_UNROLL
      for (int c = 0; c < 6; ++c) s = fma(W[a * 6 + c], W[b * 6 + c], s);
      acc[tri(NI, a, b)] += s;
    }
    double s = 0.0;
    CC_UNROLL
    for (int c = 0; c < 6; ++c) s = fma(W[a * 6 + c], g[c], s);
    acc[trisz(NI) + a] += s;
  }
}
template <int NI>
CC_HD void lane_schur_regs(double V[21], double* W, double g[6], double lam, double* acc) {
  double id[6];
  lane_schur_regs<NI>(V, W, g, lam, acc, id);
}

// First-order response of a view's pose to an intrinsic step.  For intrinsics moved by -s
// (K <- K - s) the pose that minimises the view's linearised residual is
//   pose - d,   V d = g - W^T s   (the back-substitution row of the joint normal equations);
// at a pose that is its view's optimum g = 0, so with  Y = V^-1 W^T  (6 x NI, Rodrigues frame)
//   pose_new = pose + Y s.
// cv::fisheye::calibrate re-estimates every pose after its intrinsic step
// (CALIB_RECOMPUTE_EXTRINSIC, camera_calib.py:76); moving the warm start of that refinement
// along Y s leaves ComputeExtrinsicRefine one Gauss-Newton step less to take.  (The V^-1 g
// term is left out on purpose: it vanishes for a converged view and would be an extra,
// unrequested Gauss-Newton step for one that ran into the step cap.)
// In: the factor and the forward-substituted columns lane_schur_regs leaves (lam = 0).
template <int NI>
CC_HD void lane_pose_response(const double U[21], const double id[6], double* W) {
  CC_UNROLL
  for (int a = 0; a < NI; ++a) chol6_bwd(U, id, W + a * 6);
}
// Yb: [NI * 6][S] component-major like W: Yb[(a * 6 + c) * S + slot] = (V^-1 W_a^T)_c.
// s[a] = the intrinsic step (K_new = K - s).
template <int NI>
CC_HD void lane_pose_predict(const double* Yb, long long S, long long slot, const double* s, double pose[6]) {
  double d[6] = {0, 0, 0, 0, 0, 0};
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    const double sa = s[a];
    CC_UNROLL
    for (int c = 0; c < 6; ++c) d[c] = fma(Yb[(a * 6 + c) * S + slot], sa, d[c]);
  }
  CC_UNROLL
  for (int c = 0; c < 6; ++c) pose[c] += d[c];
}

// ---------------------------------------------------------------------------
// Evaluation A: cost, intrinsic normal block U, b (global sums) and -- MODE 2
// -- the view's pose block V and gradient g (Rodrigues frame, through V/g).
// MODE 0: cost only; 1: U, b, cost; 2: U, b, cost, V, g.  Returns the lane's
// own view cost (sum of squared residuals).
template <int MODEL, int MODE, bool PLANAR>
CC_HD void intr_normal_obs(const double R[9], const double t[3], const double* intr, const Obs& o,
                           IntrAcc<MODEL>& acc, double* V, double* g, double& vc) {
  double Yr[3], Yc[3];
  transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
  PointEval<MODEL> p;
  eval_point<(MODE > 0)>(Yc, intr, p);
  const double eu = p.pu - (double)o.u, ev = p.pv - (double)o.v;
  if (MODE > 0) accumulate_intr(acc, p, eu, ev);
  if (MODE == 2) {
    double ju[6], jv[6];
    pose_row(Yr, p.gu, ju);
    pose_row(Yr, p.gv, jv);
    accumulate_vg(V, g, ju, jv, eu, ev);
  }
  vc = fma(eu, eu, vc);
  vc = fma(ev, ev, vc);
}

template <int MODEL, int MODE, bool PLANAR>
CC_HD double lane_intr_normal(const LaneTile& tv, const double* intr, const double pose[6],
                              IntrAcc<MODEL>& acc, double* V, double* g) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  double vc = 0.0;
  if (MODE == 2) {
    CC_UNROLL
    for (int i = 0; i < 21; ++i) V[i] = 0.0;
    CC_UNROLL
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
  }
  for_each_obs<PLANAR>(tv, [&](const Obs& o) { intr_normal_obs<MODEL, MODE, PLANAR>(R, t, intr, o, acc, V, g, vc); });
  if (MODE > 0) acc.add_count(tv.n);
  else acc.v[IntrAcc<MODEL>::COST] += vc;
  if (MODE == 2) {
    double Jl[9];
    left_jacobian(pose, Jl);
    vg_to_rodrigues_frame(Jl, V, g);
  }
  return vc;
}

// Evaluation B: the view's cross block W = Ji^T Je (Rodrigues frame).
template <int MODEL, bool PLANAR>
CC_HD void cross_block_obs(const double R[9], const double t[3], const double* intr, const Obs& o, double* W) {
  double Yr[3], Yc[3];
  transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
  PointEval<MODEL> p;
  eval_point<true>(Yc, intr, p);
  double ju[6], jv[6];
  pose_row(Yr, p.gu, ju);
  pose_row(Yr, p.gv, jv);
  accumulate_w(W, p, ju, jv);
}

template <int MODEL, bool PLANAR>
CC_HD void lane_cross_block(const LaneTile& tv, const double* intr, const double pose[6], double* W) {
  constexpr int NI = MT<MODEL>::NI;
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  CC_UNROLL
  for (int i = 0; i < NI * 6; ++i) W[i] = 0.0;
  for_each_obs<PLANAR>(tv, [&](const Obs& o) { cross_block_obs<MODEL, PLANAR>(R, t, intr, o, W); });
  double Jl[9];
  left_jacobian(pose, Jl);
  w_to_rodrigues_frame<NI>(Jl, W);
}

// ---------------------------------------------------------------------------
// K4: per-view pose refinement with fixed intrinsics (Gauss-Newton), the
// restatement of cv::internal::ComputeExtrinsicRefine: x += -(J^T J)^-1 J^T e,
// stop when ||d|| / ||x + d|| <= tol, when cond(J) > 1e6 (fisheye: the view keeps
// the pose it has, as in cv2) or after max_iter steps.  Returns the number of steps taken.

// Extreme eigenvalues of the symmetric 6x6 matrix U^T U (U upper, packed): cyclic Jacobi,
// 6 sweeps, to ~1e-16 of the largest.  Written with run-time loops on purpose: the matrix
// then lives in scratch memory and is touched only by the rare view that gets here; fully
// unrolled it sat in 42 registers and made the refinement kernel spill on its hot path.
CC_HD void gram6_eig_range(const double U[21], double& lo, double& hi) {
  double A[6][6];
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) {
      double sm = 0.0;
      CC_UNROLL
      for (int k = 0; k <= i; ++k) sm = fma(U[tri(6, k, i)], U[tri(6, k, j)], sm);
      A[i][j] = sm;
      A[j][i] = sm;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int sweep = 0; sweep < 6; ++sweep) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int p = 0; p < 5; ++p) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int q = p + 1; q < 6; ++q) {
        const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
        // a rotation by zero where the entry is already negligible
        const bool live = fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * (fabs(app) + fabs(aqq));
        const double th = live ? (aqq - app) / (2.0 * apq) : 0.0;
        double t = 1.0 / (fabs(th) + sqrt(th * th + 1.0));
        t = th < 0.0 ? -t : t;
        t = live ? t : 0.0;
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int k = 0; k < 6; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int k = 0; k < 6; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
      }
    }
  }
  lo = A[0][0];
  hi = lo;
  for (int i = 1; i < 6; ++i) {
    lo = fmin(lo, A[i][i]);
    hi = fmax(hi, A[i][i]);
  }
}

// cv2's guard of ComputeExtrinsicRefine: cond(J) = sqrt(lmax / lmin of J^T J) > 1e6.
// U: Cholesky factor of J^T J (U^T U), inv_diag = 1 / U_ii, tr / dmax: trace and largest
// diagonal entry of J^T J.  Two bounds decide almost every view from the inverse factor,
//   dmax * max diag((J^T J)^-1)  <=  lmax / lmin  <=  tr * tr((J^T J)^-1),
// and only a view whose ratio lies within a factor 36 of the threshold pays for eigenvalues.
CC_HD bool pose_cond_tripped(const double U[21], const double inv_diag[6], double tr, double dmax) {
  const double thresh2 = 1e12;
  // U^-1 column by column (a column needs only itself), squared into the row sums
  // rs[i] = sum_j (U^-1)_ij^2 = ((J^T J)^-1)_ii
  double rs[6] = {0, 0, 0, 0, 0, 0};
  CC_UNROLL
  for (int j = 0; j < 6; ++j) {
    double col[6];
    col[j] = inv_diag[j];
    CC_UNROLL
    for (int i = j - 1; i >= 0; --i) {
      double sm = 0.0;
      CC_UNROLL
      for (int k = i + 1; k <= j; ++k) sm = fma(U[tri(6, i, k)], col[k], sm);
      col[i] = -sm * inv_diag[i];
    }
    CC_UNROLL
    for (int i = 0; i <= j; ++i) rs[i] = fma(col[i], col[i], rs[i]);
  }
  double tinv = 0.0, dinv = 0.0;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    tinv += rs[i];
    dinv = fmax(dinv, rs[i]);
  }
  if (dmax * dinv > thresh2) return true;
  if (tr * tinv <= thresh2) return false;
  double lo, hi;
  gram6_eig_range(U, lo, hi);
  return !(lo > 0.0) || hi > thresh2 * lo;
}

// finish of one Gauss-Newton pose step from the local-frame sums V, g: map to
// Rodrigues coordinates, solve, update; returns false (pose untouched) on a non-positive
// pivot or, with COND, when cv2's condition-number guard trips
template <bool COND>
CC_HD bool pose_gn_update(double V[21], double g[6], double pose[6], double& change) {
  double Jl[9];
  left_jacobian(pose, Jl);
  vg_to_rodrigues_frame(Jl, V, g);
  double tr = 0.0, dmax = 0.0;
  if (COND) {
    CC_UNROLL
    for (int i = 0; i < 6; ++i) {
      tr += V[tri(6, i, i)];
      dmax = fmax(dmax, V[tri(6, i, i)]);
    }
  }
  double id[6];
  const bool ok = chol6(V, id);
  if (!ok) return false;
  if (COND && pose_cond_tripped(V, id, tr, dmax)) return false;
  chol6_fwd(V, id, g);
  chol6_bwd(V, id, g);
  double d2 = 0.0, x2 = 0.0;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    const double xn = pose[i] - g[i];
    d2 += g[i] * g[i];
    x2 += xn * xn;
    pose[i] = xn;
  }
  change = sqrt(d2) / sqrt(x2);
  return true;
}

// one Gauss-Newton step of one view walked by one lane
template <int MODEL, bool PLANAR>
CC_HD bool lane_pose_gn_step(const LaneTile& tv, const double* intr, double pose[6], double& change) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  double V[21], g[6];
  CC_UNROLL
  for (int i = 0; i < 21; ++i) V[i] = 0.0;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
#ifndef CAMCAL_REFINE_WALK
#define CAMCAL_REFINE_WALK for_each_obs
#endif
  CAMCAL_REFINE_WALK<PLANAR>(tv, [&](const Obs& o) {
    double Yr[3], Yc[3];
    transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
    PointEval<MODEL> p;
    eval_point<true>(Yc, intr, p);
    const double eu = p.pu - (double)o.u, ev = p.pv - (double)o.v;
    double ju[6], jv[6];
    pose_row(Yr, p.gu, ju);
    pose_row(Yr, p.gv, jv);
    accumulate_vg(V, g, ju, jv, eu, ev);
  });
  return pose_gn_update<MODEL == FISHEYE>(V, g, pose, change);
}

template <int MODEL, bool PLANAR>
CC_HD int lane_pose_refine(const LaneTile& tv, const double* intr, double pose[6], int max_iter,
                           double tol) {
  int it = 0;
  double change = 1.0;
  canonical_rvec(pose);
  while (change > tol && it < max_iter) {
    if (!lane_pose_gn_step<MODEL, PLANAR>(tv, intr, pose, change)) break;
    ++it;
  }
  return it;
}

// ---------------------------------------------------------------------------
// K3: Schur complement contribution of one view.
//   V' = V with diag * (1 + lam)   (lam = 0: undamped Gauss-Newton)
//   C += W V'^-1 W^T (upper, NI(NI+1)/2),  d += W V'^-1 g
// blocks are read component-major with stride S.  acc = [C | d].
template <int NI>
CC_HD void lane_schur(const double* Vb, const double* Wb, const double* gb, long long S,
                      long long slot, double lam, double* acc) {
  double V[21], z[6];
  CC_UNROLL
  for (int e = 0; e < 21; ++e) V[e] = Vb[e * S + slot];
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    V[tri(6, i, i)] *= 1.0 + lam;
    z[i] = gb[i * S + slot];
  }
  double id[6];
  chol6(V, id);
  chol6_fwd(V, id, z);
  double Y[NI * 6];
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    double y[6];
    CC_UNROLL
    for (int c = 0; c < 6; ++c) y[c] = Wb[(a * 6 + c) * S + slot];
    chol6_fwd(V, id, y);
    CC_UNROLL
    for (int c = 0; c < 6; ++c) Y[a * 6 + c] = y[c];
  }
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    CC_UNROLL
    for (int b = a; b < NI; ++b) {
      double s = 0.0;
      CC_UNROLL
      for (int c = 0; c < 6; ++c) s += Y[a * 6 + c] * Y[b * 6 + c];
      acc[tri(NI, a, b)] += s;
    }
    double s = 0.0;
    CC_UNROLL
    for (int c = 0; c < 6; ++c) s += Y[a * 6 + c] * z[c];
    acc[trisz(NI) + a] += s;
  }
}

// ---------------------------------------------------------------------------
// K5: back-substitution  d_i = V'^-1 (g - W^T dK);  pose_new = pose_prev - d_i.
// Returns ||d_i||^2 through dd and ||pose_prev||^2 through pp.
template <int NI>
CC_HD void lane_backsub(const double* Vb, const double* Wb, const double* gb, long long S,
                        long long slot, double lam, const double* dK, const double prev[6],
                        double out[6], double& dd, double& pp) {
  double V[21], rhs[6];
  CC_UNROLL
  for (int e = 0; e < 21; ++e) V[e] = Vb[e * S + slot];
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    V[tri(6, i, i)] *= 1.0 + lam;
    rhs[i] = gb[i * S + slot];
  }
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    const double dk = dK[a];
    CC_UNROLL
    for (int c = 0; c < 6; ++c) rhs[c] -= Wb[(a * 6 + c) * S + slot] * dk;
  }
  double id[6];
  chol6(V, id);
  chol6_fwd(V, id, rhs);
  chol6_bwd(V, id, rhs);
  dd = 0.0; pp = 0.0;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    out[i] = prev[i] - rhs[i];
    dd += rhs[i] * rhs[i];
    pp += prev[i] * prev[i];
  }
}

// ---------------------------------------------------------------------------
// Pixel -> normalised, distortion removed (pose initialisation only).
template <int MODEL> CC_HD void undistort_px(double u, double v, const double* intr, double& mx, double& my);

// cvUndistortPointsInternal: 5 fixed-point sweeps
template <> CC_HD void undistort_px<PINHOLE>(double u, double v, const double* intr, double& mx, double& my) {
  const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
  const double ifx = 1.0 / intr[0], ify = 1.0 / intr[1];      // invariant over the points of a launch
  const double x0 = (u - intr[2]) * ifx, y0 = (v - intr[3]) * ify;
  double x = x0, y = y0;
  for (int it = 0; it < 5; ++it) {
    const double r2 = x * x + y * y;
    const double icd = fast_rcp(1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
    if (icd < 0) { x = x0; y = y0; break; }
    const double dX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double dY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    x = (x0 - dX) * icd;
    y = (y0 - dY) * icd;
  }
  mx = x; my = y;
}

// cv::fisheye::undistortPoints: Newton on theta (<= 10 steps, |fix| < 1e-8)
template <> CC_HD void undistort_px<FISHEYE>(double u, double v, const double* intr, double& mx, double& my) {
  const double k1 = intr[4], k2 = intr[5], k3 = intr[6], k4 = intr[7];
  const double ifx = 1.0 / intr[0], ify = 1.0 / intr[1];      // invariant over the points of a launch
  const double px = (u - intr[2]) * ifx, py = (v - intr[3]) * ify;
  const double p2 = px * px + py * py;
  double thd = 0.0;
  if (p2 > 0.0) fast_rsqrt(p2, thd);
  const double hp = 1.5707963267948966;
  thd = thd > hp ? hp : thd;
  double scale = 1.0;
  if (thd > 1e-8) {
    double th = thd;
    for (int j = 0; j < 10; ++j) {
      const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      const double fix = (th * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8) - thd) *
                         fast_rcp(1 + 3 * k1 * t2 + 5 * k2 * t4 + 7 * k3 * t6 + 9 * k4 * t8);
      th -= fix;
      if (fabs(fix) < 1e-8) break;
    }
    scale = tan_acute(th) * fast_rcp(thd);
  }
  mx = px * scale; my = py * scale;
}

// packed SPD solve A x = b, A upper (N(N+1)/2), in place; returns false on a bad pivot
template <int N>
CC_HD bool chol_solve(double* A, double* b) {
  bool ok = true;
  CC_UNROLL
  for (int i = 0; i < N; ++i) {
    double d = A[tri(N, i, i)];
    CC_UNROLL
    for (int k = 0; k < i; ++k) d -= A[tri(N, k, i)] * A[tri(N, k, i)];
    ok = ok && (d > 0.0);
    const double u = sqrt(d > 0.0 ? d : 1.0);
    A[tri(N, i, i)] = u;
    CC_UNROLL
    for (int j = i + 1; j < N; ++j) {
      double s = A[tri(N, i, j)];
      CC_UNROLL
      for (int k = 0; k < i; ++k) s -= A[tri(N, k, i)] * A[tri(N, k, j)];
      A[tri(N, i, j)] = s / u;
    }
  }
  CC_UNROLL
  for (int i = 0; i < N; ++i) {
    double s = b[i];
    CC_UNROLL
    for (int k = 0; k < i; ++k) s -= A[tri(N, k, i)] * b[k];
    b[i] = s / A[tri(N, i, i)];
  }
  CC_UNROLL
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
    CC_UNROLL
    for (int k = i + 1; k < N; ++k) s -= A[tri(N, i, k)] * b[k];
    b[i] = s / A[tri(N, i, i)];
  }
  return ok;
}

// rotation matrix (row-major) -> Rodrigues vector, via the unit quaternion
CC_HD void rot_to_rvec(const double R[9], double r[3]) {
  const double tr = R[0] + R[4] + R[8];
  double qw, qx, qy, qz;
  if (tr > 0.0) {
    const double S = sqrt(tr + 1.0) * 2.0;
    qw = 0.25 * S; qx = (R[7] - R[5]) / S; qy = (R[2] - R[6]) / S; qz = (R[3] - R[1]) / S;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double S = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
    qw = (R[7] - R[5]) / S; qx = 0.25 * S; qy = (R[1] + R[3]) / S; qz = (R[2] + R[6]) / S;
  } else if (R[4] > R[8]) {
    const double S = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
    qw = (R[2] - R[6]) / S; qx = (R[1] + R[3]) / S; qy = 0.25 * S; qz = (R[5] + R[7]) / S;
  } else {
    const double S = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
    qw = (R[3] - R[1]) / S; qx = (R[2] + R[6]) / S; qy = (R[5] + R[7]) / S; qz = 0.25 * S;
  }
  if (qw < 0.0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
  const double vn = sqrt(qx * qx + qy * qy + qz * qz);
  const double sc = vn > 1e-12 ? 2.0 * atan2(vn, qw) / vn : 2.0;
  r[0] = qx * sc; r[1] = qy * sc; r[2] = qz * sc;
}

// ---------------------------------------------------------------------------
// K6: pose initialisation of one planar (Z = 0) view, the structure of
// cv::internal::InitExtrinsics / cvFindExtrinsicCameraParams2 (planar branch):
// undistort -> homography (normalised DLT, then Gauss-Newton on the transfer
// error, <= 10 steps) -> (u1, u2 - (u1.u2)u1, u1 x u2) -> Rodrigues vector,
// t = h3 / sc - R [mx my 0]^T.  Returns 0 on success.
// Normal equations of the homography fit.  The two rows of a corner are [a 0 c0 | e0] and
// [0 a c1 | e1] (a: 3 entries, c0, c1: 2), so of the 36 + 8 sums only 29 are distinct and
// none of the structural zeros is multiplied: S = [a a^T (6) | a c0^T (6) | a c1^T (6) |
// c0 c0^T + c1 c1^T (3) | a e0 (3) | a e1 (3) | c0 e0 + c1 e1 (2)].
CC_HD void homography_accumulate(const double a[3], const double c0[2], const double c1[2], double e0, double e1,
                                 double S[29]) {
  S[0] = fma(a[0], a[0], S[0]); S[1] = fma(a[0], a[1], S[1]); S[2] = fma(a[0], a[2], S[2]);
  S[3] = fma(a[1], a[1], S[3]); S[4] = fma(a[1], a[2], S[4]); S[5] = fma(a[2], a[2], S[5]);
  CC_UNROLL
  for (int i = 0; i < 3; ++i) {
    S[6 + i * 2] = fma(a[i], c0[0], S[6 + i * 2]);
    S[7 + i * 2] = fma(a[i], c0[1], S[7 + i * 2]);
    S[12 + i * 2] = fma(a[i], c1[0], S[12 + i * 2]);
    S[13 + i * 2] = fma(a[i], c1[1], S[13 + i * 2]);
    S[21 + i] = fma(a[i], e0, S[21 + i]);
    S[24 + i] = fma(a[i], e1, S[24 + i]);
  }
  S[18] = fma(c0[0], c0[0], fma(c1[0], c1[0], S[18]));
  S[19] = fma(c0[0], c0[1], fma(c1[0], c1[1], S[19]));
  S[20] = fma(c0[1], c0[1], fma(c1[1], c1[1], S[20]));
  S[27] = fma(c0[0], e0, fma(c1[0], e1, S[27]));
  S[28] = fma(c0[1], e0, fma(c1[1], e1, S[28]));
}
// the 29 sums -> packed upper triangle of the 8 x 8 matrix and the right-hand side
CC_HD void homography_expand(const double S[29], double A[36], double b[8]) {
  const double aa[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
  CC_UNROLL
  for (int i = 0; i < 3; ++i) {
    CC_UNROLL
    for (int j = i; j < 3; ++j) { A[tri(8, i, j)] = aa[i][j]; A[tri(8, 3 + i, 3 + j)] = aa[i][j]; }
    CC_UNROLL
    for (int j = 0; j < 3; ++j) A[tri(8, i, 3 + j)] = 0.0;
    A[tri(8, i, 6)] = S[6 + i * 2]; A[tri(8, i, 7)] = S[7 + i * 2];
    A[tri(8, 3 + i, 6)] = S[12 + i * 2]; A[tri(8, 3 + i, 7)] = S[13 + i * 2];
    b[i] = S[21 + i]; b[3 + i] = S[24 + i];
  }
  A[tri(8, 6, 6)] = S[18]; A[tri(8, 6, 7)] = S[19]; A[tri(8, 7, 7)] = S[20];
  b[6] = S[27]; b[7] = S[28];
}

// The corners k0, k0 + step, ... of the view are walked by this lane; red.sum<M>(a) makes
// the M sums in a[] complete (nothing to do when one lane walks the whole view, a
// butterfly over the lanes that share the view otherwise -- every lane then holds the
// same sums).
struct NoReduce {
  template <int M> CC_HD void sum(double*) const {}
};

template <int MODEL, class Red>
CC_HD int pose_init_walk(const float* base, int n, int k0, int step, const double* intr, double pose[6],
                         const Red& red) {
  if (n < 4) return 1;
  // pass 1: centroid / spread of the board points
  double m4[4] = {0, 0, 0, 0};
  for (int k = k0; k < n; k += step) {
    const Obs o = load_obs<true>(base, k);
    m4[0] += o.X; m4[1] += o.Y; m4[2] += (double)o.X * o.X; m4[3] += (double)o.Y * o.Y;
  }
  red.template sum<4>(m4);
  const double sx = m4[0], sy = m4[1], sxx = m4[2], syy = m4[3];
  const double inv_n = 1.0 / n;
  const double mX = sx * inv_n, mY = sy * inv_n;
  double dX = sqrt(fmax(sxx * inv_n - mX * mX, 0.0)), dY = sqrt(fmax(syy * inv_n - mY * mY, 0.0));
  const double iX = dX > 1e-300 ? 1.0 / dX : 1.0, iY = dY > 1e-300 ? 1.0 / dY : 1.0;
  // pass 2: inhomogeneous DLT (h33 = 1) in normalised board coordinates
  double A[36], b[8], S[29];
  CC_UNROLL
  for (int i = 0; i < 29; ++i) S[i] = 0.0;
  for (int k = k0; k < n; k += step) {
    const Obs o = load_obs<true>(base, k);
    double x, y;
    undistort_px<MODEL>(o.u, o.v, intr, x, y);
    const double X = (o.X - mX) * iX, Y = (o.Y - mY) * iY;
    // rows: [X Y 1 0 0 0 -xX -xY | x], [0 0 0 X Y 1 -yX -yY | y]
    const double a[3] = {X, Y, 1.0}, c0[2] = {-x * X, -x * Y}, c1[2] = {-y * X, -y * Y};
    homography_accumulate(a, c0, c1, x, y, S);
  }
  red.template sum<29>(S);
  homography_expand(S, A, b);
  if (!chol_solve<8>(A, b)) return 2;
  double h[8];
  CC_UNROLL
  for (int i = 0; i < 8; ++i) h[i] = b[i];
  // Gauss-Newton on the transfer error  sum |H X - m|^2
  for (int it = 0; it < 10; ++it) {
    CC_UNROLL
    for (int i = 0; i < 29; ++i) S[i] = 0.0;
    for (int k = k0; k < n; k += step) {
      const Obs o = load_obs<true>(base, k);
      double x, y;
      undistort_px<MODEL>(o.u, o.v, intr, x, y);
      const double X = (o.X - mX) * iX, Y = (o.Y - mY) * iY;
      const double iw = fast_rcp(h[6] * X + h[7] * Y + 1.0);
      const double px = (h[0] * X + h[1] * Y + h[2]) * iw, py = (h[3] * X + h[4] * Y + h[5]) * iw;
      const double Xw = X * iw, Yw = Y * iw;
      const double a[3] = {Xw, Yw, iw}, c0[2] = {-px * Xw, -px * Yw}, c1[2] = {-py * Xw, -py * Yw};
      homography_accumulate(a, c0, c1, px - x, py - y, S);
    }
    red.template sum<29>(S);
    homography_expand(S, A, b);
    if (!chol_solve<8>(A, b)) break;
    double dmax = 0.0, hmax = 0.0;
    CC_UNROLL
    for (int i = 0; i < 8; ++i) {
      h[i] -= b[i];
      dmax = fmax(dmax, fabs(b[i]));
      hmax = fmax(hmax, fabs(h[i]));
    }
    // Near its minimum this Gauss-Newton contracts by the relative transfer residual per step
    // (1e-3 or better on calibration boards), so the step after one of 1e-5 would move the
    // homography by about 1e-8: below FLT_EPSILON, where cv2's own LM refinement of the
    // homography stops.  The pass that would only confirm it is not run.
    if (dmax <= 1e-5 * hmax) break;
  }
  // columns of the homography in centred board millimetres
  double h1[3] = {h[0] * iX, h[3] * iX, h[6] * iX};
  double h2[3] = {h[1] * iY, h[4] * iY, h[7] * iY};
  const double h3[3] = {h[2], h[5], 1.0};
  const double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]);
  const double n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
  if (!(n1 > 0.0) || !(n2 > 0.0)) return 3;
  const double isc = 2.0 / (n1 + n2);
  double u1[3] = {h1[0] / n1, h1[1] / n1, h1[2] / n1};
  const double dot = u1[0] * h2[0] + u1[1] * h2[1] + u1[2] * h2[2];
  double u2[3] = {h2[0] - dot * u1[0], h2[1] - dot * u1[1], h2[2] - dot * u1[2]};
  const double m2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (!(m2 > 0.0)) return 3;
  u2[0] /= m2; u2[1] /= m2; u2[2] /= m2;
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double R[9] = {u1[0], u2[0], u3[0], u1[1], u2[1], u3[1], u1[2], u2[2], u3[2]};
  rot_to_rvec(R, pose);
  pose[3] = h3[0] * isc - (R[0] * mX + R[1] * mY);
  pose[4] = h3[1] * isc - (R[3] * mX + R[4] * mY);
  pose[5] = h3[2] * isc - (R[6] * mX + R[7] * mY);
  return 0;
}

template <int MODEL>
CC_HD int lane_pose_init(const LaneTile& tv, const double* intr, double pose[6]) {
  return pose_init_walk<MODEL>(tv.base, tv.n, 0, 1, intr, pose, NoReduce());
}

// ---------------------------------------------------------------------------
// Reduced solve (U' - C) dK = b - d on the free intrinsics: symmetric Jacobi
// scaling + Gaussian elimination with partial pivoting (single lane; NI <= 9).
// glob = [U | b | cost], schur = [C | d]; intr_out = intr_in + scale * dK.
// The eliminated form of CvLevMarq::step / fisheye::calibrate's solve(JJ2, ex3).
// ws: REDUCED_WS doubles of working storage (the kernel passes LDS: the pivoting makes
// every index dynamic, and private arrays indexed dynamically live in scratch memory).
// Lanes per view of a set (camcal_tiles.lanes_per_view: 0 = by size) and the partial records a
// single-pass launch writes: shared by the HIP build and the tests' emulator so that both
// reject the same sets in camcal_normal_eq_fused_static.
constexpr int GROUP64_MAX_TILES = 32;      // <= 2048 views: one wavefront per view
constexpr int GROUP16_MAX_TILES = 448;     // <= 28672 views: four views per wavefront (tools/gpu_lanes.py: ahead of one lane per
                                           // view up to ~30 000 views, full boards and ragged ones alike; 2.4 vs 3.9 ms at 12 000)
constexpr int STATIC_MAX_PARTIALS = 16384; // tiles (one lane per view) or wavefronts (grouped sets)
inline int lanes_per_view_of(int n_tiles, int planar, int want) {
  if (!planar || want == 1) return 1;
  if (want == 64) return n_tiles <= GROUP64_MAX_TILES ? 64 : 1;
  if (want == 16) return n_tiles <= GROUP16_MAX_TILES ? 16 : 1;
  return n_tiles <= GROUP64_MAX_TILES ? 64 : n_tiles <= GROUP16_MAX_TILES ? 16 : 1;
}
// Which sets run the folded stages (camcal_refine_normal_eq_partial ... camcal_backsub_intr_normal).  Pinhole LM
// step: every set in a several-lanes-per-view mapping (loop time separate / folded, C1R sets, one box,
// tools/gpu_pinhole_mid.py: 1000 views 0.57 / 0.50 ms, 6000 0.68 / 0.51, 12 000 0.81 / 0.64, 28 000 1.20 / 1.11).
// Fisheye iteration: up to FOLD_MAX_WAVES wavefronts per launch -- its fused refinement + evaluation kernel runs at
// the evaluation's one wavefront per SIMD, which costs the refinement its second wavefront once there is more than
// a SIMD's worth of them (C2 sets, tools/gpu_c3_phases.py: 1000 views / 1000 wavefronts 0.59 / 0.52 ms, 2000 / 2048
// 0.74 / 0.76, 3000 / 768 1.01 / 0.85, 6000 / 1536 1.15 / 1.01, 12 000 / 3072 1.55 / 1.58, 20 000 / 5120 1.97 / 2.28).
// CAMCAL_FOLD_MAX_WAVES (read once) overrides the limit for both models (0: separate stages everywhere).
constexpr int FOLD_MAX_WAVES = 1600;
inline bool folded_set_of(int n_tiles, int planar, int want, bool fisheye) {
  static const long forced = [] {
    const char* e = std::getenv("CAMCAL_FOLD_MAX_WAVES");
    return e ? std::atol(e) : -1L;
  }();
  const int G = lanes_per_view_of(n_tiles, planar, want);
  if (G <= 1) return false;
  if (forced >= 0) return (long)n_tiles * G <= forced;
  return !fisheye || (long)n_tiles * G <= FOLD_MAX_WAVES;
}
inline long long static_partials(int n_tiles, int planar, int want) {
  return (long long)n_tiles * lanes_per_view_of(n_tiles, planar, want);
}

constexpr int REDUCED_WS = 128;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void reduced_solve(int NI, const double* glob, const double* schur, unsigned mask, double lam,
                          double scale, const double* intr_in, double* intr_out, double* dK,
                          double* stats, double* ws) {
  const int NU = NI * (NI + 1) / 2;
  double* A = ws;            // 9 x 10
  double* sc = ws + 90;      // 9
  double* x = ws + 99;       // 9
  double* d = ws + 108;      // 9
  int* idx = (int*)(ws + 117);
  int nf = 0;
  for (int a = 0; a < NI; ++a)
    if (mask & (1u << a)) idx[nf++] = a;
  auto tr = [NI](int i, int j) { return i <= j ? i * NI - (i * (i - 1)) / 2 + (j - i) : j * NI - (j * (j - 1)) / 2 + (i - j); };
  for (int i = 0; i < nf; ++i) {
    const int a = idx[i];
    const double dd = glob[tr(a, a)] * (1.0 + lam) - schur[tr(a, a)];
    sc[i] = dd > 0.0 ? 1.0 / sqrt(dd) : 1.0;
  }
  for (int i = 0; i < nf; ++i) {
    const int a = idx[i];
    for (int j = 0; j < nf; ++j) {
      const int b = idx[j];
      double u = glob[tr(a, b)];
      if (a == b) u *= 1.0 + lam;
      A[i * 10 + j] = (u - schur[tr(a, b)]) * sc[i] * sc[j];
    }
    A[i * 10 + 9] = (glob[NU + a] - schur[NU + a]) * sc[i];
  }
  double okf = 1.0;
  for (int c = 0; c < nf; ++c) {
    int piv = c;
    double best = fabs(A[c * 10 + c]);
    for (int r = c + 1; r < nf; ++r)
      if (fabs(A[r * 10 + c]) > best) { best = fabs(A[r * 10 + c]); piv = r; }
    if (!(best > 0.0)) { okf = 0.0; continue; }
    if (piv != c)
      for (int k = 0; k < 10; ++k) { const double tmp = A[c * 10 + k]; A[c * 10 + k] = A[piv * 10 + k]; A[piv * 10 + k] = tmp; }
    const double ip = 1.0 / A[c * 10 + c];
    for (int r = c + 1; r < nf; ++r) {
      const double f = A[r * 10 + c] * ip;
      if (f != 0.0) {
        for (int k = c; k < nf; ++k) A[r * 10 + k] -= f * A[c * 10 + k];
        A[r * 10 + 9] -= f * A[c * 10 + 9];
      }
    }
  }
  for (int c = nf - 1; c >= 0; --c) {
    double s = A[c * 10 + 9];
    for (int k = c + 1; k < nf; ++k) s -= A[c * 10 + k] * x[k];
    x[c] = A[c * 10 + c] != 0.0 ? s / A[c * 10 + c] : 0.0;
  }
  for (int a = 0; a < NI; ++a) d[a] = 0.0;
  for (int i = 0; i < nf; ++i) d[idx[i]] = x[i] * sc[i];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int a = 0; a < NI; ++a) {
    const double in = intr_in[a];
    const double st = scale * d[a];
    const double out = in + st;
    dK[a] = d[a];
    intr_out[a] = out;
    s0 += st * st;
    s1 += in * in;
    if (a < 4) { s2 += st * st; s3 += out * out; }
  }
  stats[0] = s0; stats[1] = s1; stats[2] = s2; stats[3] = s3; stats[4] = okf;
}

}  // namespace camcal
