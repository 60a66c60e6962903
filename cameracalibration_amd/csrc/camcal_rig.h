// Multi-camera rig (BASELINE.json config 4): per-lane bodies.
//
// Residual of camera c seeing board pose p:  proj_c( R_c (R_p X + t_p) + t_c ) - obs.
// Unknowns: intrinsics of every camera, camera-to-rig transforms (r_c, t_c) of
// cameras 1.. (camera 0 is the rig frame) and one 6-dof pose per board
// placement.  No reference counterpart: the tool calibrates its four cameras
// one after another (controller/controller_internal_calibration.py:375-399);
// SURVEY.md 8a defines this joint problem as a build-side capability.
//
// Reuse: a (pose, camera) pair is an ordinary "view" whose pose is the
// COMPOSED transform (R_c R_p, R_c t_p + t_c).  The single-camera kernels give
// its blocks in the local tangent frame of that composed pose
//   Y' = exp(w) (R X) + t + tau,      xi = (w, tau)
// and the chain rule to the rig parameters is linear and point-independent:
//   xi = B d(pose_p) + A d(ext_c)
//   B = blkdiag(R_c Jl(r_p), R_c)
//   A = [ Jl(r_c)             0 ]
//       [ -[R_c t_p]x Jl(r_c) I ]
// so J_pose = J_loc B, J_ext = J_loc A and every rig block is a 6x6 congruence
// of the local blocks (done here, once per view).
#pragma once
#include "camcal_lane.h"

namespace camcal {

// C = A^T B (all 6x6 row-major)
CC_HD void mat6_atb(const double* A, const double* B, double* C) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) {
      double s = 0.0;
      CC_UNROLL
      for (int k = 0; k < 6; ++k) s = fma(A[k * 6 + i], B[k * 6 + j], s);
      C[i * 6 + j] = s;
    }
}
// C = A B
CC_HD void mat6_ab(const double* A, const double* B, double* C) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) {
      double s = 0.0;
      CC_UNROLL
      for (int k = 0; k < 6; ++k) s = fma(A[i * 6 + k], B[k * 6 + j], s);
      C[i * 6 + j] = s;
    }
}

CC_HD void mat3_mul(const double* A, const double* B, double* C) {
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// composed pose of (board pose p, camera extrinsic c): rotation vector + translation
CC_HD void rig_compose(const double pose_p[6], const double ext[6], double out[6]) {
  double Rp[9], Rc[9], R[9];
  rodrigues(pose_p, Rp);
  rodrigues(ext, Rc);
  mat3_mul(Rc, Rp, R);
  rot_to_rvec(R, out);
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    out[3 + i] = Rc[i * 3] * pose_p[3] + Rc[i * 3 + 1] * pose_p[4] + Rc[i * 3 + 2] * pose_p[5] + ext[3 + i];
}

// chain matrices A (d xi / d ext) and B (d xi / d pose), 6x6 row-major
CC_HD void rig_chain(const double pose_p[6], const double ext[6], double A[36], double B[36]) {
  double Rc[9], Jlp[9], Jlc[9], M1[9];
  rodrigues(ext, Rc);
  left_jacobian(pose_p, Jlp);
  left_jacobian(ext, Jlc);
  mat3_mul(Rc, Jlp, M1);
  double s[3];
  CC_UNROLL
  for (int i = 0; i < 3; ++i) s[i] = Rc[i * 3] * pose_p[3] + Rc[i * 3 + 1] * pose_p[4] + Rc[i * 3 + 2] * pose_p[5];
  const double N[9] = {0.0, s[2], -s[1], -s[2], 0.0, s[0], s[1], -s[0], 0.0};   // -[s]x
  double NJ[9];
  mat3_mul(N, Jlc, NJ);
  CC_UNROLL
  for (int i = 0; i < 36; ++i) { A[i] = 0.0; B[i] = 0.0; }
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) {
      A[i * 6 + j] = Jlc[i * 3 + j];
      A[(3 + i) * 6 + j] = NJ[i * 3 + j];
      B[i * 6 + j] = M1[i * 3 + j];
      B[(3 + i) * 6 + 3 + j] = Rc[i * 3 + j];
    }
  A[3 * 6 + 3] = 1.0; A[4 * 6 + 4] = 1.0; A[5 * 6 + 5] = 1.0;
}

// One (pose, camera) view: local blocks -> rig blocks.
//   Vp (21 upper)  = B^T V B          pose block contribution
//   gp (6)         = B^T g
//   Wp ((NI+6) x 6): rows 0..NI-1 = W B (intrinsics x pose), rows NI.. = A^T V B (extrinsic x pose)
//   ge: [ U_ie = W A (NI x 6) | U_ee = A^T V A (21 upper) | b_e = A^T g (6) ]   (added into acc)
template <int NI>
CC_HD void rig_assemble_view(const double* Vl, const double* Wl, const double* gl, const double A[36],
                             const double B[36], double* Vp, double* gp, double* Wp, double* acc) {
  double V[36];
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) V[i * 6 + j] = Vl[i <= j ? tri(6, i, j) : tri(6, j, i)];
  double VB[36], VA[36], T[36];
  mat6_ab(V, B, VB);
  mat6_ab(V, A, VA);
  mat6_atb(B, VB, T);
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) Vp[tri(6, i, j)] = T[i * 6 + j];
  mat6_atb(A, VB, T);                       // extrinsic x pose
  CC_UNROLL
  for (int i = 0; i < 36; ++i) Wp[NI * 6 + i] = T[i];
  mat6_atb(A, VA, T);                       // extrinsic x extrinsic
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) acc[NI * 6 + tri(6, i, j)] += T[i * 6 + j];
  CC_UNROLL
  for (int j = 0; j < 6; ++j) {
    double sb = 0.0, sa = 0.0;
    CC_UNROLL
    for (int k = 0; k < 6; ++k) { sb = fma(B[k * 6 + j], gl[k], sb); sa = fma(A[k * 6 + j], gl[k], sa); }
    gp[j] = sb;
    acc[NI * 6 + 21 + j] += sa;
  }
  CC_UNROLL
  for (int a = 0; a < NI; ++a)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) {
      double sb = 0.0, sa = 0.0;
      CC_UNROLL
      for (int k = 0; k < 6; ++k) { sb = fma(Wl[a * 6 + k], B[k * 6 + j], sb); sa = fma(Wl[a * 6 + k], A[k * 6 + j], sa); }
      Wp[a * 6 + j] = sb;
      acc[a * 6 + j] += sa;
    }
}

// Pose-level elimination for one board pose: V = sum_c Vp_c (damped), Cholesky
// V = U^T U, Y_c = U^-T Wp_c^T for every camera (6 x G rows into Y), z = U^-T g.
// Returns false when no camera sees the pose (all-zero block).
template <int G>
CC_HD bool rig_pose_factor(int n_cams, const double* Vp_all, const double* gp_all, long long P, long long p,
                           double lam, double U[21], double g[6]) {
  CC_UNROLL
  for (int e = 0; e < 21; ++e) U[e] = 0.0;
  CC_UNROLL
  for (int e = 0; e < 6; ++e) g[e] = 0.0;
  for (int c = 0; c < n_cams; ++c) {
    CC_UNROLL
    for (int e = 0; e < 21; ++e) U[e] += Vp_all[((long long)c * 21 + e) * P + p];
    CC_UNROLL
    for (int e = 0; e < 6; ++e) g[e] += gp_all[((long long)c * 6 + e) * P + p];
  }
  double tr = 0.0;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) { tr += U[tri(6, i, i)]; U[tri(6, i, i)] *= 1.0 + lam; }
  if (!(tr > 0.0)) return false;
  return chol6(U);
}

// ---------------------------------------------------------------------------
// Reduced system over the rig's n = n_cams * G globals (G = NI + 6, n <= 64):
//   M = blockdiag_c [[U_ii U_ie], [U_ie^T U_ee]] (diag * (1 + lam))  -  C,   rhs = [b_i ; b_e] - d
// from the record  rec = [ per camera: glob (NV = [U|b|cost]) | rext (NI*6 + 21 + 6) ] [ C (n x n) | d (n) ],
// fixed parameters (mask bit clear) replaced by identity rows, Jacobi scaling,
// Cholesky, solve.  Written as phases over a worker index (`lane` of `nl`) so
// that the kernel runs one wavefront with a barrier between phases and the CPU
// emulator loops the workers; element (i, j) sees the same operations in the
// same order either way.  A (n x RIG_LD) and the vectors live in LDS / host memory.
constexpr int RIG_MAXN = 64;
constexpr int RIG_LD = RIG_MAXN + 1;

struct RigSolveWs {
  double A[RIG_MAXN * RIG_LD];
  double rhs[RIG_MAXN], sc[RIG_MAXN], x[RIG_MAXN];
  int ok;
};

CC_HD void rig_solve_build(int lane, int nl, int NI, int n_cams, const double* rec, unsigned long long mask,
                           double lam, RigSolveWs& w) {
  const int G = NI + 6, n = n_cams * G;
  const int NU = NI * (NI + 1) / 2, NV = NU + NI + 1, NX = NI * 6 + 27;
  const double* Cm = rec + (long long)n_cams * (NV + NX);
  const double* d = Cm + (long long)n * n;
  for (int i = lane; i < n; i += nl) {
    const int c = i / G, a = i - c * G;
    const double* glob = rec + (long long)c * (NV + NX);
    const double* rx = glob + NV;
    const bool fi = (mask >> i) & 1ull;
    for (int j = 0; j < n; ++j) {
      const int cj = j / G, b = j - cj * G;
      const bool fj = (mask >> j) & 1ull;
      double v = 0.0;
      if (!fi || !fj) {
        v = (i == j) ? 1.0 : 0.0;
      } else {
        if (c == cj) {
          const int lo = a < b ? a : b, hi = a < b ? b : a;
          if (hi < NI) v = glob[tri(NI, lo, hi)];
          else if (lo < NI) v = rx[lo * 6 + (hi - NI)];
          else v = rx[NI * 6 + tri(6, lo - NI, hi - NI)];
          if (i == j) v *= 1.0 + lam;
        }
        v -= Cm[(long long)i * n + j];
      }
      w.A[i * RIG_LD + j] = v;
    }
    double r = 0.0;
    if (fi) r = (a < NI ? glob[NU + a] : rx[NI * 6 + 21 + (a - NI)]) - d[i];
    w.rhs[i] = r;
    const double dii = w.A[i * RIG_LD + i];
    w.sc[i] = dii > 0.0 ? 1.0 / sqrt(dii) : 1.0;
  }
  if (lane == 0) w.ok = 1;
}
CC_HD void rig_solve_scale(int lane, int nl, int n, RigSolveWs& w) {
  for (int i = lane; i < n; i += nl) {
    for (int j = 0; j < n; ++j) w.A[i * RIG_LD + j] *= w.sc[i] * w.sc[j];
    w.rhs[i] *= w.sc[i];
  }
}
// Cholesky step k, part 1: pivot and column k of L (lower triangle, in place)
CC_HD void rig_solve_chol_col(int lane, int nl, int n, int k, RigSolveWs& w) {
  const double p = w.A[k * RIG_LD + k];
  const bool good = p > 0.0;
  const double inv = good ? 1.0 / sqrt(p) : 0.0;
  for (int i = k + 1 + lane; i < n; i += nl) w.A[i * RIG_LD + k] *= inv;
  if (lane == 0 && !good) w.ok = 0;
}
CC_HD void rig_solve_chol_diag(int lane, int n, int k, RigSolveWs& w) {
  if (lane == 0) {
    const double p = w.A[k * RIG_LD + k];
    w.A[k * RIG_LD + k] = p > 0.0 ? sqrt(p) : 1.0;
  }
}
// part 2: trailing update, worker per column j > k
CC_HD void rig_solve_chol_update(int lane, int nl, int n, int k, RigSolveWs& w) {
  for (int j = k + 1 + lane; j < n; j += nl) {
    const double ljk = w.A[j * RIG_LD + k];
    for (int i = j; i < n; ++i) w.A[i * RIG_LD + j] = fma(-w.A[i * RIG_LD + k], ljk, w.A[i * RIG_LD + j]);
  }
}
// triangular solves are short (n <= 64): one worker
CC_HD void rig_solve_substitute(int n, RigSolveWs& w) {
  for (int i = 0; i < n; ++i) {
    double s = w.rhs[i];
    for (int j = 0; j < i; ++j) s = fma(-w.A[i * RIG_LD + j], w.x[j], s);
    w.x[i] = s / w.A[i * RIG_LD + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = w.x[i];
    for (int j = i + 1; j < n; ++j) s = fma(-w.A[j * RIG_LD + i], w.x[j], s);
    w.x[i] = s / w.A[i * RIG_LD + i];
  }
}
// parameter update: out = in - dG; stats = [||dG||^2, ||globals_in||^2]; a failed
// factorisation gives NaN so that the LM driver rejects the step
CC_HD void rig_solve_finish(int NI, int n_cams, unsigned long long mask, RigSolveWs& w, const double* intr_in,
                            const double* ext_in, double* intr_out, double* ext_out, double* dG, double* stats) {
  const int G = NI + 6, n = n_cams * G;
  double s0 = 0.0, s1 = 0.0;
  const double bad = w.ok ? 0.0 : (0.0 / 0.0);
  for (int i = 0; i < n; ++i) {
    const int c = i / G, a = i - c * G;
    const double dx = ((mask >> i) & 1ull) ? w.x[i] * w.sc[i] + bad : 0.0;
    dG[i] = dx;
    const double prev = a < NI ? intr_in[c * 16 + a] : ext_in[c * 6 + (a - NI)];
    if (a < NI) intr_out[c * 16 + a] = prev - dx;
    else ext_out[c * 6 + (a - NI)] = prev - dx;
    s0 += dx * dx;
    s1 += prev * prev;
  }
  stats[0] = s0;
  stats[1] = s1;
}

}  // namespace camcal
