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
#include <cmath>
#include "../../include/camcal.h"
#include "camcal_lane.h"

namespace camcal {

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

// Chain matrices A (d xi / d ext) and B (d xi / d pose) by their 3x3 blocks (row-major):
//   A = [ J 0 ]      B = [ M 0 ]      J = Jl(r_c), N = -[R_c t_p]x Jl(r_c), M = R_c Jl(r_p), R = R_c
//       [ N I ]          [ 0 R ]
struct RigChain {
  double J[9], N[9], M[9], R[9];
};
CC_HD void rig_chain(const double pose_p[6], const double ext[6], RigChain& c) {
  double Jlp[9];
  rodrigues(ext, c.R);
  left_jacobian(pose_p, Jlp);
  left_jacobian(ext, c.J);
  mat3_mul(c.R, Jlp, c.M);
  double s[3];
  CC_UNROLL
  for (int i = 0; i < 3; ++i) s[i] = c.R[i * 3] * pose_p[3] + c.R[i * 3 + 1] * pose_p[4] + c.R[i * 3 + 2] * pose_p[5];
  const double X[9] = {0.0, s[2], -s[1], -s[2], 0.0, s[0], s[1], -s[0], 0.0};   // -[s]x
  mat3_mul(X, c.J, c.N);
}

// One (pose, camera) view: local blocks -> rig blocks, every product written out by the
// block structure of A and B (half the multiplications of dense 6x6 congruences).  Two
// halves, so that a device kernel can give them to different wavefronts (together they
// need more registers than a lane has).
//
// Pose half:  Vp (21 upper) = B^T V B,  gp (6) = B^T g,  We (6 x 6) = A^T V B  (the
// extrinsic rows NI .. NI+5 of Wp),  acc_e += [ U_ee = A^T V A (21 upper) | b_e = A^T g (6) ].
CC_HD void rig_assemble_pose(const double* Vl, const double* gl, const RigChain& c, double* Vp, double* gp,
                             double* We, double* acc_e) {
  double V[36];
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) V[i * 6 + j] = Vl[i <= j ? tri(6, i, j) : tri(6, j, i)];
  double VB[36];
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) {
      double sm = 0.0, sr = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) {
        sm = fma(V[i * 6 + k], c.M[k * 3 + j], sm);
        sr = fma(V[i * 6 + 3 + k], c.R[k * 3 + j], sr);
      }
      VB[i * 6 + j] = sm;
      VB[i * 6 + 3 + j] = sr;
    }
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) {
      double sv = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = i < 3 ? fma(c.M[k * 3 + i], VB[k * 6 + j], sv) : fma(c.R[k * 3 + i - 3], VB[(3 + k) * 6 + j], sv);
      Vp[tri(6, i, j)] = sv;
    }
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 6; ++j) {
      double sv = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(c.J[k * 3 + i], VB[k * 6 + j], sv);
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(c.N[k * 3 + i], VB[(3 + k) * 6 + j], sv);
      We[i * 6 + j] = sv;
      We[(3 + i) * 6 + j] = VB[(3 + i) * 6 + j];
    }
  // V A: the left 6 x 3 block; the right one is V itself
  double VA[18];
  CC_UNROLL
  for (int i = 0; i < 6; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) {
      double sv = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(V[i * 6 + k], c.J[k * 3 + j], sv);
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(V[i * 6 + 3 + k], c.N[k * 3 + j], sv);
      VA[i * 3 + j] = sv;
    }
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) {
      double sv = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(c.J[k * 3 + i], j < 3 ? VA[k * 3 + j] : V[k * 6 + j], sv);
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sv = fma(c.N[k * 3 + i], j < 3 ? VA[(3 + k) * 3 + j] : V[(3 + k) * 6 + j], sv);
      acc_e[tri(6, i, j)] += sv;
    }
  CC_UNROLL
  for (int i = 3; i < 6; ++i)
    CC_UNROLL
    for (int j = i; j < 6; ++j) acc_e[tri(6, i, j)] += V[i * 6 + j];
  CC_UNROLL
  for (int j = 0; j < 3; ++j) {
    double sm = 0.0, sr = 0.0, sa = 0.0;
    CC_UNROLL
    for (int k = 0; k < 3; ++k) {
      sm = fma(c.M[k * 3 + j], gl[k], sm);
      sr = fma(c.R[k * 3 + j], gl[3 + k], sr);
      sa = fma(c.J[k * 3 + j], gl[k], sa);
    }
    CC_UNROLL
    for (int k = 0; k < 3; ++k) sa = fma(c.N[k * 3 + j], gl[3 + k], sa);
    gp[j] = sm;
    gp[3 + j] = sr;
    acc_e[21 + j] += sa;
    acc_e[21 + 3 + j] += gl[3 + j];
  }
}

// Intrinsics half:  Wi (NI x 6) = W B  (rows 0 .. NI-1 of Wp),  acc_i += U_ie = W A (NI x 6).
template <int NI>
CC_HD void rig_assemble_intr(const double* Wl, const RigChain& c, double* Wi, double* acc_i) {
  CC_UNROLL
  for (int a = 0; a < NI; ++a)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) {
      double sm = 0.0, sr = 0.0, sa = 0.0;
      CC_UNROLL
      for (int k = 0; k < 3; ++k) {
        sm = fma(Wl[a * 6 + k], c.M[k * 3 + j], sm);
        sr = fma(Wl[a * 6 + 3 + k], c.R[k * 3 + j], sr);
        sa = fma(Wl[a * 6 + k], c.J[k * 3 + j], sa);
      }
      CC_UNROLL
      for (int k = 0; k < 3; ++k) sa = fma(Wl[a * 6 + 3 + k], c.N[k * 3 + j], sa);
      Wi[a * 6 + j] = sm;
      Wi[a * 6 + 3 + j] = sr;
      acc_i[a * 6 + j] += sa;
      acc_i[a * 6 + 3 + j] += Wl[a * 6 + 3 + j];
    }
}

// Pose-level elimination for one board pose: V = sum_c Vp_c (damped), Cholesky
// V = U^T U, Y_c = U^-T Wp_c^T for every camera (6 x G rows into Y), z = U^-T g.
// Returns false when no camera sees the pose (all-zero block).
template <int G>
CC_HD bool rig_pose_factor(int n_cams, const double* Vp_all, const double* gp_all, long long P, long long p,
                           double lam, double U[21], double g[6], double inv_diag[6]) {
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
  return chol6(U, inv_diag);
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
  const double* Cm = rec + (long long)n_cams * (NV + NX) + 5;   // RigLayout::o_C
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

// ---------------------------------------------------------------------------
// Record layout of one parameter buffer (include/camcal.h: camcal_rig.rec)
struct RigLayout {
  int NI, G, n, NV, NX, per, rec_len, o_cost, o_norms, o_gstats, o_C, o_d, total;
  CC_HD RigLayout(int ni, int n_cams) {
    NI = ni; G = ni + 6; n = n_cams * G;
    NV = ni * (ni + 1) / 2 + ni + 1; NX = ni * 6 + 27; per = NV + NX;
    rec_len = n_cams * per;
    o_cost = rec_len; o_norms = rec_len + 1; o_gstats = rec_len + 3;
    o_C = rec_len + 5; o_d = o_C + n * n; total = o_d + n;
  }
};

// ---------------------------------------------------------------------------
// Initialisation of the camera-from-rig transforms.  A board pose p seen by cameras c and
// k with per-view poses q_c, q_k (camera-from-board) gives one sample of T_c o T_k^-1:
//   R_ck = R_pc R_pk^T,   t_ck = t_pc - R_ck t_pk.
// out[0..8] = R_ck (row-major), out[9..11] = t_ck.
CC_HD void rig_pair_terms(const double qc[6], const double qk[6], double out[12]) {
  double Rc[9], Rk[9];
  rodrigues(qc, Rc);
  rodrigues(qk, Rk);
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j)
      out[i * 3 + j] = Rc[i * 3] * Rk[j * 3] + Rc[i * 3 + 1] * Rk[j * 3 + 1] + Rc[i * 3 + 2] * Rk[j * 3 + 2];
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    out[9 + i] = qc[3 + i] - (out[i * 3] * qk[3] + out[i * 3 + 1] * qk[4] + out[i * 3 + 2] * qk[5]);
}

// Board pose in the rig frame from camera c's view of it: R_p = R_c^T R_pc, t_p = R_c^T (t_pc - t_c).
CC_HD void rig_pose_from_cam(const double q[6], const double ext[6], double out[6]) {
  double Rc[9], Rq[9], R[9];
  rodrigues(ext, Rc);
  rodrigues(q, Rq);
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = Rc[i] * Rq[j] + Rc[3 + i] * Rq[3 + j] + Rc[6 + i] * Rq[6 + j];
  rot_to_rvec(R, out);
  const double d[3] = {q[3] - ext[3], q[4] - ext[4], q[5] - ext[5]};
  CC_UNROLL
  for (int i = 0; i < 3; ++i) out[3 + i] = Rc[i] * d[0] + Rc[3 + i] * d[1] + Rc[6 + i] * d[2];
}

// ---------------------------------------------------------------------------
// Host side of the initialisation (runs once per solve on 13 n_cams^2 numbers).
namespace righost {

// nearest rotation to M (row-major 3x3): U V^T of its SVD with the determinant fixed,
// by one-sided Jacobi (columns of M orthogonalised by plane rotations accumulated in V)
inline void nearest_rotation(const double M[9], double R[9]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = M[i];
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < 3; ++i) { a += A[i * 3 + p] * A[i * 3 + p]; b += A[i * 3 + q] * A[i * 3 + q]; c += A[i * 3 + p] * A[i * 3 + q]; }
        off += c * c;
        if (c * c <= 1e-300 || c * c <= 1e-32 * a * b) continue;
        const double zeta = (b - a) / (2.0 * c);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < 3; ++i) {
          const double x = A[i * 3 + p], y = A[i * 3 + q];
          A[i * 3 + p] = cs * x - sn * y; A[i * 3 + q] = sn * x + cs * y;
          const double vx = V[i * 3 + p], vy = V[i * 3 + q];
          V[i * 3 + p] = cs * vx - sn * vy; V[i * 3 + q] = sn * vx + cs * vy;
        }
      }
    if (off < 1e-300) break;
  }
  // A = U S (columns), singular values = column norms; smallest one last for the sign fix
  double sv[3], U[9];
  for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
  int order[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sv[order[b]] > sv[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  for (int jj = 0; jj < 3; ++jj) {
    const int j = order[jj];
    const double inv = sv[j] > 0 ? 1.0 / sv[j] : 0.0;
    for (int i = 0; i < 3; ++i) U[i * 3 + jj] = A[i * 3 + j] * inv;
  }
  double Vs[9];
  for (int jj = 0; jj < 3; ++jj)
    for (int i = 0; i < 3; ++i) Vs[i * 3 + jj] = V[i * 3 + order[jj]];
  if (!(sv[order[2]] > 0)) {        // rank 2: complete U with the cross product of its first two columns
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  }
  auto det3 = [](const double* X) {
    return X[0] * (X[4] * X[8] - X[5] * X[7]) - X[1] * (X[3] * X[8] - X[5] * X[6]) + X[2] * (X[3] * X[7] - X[4] * X[6]);
  };
  const double sgn = det3(U) * det3(Vs) < 0 ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = U[i * 3] * Vs[j * 3] + U[i * 3 + 1] * Vs[j * 3 + 1] + sgn * U[i * 3 + 2] * Vs[j * 3 + 2];
}

inline void rot_to_rvec_host(const double R[9], double r[3]) { rot_to_rvec(R, r); }

// sums [c][k][13] = sum over shared poses of (R_ck (9), t_ck (3), 1)  ->  ext [n_cams][6].
// Spanning tree from camera 0: repeatedly attach the camera outside the tree that shares
// the most poses with a camera inside it.  Returns false if some camera shares none.
inline bool ext_from_pair_sums(int n_cams, const double* sums, double* ext) {
  double Rc[CAMCAL_RIG_MAX_CAMS][9], tc[CAMCAL_RIG_MAX_CAMS][3];
  bool in[CAMCAL_RIG_MAX_CAMS] = {false, false, false, false};
  for (int i = 0; i < 9; ++i) Rc[0][i] = (i % 4 == 0) ? 1.0 : 0.0;
  tc[0][0] = tc[0][1] = tc[0][2] = 0.0;
  in[0] = true;
  for (int done = 1; done < n_cams; ++done) {
    double best = 0.0;
    int bc = -1, bk = -1;
    for (int c = 0; c < n_cams; ++c) {
      if (in[c]) continue;
      for (int k = 0; k < n_cams; ++k) {
        if (!in[k]) continue;
        const double n = sums[(c * n_cams + k) * 13 + 12];
        if (n > 0 && n > best) { best = n; bc = c; bk = k; }
      }
    }
    if (bc < 0) return false;
    const double* s = sums + (bc * n_cams + bk) * 13;
    double R[9];
    nearest_rotation(s, R);
    const double t[3] = {s[9] / best, s[10] / best, s[11] / best};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        Rc[bc][i * 3 + j] = R[i * 3] * Rc[bk][j] + R[i * 3 + 1] * Rc[bk][3 + j] + R[i * 3 + 2] * Rc[bk][6 + j];
      tc[bc][i] = R[i * 3] * tc[bk][0] + R[i * 3 + 1] * tc[bk][1] + R[i * 3 + 2] * tc[bk][2] + t[i];
    }
    in[bc] = true;
  }
  for (int c = 0; c < n_cams; ++c) {
    double r[3] = {0, 0, 0};
    if (c > 0) rot_to_rvec_host(Rc[c], r);
    for (int i = 0; i < 3; ++i) { ext[c * 6 + i] = c > 0 ? r[i] : 0.0; ext[c * 6 + 3 + i] = c > 0 ? tc[c][i] : 0.0; }
  }
  return true;
}

}  // namespace righost

}  // namespace camcal
