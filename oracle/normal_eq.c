/* C restatement of one normal-equation evaluation of the calibration solvers
 * (oracle; TEST INFRASTRUCTURE + bench.py's cpu_baseline "port" leg ONLY).
 *
 * PARITY UNPINNED: OpenCV (opencv-contrib-python==4.10.0.84, reference
 * requirements.txt:11) is not available offline; this follows the published
 * algorithm as recalled in SURVEY.md 8(c)-appendix and is pinned against the
 * numpy oracle (oracle/cvcalib.py::normal_blocks / _schur_step) in
 * tests/test_oracle_c.py.
 *
 * What it restates, for the calls the reference makes at
 * server/aruco/camera_calib.py:74-77 (fisheye) and :153-156 (pinhole):
 *   - cvRodrigues2 incl. the 3x9 derivative
 *   - cvProjectPoints2Internal / cv::fisheye::projectPoints with Jacobians
 *   - the per-view accumulation of cvCalibrateCamera2Internal /
 *     internal::ComputeJacobians (JtJ blocks, JtErr, errNorm), written the way
 *     cv2 does it: dense (NI+6)-wide Jacobian rows, no sparsity tricks
 *   - elimination of the 6x6 pose blocks (the O(N) form of cv2's dense solve)
 *
 * Layouts: obj [n_obs][3] f32, img [n_obs][2] f32, view_ptr CSR, poses [N][6],
 * V [N][6][6], W [N][NI][6], g [N][6], U [NI][NI], b [NI], C [NI][NI], d [NI].
 * Threads over views with OpenMP when built with -fopenmp (nthreads > 1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXP 15

static void rodrigues(const double r[3], double R[9], double J[27]) {
  static const double G[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 2.220446049250313e-16) {
    memcpy(R, I, sizeof I);
    memcpy(J, G, sizeof G);
    return;
  }
  double c = cos(th), s = sin(th), c1 = 1 - c, it = 1 / th;
  double n[3] = {r[0] * it, r[1] * it, r[2] * it};
  double rrt[9], rx[9] = {0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rrt[i * 3 + j] = n[i] * n[j];
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
  double drrt[27] = {n[0] + n[0], n[1], n[2], n[1], 0, 0, n[2], 0, 0,
                     0, n[0], 0, n[0], n[1] + n[1], n[2], 0, n[2], 0,
                     0, 0, n[0], 0, 0, n[1], n[0], n[1], n[2] + n[2]};
  for (int i = 0; i < 3; i++) {
    double ri = n[i];
    double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
    for (int k = 0; k < 9; k++)
      J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] + a4 * G[i * 9 + k];
  }
}

/* rows ju, jv of the Jacobian wrt [intr(NI) | r(3) | t(3)] and the projection */
static void project(int model, const double* in, const double R[9], const double dR[27], const double t[3],
                    const float X[3], double p[2], double ju[MAXP], double jv[MAXP]) {
  const int NI = model == 0 ? 9 : 8;
  double Xd[3] = {X[0], X[1], X[2]};
  double Y[3], dY[3][6];
  for (int i = 0; i < 3; i++) {
    Y[i] = R[i * 3] * Xd[0] + R[i * 3 + 1] * Xd[1] + R[i * 3 + 2] * Xd[2] + t[i];
    for (int j = 0; j < 3; j++) {
      const double* d = dR + j * 9 + i * 3;
      dY[i][j] = d[0] * Xd[0] + d[1] * Xd[1] + d[2] * Xd[2];
      dY[i][3 + j] = i == j ? 1.0 : 0.0;
    }
  }
  double fx = in[0], fy = in[1], cx = in[2], cy = in[3];
  double gu[3], gv[3];
  for (int k = 0; k < MAXP; k++) ju[k] = jv[k] = 0.0;
  if (model == 0) {
    double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    double z = Y[2] != 0 ? 1 / Y[2] : 1, x = Y[0] * z, y = Y[1] * z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cd = 1 + k1 * r2 + k2 * r4 + k3 * r6;
    double xd = x * cd + p1 * a1 + p2 * a2, yd = y * cd + p1 * a3 + p2 * a1;
    p[0] = xd * fx + cx; p[1] = yd * fy + cy;
    ju[0] = xd; jv[1] = yd; ju[2] = 1; jv[3] = 1;
    ju[4] = fx * x * r2; jv[4] = fy * y * r2; ju[5] = fx * x * r4; jv[5] = fy * y * r4;
    ju[6] = fx * a1; jv[6] = fy * a3; ju[7] = fx * a2; jv[7] = fy * a1;
    ju[8] = fx * x * r6; jv[8] = fy * y * r6;
    double g = k1 + 2 * k2 * r2 + 3 * k3 * r4;
    double dxdx = cd + 2 * x * x * g + 2 * p1 * y + 6 * p2 * x, dxdy = 2 * x * y * g + 2 * p1 * x + 2 * p2 * y;
    double dydy = cd + 2 * y * y * g + 6 * p1 * y + 2 * p2 * x;
    gu[0] = fx * z * dxdx; gu[1] = fx * z * dxdy; gu[2] = -fx * z * (x * dxdx + y * dxdy);
    gv[0] = fy * z * dxdy; gv[1] = fy * z * dydy; gv[2] = -fy * z * (x * dxdy + y * dydy);
  } else {
    double k1 = in[4], k2 = in[5], k3 = in[6], k4 = in[7];
    double iz = 1 / Y[2], x = Y[0] * iz, y = Y[1] * iz, r2 = x * x + y * y, r = sqrt(r2);
    int big = r > 1e-8;
    double ir = big ? 1 / r : 1, th = atan(r), t2 = th * th, t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    double thd = th + k1 * t3 + k2 * t5 + k3 * t7 + k4 * t9;
    double cdist = big ? thd * ir : 1, xd = x * cdist, yd = y * cdist;
    p[0] = xd * fx + cx; p[1] = yd * fy + cy;
    ju[0] = xd; jv[1] = yd; ju[2] = 1; jv[3] = 1;
    double tp[4] = {t3, t5, t7, t9};
    for (int j = 0; j < 4; j++) { ju[4 + j] = fx * x * tp[j] * ir; jv[4 + j] = fy * y * tp[j] * ir; }
    double dthd = 1 + 3 * k1 * t2 + 5 * k2 * t2 * t2 + 7 * k3 * t2 * t2 * t2 + 9 * k4 * t2 * t2 * t2 * t2;
    double q = big ? (dthd / (1 + r2) - cdist) * ir * ir : 0;
    double a = cdist + x * x * q, bq = x * y * q, c = cdist + y * y * q;
    gu[0] = fx * iz * a; gu[1] = fx * iz * bq; gu[2] = -fx * iz * (x * a + y * bq);
    gv[0] = fy * iz * bq; gv[1] = fy * iz * c; gv[2] = -fy * iz * (x * bq + y * c);
  }
  for (int j = 0; j < 6; j++) {
    ju[NI + j] = gu[0] * dY[0][j] + gu[1] * dY[1][j] + gu[2] * dY[2][j];
    jv[NI + j] = gv[0] * dY[0][j] + gv[1] * dY[1][j] + gv[2] * dY[2][j];
  }
}

/* solve the 6x6 SPD system A X = B (B has m columns, row-major 6 x m) by Cholesky */
static int chol_solve6(double A[36], double* B, int m) {
  double L[36] = {0};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) { if (!(s > 0)) return 1; L[i * 6 + i] = sqrt(s); }
      else L[i * 6 + j] = s / L[j * 6 + j];
    }
  for (int c = 0; c < m; c++) {
    for (int i = 0; i < 6; i++) {
      double s = B[i * m + c];
      for (int k = 0; k < i; k++) s -= L[i * 6 + k] * B[k * m + c];
      B[i * m + c] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
      double s = B[i * m + c];
      for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * B[k * m + c];
      B[i * m + c] = s / L[i * 6 + i];
    }
  }
  return 0;
}

int oracle_normal_eq(int model, int64_t n_views, const int64_t* view_ptr, const float* obj, const float* img,
                     const double* intr, const double* poses, double lam, double* U, double* b, double* cost,
                     double* V, double* W, double* g, double* C, double* d, int nthreads) {
  if (model != 0 && model != 1) return -2;
  const int NI = model == 0 ? 9 : 8, P = NI + 6;
  if (nthreads < 1) nthreads = 1;
  const int NACC = NI * NI + NI + 1 + NI * NI + NI;
  double* tacc = (double*)calloc((size_t)nthreads * NACC, sizeof(double));
  if (!tacc) return -3;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int64_t v = 0; v < n_views; v++) {
#ifdef _OPENMP
    double* acc = tacc + (size_t)omp_get_thread_num() * NACC;
#else
    double* acc = tacc;
#endif
    double* aU = acc, *ab = aU + NI * NI, *ac = ab + NI, *aC = ac + 1, *ad = aC + NI * NI;
    double R[9], dR[27], JtJ[MAXP * MAXP] = {0}, Jte[MAXP] = {0};
    const double* ps = poses + v * 6;
    rodrigues(ps, R, dR);
    for (int64_t k = view_ptr[v]; k < view_ptr[v + 1]; k++) {
      double p[2], ju[MAXP], jv[MAXP];
      project(model, intr, R, dR, ps + 3, obj + 3 * k, p, ju, jv);
      double eu = p[0] - img[2 * k], ev = p[1] - img[2 * k + 1];
      for (int i = 0; i < P; i++) {
        for (int j = i; j < P; j++) JtJ[i * MAXP + j] += ju[i] * ju[j] + jv[i] * jv[j];
        Jte[i] += ju[i] * eu + jv[i] * ev;
      }
      *ac += eu * eu + ev * ev;
    }
    for (int i = 0; i < P; i++)
      for (int j = 0; j < i; j++) JtJ[i * MAXP + j] = JtJ[j * MAXP + i];
    double Vd[36], Y[6 * (MAXP + 1)];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) {
        double x = JtJ[(NI + i) * MAXP + NI + j];
        if (V) V[v * 36 + i * 6 + j] = x;
        Vd[i * 6 + j] = i == j ? x * (1 + lam) : x;
      }
      if (g) g[v * 6 + i] = Jte[NI + i];
      for (int a = 0; a < NI; a++) Y[i * (NI + 1) + a] = JtJ[a * MAXP + NI + i];
      Y[i * (NI + 1) + NI] = Jte[NI + i];
    }
    for (int a = 0; a < NI; a++) {
      for (int c = 0; c < NI; c++) aU[a * NI + c] += JtJ[a * MAXP + c];
      ab[a] += Jte[a];
      if (W) for (int j = 0; j < 6; j++) W[(v * NI + a) * 6 + j] = JtJ[a * MAXP + NI + j];
    }
    if (chol_solve6(Vd, Y, NI + 1) == 0) {
      for (int a = 0; a < NI; a++) {
        for (int c = 0; c < NI; c++) {
          double s = 0;
          for (int j = 0; j < 6; j++) s += JtJ[a * MAXP + NI + j] * Y[j * (NI + 1) + c];
          aC[a * NI + c] += s;
        }
        double s = 0;
        for (int j = 0; j < 6; j++) s += JtJ[a * MAXP + NI + j] * Y[j * (NI + 1) + NI];
        ad[a] += s;
      }
    }
  }
  memset(U, 0, sizeof(double) * NI * NI); memset(b, 0, sizeof(double) * NI);
  memset(C, 0, sizeof(double) * NI * NI); memset(d, 0, sizeof(double) * NI);
  *cost = 0;
  for (int t = 0; t < nthreads; t++) {
    double* acc = tacc + (size_t)t * NACC;
    for (int i = 0; i < NI * NI; i++) { U[i] += acc[i]; C[i] += acc[NI * NI + NI + 1 + i]; }
    for (int i = 0; i < NI; i++) { b[i] += acc[NI * NI + i]; d[i] += acc[NI * NI + NI + 1 + NI * NI + i]; }
    *cost += acc[NI * NI + NI];
  }
  free(tacc);
  return 0;
}
