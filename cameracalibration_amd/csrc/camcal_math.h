// Lane-level math of the calibration hot path (fp64), shared by the HIP kernels
// (camcal_kernels.hip) and by the CPU lane emulator that only the tests build
// (tests/hostemu/).  One "lane" owns one view: it walks that view's corner
// observations and keeps every accumulator in registers, so no cross-lane
// traffic is needed until the final block reduction.
//
// Replaces the arithmetic that the reference delegates to cv2.calibrateCamera
// (server/aruco/camera_calib.py:153-156) and cv2.fisheye.calibrate
// (server/aruco/camera_calib.py:74-77); model definitions follow OpenCV 4.10
// cvProjectPoints2Internal / cv::fisheye::projectPoints as recalled in
// SURVEY.md 8(c)-appendix.
//
// Conventions
//   residual        e = projected - observed
//   pinhole intr    [fx fy cx cy k1 k2 p1 p2 k3]     NI = 9
//   fisheye intr    [fx fy cx cy k1 k2 k3 k4]        NI = 8  (skew fixed at 0)
//   pose            [r0 r1 r2 t0 t1 t2]  (Rodrigues vector, translation)
//   packed points   tile = 64 consecutive slots (views); float at
//                   tile_off + (k*5 + c)*64 + lane,  c in {X,Y,Z,u,v}
//   per-view blocks V (6x6 upper, 21), W (NI x 6 row-major), g (6), stored
//                   component-major: block[e * n_slots + slot]
//
// The pose columns are accumulated in the local tangent frame
// (dY/dw = -[R X]x, dY/dt = I) and mapped once per view to Rodrigues
// coordinates with the SO(3) left Jacobian Jl(r):  d(RX)/dr = -[R X]x Jl(r),
// which equals cv2's (dR/dr) X chain exactly (checked in tests against
// cvRodrigues2's 3x9 derivative).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define CC_HD __host__ __device__ __forceinline__
#define CC_UNROLL _Pragma("unroll")
#else
#define CC_HD inline
#define CC_UNROLL
#endif

namespace camcal {

constexpr int PINHOLE = 0;
constexpr int FISHEYE = 1;
constexpr int LANES = 64;      // wavefront width == views per tile
constexpr int PT_COMP = 5;     // X Y Z u v

template <int MODEL> struct MT;
template <> struct MT<PINHOLE> {
  static constexpr int NI = 9;   // free-able intrinsics
  static constexpr int NA = 7;   // non-zero intrinsic columns per image row
  // columns touched by the u row / the v row; entry 1 is the unit cx / cy column
  static constexpr int ucol(int a) { return a == 0 ? 0 : a == 1 ? 2 : a + 2; }
  static constexpr int vcol(int a) { return a == 0 ? 1 : a == 1 ? 3 : a + 2; }
};
template <> struct MT<FISHEYE> {
  static constexpr int NI = 8;
  static constexpr int NA = 6;
  static constexpr int ucol(int a) { return a == 0 ? 0 : a == 1 ? 2 : a + 2; }
  static constexpr int vcol(int a) { return a == 0 ? 1 : a == 1 ? 3 : a + 2; }
};

// upper-triangle index of an n x n symmetric matrix, i <= j
CC_HD constexpr int tri(int n, int i, int j) { return i * n - (i * (i - 1)) / 2 + (j - i); }
CC_HD constexpr int trisz(int n) { return n * (n + 1) / 2; }

// ---------------------------------------------------------------------------
// Reciprocal and reciprocal square root for the per-observation path.  On the
// device: the hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-26) plus two Newton
// steps, i.e. full double accuracy (<= 1-2 ulp) in 5-7 instructions instead of
// the ~12/~22 of the IEEE division / sqrt expansions.  Arguments here are
// finite, normal and positive (depths, radii), so the special-case handling of
// the IEEE expansions is not needed.  The CPU emulator uses the plain forms.
CC_HD double fast_rcp(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(a);
  double e = fma(-a, y, 1.0);
  y = fma(y, e, y);
  e = fma(-a, y, 1.0);
  return fma(y, e, y);
#else
  return 1.0 / a;
#endif
}
// returns 1/sqrt(a) and sqrt(a) (through rt); a > 0
CC_HD double fast_rsqrt(double a, double& rt) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  double r = a * y;
  r = fma(fma(-r, r, a), 0.5 * y, r);
  rt = r;
  return y;
#else
  rt = sqrt(a);
  return 1.0 / rt;
#endif
}

// atan(x) for x >= 0: reduction to |t| <= tan(pi/8) (t = x, (x-1)/(x+1) or -1/x)
// and an 11-term odd polynomial (Chebyshev-node fit, max relative error
// 5e-18 before rounding; coefficients from mpmath, tools/atan_coeffs.py).
// Fewer constants than the 20-term libm form, which matters for registers.
// On the device the coefficients are read from a (non-const) table: the loads are
// uniform and loop invariant, so they become scalar loads hoisted out of the corner
// loop and the coefficients stay in SGPRs, where an FMA can read them directly.  As
// literals they occupy 20 VGPRs for the whole loop plus a copy per Horner step.
#define CAMCAL_ATAN_COEFFS                                                                            \
  -0.019176887119062258989, 0.039231658295587191303, -0.050854497379402598613,                      \
  0.058581489128022098816, -0.066645114473819480001, 0.076921831908260865637,                       \
  -0.090909045781239018905, 0.11111111015256361727, -0.14285714284666542922,                        \
  0.19999999999995520678, -0.3333333333333333016
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ double camcal_atan_c[11] = {CAMCAL_ATAN_COEFFS};
#else
static const double camcal_atan_c[11] = {CAMCAL_ATAN_COEFFS};
#endif

CC_HD double atan_pos(double x) {
  // t = x, (x-1)/(x+1), -1/x on the three ranges, written t = (a x + b) / (c x + a) so that
  // only the constants 0 / +-1 are selected (one conditional move each); the offset is
  // (c + 1 - a) pi/4 = 0, pi/4, pi/2, an exact product
  const bool hi = x > 2.414213562373095, mid = x > 0.41421356237309503;
  const double a = hi ? 0.0 : 1.0, b = mid ? -1.0 : 0.0, c = mid ? 1.0 : 0.0;
  const double t = fma(x, a, b) * fast_rcp(fma(x, c, a));
  const double z = t * t;
  double p = camcal_atan_c[0];
  CC_UNROLL
  for (int i = 1; i < 11; ++i) p = fma(p, z, camcal_atan_c[i]);
  return fma(c + (1.0 - a), 0.78539816339744831, fma(t * z, p, t));
}

// tan(x) for the angles of the fisheye undistortion (0 <= x <~ pi/2): sin and cos of
// y = x or pi/2 - x (|y| <= pi/4) from their Taylor series to y^19 / y^20 (truncation
// < 2e-22), one reciprocal.  About 30 instructions instead of libm's ~150 with its
// large-argument reduction; anything outside the range, or within 1e-8 of the pole,
// goes to libm.
CC_HD double tan_acute(double x) {
  if (!(x >= 0.0 && x <= 2.35)) return tan(x);
  const bool big = x > 0.78539816339744831;
  const double y = big ? (1.5707963267948966 - x) + 6.123233995736766e-17 : x;
  if (big && fabs(y) < 1e-8) return tan(x);
  const double z = y * y;
  double sp = -8.2206352466243297e-18;        // -1/19!
  sp = fma(sp, z, 2.8114572543455208e-15);    //  1/17!
  sp = fma(sp, z, -7.6471637318198165e-13);   // -1/15!
  sp = fma(sp, z, 1.6059043836821615e-10);    //  1/13!
  sp = fma(sp, z, -2.5052108385441719e-08);   // -1/11!
  sp = fma(sp, z, 2.7557319223985891e-06);    //  1/9!
  sp = fma(sp, z, -1.9841269841269841e-04);   // -1/7!
  sp = fma(sp, z, 8.3333333333333333e-03);    //  1/5!
  sp = fma(sp, z, -1.6666666666666667e-01);   // -1/3!
  const double sn = fma(y * z, sp, y);
  double cp = 4.1103176233121649e-19;         //  1/20!
  cp = fma(cp, z, -1.5619206968586226e-16);   // -1/18!
  cp = fma(cp, z, 4.7794773323873853e-14);    //  1/16!
  cp = fma(cp, z, -1.1470745597729725e-11);   // -1/14!
  cp = fma(cp, z, 2.0876756987868099e-09);    //  1/12!
  cp = fma(cp, z, -2.7557319223985891e-07);   // -1/10!
  cp = fma(cp, z, 2.4801587301587302e-05);    //  1/8!
  cp = fma(cp, z, -1.3888888888888889e-03);   // -1/6!
  cp = fma(cp, z, 4.1666666666666667e-02);    //  1/4!
  cp = fma(cp, z, -0.5);
  const double cs = fma(cp, z, 1.0);
  return big ? cs * fast_rcp(sn) : sn * fast_rcp(cs);
}

// ---------------------------------------------------------------------------
// The three SO(3) series of a rotation vector with theta^2 = t:
//   A = sin(theta)/theta, B = (1 - cos theta)/theta^2, C = (theta - sin theta)/theta^3.
// Below theta = 0.5 their Taylor series; above, sin and cos of theta/2 from the Taylor
// series on |y| <= pi/4 (as tan_acute), sin theta = 2 sh ch and 1 - cos theta = 2 sh^2
// (no cancellation), so one reduction serves the rotation matrix and the left Jacobian of
// the same vector (the compiler merges the two evaluations).  Rotation vectors longer than
// 4.7 (1.5 pi; none occurs for a board in front of a camera) take the libm route.
CC_HD void so3_series(double t, double& A, double& B, double& C) {
  if (t < 0.25) {
    A = 1.0 + t * (-1.0 / 6 + t * (1.0 / 120 + t * (-1.0 / 5040 + t * (1.0 / 362880 + t * (-1.0 / 39916800 +
        t * (1.0 / 6227020800.0))))));
    B = 0.5 + t * (-1.0 / 24 + t * (1.0 / 720 + t * (-1.0 / 40320 + t * (1.0 / 3628800 +
        t * (-1.0 / 479001600 + t * (1.0 / 87178291200.0))))));
    C = 1.0 / 6 + t * (-1.0 / 120 + t * (1.0 / 5040 + t * (-1.0 / 362880 + t * (1.0 / 39916800 +
        t * (-1.0 / 6227020800.0 + t * (1.0 / 1307674368000.0))))));
    return;
  }
  double th;
  const double ith = fast_rsqrt(t, th);          // 1/theta, theta
  double sn, omc;                                // sin theta, 1 - cos theta
  const double h = 0.5 * th;
  if (h <= 2.35) {
    const bool big = h > 0.78539816339744831;
    const double y = big ? (1.5707963267948966 - h) + 6.123233995736766e-17 : h;
    const double z = y * y;
    double sp = -8.2206352466243297e-18;
    sp = fma(sp, z, 2.8114572543455208e-15);
    sp = fma(sp, z, -7.6471637318198165e-13);
    sp = fma(sp, z, 1.6059043836821615e-10);
    sp = fma(sp, z, -2.5052108385441719e-08);
    sp = fma(sp, z, 2.7557319223985891e-06);
    sp = fma(sp, z, -1.9841269841269841e-04);
    sp = fma(sp, z, 8.3333333333333333e-03);
    sp = fma(sp, z, -1.6666666666666667e-01);
    const double sy = fma(y * z, sp, y);
    double cp = 4.1103176233121649e-19;
    cp = fma(cp, z, -1.5619206968586226e-16);
    cp = fma(cp, z, 4.7794773323873853e-14);
    cp = fma(cp, z, -1.1470745597729725e-11);
    cp = fma(cp, z, 2.0876756987868099e-09);
    cp = fma(cp, z, -2.7557319223985891e-07);
    cp = fma(cp, z, 2.4801587301587302e-05);
    cp = fma(cp, z, -1.3888888888888889e-03);
    cp = fma(cp, z, 4.1666666666666667e-02);
    cp = fma(cp, z, -0.5);
    const double cy = fma(cp, z, 1.0);
    const double sh = big ? cy : sy, ch = big ? sy : cy;     // sin, cos of theta / 2
    sn = 2.0 * sh * ch;
    omc = 2.0 * sh * sh;
  } else {
    sn = sin(th);
    const double sh = sin(h);
    omc = 2.0 * sh * sh;
  }
  const double it2 = ith * ith;
  A = sn * ith;
  B = omc * it2;
  C = (th - sn) * it2 * ith;
}

// Rodrigues vector -> rotation matrix (row-major), as cvRodrigues2:
// R = cos I + (1 - cos) n n^T + sin [n]x = I + A [r]x + B [r]x^2.
CC_HD void rodrigues(const double r[3], double R[9]) {
  const double x = r[0], y = r[1], z = r[2];
  const double th2 = x * x + y * y + z * z;
  double A, B, C;
  so3_series(th2, A, B, C);
  // [r]x^2 = r r^T - th2 I
  R[0] = 1.0 + B * (x * x - th2); R[1] = -A * z + B * x * y;        R[2] = A * y + B * x * z;
  R[3] = A * z + B * x * y;       R[4] = 1.0 + B * (y * y - th2);   R[5] = -A * x + B * y * z;
  R[6] = -A * y + B * x * z;      R[7] = A * x + B * y * z;         R[8] = 1.0 + B * (z * z - th2);
}

// The Rodrigues vector of the same rotation with |r| <= pi.  cv2 re-derives every pose from
// a rotation MATRIX at each outer iteration of fisheye::calibrate (InitExtrinsics), so its
// vectors start every refinement inside that ball; a refinement that is warm-started from
// the previous vector instead can drift across |r| = pi and would hand back the other
// representation (r' = r (1 - 2 pi / |r|)) of an identical rotation.
CC_HD void canonical_rvec(double r[3]) {
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double pi = 3.141592653589793;
  if (th2 > pi * pi) {
    const double th = sqrt(th2);
    const double k = th - 6.283185307179586 * floor((th + pi) * (1.0 / 6.283185307179586));   // angle in (-pi, pi]
    const double f = k / th;
    r[0] *= f; r[1] *= f; r[2] *= f;
  }
}

// SO(3) left Jacobian Jl(r) = I + B [r]x + C [r]x^2 (row-major).
CC_HD void left_jacobian(const double r[3], double J[9]) {
  const double x = r[0], y = r[1], z = r[2];
  const double th2 = x * x + y * y + z * z;
  double A, B, C;
  so3_series(th2, A, B, C);
  J[0] = 1.0 + C * (x * x - th2); J[1] = -B * z + C * x * y;        J[2] = B * y + C * x * z;
  J[3] = B * z + C * x * y;       J[4] = 1.0 + C * (y * y - th2);   J[5] = -B * x + C * y * z;
  J[6] = -B * y + C * x * z;      J[7] = B * x + C * y * z;         J[8] = 1.0 + C * (z * z - th2);
}

// ---------------------------------------------------------------------------
// Per-observation evaluation in the camera frame.
template <int MODEL> struct PointEval {
  double pu, pv;             // projected pixel
  double gu[3], gv[3];       // d pu / dY, d pv / dY   (Y = camera-frame point)
  double au[MT<MODEL>::NA];  // intrinsic columns of the u row (au[1] == 1: cx)
  double av[MT<MODEL>::NA];  // intrinsic columns of the v row (av[1] == 1: cy)
  double th2;                // fisheye only: theta^2 (au[2+j] = au[2] * th2^j)
};

// pinhole, 5 coefficients: intr = fx fy cx cy k1 k2 p1 p2 k3
template <bool JAC>
CC_HD void eval_point(const double Yc[3], const double* intr, PointEval<PINHOLE>& o) {
  const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
  const double z = Yc[2] != 0.0 ? fast_rcp(Yc[2]) : 1.0;
  const double x = Yc[0] * z, y = Yc[1] * z;
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
  const double cdist = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  const double xd = x * cdist + p1 * a1 + p2 * a2;
  const double yd = y * cdist + p1 * a3 + p2 * a1;
  o.pu = xd * fx + cx;
  o.pv = yd * fy + cy;
  if (JAC) {
    o.au[0] = xd; o.au[1] = 1.0;
    o.av[0] = yd; o.av[1] = 1.0;
    const double fxx = fx * x, fyy = fy * y;
    o.au[2] = fxx * r2; o.av[2] = fyy * r2;
    o.au[3] = fxx * r4; o.av[3] = fyy * r4;
    o.au[4] = fx * a1;  o.av[4] = fy * a3;
    o.au[5] = fx * a2;  o.av[5] = fy * a1;
    o.au[6] = fxx * r6; o.av[6] = fyy * r6;
    const double g = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    const double dxdx = cdist + 2.0 * x * x * g + 2.0 * p1 * y + 6.0 * p2 * x;
    const double dxdy = 2.0 * x * y * g + 2.0 * p1 * x + 2.0 * p2 * y;
    const double dydy = cdist + 2.0 * y * y * g + 6.0 * p1 * y + 2.0 * p2 * x;
    const double fxz = fx * z, fyz = fy * z;
    o.gu[0] = fxz * dxdx; o.gu[1] = fxz * dxdy; o.gu[2] = -fxz * (x * dxdx + y * dxdy);
    o.gv[0] = fyz * dxdy; o.gv[1] = fyz * dydy; o.gv[2] = -fyz * (x * dxdy + y * dydy);
  }
}

// Kannala-Brandt fisheye: intr = fx fy cx cy k1 k2 k3 k4 (skew 0)
template <bool JAC>
CC_HD void eval_point(const double Yc[3], const double* intr, PointEval<FISHEYE>& o) {
  const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const double k1 = intr[4], k2 = intr[5], k3 = intr[6], k4 = intr[7];
  const double iz = fast_rcp(Yc[2]);
  const double x = Yc[0] * iz, y = Yc[1] * iz;
  const double r2 = x * x + y * y;
  // cv2 switches to cdist = 1 for r <= 1e-8.  Here such a point is evaluated by the general
  // formulas at r = 1e-8 instead (one v_max instead of a compare and fourteen selects per
  // corner): theta_d / r is then 1 to rounding, and the terms that are inaccurate at that
  // radius (q) only enter multiplied by x^2, x y, y^2 <= 1e-16 -- the result differs from
  // cv2's special case by less than 1e-15 relative.
  const double r2c = fmax(r2, 1e-16);
  double rr;
  const double inv_r = fast_rsqrt(r2c, rr);
  const double th = atan_pos(rr);
  const double th2 = th * th;
  const double poly = 1.0 + th2 * (k1 + th2 * (k2 + th2 * (k3 + th2 * k4)));
  const double cdist = th * poly * inv_r;
  const double xd = x * cdist, yd = y * cdist;
  o.pu = xd * fx + cx;
  o.pv = yd * fy + cy;
  if (JAC) {
    o.au[0] = xd; o.au[1] = 1.0;
    o.av[0] = yd; o.av[1] = 1.0;
    o.th2 = th2;
    double tp = th * th2 * inv_r;       // theta^3 / r
    const double fxx = fx * x, fyy = fy * y;
    o.au[2] = fxx * tp; o.av[2] = fyy * tp; tp *= th2;
    o.au[3] = fxx * tp; o.av[3] = fyy * tp; tp *= th2;
    o.au[4] = fxx * tp; o.av[4] = fyy * tp; tp *= th2;
    o.au[5] = fxx * tp; o.av[5] = fyy * tp;
    const double dpoly = 1.0 + th2 * (3.0 * k1 + th2 * (5.0 * k2 + th2 * (7.0 * k3 + th2 * 9.0 * k4)));
    // s(r) = theta_d / r ;  q = (ds/dr)/r
    const double q = (dpoly * fast_rcp(1.0 + r2) - cdist) * inv_r * inv_r;
    const double a = cdist + x * x * q, b = x * y * q, c = cdist + y * y * q;
    const double fxz = fx * iz, fyz = fy * iz;
    o.gu[0] = fxz * a; o.gu[1] = fxz * b; o.gu[2] = -fxz * (x * a + y * b);
    o.gv[0] = fyz * b; o.gv[1] = fyz * c; o.gv[2] = -fyz * (x * b + y * c);
  }
}

// camera-frame point and its rotated part (PLANAR: board points with Z == 0)
template <bool PLANAR>
CC_HD void transform_point(const double R[9], const double t[3], float X, float Y, float Z,
                           double Yr[3], double Yc[3]) {
  const double Xd = X, Yd = Y;
  if (PLANAR) {
    Yr[0] = R[0] * Xd + R[1] * Yd;
    Yr[1] = R[3] * Xd + R[4] * Yd;
    Yr[2] = R[6] * Xd + R[7] * Yd;
  } else {
    const double Zd = Z;
    Yr[0] = R[0] * Xd + R[1] * Yd + R[2] * Zd;
    Yr[1] = R[3] * Xd + R[4] * Yd + R[5] * Zd;
    Yr[2] = R[6] * Xd + R[7] * Yd + R[8] * Zd;
  }
  Yc[0] = Yr[0] + t[0]; Yc[1] = Yr[1] + t[1]; Yc[2] = Yr[2] + t[2];
}

// local-frame pose row j = [Yr x g, g]
CC_HD void pose_row(const double Yr[3], const double g[3], double j[6]) {
  j[0] = Yr[1] * g[2] - Yr[2] * g[1];
  j[1] = Yr[2] * g[0] - Yr[0] * g[2];
  j[2] = Yr[0] * g[1] - Yr[1] * g[0];
  j[3] = g[0]; j[4] = g[1]; j[5] = g[2];
}

// ---------------------------------------------------------------------------
// Accumulator sets
template <int MODEL> struct ViewAcc {   // per-view blocks in the local pose frame
  static constexpr int NI = MT<MODEL>::NI;
  double V[21];
  double W[NI * 6];
  double g[6];
  CC_HD void zero() {
    CC_UNROLL
    for (int i = 0; i < 21; ++i) V[i] = 0.0;
    CC_UNROLL
    for (int i = 0; i < NI * 6; ++i) W[i] = 0.0;
    CC_UNROLL
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
  }
};

// Global intrinsic normal block.  NV = length of the record [U | b | cost];
// NR = number of values a lane actually accumulates (and the wave reduces);
// expand() maps the reduced values to the record.
template <int MODEL> struct IntrAcc {
  static constexpr int NI = MT<MODEL>::NI;
  static constexpr int NU = trisz(NI);
  static constexpr int NV = NU + NI + 1;   // U | b | cost
  static constexpr int NR = NV;
  static constexpr int COST = NV - 1;      // index of the cost inside v[]
  double v[NR];
  CC_HD void zero() { for (int i = 0; i < NR; ++i) v[i] = 0.0; }
  CC_HD void add_count(int) {}
  CC_HD static void expand(const double* r, double* out) {
    for (int i = 0; i < NV; ++i) out[i] = r[i];
  }
};

// Kannala-Brandt: the intrinsic rows are (xd, 1, m t^0..t^3) and (yd, 1, n t^0..t^3)
// with t = theta^2, so U has Hankel structure in the k block and needs only 27
// distinct sums (not 36); fx/fy and cx/cy never mix.
//   v[0..6]   H_p  = sum (m^2 + n^2) t^p          -> U(k_i, k_j) = H_{i+j}
//   v[7..10]  sum xd^2, sum xd, sum yd^2, sum yd  -> U(fx,fx) U(fx,cx) U(fy,fy) U(fy,cy)
//   v[11..14] sum xd m t^j  -> U(fx,k_j)    v[15..18] sum m t^j -> U(cx,k_j)
//   v[19..22] sum yd n t^j  -> U(fy,k_j)    v[23..26] sum n t^j -> U(cy,k_j)
//   v[27]     number of observations        -> U(cx,cx) = U(cy,cy)
//   v[28..35] b (fx fy cx cy k1..k4), v[36] cost
template <> struct IntrAcc<FISHEYE> {
  static constexpr int NI = 8;
  static constexpr int NU = 36;
  static constexpr int NV = 45;
  static constexpr int NR = 37;
  static constexpr int COST = 36;
  double v[NR];
  CC_HD void zero() { for (int i = 0; i < NR; ++i) v[i] = 0.0; }
  CC_HD void add_count(int n) { v[27] += (double)n; }
  CC_HD static void expand(const double* r, double* out) {
    for (int i = 0; i < NV; ++i) out[i] = 0.0;
    out[tri(8, 0, 0)] = r[7];  out[tri(8, 0, 2)] = r[8];
    out[tri(8, 1, 1)] = r[9];  out[tri(8, 1, 3)] = r[10];
    out[tri(8, 2, 2)] = r[27]; out[tri(8, 3, 3)] = r[27];
    for (int j = 0; j < 4; ++j) {
      out[tri(8, 0, 4 + j)] = r[11 + j];
      out[tri(8, 2, 4 + j)] = r[15 + j];
      out[tri(8, 1, 4 + j)] = r[19 + j];
      out[tri(8, 3, 4 + j)] = r[23 + j];
      for (int i = 0; i <= j; ++i) out[tri(8, 4 + i, 4 + j)] = r[i + j];
    }
    for (int i = 0; i < 8; ++i) out[36 + i] = r[28 + i];
    out[44] = r[36];
  }
};

// pose block V (6x6 upper) and gradient g from the two local-frame pose rows
CC_HD void accumulate_vg(double V[21], double g[6], const double ju[6], const double jv[6], double eu, double ev) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    CC_UNROLL
    for (int j = i; j < 6; ++j) {
      V[tri(6, i, j)] = fma(ju[i], ju[j], V[tri(6, i, j)]);
      V[tri(6, i, j)] = fma(jv[i], jv[j], V[tri(6, i, j)]);
    }
    g[i] = fma(eu, ju[i], g[i]);
    g[i] = fma(ev, jv[i], g[i]);
  }
}

// intrinsic x pose cross block W (NI x 6)
template <int MODEL>
CC_HD void accumulate_w(double* W, const PointEval<MODEL>& p, const double ju[6], const double jv[6]) {
  using T = MT<MODEL>;
  CC_UNROLL
  for (int a = 0; a < T::NA; ++a) {
    CC_UNROLL
    for (int c = 0; c < 6; ++c) {
      W[T::ucol(a) * 6 + c] = fma(p.au[a], ju[c], W[T::ucol(a) * 6 + c]);
      W[T::vcol(a) * 6 + c] = fma(p.av[a], jv[c], W[T::vcol(a) * 6 + c]);
    }
  }
}
// Kannala-Brandt: the k rows share the 6-vector m6 = m ju + n jv, scaled by t^j
CC_HD void accumulate_w(double* W, const PointEval<FISHEYE>& p, const double ju[6], const double jv[6]) {
  const double xd = p.au[0], yd = p.av[0], m = p.au[2], n = p.av[2], t1 = p.th2;
  const double t2 = t1 * t1, t3 = t2 * t1;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    W[0 * 6 + i] = fma(xd, ju[i], W[0 * 6 + i]);
    W[1 * 6 + i] = fma(yd, jv[i], W[1 * 6 + i]);
    W[2 * 6 + i] += ju[i];
    W[3 * 6 + i] += jv[i];
    const double m6 = fma(m, ju[i], n * jv[i]);
    W[4 * 6 + i] += m6;
    W[5 * 6 + i] = fma(t1, m6, W[5 * 6 + i]);
    W[6 * 6 + i] = fma(t2, m6, W[6 * 6 + i]);
    W[7 * 6 + i] = fma(t3, m6, W[7 * 6 + i]);
  }
}

template <int MODEL>
CC_HD void accumulate_view(ViewAcc<MODEL>& A, const PointEval<MODEL>& p, const double Yr[3],
                           double eu, double ev) {
  double ju[6], jv[6];
  pose_row(Yr, p.gu, ju);
  pose_row(Yr, p.gv, jv);
  accumulate_vg(A.V, A.g, ju, jv, eu, ev);
  accumulate_w(A.W, p, ju, jv);
}

template <int MODEL>
CC_HD void accumulate_intr(IntrAcc<MODEL>& A, const PointEval<MODEL>& p, double eu, double ev) {
  using T = MT<MODEL>;
  constexpr int NI = T::NI;
  constexpr int NU = trisz(NI);
CC_UNROLL
  for (int a = 0; a < T::NA; ++a) {
CC_UNROLL
    for (int b = a; b < T::NA; ++b) {
      A.v[tri(NI, T::ucol(a), T::ucol(b))] = fma(p.au[a], p.au[b], A.v[tri(NI, T::ucol(a), T::ucol(b))]);
      A.v[tri(NI, T::vcol(a), T::vcol(b))] = fma(p.av[a], p.av[b], A.v[tri(NI, T::vcol(a), T::vcol(b))]);
    }
    A.v[NU + T::ucol(a)] = fma(p.au[a], eu, A.v[NU + T::ucol(a)]);
    A.v[NU + T::vcol(a)] = fma(p.av[a], ev, A.v[NU + T::vcol(a)]);
  }
  A.v[NU + NI] = fma(eu, eu, A.v[NU + NI]);
  A.v[NU + NI] = fma(ev, ev, A.v[NU + NI]);
}
// (pinhole counts cx/cy through its unit columns; add_count is a no-op there)

CC_HD void accumulate_intr(IntrAcc<FISHEYE>& A, const PointEval<FISHEYE>& p, double eu, double ev) {
  const double xd = p.au[0], yd = p.av[0], m = p.au[2], n = p.av[2], t1 = p.th2;
  double tp[7];
  tp[0] = 1.0; tp[1] = t1; tp[2] = t1 * t1; tp[3] = tp[2] * t1; tp[4] = tp[2] * tp[2]; tp[5] = tp[4] * t1;
  tp[6] = tp[3] * tp[3];
  const double w = fma(m, m, n * n);
  A.v[0] += w;
  CC_UNROLL
  for (int q = 1; q < 7; ++q) A.v[q] = fma(w, tp[q], A.v[q]);
  A.v[7] = fma(xd, xd, A.v[7]);  A.v[8] += xd;
  A.v[9] = fma(yd, yd, A.v[9]);  A.v[10] += yd;
  const double xm = xd * m, yn = yd * n, me = fma(m, eu, n * ev);
  A.v[11] += xm; A.v[15] += m; A.v[19] += yn; A.v[23] += n; A.v[32] += me;
  CC_UNROLL
  for (int j = 1; j < 4; ++j) {
    A.v[11 + j] = fma(xm, tp[j], A.v[11 + j]);
    A.v[15 + j] = fma(m, tp[j], A.v[15 + j]);
    A.v[19 + j] = fma(yn, tp[j], A.v[19 + j]);
    A.v[23 + j] = fma(n, tp[j], A.v[23 + j]);
    A.v[32 + j] = fma(me, tp[j], A.v[32 + j]);
  }
  A.v[28] = fma(xd, eu, A.v[28]); A.v[29] = fma(yd, ev, A.v[29]);
  A.v[30] += eu; A.v[31] += ev;
  A.v[36] = fma(eu, eu, A.v[36]);
  A.v[36] = fma(ev, ev, A.v[36]);
}

// Map the local-frame blocks of one view to Rodrigues coordinates (in place):
// V <- T^T V T, W <- W T, g <- T^T g with T = diag(Jl, I3).
template <int NI>
CC_HD void to_rodrigues_frame(const double Jl[9], double V[21], double* W, double g[6]) {
  // A = V[0:3,0:3] (sym), B = V[0:3,3:6]
  double A[9], B[9];
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j) {
      A[i * 3 + j] = V[i <= j ? tri(6, i, j) : tri(6, j, i)];
      B[i * 3 + j] = V[tri(6, i, 3 + j)];
    }
  double AJ[9];
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j)
      AJ[i * 3 + j] = A[i * 3 + 0] * Jl[0 * 3 + j] + A[i * 3 + 1] * Jl[1 * 3 + j] + A[i * 3 + 2] * Jl[2 * 3 + j];
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = i; j < 3; ++j)
      V[tri(6, i, j)] = Jl[0 * 3 + i] * AJ[0 * 3 + j] + Jl[1 * 3 + i] * AJ[1 * 3 + j] + Jl[2 * 3 + i] * AJ[2 * 3 + j];
  CC_UNROLL
  for (int i = 0; i < 3; ++i)
    CC_UNROLL
    for (int j = 0; j < 3; ++j)
      V[tri(6, i, 3 + j)] = Jl[0 * 3 + i] * B[0 * 3 + j] + Jl[1 * 3 + i] * B[1 * 3 + j] + Jl[2 * 3 + i] * B[2 * 3 + j];
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    const double w0 = W[a * 6 + 0], w1 = W[a * 6 + 1], w2 = W[a * 6 + 2];
    CC_UNROLL
    for (int j = 0; j < 3; ++j) W[a * 6 + j] = w0 * Jl[0 * 3 + j] + w1 * Jl[1 * 3 + j] + w2 * Jl[2 * 3 + j];
  }
  const double g0 = g[0], g1 = g[1], g2 = g[2];
  CC_UNROLL
  for (int j = 0; j < 3; ++j) g[j] = Jl[0 * 3 + j] * g0 + Jl[1 * 3 + j] * g1 + Jl[2 * 3 + j] * g2;
}

// The same map split in two: pose block + gradient, and the cross block.
CC_HD void vg_to_rodrigues_frame(const double Jl[9], double V[21], double g[6]) {
  double Wd[6] = {0, 0, 0, 0, 0, 0};
  to_rodrigues_frame<1>(Jl, V, Wd, g);
}
template <int NI>
CC_HD void w_to_rodrigues_frame(const double Jl[9], double* W) {
  CC_UNROLL
  for (int a = 0; a < NI; ++a) {
    const double w0 = W[a * 6 + 0], w1 = W[a * 6 + 1], w2 = W[a * 6 + 2];
    CC_UNROLL
    for (int j = 0; j < 3; ++j) W[a * 6 + j] = w0 * Jl[0 * 3 + j] + w1 * Jl[1 * 3 + j] + w2 * Jl[2 * 3 + j];
  }
}

// In-place Cholesky of a 6x6 SPD matrix stored as upper triangle (21): on
// return V holds U with V = U^T U.  Returns false if a pivot is not positive.
// inv_diag receives 1 / U_ii: the elimination substitutes NI + 1 vectors through one
// factor, and the six reciprocals the factorisation computes anyway replace 6 (NI + 1)
// divisions.  Square root and reciprocal come from one hardware-seeded rsqrt (<= 2 ulp).
CC_HD bool chol6(double V[21], double inv_diag[6]) {
  bool ok = true;
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    double d = V[tri(6, i, i)];
    CC_UNROLL
    for (int k = 0; k < i; ++k) d -= V[tri(6, k, i)] * V[tri(6, k, i)];
    ok = ok && (d > 0.0);
    double u;
    const double iu = fast_rsqrt(d > 0.0 ? d : 1.0, u);
    inv_diag[i] = iu;
    V[tri(6, i, i)] = u;
    CC_UNROLL
    for (int j = i + 1; j < 6; ++j) {
      double s = V[tri(6, i, j)];
      CC_UNROLL
      for (int k = 0; k < i; ++k) s -= V[tri(6, k, i)] * V[tri(6, k, j)];
      V[tri(6, i, j)] = s * iu;
    }
  }
  return ok;
}
CC_HD bool chol6(double V[21]) {
  double inv_diag[6];
  return chol6(V, inv_diag);
}
CC_HD void chol6_fwd(const double U[21], const double inv_diag[6], double y[6]) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    double s = y[i];
    CC_UNROLL
    for (int k = 0; k < i; ++k) s -= U[tri(6, k, i)] * y[k];
    y[i] = s * inv_diag[i];
  }
}
CC_HD void chol6_bwd(const double U[21], const double inv_diag[6], double y[6]) {
  CC_UNROLL
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    CC_UNROLL
    for (int k = i + 1; k < 6; ++k) s -= U[tri(6, i, k)] * y[k];
    y[i] = s * inv_diag[i];
  }
}
// y <- U^-T y   (forward substitution with the upper factor)
CC_HD void chol6_fwd(const double U[21], double y[6]) {
  CC_UNROLL
  for (int i = 0; i < 6; ++i) {
    double s = y[i];
    CC_UNROLL
    for (int k = 0; k < i; ++k) s -= U[tri(6, k, i)] * y[k];
    y[i] = s / U[tri(6, i, i)];
  }
}
// y <- U^-1 y   (back substitution)
CC_HD void chol6_bwd(const double U[21], double y[6]) {
  CC_UNROLL
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    CC_UNROLL
    for (int k = i + 1; k < 6; ++k) s -= U[tri(6, i, k)] * y[k];
    y[i] = s / U[tri(6, i, i)];
  }
}

}  // namespace camcal
