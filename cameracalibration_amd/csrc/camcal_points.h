// Point-level operations either side of the solver:
// projection, the tool's per-view reprojection error, undistortion of pixels and
// the undistort/rectify lookup maps.  One lane = one point / one view / one pixel.
#pragma once
#include "camcal_lane.h"

namespace camcal {

// cv2.projectPoints / cv2.fisheye.projectPoints of one point (double precision in,
// as cv2 converts its input before any arithmetic).
template <int MODEL>
CC_HD void project_point(const double X[3], const double R[9], const double t[3], const double* intr, double& u,
                         double& v) {
  double Yc[3];
  CC_UNROLL
  for (int i = 0; i < 3; ++i) Yc[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
  PointEval<MODEL> p;
  eval_point<false>(Yc, intr, p);
  u = p.pu;
  v = p.pv;
}

// The acceptance number of the tool for one view (camera_calib.py:96-104,174-181;
// intrinsicCalib.py:137-146,229-238):
//     cv2.norm(projectPoints(board, rvec, tvec, K, D), corners, NORM_L2) / len(corners)
// With F32 the projection is rounded to float32 and subtracted in float32, as
// cv2 does for float32 inputs (projectPoints returns the depth of objectPoints;
// norm() subtracts in the element type and accumulates in double).
template <int MODEL, bool PLANAR, bool F32>
CC_HD double lane_reproj_err(const LaneTile& tv, const double* intr, const double pose[6]) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  double s = 0.0;
  for (int k = 0; k < tv.n; ++k) {
    const Obs o = load_obs<PLANAR>(tv.base, k);
    double Yr[3], Yc[3];
    transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
    PointEval<MODEL> p;
    eval_point<false>(Yc, intr, p);
    double eu, ev;
    if (F32) {
      eu = (double)((float)p.pu - o.u);
      ev = (double)((float)p.pv - o.v);
    } else {
      eu = p.pu - (double)o.u;
      ev = p.pv - (double)o.v;
    }
    s = fma(eu, eu, s);
    s = fma(ev, ev, s);
  }
  return tv.n > 0 ? sqrt(s) / (double)tv.n : 0.0;
}

// cv2.undistortPoints (pinhole: 5 fixed-point sweeps) / cv2.fisheye.undistortPoints
// (Newton on theta, <= 10 steps, eps 1e-8; a point whose iteration does not
// converge or whose theta changes sign becomes (-1e6, -1e6)), followed by the
// optional 3x3 matrix M = P[:, :3] * R (row-major; NULL = identity).
template <int MODEL>
CC_HD void undistort_point(double u, double v, const double* intr, const double* M, double& ox, double& oy);

template <>
CC_HD void undistort_point<PINHOLE>(double u, double v, const double* intr, const double* M, double& ox, double& oy) {
  double x, y;
  undistort_px<PINHOLE>(u, v, intr, x, y);
  if (M) {
    const double xx = M[0] * x + M[1] * y + M[2], yy = M[3] * x + M[4] * y + M[5];
    const double ww = 1.0 / (M[6] * x + M[7] * y + M[8]);
    x = xx * ww;
    y = yy * ww;
  }
  ox = x;
  oy = y;
}

template <>
CC_HD void undistort_point<FISHEYE>(double u, double v, const double* intr, const double* M, double& ox, double& oy) {
  const double k1 = intr[4], k2 = intr[5], k3 = intr[6], k4 = intr[7];
  const double ifx = 1.0 / intr[0], ify = 1.0 / intr[1];
  const double px = (u - intr[2]) * ifx, py = (v - intr[3]) * ify;
  const double p2 = px * px + py * py;
  double thd = 0.0;
  if (p2 > 0.0) fast_rsqrt(p2, thd);
  const double hp = 1.5707963267948966;
  thd = thd > hp ? hp : thd;
  bool converged = false;
  double th = thd, scale = 0.0;
  if (thd > 1e-8) {
    for (int j = 0; j < 10; ++j) {
      const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      const double fix = (th * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8) - thd) *
                         fast_rcp(1 + 3 * k1 * t2 + 5 * k2 * t4 + 7 * k3 * t6 + 9 * k4 * t8);
      th -= fix;
      if (fabs(fix) < 1e-8) { converged = true; break; }
    }
    scale = tan_acute(th) * fast_rcp(thd);
  } else {
    converged = true;
  }
  const bool flipped = thd > 0 && th < 0;
  if (!converged || flipped) { ox = -1000000.0; oy = -1000000.0; return; }
  double x = px * scale, y = py * scale;
  if (M) {
    const double xx = M[0] * x + M[1] * y + M[2], yy = M[3] * x + M[4] * y + M[5];
    const double ww = M[6] * x + M[7] * y + M[8];
    x = xx / ww;
    y = yy / ww;
  }
  ox = x;
  oy = y;
}

// cv2.initUndistortRectifyMap / cv2.fisheye.initUndistortRectifyMap: source pixel
// of destination pixel (i = row, j = column); iR = (newK * R)^-1 row-major.
// Returns false where the fisheye ray has w <= 0 (map entry -1).
template <int MODEL>
CC_HD bool rectify_map_px(int i, int j, const double iR[9], const double* intr, double& u, double& v) {
  const double _x = j * iR[0] + i * iR[1] + iR[2];
  const double _y = j * iR[3] + i * iR[4] + iR[5];
  const double _w = j * iR[6] + i * iR[7] + iR[8];
  if (MODEL == FISHEYE) {
    if (_w <= 0) { u = -1.0; v = -1.0; return false; }
    const double x = _x / _w, y = _y / _w;
    const double r = sqrt(x * x + y * y);
    const double th = atan(r);
    const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double thd = th * (1 + intr[4] * t2 + intr[5] * t4 + intr[6] * t6 + intr[7] * t8);
    const double sc = r == 0 ? 1.0 : thd / r;
    u = intr[0] * x * sc + intr[2];
    v = intr[1] * y * sc + intr[3];
  } else {
    const double w = 1.0 / _w, x = _x * w, y = _y * w;
    const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
    const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
    const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
    u = intr[0] * xd + intr[2];
    v = intr[1] * yd + intr[3];
  }
  return true;
}

// CV_16SC2 fixed-point encoding of a map entry (INTER_BITS = 5): integer part
// in map1, the two 5-bit fractions in map2.  cvRound = round half to even.
CC_HD void encode_fixed_map(double u, double v, short& mx, short& my, unsigned short& frac) {
  auto sat = [](double x) -> int {
    if (!(x > -2147483648.0)) return -2147483647 - 1;      // also NaN (cvRound of NaN: INT_MIN)
    if (x >= 2147483647.0) return 2147483647;
    return (int)nearbyint(x);
  };
  const int iu = sat(u * 32.0), iv = sat(v * 32.0);
  mx = (short)(iu >> 5);                                    // plain narrowing, as cv2
  my = (short)(iv >> 5);
  frac = (unsigned short)((iv & 31) * 32 + (iu & 31));
}

// cv2.remap, INTER_LINEAR, BORDER_CONSTANT, 8-bit pixels (video_server.py:504;
// intrinsicCalib.py:369): source position in 1/32 pixel fixed point (sx, sy =
// cvRound(map * 32) for float maps, or map1 * 32 + the fractions of map2), the four
// neighbours weighted with (32 - fx)(32 - fy) * 32 ... (exact products of cv2's
// 15-bit weight table), rounded by (sum + 2^14) >> 15.
CC_HD int remap_fixed_coord(float m) {
  const double x = (double)m * 32.0;
  if (!(x > -2147483648.0)) return -2147483647 - 1;
  if (x >= 2147483647.0) return 2147483647;
  return (int)nearbyint(x);
}
CC_HD int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

CC_HD void remap_pixel_u8(const unsigned char* src, int sw, int sh, long long sstride, int cn, int ix, int iy,
                          int fx, int fy, unsigned char cval, unsigned char* dst) {
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32,
            w11 = fx * fy * 32;
  const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw;
  const bool y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
  // offsets are formed only for neighbours inside the image
  const long long o00 = (x0 && y0) ? (long long)iy * sstride + (long long)ix * cn : -1;
  const long long o01 = (x1 && y0) ? (long long)iy * sstride + (long long)(ix + 1) * cn : -1;
  const long long o10 = (x0 && y1) ? (long long)(iy + 1) * sstride + (long long)ix * cn : -1;
  const long long o11 = (x1 && y1) ? (long long)(iy + 1) * sstride + (long long)(ix + 1) * cn : -1;
  for (int c = 0; c < cn; ++c) {
    const int p00 = o00 >= 0 ? src[o00 + c] : cval, p01 = o01 >= 0 ? src[o01 + c] : cval;
    const int p10 = o10 >= 0 ? src[o10 + c] : cval, p11 = o11 >= 0 ? src[o11 + c] : cval;
    const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
    dst[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace camcal
