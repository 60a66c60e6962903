// Host program that calibrates a synthetic Kannala-Brandt camera through the C ABI
// alone (include/camcal.h + the HIP runtime for memory): no Python, no torch.
// It is cv2.fisheye.calibrate as the reference calls it at
// server/aruco/camera_calib.py:74-77 (flags FIX_SKEW | RECOMPUTE_EXTRINSIC |
// USE_INTRINSIC_GUESS, criteria (EPS + COUNT, 1000, 1e-6)) through the one-call entry
// point camcal_calibrate: host arrays in, host results out, one caller-owned workspace.
// (examples/cabi_rig.cpp drives a solve stage by stage instead.)
//
//   make examples && examples/cabi_fisheye        (prints the recovered intrinsics, exit 0 on success)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/camcal.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CC_OK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d\n", #x, r_); return 3; } } while (0)

static void rodrigues(const double r[3], double R[9]) {
  const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 1e-12) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0); return; }
  const double c = std::cos(th), s = std::sin(th), x = r[0] / th, y = r[1] / th, z = r[2] / th, c1 = 1 - c;
  const double M[9] = {c + c1 * x * x, c1 * x * y - s * z, c1 * x * z + s * y, c1 * x * y + s * z, c + c1 * y * y,
                       c1 * y * z - s * x, c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z};
  for (int i = 0; i < 9; ++i) R[i] = M[i];
}

int main() {
  const double truth[8] = {1066.58, 1066.31, 1478.65, 791.19, -0.02334, 0.007086, -0.005574, 0.000994};
  const double guess[8] = {1062.26756, 1062.212564, 1501.057278, 867.887492, -0.018123981144576858,
                           -0.0026101267970621823, 0.0006215782316954739, -0.00030905667520437564};
  const int n_views = 300, bw = 11, bh = 8;
  const double sq = 25.0;
  // synthetic views: ragged (views keep 60..88 corners), poses in front of the camera
  std::vector<int32_t> n_per_view(n_views);
  std::vector<int64_t> view_ptr(n_views + 1, 0);
  std::vector<float> obj, img;
  unsigned long long rng = 88172645463325252ull;
  auto uni = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
  for (int v = 0; v < n_views; ++v) {
    const double r[3] = {(uni() - 0.5) * 1.0, (uni() - 0.5) * 1.0, (uni() - 0.5) * 3.0};
    const double t[3] = {(uni() - 0.5) * 900.0 - 125.0, (uni() - 0.5) * 500.0 - 90.0, 500.0 + uni() * 700.0};
    double R[9];
    rodrigues(r, R);
    const int keep = 60 + (int)(uni() * 29.0);
    int cnt = 0;
    for (int k = 0; k < bw * bh && cnt < keep; ++k) {
      const double X = (k % bw) * sq, Y = (k / bw) * sq;
      const double Yc[3] = {R[0] * X + R[1] * Y + t[0], R[3] * X + R[4] * Y + t[1], R[6] * X + R[7] * Y + t[2]};
      const double x = Yc[0] / Yc[2], y = Yc[1] / Yc[2], rr = std::sqrt(x * x + y * y), th = std::atan(rr), t2 = th * th;
      const double thd = th * (1 + t2 * (truth[4] + t2 * (truth[5] + t2 * (truth[6] + t2 * truth[7]))));
      const double sc = rr > 1e-8 ? thd / rr : 1.0;
      obj.push_back((float)X); obj.push_back((float)Y); obj.push_back(0.f);
      img.push_back((float)(truth[0] * x * sc + truth[2])); img.push_back((float)(truth[1] * y * sc + truth[3]));
      ++cnt;
    }
    n_per_view[v] = cnt;
    view_ptr[v + 1] = view_ptr[v] + cnt;
  }
  // one workspace, one call: cv2.fisheye.calibrate(obj, img, size, K, D, flags = FIX_SKEW | RECOMPUTE_EXTRINSIC |
  // USE_INTRINSIC_GUESS, criteria = (EPS + COUNT, 1000, 1e-6)) -- camera_calib.py:74-77
  const int64_t ws_bytes = camcal_workspace_len(CAMCAL_FISHEYE, n_views, view_ptr.data());
  if (ws_bytes <= 0) { std::fprintf(stderr, "camcal_workspace_len -> %lld\n", (long long)ws_bytes); return 3; }
  void* ws = nullptr;
  HIP_OK(hipMalloc(&ws, (size_t)ws_bytes));
  std::vector<double> rt((size_t)n_views * 6);
  std::vector<int32_t> view_status(n_views);
  double K[9], D[4], rms = 0.0;
  int32_t it = 0;
  double wall_ms = 0.0;
  for (int pass = 0; pass < 2; ++pass) {       // the second pass is on the clock (the first pays for code-object loading)
    const double K0[9] = {guess[0], 0, guess[2], 0, guess[1], guess[3], 0, 0, 1};
    for (int i = 0; i < 9; ++i) K[i] = K0[i];
    for (int i = 0; i < 4; ++i) D[i] = guess[4 + i];
    HIP_OK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    CC_OK(camcal_calibrate(CAMCAL_FISHEYE, n_views, view_ptr.data(), obj.data(), img.data(), 2960, 1664, K, D, 4,
                           rt.data(), view_status.data(), /*flags*/ 8 | 2 | 1, /*criteria*/ 3, 1000, 1e-6, ws, ws_bytes,
                           /*comm*/ nullptr, /*stream*/ nullptr, &rms, &it));
    wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  double intr16[16] = {K[0], K[4], K[2], K[5], D[0], D[1], D[2], D[3]};
  std::printf("solve wall clock %.3f ms (%d views, host arrays in, host results out)\n", wall_ms, n_views);
  std::printf("iterations %d  rms %.3e px  first view: r = (%.4f %.4f %.4f) t = (%.2f %.2f %.2f)\n", it, rms, rt[0], rt[1],
              rt[2], rt[3], rt[4], rt[5]);
  double worst = 0.0;
  for (int i = 0; i < 4; ++i) {
    std::printf("  intr[%d] = %.6f (truth %.6f)\n", i, intr16[i], truth[i]);
    worst = std::fmax(worst, std::fabs(intr16[i] / truth[i] - 1.0));
  }
  for (int i = 4; i < 8; ++i) std::printf("  k%d = %.7f (truth %.7f)\n", i - 3, intr16[i], truth[i]);
  // observations were rounded to float32 (~6e-5 px), so the focal lengths come back to ~1e-6 relative
  const bool ok = rms < 1e-3 && worst < 1e-5;
  std::printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
