// Host program that calibrates a synthetic Kannala-Brandt camera through the C ABI
// alone (include/camcal.h + the HIP runtime for memory): no Python, no torch.
// It is the sequence cv2.fisheye.calibrate runs at the reference's call site
// server/aruco/camera_calib.py:74-77 (flags FIX_SKEW | RECOMPUTE_EXTRINSIC |
// USE_INTRINSIC_GUESS, criteria (EPS + COUNT, 1000, 1e-6)).
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

template <typename T> static T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
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
  // layout (host) and device buffers
  int32_t n_tiles = 0;
  int64_t n_slots = 0, total = 0;
  CC_OK(camcal_layout(n_views, n_per_view.data(), &n_tiles, &n_slots, &total, nullptr, nullptr, nullptr, nullptr));
  std::vector<int32_t> slot_view(n_slots), lane_n(n_slots), tile_nmax(n_tiles);
  std::vector<int64_t> tile_off(n_tiles);
  CC_OK(camcal_layout(n_views, n_per_view.data(), &n_tiles, &n_slots, &total, slot_view.data(), lane_n.data(),
                      tile_nmax.data(), tile_off.data()));
  float* d_obj = to_device(obj); float* d_img = to_device(img);
  int64_t* d_vp = to_device(view_ptr); int32_t* d_sv = to_device(slot_view);
  int32_t* d_ln = to_device(lane_n); int32_t* d_nm = to_device(tile_nmax); int64_t* d_to = to_device(tile_off);
  float* d_pts = nullptr;
  HIP_OK(hipMalloc(&d_pts, sizeof(float) * (total > 0 ? total : 1)));
  camcal_tiles tiles = {d_pts, d_to, d_nm, d_ln, n_tiles, n_slots, 1};
  CC_OK(camcal_pack_tiles(d_obj, d_img, d_vp, d_sv, &tiles, d_pts, nullptr));
  const int ng = camcal_glob_len(CAMCAL_FISHEYE), ns = camcal_schur_len(CAMCAL_FISHEYE);
  double *d_intr, *d_pose, *d_rec, *d_dK, *d_stats, *d_scratch;
  int32_t *d_status, *d_iters;
  HIP_OK(hipMalloc(&d_intr, 16 * sizeof(double)));
  HIP_OK(hipMalloc(&d_pose, 6 * n_slots * sizeof(double)));
  HIP_OK(hipMalloc(&d_rec, (ng + ns) * sizeof(double)));
  HIP_OK(hipMalloc(&d_dK, 16 * sizeof(double)));
  HIP_OK(hipMalloc(&d_stats, 8 * sizeof(double)));
  HIP_OK(hipMalloc(&d_scratch, camcal_scratch_len(CAMCAL_FISHEYE, n_tiles) * sizeof(double)));
  HIP_OK(hipMalloc(&d_status, n_slots * sizeof(int32_t)));
  HIP_OK(hipMalloc(&d_iters, n_slots * sizeof(int32_t)));
  double intr16[16] = {0};
  for (int i = 0; i < 8; ++i) intr16[i] = guess[i];
  HIP_OK(hipMemcpy(d_intr, intr16, sizeof(intr16), hipMemcpyHostToDevice));
  // cv::fisheye::calibrate: CalibrateExtrinsics, then damped Gauss-Newton with pose re-estimation;
  // run twice, the second time on the clock (the first pays for code-object loading)
  double change = 1.0, stats[8], rec[64];
  int it = 0;
  double wall_ms = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
  HIP_OK(hipMemcpy(d_intr, intr16, sizeof(intr16), hipMemcpyHostToDevice));
  HIP_OK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  CC_OK(camcal_pose_init(CAMCAL_FISHEYE, &tiles, d_intr, d_pose, d_status, nullptr));
  CC_OK(camcal_pose_refine(CAMCAL_FISHEYE, &tiles, d_intr, d_pose, 20, 1e-10, d_iters, nullptr));
  const uint8_t mask[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  change = 1.0;
  for (it = 0; it < 1000 && change > 1e-6; ++it) {
    CC_OK(camcal_normal_eq_fused(CAMCAL_FISHEYE, &tiles, d_intr, d_pose, 0.0, nullptr, nullptr, nullptr, d_rec,
                                 d_rec + ng, d_scratch, nullptr));
    CC_OK(camcal_solve_reduced(CAMCAL_FISHEYE, d_rec, d_rec + ng, mask, 0.0, -(1.0 - std::pow(0.6, it + 1.0)), d_intr,
                               d_intr, d_dK, d_stats, nullptr));
    HIP_OK(hipMemcpy(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost));
    change = std::sqrt(stats[2]) / std::sqrt(stats[3]);
    CC_OK(camcal_pose_refine(CAMCAL_FISHEYE, &tiles, d_intr, d_pose, 20, 1e-10, d_iters, nullptr));
  }
  CC_OK(camcal_intr_normal(CAMCAL_FISHEYE, &tiles, d_intr, d_pose, 0, d_rec, nullptr, nullptr, nullptr, d_scratch, nullptr));
  HIP_OK(hipMemcpy(rec, d_rec, ng * sizeof(double), hipMemcpyDeviceToHost));
  wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  HIP_OK(hipMemcpy(intr16, d_intr, sizeof(intr16), hipMemcpyDeviceToHost));
  std::printf("solve wall clock %.3f ms (%d views, points resident)\n", wall_ms, n_views);
  const double rms = std::sqrt(rec[ng - 1] / (double)view_ptr[n_views]);
  std::printf("iterations %d  rms %.3e px\n", it, rms);
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
