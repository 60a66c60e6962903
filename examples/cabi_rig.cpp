// Host program that calibrates a synthetic 3-camera Kannala-Brandt rig through the C ABI
// alone (include/camcal.h + the HIP runtime for memory): no Python, no torch, no BLAS.
// BASELINE.json config 4 in small: shared board poses, joint intrinsics + camera-from-rig
// transforms + one pose per board placement; camcal_rig_init for the starting point,
// camcal_rig_lm for the Levenberg-Marquardt loop (C = Y^T Y on v_mfma_f64_16x16x4 inside).
// The reference has no counterpart (it calibrates its cameras one after another,
// controller/controller_internal_calibration.py:375-399).
//
//   make examples && examples/cabi_rig        (prints the recovered parameters, exit 0 on success)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/camcal.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CC_OK(x) do { int r_ = (int)(x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d\n", #x, r_); return 3; } } while (0)

static void rodrigues(const double r[3], double R[9]) {
  const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 1e-12) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0); return; }
  const double c = std::cos(th), s = std::sin(th), x = r[0] / th, y = r[1] / th, z = r[2] / th, c1 = 1 - c;
  const double M[9] = {c + c1 * x * x, c1 * x * y - s * z, c1 * x * z + s * y, c1 * x * y + s * z, c + c1 * y * y,
                       c1 * y * z - s * x, c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z};
  for (int i = 0; i < 9; ++i) R[i] = M[i];
}

template <typename T> static T* dev_copy(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
static double* dev_zeros(size_t n) {
  double* d = nullptr;
  if (hipMalloc(&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
  if (hipMemset(d, 0, sizeof(double) * (n ? n : 1)) != hipSuccess) return nullptr;
  return d;
}

int main() {
  const int nC = 3, NI = 8, G = NI + 6, n = nC * G, bw = 11, bh = 8;
  const int64_t P = 200;
  const double sq = 25.0;
  const double truth[nC][8] = {{1066.58, 1066.31, 1478.65, 791.19, -0.02334, 0.007086, -0.005574, 0.000994},
                               {1055.05, 1054.64, 1476.66, 887.57, -0.0478, 0.0932, -0.2709, 0.2967},
                               {1087.9, 1087.6, 1449.1, 807.0, -0.0238, 0.0072, -0.0057, 0.0010}};
  const double ext_true[nC][6] = {{0, 0, 0, 0, 0, 0}, {0.02, 0.35, -0.01, -180.0, 4.0, 15.0}, {-0.015, -0.33, 0.02, 175.0, -6.0, 12.0}};
  // board poses in the rig frame; every camera records the corners that fall inside its image
  unsigned long long rng = 0x9E3779B97F4A7C15ull;
  auto uni = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
  std::vector<double> pose_true(P * 6);
  std::vector<float> obj[nC], img[nC];
  std::vector<int64_t> view_ptr[nC];
  std::vector<int32_t> n_per_view[nC], pose_of_view[nC];
  for (int c = 0; c < nC; ++c) view_ptr[c].push_back(0);
  for (int64_t p = 0; p < P; ++p) {
    const double r[3] = {(uni() - 0.5) * 0.9, (uni() - 0.5) * 0.9, (uni() - 0.5) * 2.5};
    const double t[3] = {(uni() - 0.5) * 700.0 - 125.0, (uni() - 0.5) * 400.0 - 90.0, 600.0 + uni() * 600.0};
    for (int i = 0; i < 3; ++i) { pose_true[p * 6 + i] = r[i]; pose_true[p * 6 + 3 + i] = t[i]; }
    double Rp[9];
    rodrigues(r, Rp);
    for (int c = 0; c < nC; ++c) {
      double Rc[9];
      rodrigues(ext_true[c], Rc);
      std::vector<float> o, m;
      for (int k = 0; k < bw * bh; ++k) {
        const double X = (k % bw) * sq, Y = (k / bw) * sq;
        const double Yr[3] = {Rp[0] * X + Rp[1] * Y + t[0], Rp[3] * X + Rp[4] * Y + t[1], Rp[6] * X + Rp[7] * Y + t[2]};
        double Yc[3];
        for (int i = 0; i < 3; ++i) Yc[i] = Rc[i * 3] * Yr[0] + Rc[i * 3 + 1] * Yr[1] + Rc[i * 3 + 2] * Yr[2] + ext_true[c][3 + i];
        if (Yc[2] < 50.0) continue;
        const double x = Yc[0] / Yc[2], y = Yc[1] / Yc[2], rr = std::sqrt(x * x + y * y), th = std::atan(rr), t2 = th * th;
        const double* K = truth[c];
        const double thd = th * (1 + t2 * (K[4] + t2 * (K[5] + t2 * (K[6] + t2 * K[7]))));
        const double sc = rr > 1e-8 ? thd / rr : 1.0;
        const double u = K[0] * x * sc + K[2], v = K[1] * y * sc + K[3];
        if (u < 0 || u >= 2960 || v < 0 || v >= 1664) continue;
        o.push_back((float)X); o.push_back((float)Y); o.push_back(0.f);
        m.push_back((float)u); m.push_back((float)v);
      }
      if (m.size() / 2 < 12) continue;                       // this camera does not see the pose
      obj[c].insert(obj[c].end(), o.begin(), o.end());
      img[c].insert(img[c].end(), m.begin(), m.end());
      n_per_view[c].push_back((int32_t)(m.size() / 2));
      pose_of_view[c].push_back((int32_t)p);
      view_ptr[c].push_back(view_ptr[c].back() + (int64_t)(m.size() / 2));
    }
  }
  // ---- the rig descriptor: every camera packed like a single camera
  camcal_rig rig;
  std::memset(&rig, 0, sizeof(rig));
  rig.model = CAMCAL_FISHEYE;
  rig.n_cams = nC;
  rig.P = P;
  int32_t n_tiles_all[nC];
  int64_t total_obs = 0, slots_all = 0;
  for (int c = 0; c < nC; ++c) {
    const int64_t nv = (int64_t)n_per_view[c].size();
    total_obs += view_ptr[c].back();
    int32_t n_tiles = 0;
    int64_t n_slots = 0, total = 0;
    CC_OK(camcal_layout(nv, n_per_view[c].data(), &n_tiles, &n_slots, &total, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> slot_view(n_slots), lane_n(n_slots), tile_nmax(n_tiles), slot_pose(n_slots, -1);
    std::vector<int64_t> tile_off(n_tiles);
    CC_OK(camcal_layout(nv, n_per_view[c].data(), &n_tiles, &n_slots, &total, slot_view.data(), lane_n.data(), tile_nmax.data(),
                        tile_off.data()));
    for (int64_t s = 0; s < n_slots; ++s)
      if (slot_view[s] >= 0) slot_pose[s] = pose_of_view[c][slot_view[s]];
    float* d_pts = nullptr;
    HIP_OK(hipMalloc(&d_pts, sizeof(float) * (total ? total : 1)));
    camcal_rig_cam& k = rig.cam[c];
    k.tiles = camcal_tiles{d_pts, dev_copy(tile_off), dev_copy(tile_nmax), dev_copy(lane_n), n_tiles, n_slots, 1, 0};
    float* d_obj = dev_copy(obj[c]);
    float* d_img = dev_copy(img[c]);
    int64_t* d_vp = dev_copy(view_ptr[c]);
    int32_t* d_sv = dev_copy(slot_view);
    CC_OK(camcal_pack_tiles(d_obj, d_img, d_vp, d_sv, &k.tiles, d_pts, nullptr));
    k.slot_pose = dev_copy(slot_pose);
    HIP_OK(hipMalloc(&k.pose_slot, sizeof(int32_t) * P));
    k.pose = dev_zeros(6 * n_slots);
    k.V = dev_zeros(21 * n_slots);
    k.W = dev_zeros((size_t)NI * 6 * n_slots);
    k.g = dev_zeros(6 * n_slots);
    n_tiles_all[c] = n_tiles;
    slots_all += n_slots;
  }
  const int64_t rec_len = camcal_rig_rec_len(rig.model, nC);
  const int64_t scratch_len = camcal_rig_scratch_len(rig.model, nC, n_tiles_all, P);
  if (rec_len <= 0 || scratch_len <= 0) return 3;
  for (int b = 0; b < 2; ++b) {
    rig.intr[b] = dev_zeros(nC * 16);
    rig.ext[b] = dev_zeros(nC * 6);
    rig.rig_pose[b] = dev_zeros(6 * P);
    rig.Vp[b] = dev_zeros((size_t)nC * 21 * P);       // zeroed ONCE: entries of unseen poses stay zero
    rig.gp[b] = dev_zeros((size_t)nC * 6 * P);
    rig.Wp[b] = dev_zeros((size_t)nC * G * 6 * P);
    rig.rec[b] = dev_zeros(rec_len);
  }
  rig.Yt = dev_zeros((size_t)n * 6 * P);
  rig.z = dev_zeros(6 * P);
  rig.dG = dev_zeros(64);
  rig.scratch = dev_zeros(scratch_len);
  HIP_OK(hipMalloc(&rig.status, sizeof(int32_t) * slots_all));
  // guesses: the truth perturbed by up to 1 % / 10 %
  std::vector<double> intr0(nC * 16, 0.0);
  for (int c = 0; c < nC; ++c)
    for (int i = 0; i < 8; ++i) intr0[c * 16 + i] = truth[c][i] * (i < 4 ? 1.0 + 0.01 * (uni() - 0.5) : 0.9);
  HIP_OK(hipMemcpy(rig.intr[0], intr0.data(), sizeof(double) * intr0.size(), hipMemcpyHostToDevice));
  // free parameters: everything but camera 0's (identity) extrinsics
  uint8_t mask[64];
  for (int c = 0; c < nC; ++c)
    for (int a = 0; a < G; ++a) mask[c * G + a] = (c == 0 && a >= NI) ? 0 : 1;

  const auto t0 = std::chrono::steady_clock::now();
  CC_OK(camcal_rig_init(&rig, nullptr, nullptr));
  double err = 0.0;
  int32_t iters = 0, cur = 0;
  CC_OK(camcal_rig_lm(&rig, mask, 100, 1e-10, nullptr, nullptr, &err, &iters, &cur));
  HIP_OK(hipDeviceSynchronize());
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  std::vector<double> intr(nC * 16), ext(nC * 6);
  HIP_OK(hipMemcpy(intr.data(), rig.intr[cur], sizeof(double) * intr.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ext.data(), rig.ext[cur], sizeof(double) * ext.size(), hipMemcpyDeviceToHost));
  const double rms = std::sqrt(err / (double)total_obs);
  std::printf("rig: %d cameras, %lld board poses, %lld observations: init + %d LM iterations in %.3f ms, rms %.3e px\n", nC,
              (long long)P, (long long)total_obs, iters, wall_ms, rms);
  double worst_f = 0.0, worst_r = 0.0, worst_t = 0.0;
  for (int c = 0; c < nC; ++c) {
    std::printf("  cam %d: fx %.5f (truth %.5f)  cx %.5f (%.5f)  r_c = (%.6f %.6f %.6f)  t_c = (%.4f %.4f %.4f)\n", c, intr[c * 16],
                truth[c][0], intr[c * 16 + 2], truth[c][2], ext[c * 6], ext[c * 6 + 1], ext[c * 6 + 2], ext[c * 6 + 3],
                ext[c * 6 + 4], ext[c * 6 + 5]);
    for (int i = 0; i < 4; ++i) worst_f = std::fmax(worst_f, std::fabs(intr[c * 16 + i] / truth[c][i] - 1.0));
    for (int i = 0; i < 3; ++i) {
      worst_r = std::fmax(worst_r, std::fabs(ext[c * 6 + i] - ext_true[c][i]));
      worst_t = std::fmax(worst_t, std::fabs(ext[c * 6 + 3 + i] - ext_true[c][3 + i]));
    }
  }
  // observations were rounded to float32 (~1e-4 px): parameters come back to ~1e-6 relative
  const bool ok = rms < 2e-3 && worst_f < 2e-5 && worst_r < 2e-5 && worst_t < 2e-2;
  std::printf("worst: focal/centre rel %.2e, r_c %.2e rad, t_c %.2e mm\n%s\n", worst_f, worst_r, worst_t, ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
