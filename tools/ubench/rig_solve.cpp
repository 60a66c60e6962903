// Timing harness for camcal_rig_solve (development tool): a fake 4-camera fisheye rig
// whose record holds a diagonally dominant reduced system; times the reduced solve alone.
//   hipcc -O2 -std=c++17 tools/ubench/rig_solve.cpp -Lcameracalibration_amd/csrc -lcamcal -ldl \
//         -Wl,-rpath,'$ORIGIN/../../cameracalibration_amd/csrc' -o tools/ubench/rig_solve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/camcal.h"
#include <dlfcn.h>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main() {
  const int nC = 4, NI = 8, G = NI + 6, n = nC * G;
  const int NU = NI * (NI + 1) / 2, NV = NU + NI + 1, NX = NI * 6 + 27;
  const int64_t total = camcal_rig_rec_len(CAMCAL_FISHEYE, nC);
  std::vector<double> rec(total, 0.0);
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
  for (int c = 0; c < nC; ++c) {
    double* glob = rec.data() + c * (NV + NX);
    double* rx = glob + NV;
    int t = 0;
    for (int i = 0; i < NI; ++i) for (int j = i; j < NI; ++j) glob[t++] = i == j ? 100.0 + i : rnd();
    for (int i = 0; i < NI; ++i) glob[NU + i] = rnd();
    for (int i = 0; i < NI * 6; ++i) rx[i] = rnd();
    t = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) rx[NI * 6 + t++] = i == j ? 100.0 + i : rnd();
    for (int i = 0; i < 6; ++i) rx[NI * 6 + 21 + i] = rnd();
  }
  double* Cm = rec.data() + nC * (NV + NX) + 5;
  for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { const double v = 0.1 * rnd(); Cm[i * n + j] = v; Cm[j * n + i] = v; }
  for (int i = 0; i < n; ++i) Cm[n * n + i] = rnd();
  double* d_all = nullptr;
  const size_t dummy = 1 << 20;
  HIP_OK(hipMalloc(&d_all, (dummy + 2 * total + 1024) * sizeof(double)));
  HIP_OK(hipMemset(d_all, 0, (dummy + 2 * total + 1024) * sizeof(double)));
  camcal_rig rig = {};
  rig.model = CAMCAL_FISHEYE; rig.n_cams = nC; rig.P = 64;
  for (int c = 0; c < nC; ++c) {
    camcal_rig_cam& k = rig.cam[c];
    k.tiles.pts = (float*)d_all; k.tiles.tile_off = (int64_t*)d_all; k.tiles.tile_nmax = (int32_t*)d_all;
    k.tiles.lane_n = (int32_t*)d_all; k.tiles.n_tiles = 1; k.tiles.n_slots = 64; k.tiles.planar = 1;
    k.slot_pose = (int32_t*)d_all; k.pose_slot = (int32_t*)d_all; k.pose = d_all; k.V = d_all; k.W = d_all; k.g = d_all;
  }
  for (int b = 0; b < 2; ++b) {
    rig.intr[b] = d_all + 1024 * (b + 1); rig.ext[b] = d_all + 4096 * (b + 1); rig.rig_pose[b] = d_all; rig.Vp[b] = d_all;
    rig.gp[b] = d_all; rig.Wp[b] = d_all; rig.rec[b] = d_all + dummy + b * total;
  }
  rig.Yt = d_all; rig.z = d_all; rig.dG = d_all + 16384; rig.scratch = d_all; rig.status = (int32_t*)d_all;
  HIP_OK(hipMemcpy(rig.rec[0], rec.data(), total * sizeof(double), hipMemcpyHostToDevice));
  std::vector<uint8_t> mask(n, 1);
  for (int i = NI; i < G; ++i) mask[i] = 0;                 // camera 0: transform fixed
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 200; ++i)
      if (camcal_rig_solve(&rig, 0, mask.data(), 1e-3, 1, nullptr)) return 3;
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("camcal_rig_solve n=%d: %.2f us per call\n", n, ms * 1e3 / 200);
  }
  std::vector<double> dG(n);
  HIP_OK(hipMemcpy(dG.data(), rig.dG, n * sizeof(double), hipMemcpyDeviceToHost));
  // residual check against the dense system on the host
  std::vector<double> M(n * n, 0.0), rhs(n, 0.0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    const int c = i / G, a = i % G, cj = j / G, b = j % G;
    double v = 0.0;
    const double* glob = rec.data() + c * (NV + NX); const double* rx = glob + NV;
    auto tri = [](int N, int lo, int hi) { return lo * N - lo * (lo - 1) / 2 + (hi - lo); };
    if (!mask[i] || !mask[j]) v = i == j ? 1.0 : 0.0;
    else {
      if (c == cj) {
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        if (hi < NI) v = glob[tri(NI, lo, hi)]; else if (lo < NI) v = rx[lo * 6 + hi - NI]; else v = rx[NI * 6 + tri(6, lo - NI, hi - NI)];
        if (i == j) v *= 1.0 + 1e-3;
      }
      v -= Cm[i * n + j];
    }
    M[i * n + j] = v;
  }
  for (int i = 0; i < n; ++i) {
    const int c = i / G, a = i % G;
    const double* glob = rec.data() + c * (NV + NX); const double* rx = glob + NV;
    rhs[i] = mask[i] ? (a < NI ? glob[NU + a] : rx[NI * 6 + 21 + a - NI]) - Cm[n * n + i] : 0.0;
  }
  double worst = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = -rhs[i];
    for (int j = 0; j < n; ++j) s += M[i * n + j] * dG[j];
    if (mask[i] && std::abs(s) > worst) worst = std::abs(s);
  }
  std::printf("max |M dG - rhs| = %.3e\n", worst);
  // an instrumented build of the library may export phase timers
  if (auto dbg = (int (*)(long long*))dlsym(RTLD_DEFAULT, "camcal_debug_clk")) {
    long long clk[8];
    dbg(clk);
    std::printf("cycles:");
    for (int i = 0; i < 8; ++i) std::printf(" %lld", clk[i]);
    std::printf("\n");
  }
  return 0;
}
