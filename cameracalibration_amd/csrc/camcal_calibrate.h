// One-call entry points (include/camcal.h: camcal_workspace_len, camcal_calibrate,
// camcal_calibrate_many): everything cv2.calibrateCamera / cv2.fisheye.calibrate do behind
// the reference's call sites (camera_calib.py:74-77,153-156; intrinsicCalib.py:102-105,
// 131-134,194-197,223-226) from host arrays in to host results out -- argument checks, view
// layout, upload, packing, per-view pose initialisation and refinement, the solver loop
// (camcal_driver.h), results in the caller's view order -- with every device byte in a
// workspace the caller owns.  camcal_calibrate_many runs several independent problems (the
// reference solves its four cameras one after another, controller/
// controller_internal_calibration.py:375-399) at once: one host thread and one stream per
// problem.  A production-sized solve is a chain of ~10 us kernels whose issue rate, not the
// device, is the limit (measured: the interleaved single-thread form spent 1.9 ms on 300
// launches that each left the device idle until the next one arrived), so the problems need
// issuing threads of their own to overlap.
//
// Included by camcal_kernels.hip and by the tests' CPU lane emulator after camcal_driver.h;
// the includer provides, with internal linkage, besides what camcal_driver.h and
// camcal_rig_driver.h ask for,
//   int camcal_upload(void* dst_dev, const void* src_host, size_t bytes, void* stream)
//   int camcal_upload_bulk(void* dst_dev, const void* src_host, size_t bytes, void* stream)
//       (a large array: may stage it through page-locked memory on several threads; the host side is free again
//        when it returns)
//   int camcal_download(void* dst_host, const void* src_dev, size_t bytes, void* stream)
//       (asynchronous, ordered on the stream; the host side must stay valid until the sync)
//   int camcal_stream_sync(void* stream)
//   int camcal_get_device(int* dev) / int camcal_set_device(int dev)   (a new host thread starts on device 0)
//   void* camcal_pinned_take(size_t bytes) / void camcal_pinned_give(void*)
//       (page-locked host staging from a pool that lives as long as the process; NULL: none to be had, the
//        copies then go from / to the caller's arrays directly)
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/camcal.h"
#include "camcal_driver.h"

namespace {
namespace camcal_cal {

// cv2 flag bits (calib3d.hpp; the Python drop-in carries the same table)
constexpr int PH_USE_GUESS = 1, PH_FIX_ASPECT = 2, PH_FIX_PP = 4, PH_ZERO_TANGENT = 8, PH_FIX_FOCAL = 16, PH_FIX_K1 = 32,
              PH_FIX_K2 = 64, PH_FIX_K3 = 128, PH_RATIONAL = 16384, PH_THIN_PRISM = 32768, PH_TILTED = 262144,
              PH_FIX_TANGENT = 2097152;
constexpr int FE_USE_GUESS = 1, FE_RECOMPUTE = 2, FE_CHECK_COND = 4, FE_FIX_SKEW = 8, FE_FIX_K1 = 16, FE_FIX_K2 = 32,
              FE_FIX_K3 = 64, FE_FIX_K4 = 128, FE_FIX_PP = 512, FE_FIX_FOCAL = 2048;

constexpr size_t ALIGN = 256;
inline size_t up(size_t x) { return (x + ALIGN - 1) & ~(ALIGN - 1); }

// Byte offsets of the pieces of a workspace.  The host-filled pieces form one block
// [tile_off | view_ptr | slot_view | lane_n | tile_nmax | img | obj | intr0] (one upload from pinned staging
// for production-sized calls) and the results one block [intr0 | pose0 | status] (one download).
struct Carve {
  size_t tile_off = 0, view_ptr = 0, slot_view = 0, lane_n = 0, tile_nmax = 0, tables_bytes = 0;
  size_t img = 0, obj = 0, intr0 = 0, up_bytes = 0;
  size_t pose0 = 0, status = 0, res_bytes = 0;
  size_t pts = 0;
  size_t intr1 = 0, pose1 = 0, V0 = 0, V1 = 0, g0 = 0, g1 = 0, W1 = 0, intr_spare = 0, W = 0, rec = 0, glob1 = 0, dK = 0, stats = 0,
         scratch = 0, iters = 0, flag = 0;
  size_t total = 0;
};

inline size_t up8(size_t x) { return (x + 7) & ~(size_t)7; }

inline Carve carve(int model, int64_t n_views, int64_t n_obs, int32_t T, int64_t S, int64_t total_floats) {
  const int ni = camcal_n_intr(model), ng = camcal_glob_len(model), ns = camcal_schur_len(model);
  Carve c;
  size_t o = 0;
  c.tile_off = o; o += (size_t)T * 8;
  c.view_ptr = o; o += (size_t)(n_views + 1) * 8;
  c.slot_view = o; o += (size_t)S * 4;
  c.lane_n = o; o += (size_t)S * 4;
  c.tile_nmax = o; o += (size_t)T * 4;
  c.tables_bytes = o;
  o = up8(o);
  c.img = o; o += (size_t)n_obs * 8;
  c.obj = o; o = (o + (size_t)n_obs * 12 + 15) & ~(size_t)15;
  c.intr0 = o; o += 16 * sizeof(double);
  c.up_bytes = o;
  c.pose0 = o; o += 6 * (size_t)S * sizeof(double);
  c.status = o; o += (size_t)S * 4;
  c.res_bytes = o - c.intr0;
  o = up(o);
  c.pts = o; o = up(o + (size_t)(total_floats > 0 ? total_floats : 1) * 4);
  auto dbl = [&](size_t n) { const size_t at = o; o = up(o + n * sizeof(double)); return at; };
  c.W = dbl((size_t)ni * 6 * S);       // pinhole: the cross blocks; fisheye: the pose response Y
  c.rec = dbl((size_t)(ng + ns));
  c.glob1 = dbl((size_t)ng + 2);       // + the two step norms right behind it (one message for a trial point)
  c.dK = dbl(16);
  c.stats = dbl(8);
  if (model == CAMCAL_PINHOLE) {       // CvLevMarq keeps a current and a trial point
    c.intr1 = dbl(16 + 6 * (size_t)S);
    c.pose1 = c.intr1 + 16 * sizeof(double);
    c.V0 = dbl(21 * (size_t)S);
    c.V1 = dbl(21 * (size_t)S);
    c.g0 = dbl(6 * (size_t)S);
    c.g1 = dbl(6 * (size_t)S);
    c.W1 = dbl((size_t)ni * 6 * S);    // spare cross blocks + intrinsics: the LM loop runs half a step ahead (camcal_driver.h)
    c.intr_spare = dbl(16);
  }
  c.flag = dbl(4);                     // camcal_points_off_plane's four words (large sets)
  c.scratch = dbl((size_t)camcal_scratch_len(model, T));
  c.iters = o; o = up(o + (size_t)S * 4);
  c.total = o;
  return c;
}

// pinned staging above this size is not worth its copy (and would pin gigabytes for the large sets)
constexpr size_t PIN_LIMIT = (size_t)8 << 20;
// CAMCAL_PIN_LIMIT (bytes, read once) overrides it: 0 sends every call down the direct-copy path (tests)
inline size_t pin_limit() {
  static const size_t v = [] {
    const char* e = std::getenv("CAMCAL_PIN_LIMIT");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : PIN_LIMIT;
  }();
  return v;
}

// CAMCAL_TRACE_SETUP (environment, read once): the host-side phases of a call's set-up with their wall clock, on stderr
struct PhaseClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseClock() {
    static const bool env = std::getenv("CAMCAL_TRACE_SETUP") != nullptr;
    on = env;
    if (on) t = std::chrono::steady_clock::now();
  }
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[camcal set-up] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// From this many points on a single-rank call tests the planarity of its target on the device, behind the upload
// (one pass at HBM speed, the verdict comes back with the results), instead of walking the host array first: at 1M
// views x 88 corners the host pass costs 28 ms of a 138 ms call.  A multi-rank call keeps the host test: a shard
// that fails must be known before the first collective (agree_on_setup).
constexpr int64_t PLANAR_ON_DEVICE_MIN_OBS = (int64_t)1 << 20;
// CAMCAL_PLANAR_ON_DEVICE_MIN (points, read once) overrides it (tests: 0 sends every single-rank call that way)
inline int64_t planar_on_device_min() {
  static const int64_t v = [] {
    const char* e = std::getenv("CAMCAL_PLANAR_ON_DEVICE_MIN");
    return e ? (int64_t)std::strtoll(e, nullptr, 10) : PLANAR_ON_DEVICE_MIN_OBS;
  }();
  return v;
}

// The point arrays of a large call on their way to the device while the host lays the views out: a thread of its
// own (joined before anything that needs the points is enqueued, and on every way out of the set-up)
struct EarlyUpload {
  std::thread th;
  int rc = 0;
  bool started = false;
  ~EarlyUpload() { if (th.joinable()) th.join(); }
  int wait() {
    if (th.joinable()) th.join();
    return rc;
  }
};
constexpr size_t EARLY_UPLOAD_MIN_BYTES = (size_t)32 << 20;
// CAMCAL_EARLY_UPLOAD_MIN (bytes of object points, read once) overrides it (tests)
inline size_t early_upload_min() {
  static const size_t v = [] {
    const char* e = std::getenv("CAMCAL_EARLY_UPLOAD_MIN");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : EARLY_UPLOAD_MIN_BYTES;
  }();
  return v;
}

struct Solve {
  camcal_problem* p = nullptr;
  const camcal_comm* comm = nullptr;
  void* stream = nullptr;
  int rc = 0;
  // layout
  int32_t T = 0;
  int64_t S = 0, total_floats = 0, n_obs = 0;
  std::vector<char> head;              // host image of the layout tables
  double intr0[16];
  void* pin = nullptr;                 // pinned staging for the one upload and the one download (or NULL: direct copies)
  bool collecting = false;             // the result copies are on the stream
  bool planar_on_device = false;
  int64_t planar_flag[4] = {0, 0, 0, 0};
  ~Solve() { if (pin) camcal_pinned_give(pin); }
  Carve c;
  camcal_tiles tiles{};
  uint8_t mask[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
  int ni = 0, crit_type = 3, max_iter = 30;
  double eps = 0.0;
  std::unique_ptr<camcal_drv::FisheyeGN> gn;
  std::unique_ptr<camcal_drv::PinholeLM> lm;
  // results
  std::vector<char> h_stage;            // [intr | pose | status] when there is no pinned staging

  char* ws() const { return (char*)p->workspace; }
  template <class Tp> Tp* at(size_t off) const { return (Tp*)(ws() + off); }

  // argument checks of the two cv2 calls, as the Python drop-in makes them (api.py: _prep_pinhole / _prep_fisheye)
  int check() {
    if (!p->view_ptr || !p->obj || !p->img || !p->K || !p->D || !p->rt || !p->workspace || p->n_views <= 0)
      return CAMCAL_E_BADARG;
    ni = camcal_n_intr(p->model);
    if (ni < 0) return CAMCAL_E_BADMODEL;
    const double* K = p->K;
    for (int i = 0; i < 16; ++i) intr0[i] = 0.0;
    intr0[0] = K[0]; intr0[1] = K[4]; intr0[2] = K[2]; intr0[3] = K[5];
    const int f = p->flags;
    crit_type = p->crit_type; max_iter = p->max_iter; eps = p->eps;
    if (p->model == CAMCAL_PINHOLE) {
      if (f & (PH_FIX_ASPECT | PH_RATIONAL | PH_THIN_PRISM | PH_TILTED)) return CAMCAL_E_UNSUPPORTED;
      if (!(f & PH_USE_GUESS)) return CAMCAL_E_UNSUPPORTED;
      if (p->n_dist != 4 && p->n_dist != 5) return CAMCAL_E_BADARG;
      if (p->width <= 0 || p->height <= 0) return CAMCAL_E_BADARG;
      if (!(K[0] > 0.0) || !(K[4] > 0.0)) return CAMCAL_E_BADARG;                            // focal lengths
      if (K[2] < 0 || K[2] >= p->width || K[5] < 0 || K[5] >= p->height) return CAMCAL_E_BADARG;   // principal point
      if (std::fabs(K[1]) > 1e-5 || std::fabs(K[3]) > 1e-5 || std::fabs(K[6]) > 1e-5 || std::fabs(K[7]) > 1e-5 ||
          std::fabs(K[8] - 1.0) > 1e-5)
        return CAMCAL_E_BADARG;
      for (int i = 0; i < p->n_dist; ++i) intr0[4 + i] = p->D[i];
      if (f & PH_ZERO_TANGENT) intr0[6] = intr0[7] = 0.0;
      if (f & PH_FIX_FOCAL) mask[0] = mask[1] = 0;
      if (f & PH_FIX_PP) mask[2] = mask[3] = 0;
      if (f & (PH_ZERO_TANGENT | PH_FIX_TANGENT)) mask[6] = mask[7] = 0;
      if (f & PH_FIX_K1) mask[4] = 0;
      if (f & PH_FIX_K2) mask[5] = 0;
      if (f & PH_FIX_K3) mask[8] = 0;
      if (!(crit_type & 1)) max_iter = 1000;
      if (!(crit_type & 2)) eps = 2.220446049250313e-16;
    } else {
      if (f & FE_CHECK_COND) return CAMCAL_E_UNSUPPORTED;
      if (!(f & FE_USE_GUESS) || !(f & FE_FIX_SKEW)) return CAMCAL_E_UNSUPPORTED;
      if (p->n_dist != 4) return CAMCAL_E_BADARG;
      if (K[1] != 0.0) return CAMCAL_E_UNSUPPORTED;
      if (crit_type < 1 || crit_type > 3) return CAMCAL_E_BADARG;
      for (int i = 0; i < 4; ++i) intr0[4 + i] = p->D[i];
      if (f & FE_FIX_FOCAL) mask[0] = mask[1] = 0;
      if (f & FE_FIX_PP) mask[2] = mask[3] = 0;
      const int fk[4] = {FE_FIX_K1, FE_FIX_K2, FE_FIX_K3, FE_FIX_K4};
      for (int j = 0; j < 4; ++j)
        if (f & fk[j]) { mask[4 + j] = 0; intr0[4 + j] = 0.0; }
    }
    return 0;
  }

  // everything up to the first step of the loop, enqueued on the stream
  int setup() {
    CAMCAL_DRV_TRY(check());
    PhaseClock clk;
    const int64_t N = p->n_views;
    if (p->view_ptr[0] != 0) return CAMCAL_E_BADARG;
    std::vector<int32_t> n_per_view((size_t)N);
    for (int64_t v = 0; v < N; ++v) {
      const int64_t n = p->view_ptr[v + 1] - p->view_ptr[v];
      if (n < 4 || n > 2147483647) return CAMCAL_E_BADARG;       // cv2: every view needs at least 4 points
      n_per_view[(size_t)v] = (int32_t)n;
    }
    n_obs = p->view_ptr[N];
    clk.lap("corner counts");
    // Tiles and slots follow from the view count alone, and with them the place of the upload block: a large
    // call starts sending its points now and lays the views out meanwhile (1M views: 4 ms of host work off the path)
    EarlyUpload early;
    {
      const int64_t T0 = (N + 63) / 64;
      const Carve c0 = carve(p->model, N, n_obs, (int32_t)T0, T0 * 64, 0);
      const size_t pin0 = c0.up_bytes > c0.res_bytes ? c0.up_bytes : c0.res_bytes;
      if (pin0 > pin_limit() && (size_t)n_obs * 12 >= early_upload_min() && T0 <= 2147483647) {
        if ((int64_t)c0.up_bytes > p->workspace_bytes) return CAMCAL_E_TOOBIG;
        if (((uintptr_t)p->workspace & (ALIGN - 1)) != 0) return CAMCAL_E_BADARG;
        int dev = 0;
        CAMCAL_DRV_TRY(camcal_get_device(&dev));
        EarlyUpload* e = &early;
        const camcal_problem* q = p;
        char* base = ws();
        void* st = stream;
        const int64_t n = n_obs;
        try {
          early.th = std::thread([e, q, base, st, n, c0, dev] {
            if (camcal_set_device(dev) != 0) { e->rc = CAMCAL_E_BADARG; return; }
            e->rc = camcal_upload_bulk(base + c0.img, q->img, (size_t)n * 8, st);
            if (!e->rc) e->rc = camcal_upload_bulk(base + c0.obj, q->obj, (size_t)n * 12, st);
          });
          early.started = true;
        } catch (...) {                              // no thread to be had: the uploads go out below, as before
        }
      }
    }
    planar_on_device = !comm && n_obs >= planar_on_device_min();
    if (!planar_on_device)
      for (int64_t i = 0; i < n_obs; ++i)
        if (p->obj[i * 3 + 2] != 0.f) return CAMCAL_E_UNSUPPORTED;  // planar targets only (the reference's boards)
    clk.lap("planarity test");
    CAMCAL_DRV_TRY(camcal_layout(N, n_per_view.data(), &T, &S, &total_floats, nullptr, nullptr, nullptr, nullptr));
    c = carve(p->model, N, n_obs, T, S, total_floats);
    if ((int64_t)c.total > p->workspace_bytes) return CAMCAL_E_TOOBIG;
    if (((uintptr_t)p->workspace & (ALIGN - 1)) != 0) return CAMCAL_E_BADARG;
    head.assign(c.tables_bytes, 0);
    CAMCAL_DRV_TRY(camcal_layout(N, n_per_view.data(), &T, &S, &total_floats, (int32_t*)(head.data() + c.slot_view),
                                 (int32_t*)(head.data() + c.lane_n), (int32_t*)(head.data() + c.tile_nmax),
                                 (int64_t*)(head.data() + c.tile_off)));
    std::memcpy(head.data() + c.view_ptr, p->view_ptr, (size_t)(N + 1) * 8);
    clk.lap("layout (sizes, then tables)");
    const size_t pin_bytes = c.up_bytes > c.res_bytes ? c.up_bytes : c.res_bytes;
    if (pin_bytes <= pin_limit()) pin = camcal_pinned_take(pin_bytes);
    if (pin) {                                    // one block, one copy
      char* u = (char*)pin;
      std::memcpy(u, head.data(), c.tables_bytes);
      std::memcpy(u + c.img, p->img, (size_t)n_obs * 8);
      std::memcpy(u + c.obj, p->obj, (size_t)n_obs * 12);
      std::memcpy(u + c.intr0, intr0, sizeof(intr0));
      CAMCAL_DRV_TRY(camcal_upload(ws(), u, c.up_bytes, stream));
    } else {
      CAMCAL_DRV_TRY(camcal_upload(ws(), head.data(), c.tables_bytes, stream));
      if (early.started) {
        CAMCAL_DRV_TRY(early.wait());
      } else {
        CAMCAL_DRV_TRY(camcal_upload_bulk(ws() + c.img, p->img, (size_t)n_obs * 8, stream));
        CAMCAL_DRV_TRY(camcal_upload_bulk(ws() + c.obj, p->obj, (size_t)n_obs * 12, stream));
      }
      CAMCAL_DRV_TRY(camcal_upload(ws() + c.intr0, intr0, sizeof(intr0), stream));
    }
    clk.lap("staging + upload calls");
    if (planar_on_device) CAMCAL_DRV_TRY(camcal_points_off_plane(n_obs, at<float>(c.obj), at<int64_t>(c.flag), stream));
    tiles.pts = at<float>(c.pts);
    tiles.tile_off = at<int64_t>(c.tile_off);
    tiles.tile_nmax = at<int32_t>(c.tile_nmax);
    tiles.lane_n = at<int32_t>(c.lane_n);
    tiles.n_tiles = T;
    tiles.n_slots = S;
    tiles.planar = 1;
    tiles.lanes_per_view = p->lanes_per_view;
    CAMCAL_DRV_TRY(camcal_pack_tiles(at<float>(c.obj), at<float>(c.img), at<int64_t>(c.view_ptr), at<int32_t>(c.slot_view),
                                     &tiles, at<float>(c.pts), stream));
    double* intr = at<double>(c.intr0);
    double* pose = at<double>(c.pose0);
    // cvFindExtrinsicCameraParams2 / CalibrateExtrinsics per view: homography start + refinement
    CAMCAL_DRV_TRY(camcal_pose_init(p->model, &tiles, intr, pose, at<int32_t>(c.status), stream));
    // (+ the first refinement: the fisheye loop folds it into its first launch where it can)
    if (p->model != CAMCAL_FISHEYE)
      CAMCAL_DRV_TRY(camcal_pose_refine(p->model, &tiles, intr, pose, 20, 1e-10, at<int32_t>(c.iters), stream));
    if (p->model == CAMCAL_FISHEYE) {
      gn.reset(new camcal_drv::FisheyeGN());
      gn->refine_first = true;
      gn->tiles = &tiles; gn->intr = intr; gn->pose = pose; gn->mask_host = mask; gn->crit_type = crit_type;
      gn->max_count = max_iter; gn->eps = eps; gn->recompute_extrinsic = (p->flags & FE_RECOMPUTE) ? 1 : 0;
      gn->rec = at<double>(c.rec); gn->dK = at<double>(c.dK); gn->stats = at<double>(c.stats);
      gn->iters = at<int32_t>(c.iters); gn->Y = at<double>(c.W); gn->scratch = at<double>(c.scratch);
      gn->comm = comm; gn->stream = stream;
      if (!gn->args_ok()) return CAMCAL_E_BADARG;
    } else {
      lm.reset(new camcal_drv::PinholeLM());
      lm->tiles = &tiles;
      lm->intr2[0] = intr; lm->intr2[1] = at<double>(c.intr1);
      lm->pose2[0] = pose; lm->pose2[1] = at<double>(c.pose1);
      lm->V2[0] = at<double>(c.V0); lm->V2[1] = at<double>(c.V1);
      lm->g2[0] = at<double>(c.g0); lm->g2[1] = at<double>(c.g1);
      lm->W = at<double>(c.W); lm->rec = at<double>(c.rec); lm->glob_trial = at<double>(c.glob1);
      lm->norms = at<double>(c.glob1) + camcal_glob_len(CAMCAL_PINHOLE);
      lm->dK = at<double>(c.dK); lm->stats = at<double>(c.stats); lm->scratch = at<double>(c.scratch);
      lm->W_alt = at<double>(c.W1); lm->intr_spare = at<double>(c.intr_spare);
      lm->mask_host = mask; lm->max_iter = max_iter; lm->eps = eps; lm->comm = comm; lm->stream = stream;
      if (!lm->args_ok()) return CAMCAL_E_BADARG;
    }
    return 0;
  }

  // Multi-rank call: did the set-up succeed on EVERY rank?  A shard can be rejected for reasons the other
  // ranks do not share (a view with three corners, a workspace that is too small); without this exchange
  // they would wait in the loop's first all-reduce for a rank that has already returned.
  int agree_on_setup(int rc_setup) {
    if (!comm) return rc_setup;
    double* d = nullptr;
    if (gn || lm) d = at<double>(c.stats);
    else if (p->workspace && p->workspace_bytes >= 256 && ((uintptr_t)p->workspace & (ALIGN - 1)) == 0) d = (double*)p->workspace;
    if (!d) return rc_setup ? rc_setup : CAMCAL_E_BADARG;      // no device memory at all to take part with (documented)
    double failed = rc_setup ? 1.0 : 0.0;
    CAMCAL_DRV_TRY(camcal_upload(d, &failed, sizeof(failed), stream));
    CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, d, 1, stream));
    CAMCAL_DRV_TRY(camcal_download(&failed, d, sizeof(failed), stream));
    CAMCAL_DRV_TRY(camcal_stream_sync(stream));
    if (rc_setup) return rc_setup;
    return failed != 0.0 ? CAMCAL_E_PEER : 0;
  }

  bool done() const { return gn ? gn->done() : lm ? lm->done() : true; }
  int launch() { return gn ? gn->launch() : lm->launch(); }
  int finish() { return gn ? gn->finish() : lm->finish(); }

  // the loop has launched its last piece of work (the results on the device are final)
  bool final_launched() const { return gn && gn->pending == camcal_drv::FisheyeGN::FINAL_COST; }

  // results: enqueue the copies ([intr | pose | status], into the pinned staging when there is one) ...
  const double* r_intr = nullptr;
  const double* r_pose = nullptr;
  const int32_t* r_status = nullptr;
  int collect_begin() {
    if (collecting) return 0;
    collecting = true;
    const int cur = lm ? lm->cur : 0;
    const size_t pose_bytes = (size_t)6 * S * sizeof(double), st_bytes = (size_t)S * sizeof(int32_t);
    char* dst = (char*)pin;
    if (!dst) {
      h_stage.assign(c.res_bytes, 0);
      dst = h_stage.data();
    }
    r_intr = (const double*)dst;
    r_pose = (const double*)(dst + 128);
    r_status = (const int32_t*)(dst + 128 + pose_bytes);
    if (planar_on_device) CAMCAL_DRV_TRY(camcal_download(planar_flag, ws() + c.flag, sizeof(planar_flag), stream));
    if (cur == 0) return camcal_download(dst, ws() + c.intr0, c.res_bytes, stream);     // one block
    CAMCAL_DRV_TRY(camcal_download(dst, ws() + c.intr1, 128 + pose_bytes, stream));
    return camcal_download(dst + 128 + pose_bytes, ws() + c.status, st_bytes, stream);
  }
  // ... and, after the stream has drained, hand them out in the caller's order
  int collect_end(int loop_rc) {
    if (planar_on_device && planar_flag[2] != 0) return CAMCAL_E_UNSUPPORTED;   // planar targets only; nothing of this solve is handed out
    PhaseClock clk_end;
    const int32_t* slot_view = (const int32_t*)(head.data() + c.slot_view);
    // slot order -> the caller's view order (every view sits in exactly one slot: the ranges write disjoint rows);
    // a large set is split over a few threads (1M views: 6 ms of a single thread's loads and stores)
    auto reorder = [&](int64_t s0, int64_t s1, int64_t* bad_out) {
      int64_t bad = 0;
      for (int64_t s = s0; s < s1; ++s) {
        const int32_t v = slot_view[s];
        if (v < 0) continue;
        for (int k = 0; k < 6; ++k) p->rt[(size_t)v * 6 + k] = r_pose[(size_t)k * S + s];
        if (p->view_status) p->view_status[v] = r_status[(size_t)s];
        bad += r_status[(size_t)s] != 0;
      }
      *bad_out = bad;
    };
    constexpr int RT = 4;
    int64_t bad_part[RT] = {0, 0, 0, 0};
    if (S >= ((int64_t)1 << 16)) {
      std::vector<std::thread> th;
      const int64_t per = (S + RT - 1) / RT;
      int started = 1;
      try {
        for (; started < RT; ++started)
          th.emplace_back(reorder, started * per, (started + 1) * per < S ? (started + 1) * per : S, &bad_part[started]);
      } catch (...) {
      }
      reorder(0, per < S ? per : S, &bad_part[0]);
      for (int t = started; t < RT; ++t) reorder(t * per, (t + 1) * per < S ? (t + 1) * per : S, &bad_part[t]);
      for (auto& x : th) x.join();
    } else {
      reorder(0, S, &bad_part[0]);
    }
    const int64_t n_bad = bad_part[0] + bad_part[1] + bad_part[2] + bad_part[3];
    clk_end.lap("view order");
    // totals over the ranks (multi-rank call): observations for the rms, degenerate views for a collective verdict
    double tot[2] = {(double)n_obs, (double)n_bad};
    if (comm) {
      double* d = at<double>(c.stats);
      CAMCAL_DRV_TRY(camcal_upload(d, tot, sizeof(tot), stream));
      CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, d, 2, stream));
      CAMCAL_DRV_TRY(camcal_download(tot, d, sizeof(tot), stream));
      CAMCAL_DRV_TRY(camcal_stream_sync(stream));
    }
    if (tot[1] != 0.0) return n_bad ? CAMCAL_E_NUMERIC : CAMCAL_E_PEER;
    if (loop_rc) return loop_rc;
    const double cost = gn ? gn->cost : lm->err;
    p->rms = std::sqrt(cost / tot[0]);
    p->iterations = gn ? gn->it : lm->iters;
    double* K = p->K;
    const double* h_intr = r_intr;
    K[0] = h_intr[0]; K[1] = 0.0; K[2] = h_intr[2];
    K[3] = 0.0; K[4] = h_intr[1]; K[5] = h_intr[3];
    K[6] = 0.0; K[7] = 0.0; K[8] = 1.0;
    const int nd = p->model == CAMCAL_PINHOLE ? 5 : 4;
    for (int i = 0; i < nd; ++i) p->D[i] = h_intr[4 + i];
    return 0;
  }
};

inline void fail(camcal_problem* p, int rc) {
  p->status = rc;
  p->rms = std::nan("");
  p->iterations = 0;
}

// one problem from its host arrays to the point where its results sit in host staging: set-up, loop,
// read-back, drained stream.  rc: the solve's own status; the return value: whether the read-back worked
inline int run_to_readback(Solve& s) {
  s.rc = s.agree_on_setup(s.setup());
  PhaseClock clk;
  if (s.rc == 0) {
    while (!s.done()) {
      int rc = s.launch();
      // the result copies go out behind the loop's last launch, before the host waits for its scalar
      if (rc == 0 && s.final_launched()) rc = s.collect_begin();
      if (rc == 0) rc = s.finish();
      if (rc) { s.rc = rc; break; }
    }
  }
  if (!(s.gn || s.lm) || s.rc == CAMCAL_E_PEER) return 0;   // rejected before any device work / with the other ranks
  clk.lap("loop (host side)");
  int rc = s.collect_begin();                  // got as far as the device: read back what there is
  if (rc == 0) rc = camcal_stream_sync(s.stream);
  clk.lap("results on the host");
  return rc;
}

}  // namespace camcal_cal
}  // namespace

extern "C" {

int64_t camcal_workspace_len(int model, int64_t n_views, const int64_t* view_ptr_host) {
  if (camcal_n_intr(model) < 0) return CAMCAL_E_BADMODEL;
  if (n_views <= 0 || !view_ptr_host) return CAMCAL_E_BADARG;
  std::vector<int32_t> n((size_t)n_views);
  for (int64_t v = 0; v < n_views; ++v) {
    const int64_t k = view_ptr_host[v + 1] - view_ptr_host[v];
    if (k < 0 || k > 2147483647) return CAMCAL_E_BADARG;
    n[(size_t)v] = (int32_t)k;
  }
  int32_t T = 0;
  int64_t S = 0, total = 0;
  const int rc = camcal_layout(n_views, n.data(), &T, &S, &total, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  return (int64_t)camcal_cal::carve(model, n_views, view_ptr_host[n_views], T, S, total).total;
}

int camcal_calibrate_many(camcal_problem* problems, int n, void* stream) {
  if (!problems || n <= 0) return CAMCAL_E_BADARG;
  using camcal_cal::Solve;
  std::vector<std::unique_ptr<Solve>> sv;
  // independent problems: one stream each (at most CAMCAL_RIG_MAX_CAMS distinct ones, shared round robin beyond)
  void* side[CAMCAL_RIG_MAX_CAMS];
  const int n_streams = n < CAMCAL_RIG_MAX_CAMS ? n : CAMCAL_RIG_MAX_CAMS;
  int dev = 0;
  CAMCAL_DRV_TRY(camcal_get_device(&dev));
  CAMCAL_DRV_TRY(camcal_fork_streams(stream, n_streams, side));
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<Solve> s(new Solve());
    s->p = &problems[i];
    s->stream = side[i % n_streams];
    problems[i].status = 0;
    sv.push_back(std::move(s));
  }
  std::vector<int> rb((size_t)n, 0);
  auto work = [&](int i, bool own_thread) {
    if (own_thread && camcal_set_device(dev) != 0) { sv[(size_t)i]->rc = CAMCAL_E_BADARG; return; }
    rb[(size_t)i] = camcal_cal::run_to_readback(*sv[(size_t)i]);
  };
  {
    std::vector<std::thread> th;
    int started = 1;                               // problem 0 runs on the calling thread
    try {
      for (; started < n; ++started) th.emplace_back(work, started, true);
    } catch (...) {                                // no more threads to be had: the rest runs here, after problem 0
    }
    work(0, false);
    for (int i = started; i < n; ++i) work(i, false);
    for (auto& t : th) t.join();
  }
  int first = camcal_join_streams(stream, n_streams, side);
  for (int i = 0; i < n; ++i) {
    Solve& s = *sv[(size_t)i];
    int rc = s.rc;
    if ((s.gn || s.lm) && s.rc != CAMCAL_E_PEER) rc = rb[(size_t)i] ? (s.rc ? s.rc : rb[(size_t)i]) : s.collect_end(s.rc);
    if (rc) camcal_cal::fail(s.p, rc);
    if (rc && !first) first = rc;
  }
  return first;
}

int camcal_calibrate(int model, int64_t n_views, const int64_t* view_ptr_host, const float* obj_host,
                     const float* img_host, int width, int height, double* K_host, double* D_host, int n_dist,
                     double* rt_host, int32_t* view_status_host, int flags, int crit_type, int max_iter, double eps,
                     void* workspace, int64_t workspace_bytes, const camcal_comm* comm, void* stream,
                     double* rms_out_host, int32_t* iterations_out_host) {
  if (!rms_out_host || !iterations_out_host) return CAMCAL_E_BADARG;
  camcal_problem p;
  std::memset(&p, 0, sizeof(p));
  p.model = model; p.flags = flags; p.width = width; p.height = height; p.crit_type = crit_type; p.max_iter = max_iter;
  p.eps = eps; p.n_views = n_views; p.view_ptr = view_ptr_host; p.obj = obj_host; p.img = img_host; p.K = K_host;
  p.D = D_host; p.n_dist = n_dist; p.rt = rt_host; p.view_status = view_status_host; p.workspace = workspace;
  p.workspace_bytes = workspace_bytes;
  camcal_cal::Solve s;
  s.p = &p;
  s.comm = comm;
  s.stream = stream;
  const int rb = camcal_cal::run_to_readback(s);
  int rc = s.rc;
  if ((s.gn || s.lm) && rc != CAMCAL_E_PEER) rc = rb ? (rc ? rc : rb) : s.collect_end(rc);
  if (rc) camcal_cal::fail(&p, rc);
  *rms_out_host = p.rms;
  *iterations_out_host = p.iterations;
  return rc;
}

}  // extern "C"
