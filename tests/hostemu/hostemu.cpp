// CPU lane emulator of libcamcal.so -- TEST INFRASTRUCTURE ONLY.
//
// Exposes the same C ABI as include/camcal.h but with HOST pointers, and runs
// the very same per-lane bodies (cameracalibration_amd/csrc/camcal_lane.h) in
// a loop over tiles and lanes.  It lets the CPU-only test tier (-m "not gpu")
// check the lane math against the oracle and drive the Python solver with the
// gloo backend at world_size 2.  The product never loads this library: its
// loader (cameracalibration_amd/_lib.py) only opens the HIP build and fails
// loudly when that is missing.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/camcal.h"
#include "../../cameracalibration_amd/csrc/camcal_lane.h"
#include "../../cameracalibration_amd/csrc/camcal_rig.h"
#include "../../cameracalibration_amd/csrc/camcal_points.h"

using namespace camcal;

// The loops over tiles / poses run on all host cores when the file is built with -fopenmp (the oracle's
// CPU-baseline build, oracle/Makefile: libcamcal_cpu.so); the tests' build is serial.  Either way sums are
// formed per tile and added up in tile order, so results do not depend on the number of threads.
#ifdef _OPENMP
#define CC_OMP_FOR _Pragma("omp parallel for schedule(dynamic, 1)")
#else
#define CC_OMP_FOR
#endif

namespace {
inline LaneTile lane_tile(const camcal_tiles* t, int tile, int lane) {
  LaneTile tv;
  tv.base = t->pts + t->tile_off[tile] + lane;
  tv.n = t->lane_n[(int64_t)tile * LANES + lane];
  tv.nmax = t->tile_nmax[tile];
  return tv;
}
inline void load_pose(const double* pose, int64_t S, int64_t slot, double p[6]) {
  for (int c = 0; c < 6; ++c) p[c] = pose[c * S + slot];
}

template <int MODEL, int MODE>
void intr_normal(const camcal_tiles* t, const double* intr, const double* pose, double* glob, double* view_cost,
                 double* Vb, double* gb) {
  constexpr bool NORMAL = MODE > 0;
  constexpr int NV = IntrAcc<MODEL>::NV, NR = IntrAcc<MODEL>::NR;
  // partial sums per tile, added up in tile order: the same result for every thread count
  std::vector<double> part((size_t)t->n_tiles * NR, 0.0), tot(NR, 0.0);
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile) {
    double* pt = part.data() + (size_t)tile * NR;
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      double p[6];
      load_pose(pose, t->n_slots, slot, p);
      IntrAcc<MODEL> acc;
      acc.zero();
      double V[21], g[6];
      const double vc = t->planar ? lane_intr_normal<MODEL, MODE, true>(lane_tile(t, tile, lane), intr, p, acc, V, g)
                                  : lane_intr_normal<MODEL, MODE, false>(lane_tile(t, tile, lane), intr, p, acc, V, g);
      if (view_cost) view_cost[slot] = vc;
      if (MODE == 2) {
        for (int e = 0; e < 21; ++e) Vb[e * t->n_slots + slot] = V[e];
        for (int e = 0; e < 6; ++e) gb[e * t->n_slots + slot] = g[e];
      }
      for (int i = 0; i < NR; ++i) pt[i] += acc.v[i];
    }
  }
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int i = 0; i < NR; ++i) tot[i] += part[(size_t)tile * NR + i];
  if (NORMAL) IntrAcc<MODEL>::expand(tot.data(), glob);
  else glob[NV - 1] = tot[IntrAcc<MODEL>::COST];
}

template <int MODEL>
void view_blocks(const camcal_tiles* t, const double* intr, const double* pose, double* V, double* W, double* g) {
  constexpr int NI = MT<MODEL>::NI;
  const int64_t S = t->n_slots;
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      double p[6];
      load_pose(pose, S, slot, p);
      ViewAcc<MODEL> acc;
      if (t->planar) lane_view_blocks<MODEL, true>(lane_tile(t, tile, lane), intr, p, acc);
      else lane_view_blocks<MODEL, false>(lane_tile(t, tile, lane), intr, p, acc);
      for (int e = 0; e < 21; ++e) V[e * S + slot] = acc.V[e];
      for (int e = 0; e < NI * 6; ++e) W[e * S + slot] = acc.W[e];
      for (int e = 0; e < 6; ++e) g[e * S + slot] = acc.g[e];
    }
}

template <int NI>
void schur(const camcal_tiles* t, const double* V, const double* W, const double* g, double lam, double* out) {
  constexpr int NV = trisz(NI) + NI;
  std::vector<double> part((size_t)t->n_tiles * NV, 0.0);
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int64_t slot = (int64_t)tile * LANES; slot < (int64_t)(tile + 1) * LANES; ++slot)
      if (t->lane_n[slot] > 0) lane_schur<NI>(V, W, g, t->n_slots, slot, lam, part.data() + (size_t)tile * NV);
  for (int i = 0; i < NV; ++i) out[i] = 0.0;
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int i = 0; i < NV; ++i) out[i] += part[(size_t)tile * NV + i];
}

template <int NI>
void backsub(const camcal_tiles* t, const double* V, const double* W, const double* g, double lam, const double* dK,
             const double* pin, double* pout, double* norms) {
  const int64_t S = t->n_slots;
  std::vector<double> part((size_t)t->n_tiles * 2, 0.0);
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int64_t slot = (int64_t)tile * LANES; slot < (int64_t)(tile + 1) * LANES; ++slot) {
      double prev[6], out[6];
      load_pose(pin, S, slot, prev);
      if (t->lane_n[slot] > 0) {
        double dd, pp;
        lane_backsub<NI>(V, W, g, S, slot, lam, dK, prev, out, dd, pp);
        part[(size_t)tile * 2] += dd; part[(size_t)tile * 2 + 1] += pp;
      } else {
        for (int c = 0; c < 6; ++c) out[c] = prev[c];
      }
      for (int c = 0; c < 6; ++c) pout[c * S + slot] = out[c];
    }
  double a0 = 0, a1 = 0;
  for (int tile = 0; tile < t->n_tiles; ++tile) { a0 += part[(size_t)tile * 2]; a1 += part[(size_t)tile * 2 + 1]; }
  norms[0] = a0; norms[1] = a1;
}

template <int MODEL>
void pose_refine(const camcal_tiles* t, const double* intr, double* pose, int max_iter, double tol, int32_t* iters) {
  const int64_t S = t->n_slots;
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const LaneTile tv = lane_tile(t, tile, lane);
      if (tv.n < 4) { if (iters) iters[slot] = 0; continue; }
      double p[6];
      load_pose(pose, S, slot, p);
      const int it = t->planar ? lane_pose_refine<MODEL, true>(tv, intr, p, max_iter, tol)
                               : lane_pose_refine<MODEL, false>(tv, intr, p, max_iter, tol);
      for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
      if (iters) iters[slot] = it;
    }
}

template <int MODEL>
void pose_init(const camcal_tiles* t, const double* intr, double* pose, int32_t* status) {
  const int64_t S = t->n_slots;
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const LaneTile tv = lane_tile(t, tile, lane);
      double p[6] = {0, 0, 0, 0, 0, 1};
      const int st = lane_pose_init<MODEL>(tv, intr, p);
      for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
      if (status) status[slot] = tv.n == 0 ? 0 : st;
    }
}
template <int MODEL>
static void cross_schur(const camcal_tiles* t, const double* intr, const double* pose, const double* Vb,
                        const double* gb, double lam, double* Wb, double* out) {
  constexpr int NI = MT<MODEL>::NI, NS = trisz(NI) + NI;
  const int64_t S = t->n_slots;
  std::vector<double> part((size_t)t->n_tiles * NS, 0.0);
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      double p[6], W[NI * 6];
      load_pose(pose, S, slot, p);
      const LaneTile tv = lane_tile(t, tile, lane);
      if (t->planar) lane_cross_block<MODEL, true>(tv, intr, p, W); else lane_cross_block<MODEL, false>(tv, intr, p, W);
      if (Wb) for (int e = 0; e < NI * 6; ++e) Wb[e * S + slot] = W[e];
      if (tv.n > 0) {
        double V[21], g[6];
        for (int e = 0; e < 21; ++e) V[e] = Vb[e * S + slot];
        for (int e = 0; e < 6; ++e) g[e] = gb[e * S + slot];
        lane_schur_regs<NI>(V, W, g, lam, part.data() + (size_t)tile * NS);
      }
    }
  for (int i = 0; i < NS; ++i) out[i] = 0.0;
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int i = 0; i < NS; ++i) out[i] += part[(size_t)tile * NS + i];
}
// ---- rig (include/camcal.h "Multi-camera rig"): the same per-lane bodies, looped -------------
template <int MODEL>
void rig_eval_cam(const camcal_rig* rg, int b, int cam, double* red) {
  constexpr int NI = MT<MODEL>::NI, NR = IntrAcc<MODEL>::NR;
  const camcal_rig_cam& k = rg->cam[cam];
  const camcal_tiles* t = &k.tiles;
  const int64_t S = t->n_slots;
  std::vector<double> tot(NR, 0.0), part((size_t)t->n_tiles * NR, 0.0);
  const double* I = rg->intr[b] + cam * 16;
  const double* e = rg->ext[b] + cam * 6;
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      double p[6] = {0, 0, 0, 0, 0, 1};
      const int idx = k.slot_pose[slot];
      if (idx >= 0) {
        double pp[6];
        for (int c = 0; c < 6; ++c) pp[c] = rg->rig_pose[b][c * rg->P + idx];
        rig_compose(pp, e, p);
      }
      ViewAcc<MODEL> acc;
      IntrAcc<MODEL> gacc;
      gacc.zero();
      if (t->planar) lane_normal_fused<MODEL, true, true>(lane_tile(t, tile, lane), I, p, acc, gacc);
      else lane_normal_fused<MODEL, false, true>(lane_tile(t, tile, lane), I, p, acc, gacc);
      for (int q = 0; q < 21; ++q) k.V[q * S + slot] = acc.V[q];
      for (int q = 0; q < NI * 6; ++q) k.W[q * S + slot] = acc.W[q];
      for (int q = 0; q < 6; ++q) k.g[q * S + slot] = acc.g[q];
      for (int i = 0; i < NR; ++i) part[(size_t)tile * NR + i] += gacc.v[i];
    }
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int i = 0; i < NR; ++i) tot[i] += part[(size_t)tile * NR + i];
  for (int i = 0; i < NR; ++i) red[i] = tot[i];
}

template <int NI>
void rig_assemble_cam(const camcal_rig* rg, int b, int cam, double* rec_ext) {
  constexpr int NACC = NI * 6 + 27;
  const camcal_rig_cam& k = rg->cam[cam];
  const int64_t S = k.tiles.n_slots, P = rg->P;
  const int64_t nch = (S + 63) / 64;
  std::vector<double> part((size_t)nch * NACC, 0.0);
  double* Vp = rg->Vp[b] + (int64_t)cam * 21 * P;
  double* gp = rg->gp[b] + (int64_t)cam * 6 * P;
  double* Wp = rg->Wp[b] + (int64_t)cam * (NI + 6) * 6 * P;
  CC_OMP_FOR
  for (int64_t ch = 0; ch < nch; ++ch)
  for (int64_t slot = ch * 64; slot < (ch + 1) * 64 && slot < S; ++slot) {
    double* acc = part.data() + (size_t)ch * NACC;
    const int idx = k.slot_pose[slot];
    if (idx < 0) continue;
    double pp[6], Vl[21], Wl[NI * 6], gl[6], oV[21], og[6], oW[(NI + 6) * 6];
    for (int c = 0; c < 6; ++c) pp[c] = rg->rig_pose[b][c * P + idx];
    for (int i = 0; i < 21; ++i) Vl[i] = k.V[i * S + slot];
    for (int i = 0; i < NI * 6; ++i) Wl[i] = k.W[i * S + slot];
    for (int i = 0; i < 6; ++i) gl[i] = k.g[i * S + slot];
    RigChain ch;
    rig_chain(pp, rg->ext[b] + cam * 6, ch);
    rig_assemble_pose(Vl, gl, ch, oV, og, oW + NI * 6, acc + NI * 6);
    rig_assemble_intr<NI>(Wl, ch, oW, acc);
    for (int i = 0; i < 21; ++i) Vp[i * P + idx] = oV[i];
    for (int i = 0; i < 6; ++i) gp[i * P + idx] = og[i];
    for (int i = 0; i < (NI + 6) * 6; ++i) Wp[i * P + idx] = oW[i];
  }
  for (int i = 0; i < NACC; ++i) rec_ext[i] = 0.0;
  for (int64_t ch = 0; ch < nch; ++ch)
    for (int i = 0; i < NACC; ++i) rec_ext[i] += part[(size_t)ch * NACC + i];
}

template <int G>
void rig_pose_schur(int n_cams, int64_t P, const double* Vp, const double* gp, const double* Wp, double lam, double* Yt,
                    double* z) {
  CC_OMP_FOR
  for (int64_t p = 0; p < P; ++p) {
    double U[21], g[6];
    double id[6];
    const bool ok = rig_pose_factor<G>(n_cams, Vp, gp, P, p, lam, U, g, id);
    if (ok) chol6_fwd(U, id, g);
    for (int k = 0; k < 6; ++k) z[k * P + p] = ok ? g[k] : 0.0;
    for (int v = 0; v < n_cams * G; ++v) {
      double y[6];
      for (int k = 0; k < 6; ++k) y[k] = Wp[((int64_t)v * 6 + k) * P + p];
      if (ok) chol6_fwd(U, id, y);
      for (int k = 0; k < 6; ++k) Yt[((int64_t)v * 6 + k) * P + p] = ok ? y[k] : 0.0;
    }
  }
}

template <int G>
void rig_backsub(int n_cams, int64_t P, const double* Vp, const double* gp, const double* Wp, double lam,
                 const double* dG, const double* pin, double* pout, double* norms) {
  const int64_t nch = (P + 63) / 64;
  std::vector<double> part((size_t)nch * 2, 0.0);
  CC_OMP_FOR
  for (int64_t ch = 0; ch < nch; ++ch)
  for (int64_t p = ch * 64; p < (ch + 1) * 64 && p < P; ++p) {
    double& a0 = part[(size_t)ch * 2];
    double& a1 = part[(size_t)ch * 2 + 1];
    double U[21], rhs[6];
    double id[6];
    const bool ok = rig_pose_factor<G>(n_cams, Vp, gp, P, p, lam, U, rhs, id);
    if (ok) {
      for (int c = 0; c < n_cams; ++c)
        for (int a = 0; a < G; ++a)
          for (int k = 0; k < 6; ++k) rhs[k] -= Wp[(((int64_t)c * G + a) * 6 + k) * P + p] * dG[c * G + a];
      chol6_fwd(U, id, rhs);
      chol6_bwd(U, id, rhs);
    }
    for (int k = 0; k < 6; ++k) {
      const double prev = pin[k * P + p], d = ok ? rhs[k] : 0.0;
      pout[k * P + p] = prev - d;
      a0 += d * d; a1 += prev * prev;
    }
  }
  double s0 = 0, s1 = 0;
  for (int64_t ch = 0; ch < nch; ++ch) { s0 += part[(size_t)ch * 2]; s1 += part[(size_t)ch * 2 + 1]; }
  norms[0] = s0; norms[1] = s1;
}
}  // namespace

namespace {
template <int MODEL>
void fused_response(const camcal_tiles* t, const double* intr, const double* pose, double* Y, double* glob,
                    double* schur_out) {
  constexpr int NI = MT<MODEL>::NI, NR = IntrAcc<MODEL>::NR, NS = trisz(NI) + NI;
  const int64_t S = t->n_slots;
  std::vector<double> tot(NR, 0.0), part((size_t)t->n_tiles * (NR + NS), 0.0);
  double sacc[NS];
  for (int i = 0; i < NS; ++i) sacc[i] = 0.0;
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      double* pt = part.data() + (size_t)tile * (NR + NS);
      const int64_t slot = (int64_t)tile * LANES + lane;
      const LaneTile tv = lane_tile(t, tile, lane);
      double p[6];
      load_pose(pose, S, slot, p);
      ViewAcc<MODEL> acc;
      IntrAcc<MODEL> gacc;
      gacc.zero();
      if (t->planar) lane_normal_fused<MODEL, true>(tv, intr, p, acc, gacc);
      else lane_normal_fused<MODEL, false>(tv, intr, p, acc, gacc);
      for (int i = 0; i < NR; ++i) pt[i] += gacc.v[i];
      if (tv.n <= 0) continue;
      double id[6];
      lane_schur_regs<NI>(acc.V, acc.W, acc.g, 0.0, pt + NR, id);
      lane_pose_response<NI>(acc.V, id, acc.W);
      for (int e = 0; e < NI * 6; ++e) Y[e * S + slot] = acc.W[e];
    }
  for (int tile = 0; tile < t->n_tiles; ++tile) {
    const double* pt = part.data() + (size_t)tile * (NR + NS);
    for (int i = 0; i < NR; ++i) tot[i] += pt[i];
    for (int i = 0; i < NS; ++i) sacc[i] += pt[NR + i];
  }
  IntrAcc<MODEL>::expand(tot.data(), glob);
  for (int i = 0; i < NS; ++i) schur_out[i] = sacc[i];
}
}  // namespace

extern "C" {
int camcal_version(void) { return -100; }   // negative: emulator

// the emulator always walks one view per lane; camcal_tiles.lanes_per_view is ignored
int camcal_n_intr(int model) { return model == 0 ? 9 : model == 1 ? 8 : CAMCAL_E_BADMODEL; }
int camcal_glob_len(int model) { const int ni = camcal_n_intr(model); return ni < 0 ? ni : ni * (ni + 1) / 2 + ni + 1; }
int camcal_schur_len(int model) { const int ni = camcal_n_intr(model); return ni < 0 ? ni : ni * (ni + 1) / 2 + ni; }
int64_t camcal_scratch_len(int, int) { return 256; }   // room for one finished record (camcal_refine_normal_eq_partial)

int camcal_layout(int64_t n_views, const int32_t* n_per_view, int32_t* n_tiles, int64_t* n_slots,
                  int64_t* total_floats, int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax, int64_t* tile_off) {
  if (n_views < 0 || (n_views > 0 && !n_per_view) || !n_tiles || !n_slots || !total_floats) return CAMCAL_E_BADARG;
  if (slot_view && (!lane_n || !tile_nmax || !tile_off)) return CAMCAL_E_BADARG;
  const int64_t T = n_views > 0 ? (n_views + LANES - 1) / LANES : 1;
  if (T > 2147483647) return CAMCAL_E_TOOBIG;
  *n_tiles = (int32_t)T;
  *n_slots = T * LANES;
  std::vector<int32_t> order((size_t)n_views);
  for (int64_t i = 0; i < n_views; ++i) {
    if (n_per_view[i] < 0) return CAMCAL_E_BADARG;
    order[(size_t)i] = (int32_t)i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n_per_view[a] > n_per_view[b]; });
  int64_t off = 0;
  for (int64_t t = 0; t < T; ++t) {
    int32_t nmax = 0;
    for (int l = 0; l < LANES; ++l) {
      const int64_t s = t * LANES + l;
      const int32_t v = s < n_views ? order[(size_t)s] : -1;
      const int32_t n = v >= 0 ? n_per_view[v] : 0;
      if (slot_view) { slot_view[s] = v; lane_n[s] = n; }
      nmax = n > nmax ? n : nmax;
    }
    if (slot_view) { tile_nmax[t] = nmax; tile_off[t] = off; }
    off += (int64_t)nmax * PT_COMP * LANES;
  }
  *total_floats = off;
  return 0;
}

// camcal_layout_device on host arrays: the counts from the offsets, then camcal_layout
int64_t camcal_layout_device_temp_len(int64_t n_views) { return n_views < 0 ? CAMCAL_E_TOOBIG : 256; }
int camcal_layout_device(int64_t n_views, const int64_t* view_ptr, int64_t n_obs, const float* obj_xyz,
                         int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax, int64_t* tile_off,
                         int64_t* head, void*, int64_t, void*) {
  if (n_views < 0 || n_obs < 0 || !view_ptr || !slot_view || !lane_n || !tile_nmax || !tile_off || !head) return CAMCAL_E_BADARG;
  std::vector<int32_t> n((size_t)n_views);
  int64_t flags = 0;
  for (int64_t i = 0; i < n_views; ++i) {
    const int64_t c = view_ptr[i + 1] - view_ptr[i];
    const bool bad = c < 0 || c > 2147483647LL;
    if (bad) flags |= CAMCAL_LAYOUT_BAD_COUNT;
    n[(size_t)i] = bad ? 0 : (int32_t)c;
  }
  const int64_t total = view_ptr[n_views] - view_ptr[0];
  if (total != n_obs || view_ptr[0] != 0) flags |= CAMCAL_LAYOUT_BAD_TOTAL;
  if (obj_xyz)
    for (int64_t i = 0; i < n_obs; ++i)
      if (!(obj_xyz[3 * i + 2] == 0.0f)) { flags |= CAMCAL_LAYOUT_NOT_PLANAR; break; }
  int32_t nt = 0;
  int64_t ns = 0, tf = 0;
  const int rc = camcal_layout(n_views, n.data(), &nt, &ns, &tf, slot_view, lane_n, tile_nmax, tile_off);
  if (rc) return rc;
  head[0] = total; head[1] = tf; head[2] = flags; head[3] = n_views > 0 ? lane_n[0] : 0;
  return 0;
}

int camcal_points_off_plane(int64_t n_obs, const float* obj_xyz, int64_t* flag, void*) {
  if (n_obs < 0 || !flag || (n_obs > 0 && !obj_xyz)) return CAMCAL_E_BADARG;
  flag[0] = flag[1] = flag[2] = flag[3] = 0;
  for (int64_t i = 0; i < n_obs; ++i)
    if (!(obj_xyz[3 * i + 2] == 0.0f)) { flag[2] = CAMCAL_LAYOUT_NOT_PLANAR; break; }
  return 0;
}

int camcal_pack_tiles(const float* obj, const float* img, const int64_t* view_ptr, const int32_t* slot_view,
                      const camcal_tiles* t, float* out, void*) {
  CC_OMP_FOR
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const int v = slot_view[slot], n = t->lane_n[slot], nmax = t->tile_nmax[tile];
      const int64_t p0 = v >= 0 ? view_ptr[v] : 0;
      float* base = out + t->tile_off[tile] + lane;
      for (int k = 0; k < nmax; ++k) {
        float* p = base + (int64_t)k * (PT_COMP * LANES);
        if (k < n) {
          p[0] = obj[(p0 + k) * 3]; p[LANES] = obj[(p0 + k) * 3 + 1]; p[2 * LANES] = obj[(p0 + k) * 3 + 2];
          p[3 * LANES] = img[(p0 + k) * 2]; p[4 * LANES] = img[(p0 + k) * 2 + 1];
        } else {
          for (int c = 0; c < 5; ++c) p[c * LANES] = 0.f;
        }
      }
    }
  return 0;
}
int camcal_pose_init(int model, const camcal_tiles* t, const double* intr, double* pose, int32_t* status, void*) {
  if (model == 0) pose_init<PINHOLE>(t, intr, pose, status); else pose_init<FISHEYE>(t, intr, pose, status);
  return 0;
}
int camcal_pose_refine(int model, const camcal_tiles* t, const double* intr, double* pose, int max_iter, double tol,
                       int32_t* iters, void*) {
  if (model == 0) pose_refine<PINHOLE>(t, intr, pose, max_iter, tol, iters);
  else pose_refine<FISHEYE>(t, intr, pose, max_iter, tol, iters);
  return 0;
}
int camcal_intr_normal(int model, const camcal_tiles* t, const double* intr, const double* pose, int with_normal,
                       double* glob, double* view_cost, double* V, double* g, double*, void*) {
  const int mode = !with_normal ? 0 : (V ? 2 : 1);
  if (model == 0) {
    if (mode == 2) intr_normal<PINHOLE, 2>(t, intr, pose, glob, view_cost, V, g);
    else if (mode == 1) intr_normal<PINHOLE, 1>(t, intr, pose, glob, view_cost, V, g);
    else intr_normal<PINHOLE, 0>(t, intr, pose, glob, view_cost, V, g);
  } else {
    if (mode == 2) intr_normal<FISHEYE, 2>(t, intr, pose, glob, view_cost, V, g);
    else if (mode == 1) intr_normal<FISHEYE, 1>(t, intr, pose, glob, view_cost, V, g);
    else intr_normal<FISHEYE, 0>(t, intr, pose, glob, view_cost, V, g);
  }
  return 0;
}
int camcal_cross_schur(int model, const camcal_tiles* t, const double* intr, const double* pose, const double* V,
                       const double* g, double lam, double* W_out, double* schur, double*, void*) {
  if (model == 0) cross_schur<PINHOLE>(t, intr, pose, V, g, lam, W_out, schur);
  else cross_schur<FISHEYE>(t, intr, pose, V, g, lam, W_out, schur);
  return 0;
}
int camcal_view_blocks(int model, const camcal_tiles* t, const double* intr, const double* pose, double* V, double* W,
                       double* g, void*) {
  if (model == 0) view_blocks<PINHOLE>(t, intr, pose, V, W, g); else view_blocks<FISHEYE>(t, intr, pose, V, W, g);
  return 0;
}
int camcal_schur(int model, const camcal_tiles* t, const double* V, const double* W, const double* g, double lam,
                 double* out, double*, void*) {
  if (model == 0) schur<9>(t, V, W, g, lam, out); else schur<8>(t, V, W, g, lam, out);
  return 0;
}
int camcal_backsub(int model, const camcal_tiles* t, const double* V, const double* W, const double* g, double lam,
                   const double* dK, const double* pin, double* pout, double* norms, double*, void*) {
  if (model == 0) backsub<9>(t, V, W, g, lam, dK, pin, pout, norms); else backsub<8>(t, V, W, g, lam, dK, pin, pout, norms);
  return 0;
}
int camcal_normal_eq(int model, const camcal_tiles* t, const double* intr, const double* pose, double lam, double* V,
                     double* W, double* g, double* glob, double* schur_out, double* scratch, void* s) {
  camcal_intr_normal(model, t, intr, pose, 1, glob, nullptr, V, g, scratch, s);
  return camcal_cross_schur(model, t, intr, pose, V, g, lam, W, schur_out, scratch, s);
}
int camcal_normal_eq_fused(int model, const camcal_tiles* t, const double* intr, const double* pose, double lam,
                           double* V, double* W, double* g, double* glob, double* schur_out, double* scratch, void* s) {
  camcal_intr_normal(model, t, intr, pose, 1, glob, nullptr, nullptr, nullptr, scratch, s);
  std::vector<double> tV, tW, tg;
  if (!V) {
    tV.resize(21 * t->n_slots); tW.resize(54 * t->n_slots); tg.resize(6 * t->n_slots);
    V = tV.data(); W = tW.data(); g = tg.data();
  }
  camcal_view_blocks(model, t, intr, pose, V, W, g, s);
  return camcal_schur(model, t, V, W, g, lam, schur_out, scratch, s);
}
int camcal_normal_eq_fused_response(int model, const camcal_tiles* t, const double* intr, const double* pose, double* Y,
                                    double* glob, double* schur_out, double*, void*) {
  if (!Y) return CAMCAL_E_BADARG;
  if (model == 0) fused_response<PINHOLE>(t, intr, pose, Y, glob, schur_out);
  else fused_response<FISHEYE>(t, intr, pose, Y, glob, schur_out);
  return 0;
}
int camcal_pose_predict(int model, const camcal_tiles* t, const double* Y, const double* dK, double scale,
                        const int32_t* iters, int cap, double* pose, void*) {
  const int ni = camcal_n_intr(model);
  if (ni < 0) return ni;
  const int64_t S = t->n_slots;
  double s[9];
  for (int a = 0; a < ni; ++a) s[a] = -scale * dK[a];
  CC_OMP_FOR
  for (int64_t slot = 0; slot < S; ++slot) {
    if (t->lane_n[slot] <= 0 || (iters && (iters[slot] == 0 || iters[slot] >= cap))) continue;
    double p[6];
    load_pose(pose, S, slot, p);
    if (ni == 9) lane_pose_predict<9>(Y, S, slot, s, p); else lane_pose_predict<8>(Y, S, slot, s, p);
    for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
  }
  return 0;
}
int camcal_pose_predict_refine(int model, const camcal_tiles* t, const double* intr, const double* Y, const double* dK,
                               double scale, int cap, double* pose, int max_iter, double tol, int32_t* iters, void* s) {
  if (!Y || !dK) return CAMCAL_E_BADARG;
  const int rc = camcal_pose_predict(model, t, Y, dK, scale, iters, cap, pose, s);
  return rc ? rc : camcal_pose_refine(model, t, intr, pose, max_iter, tol, iters, s);
}
// the two-launch iteration of small sets: same arithmetic as the stages; the "partial records" of the
// emulator are the finished record, kept in scratch between the two calls
int camcal_refine_normal_eq_partial(int model, const camcal_tiles* t, const double* intr, double* pose, int max_iter,
                                    double tol, int32_t* iters, double* Y, int predict, const double* dK, double scale,
                                    int cap, int, double* scratch, void* s) {
  if (!t || !intr || !pose || !scratch || max_iter < 0 || (predict && (!Y || !dK))) return CAMCAL_E_BADARG;
  if (model != 0 && model != 1) return CAMCAL_E_BADMODEL;
  if (model != 1) return CAMCAL_E_UNSUPPORTED;
  if (max_iter > 0) {
    const int rc = predict ? camcal_pose_predict_refine(model, t, intr, Y, dK, scale, cap, pose, max_iter, tol, iters, s)
                           : camcal_pose_refine(model, t, intr, pose, max_iter, tol, iters, s);
    if (rc) return rc;
  }
  const int ng = camcal_glob_len(model);
  if (Y) return camcal_normal_eq_fused_response(model, t, intr, pose, Y, scratch, scratch + ng, nullptr, s);
  return camcal_normal_eq_fused(model, t, intr, pose, 0.0, nullptr, nullptr, nullptr, scratch, scratch + ng, nullptr, s);
}
int camcal_reduce_solve(int model, const camcal_tiles* t, double* scratch, double* rec, int do_solve, const uint8_t* mask_host,
                        double lam, double scale, const double* intr_in, double* intr_out, double* dK, double* stats,
                        double* notify, double seq, void* s) {
  if (!t || !scratch || !rec || (do_solve && (!mask_host || !intr_in || !intr_out || !dK || !stats))) return CAMCAL_E_BADARG;
  if (model != 0 && model != 1) return CAMCAL_E_BADMODEL;
  if (model != 1) return CAMCAL_E_UNSUPPORTED;
  const int ng = camcal_glob_len(model), ns = camcal_schur_len(model);
  for (int i = 0; i < ng + ns; ++i) rec[i] = scratch[i];
  if (do_solve) {
    const int rc = camcal_solve_reduced(model, rec, rec + ng, mask_host, lam, scale, intr_in, intr_out, dK, stats, s);
    if (rc) return rc;
  }
  if (notify) {
    for (int i = 0; i < 5; ++i) notify[i] = do_solve ? stats[i] : 0.0;
    notify[5] = rec[ng - 1];
    notify[CAMCAL_NOTIFY_SEQ] = seq;
  }
  return 0;
}
// the folded pinhole LM step of small sets: compositions of the stages
static int small_pinhole(int model, const camcal_tiles* t) {
  if (model != 0 && model != 1) return CAMCAL_E_BADMODEL;
  return (model != 0 || !folded_set_of(t->n_tiles, t->planar, t->lanes_per_view, false)) ? CAMCAL_E_UNSUPPORTED : 0;
}
int camcal_small_set_begin(int model, const camcal_tiles* t, double* scratch, void*) {
  if (!t || !scratch) return CAMCAL_E_BADARG;
  return small_pinhole(model, t);
}
int camcal_cross_schur_solve(int model, const camcal_tiles* t, const double* intr, const double* pose, const double* V,
                             const double* g, double lam, double* W_out, const double* glob, double* schur_out,
                             const uint8_t* mask_host, double scale, double* intr_out, double* dK, double* stats,
                             double* scratch, void* s) {
  if (!t || !intr || !pose || !V || !g || !glob || !schur_out || !mask_host || !intr_out || !dK || !stats || !scratch)
    return CAMCAL_E_BADARG;
  int rc = small_pinhole(model, t);
  if (rc) return rc;
  rc = camcal_cross_schur(model, t, intr, pose, V, g, lam, W_out, schur_out, scratch, s);
  return rc ? rc : camcal_solve_reduced(model, glob, schur_out, mask_host, lam, scale, intr, intr_out, dK, stats, s);
}
int camcal_schur_solve(int model, const camcal_tiles* t, const double* V, const double* W, const double* g, double lam,
                       const double* glob, double* schur_out, const uint8_t* mask_host, double scale, const double* intr_in,
                       double* intr_out, double* dK, double* stats, double* scratch, void* s) {
  if (!t || !V || !W || !g || !glob || !schur_out || !mask_host || !intr_in || !intr_out || !dK || !stats || !scratch)
    return CAMCAL_E_BADARG;
  int rc = small_pinhole(model, t);
  if (rc) return rc;
  rc = camcal_schur(model, t, V, W, g, lam, schur_out, scratch, s);
  return rc ? rc : camcal_solve_reduced(model, glob, schur_out, mask_host, lam, scale, intr_in, intr_out, dK, stats, s);
}
int camcal_backsub_intr_normal(int model, const camcal_tiles* t, const double* V, const double* W, const double* g, double lam,
                               const double* dK, const double* pose_in, double* pose_out, const double* intr_trial,
                               double* glob_trial, double* V_trial, double* g_trial, double* norms, const double* stats,
                               double* notify, double seq, double* scratch, void* s) {
  if (!t || !V || !W || !g || !dK || !pose_in || !pose_out || !intr_trial || !glob_trial || !V_trial || !g_trial || !norms ||
      !stats || !scratch)
    return CAMCAL_E_BADARG;
  int rc = small_pinhole(model, t);
  if (rc) return rc;
  rc = camcal_backsub(model, t, V, W, g, lam, dK, pose_in, pose_out, norms, scratch, s);
  if (rc == 0) rc = camcal_intr_normal(model, t, intr_trial, pose_out, 1, glob_trial, nullptr, V_trial, g_trial, scratch, s);
  if (rc) return rc;
  if (notify) {
    for (int i = 0; i < 5; ++i) notify[i] = stats[i];
    notify[5] = norms[0]; notify[6] = norms[1]; notify[7] = glob_trial[camcal_glob_len(model) - 1];
    notify[CAMCAL_NOTIFY_SEQ] = seq;
  }
  return 0;
}
int camcal_normal_eq_fused_static(int model, const camcal_tiles* t, const double* intr, const double* pose, double lam,
                                  double* glob, double* schur_out, void* s) {
  // the same checks, in the same order, as the HIP build (the emulator walks one view per lane, but the
  // limit counts the partial records of the mapping the HIP build would choose: camcal_static_partials)
  if (!t || t->n_tiles <= 0 || t->n_slots != (int64_t)t->n_tiles * LANES || !t->tile_off || !t->tile_nmax || !t->lane_n ||
      (t->lanes_per_view != 0 && t->lanes_per_view != 1 && t->lanes_per_view != 16 && t->lanes_per_view != 64))
    return CAMCAL_E_BADARG;
  if (camcal_glob_len(model) < 0) return CAMCAL_E_BADMODEL;
  if (static_partials(t->n_tiles, t->planar, t->lanes_per_view) > STATIC_MAX_PARTIALS) return CAMCAL_E_TOOBIG;
  return camcal_normal_eq_fused(model, t, intr, pose, lam, nullptr, nullptr, nullptr, glob, schur_out, nullptr, s);
}
int camcal_time_stage(int, int, int, const camcal_tiles*, const double*, const double*, double, double*, double*,
                      double*, double*, double*, double*, void*, float* ms) { if (ms) *ms = 0.f; return 0; }
int camcal_solve_reduced(int model, const double* glob, const double* schur, const uint8_t* mask_host, double lam,
                         double scale, const double* intr_in, double* intr_out, double* dK, double* stats, void*) {
  const int ni = camcal_n_intr(model);
  if (ni < 0) return ni;
  unsigned mask = 0;
  for (int a = 0; a < ni; ++a) if (mask_host[a]) mask |= 1u << a;
  double ws[REDUCED_WS];
  reduced_solve(ni, glob, schur, mask, lam, scale, intr_in, intr_out, dK, stats, ws);
  return 0;
}
int64_t camcal_rig_rec_len(int model, int n_cams) {
  const int ni = camcal_n_intr(model);
  if (ni < 0 || n_cams < 1 || n_cams > CAMCAL_RIG_MAX_CAMS) return CAMCAL_E_BADARG;
  return RigLayout(ni, n_cams).total;
}
int64_t camcal_rig_scratch_len(int, int, const int32_t*, int64_t) { return 16 * 13 + 64; }
int camcal_rig_residuals(const camcal_rig* rg, int b, void*) {
  const int ni = camcal_n_intr(rg->model);
  const RigLayout L(ni, rg->n_cams);
  double cost = 0.0;
  for (int c = 0; c < rg->n_cams; ++c) {
    double* glob = rg->rec[b] + c * L.per;
    if (rg->model == 0) {
      double red[IntrAcc<PINHOLE>::NR];
      rig_eval_cam<PINHOLE>(rg, b, c, red);
      IntrAcc<PINHOLE>::expand(red, glob);
    } else {
      double red[IntrAcc<FISHEYE>::NR];
      rig_eval_cam<FISHEYE>(rg, b, c, red);
      IntrAcc<FISHEYE>::expand(red, glob);
    }
    cost += glob[L.NV - 1];
  }
  rg->rec[b][L.o_cost] = cost;
  return 0;
}
int camcal_rig_residuals_notify(const camcal_rig* rg, int b, double* notify, double seq, void* stream) {
  const int rc = camcal_rig_residuals(rg, b, stream);
  if (rc || !notify) return rc;
  const camcal::RigLayout L(camcal_n_intr(rg->model), rg->n_cams);
  for (int k = 0; k < 5; ++k) notify[k] = rg->rec[b][L.o_cost + k];
  notify[CAMCAL_NOTIFY_SEQ] = seq;
  return 0;
}
int camcal_rig_chain(const camcal_rig* rg, int b, void*) {
  const RigLayout L(camcal_n_intr(rg->model), rg->n_cams);
  for (int c = 0; c < rg->n_cams; ++c) {
    double* glob = rg->rec[b] + c * L.per;
    if (rg->model == 0) rig_assemble_cam<9>(rg, b, c, glob + L.NV);
    else rig_assemble_cam<8>(rg, b, c, glob + L.NV);
  }
  return 0;
}
int camcal_rig_evaluate(const camcal_rig* rg, int b, void* stream) {
  const int rc = camcal_rig_residuals(rg, b, stream);
  return rc ? rc : camcal_rig_chain(rg, b, stream);
}
int camcal_rig_eliminate(const camcal_rig* rg, int b, double lam, void*) {
  const int ni = camcal_n_intr(rg->model);
  const RigLayout L(ni, rg->n_cams);
  const int64_t P = rg->P;
  if (ni == 9) rig_pose_schur<15>(rg->n_cams, P, rg->Vp[b], rg->gp[b], rg->Wp[b], lam, rg->Yt, rg->z);
  else rig_pose_schur<14>(rg->n_cams, P, rg->Vp[b], rg->gp[b], rg->Wp[b], lam, rg->Yt, rg->z);
  double* Cm = rg->rec[b] + L.o_C;
  double* d = rg->rec[b] + L.o_d;
  const int n = L.n;
  CC_OMP_FOR
  for (int i = 0; i < n; ++i) {
    for (int j = i; j < n; ++j) {
      double s = 0.0;
      for (int64_t r = 0; r < 6 * P; ++r) s += rg->Yt[(int64_t)i * 6 * P + r] * rg->Yt[(int64_t)j * 6 * P + r];
      Cm[i * n + j] = s;
      Cm[j * n + i] = s;
    }
    double s = 0.0;
    for (int64_t r = 0; r < 6 * P; ++r) s += rg->Yt[(int64_t)i * 6 * P + r] * rg->z[r];
    d[i] = s;
  }
  return 0;
}
int camcal_rig_solve(const camcal_rig* rg, int b, const uint8_t* mask_host, double lam, int out, void*) {
  const int ni = camcal_n_intr(rg->model);
  if (ni < 0) return ni;
  const RigLayout L(ni, rg->n_cams);
  const int n = L.n;
  if (n > RIG_MAXN - 1) return CAMCAL_E_BADARG;
  unsigned long long mask = 0;
  for (int i = 0; i < n; ++i) if (mask_host[i]) mask |= 1ull << i;
  static thread_local RigSolveWs w;
  const int nl = 64;
  for (int l = 0; l < nl; ++l) rig_solve_build(l, nl, ni, rg->n_cams, rg->rec[b], mask, lam, w);
  for (int l = 0; l < nl; ++l) rig_solve_scale(l, nl, n, w);
  for (int k = 0; k < n; ++k) {
    for (int l = 0; l < nl; ++l) rig_solve_chol_col(l, nl, n, k, w);
    rig_solve_chol_diag(0, n, k, w);
    for (int l = 0; l < nl; ++l) rig_solve_chol_update(l, nl, n, k, w);
  }
  rig_solve_substitute(n, w);
  rig_solve_finish(ni, rg->n_cams, mask, w, rg->intr[b], rg->ext[b], rg->intr[out], rg->ext[out], rg->dG,
                   rg->rec[out] + L.o_gstats);
  return 0;
}
int camcal_rig_backsub(const camcal_rig* rg, int b, double lam, int out, void*) {
  const int ni = camcal_n_intr(rg->model);
  const RigLayout L(ni, rg->n_cams);
  if (ni == 9) rig_backsub<15>(rg->n_cams, rg->P, rg->Vp[b], rg->gp[b], rg->Wp[b], lam, rg->dG, rg->rig_pose[b], rg->rig_pose[out], rg->rec[out] + L.o_norms);
  else rig_backsub<14>(rg->n_cams, rg->P, rg->Vp[b], rg->gp[b], rg->Wp[b], lam, rg->dG, rg->rig_pose[b], rg->rig_pose[out], rg->rec[out] + L.o_norms);
  return 0;
}
int camcal_project_points(int model, int64_t n, const double* xyz, const double* pose6, const double* intr,
                          double* uv, void*) {
  double R[9];
  rodrigues(pose6, R);
  for (int64_t i = 0; i < n; ++i) {
    if (model == 0) project_point<PINHOLE>(xyz + 3 * i, R, pose6 + 3, intr, uv[2 * i], uv[2 * i + 1]);
    else project_point<FISHEYE>(xyz + 3 * i, R, pose6 + 3, intr, uv[2 * i], uv[2 * i + 1]);
  }
  return 0;
}
int camcal_reproj_err(int model, const camcal_tiles* t, const double* intr, const double* pose, int f32_round,
                      double* err, double* sum_out, double*, void*) {
  double s = 0, cnt = 0;
  for (int tile = 0; tile < t->n_tiles; ++tile)
    for (int lane = 0; lane < LANES; ++lane) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const LaneTile tv = lane_tile(t, tile, lane);
      double p[6], e;
      load_pose(pose, t->n_slots, slot, p);
      const bool pl = t->planar != 0, f = f32_round != 0;
#define RE(M) (pl ? (f ? lane_reproj_err<M, true, true>(tv, intr, p) : lane_reproj_err<M, true, false>(tv, intr, p)) \
                  : (f ? lane_reproj_err<M, false, true>(tv, intr, p) : lane_reproj_err<M, false, false>(tv, intr, p)))
      e = model == 0 ? RE(PINHOLE) : RE(FISHEYE);
#undef RE
      err[slot] = e;
      s += e;
      cnt += tv.n > 0 ? 1.0 : 0.0;
    }
  sum_out[0] = s; sum_out[1] = cnt;
  return 0;
}
int camcal_undistort_points(int model, int64_t n, const double* uv, const double* intr, const double* M, double* out,
                            void*) {
  for (int64_t i = 0; i < n; ++i) {
    if (model == 0) undistort_point<PINHOLE>(uv[2 * i], uv[2 * i + 1], intr, M, out[2 * i], out[2 * i + 1]);
    else undistort_point<FISHEYE>(uv[2 * i], uv[2 * i + 1], intr, M, out[2 * i], out[2 * i + 1]);
  }
  return 0;
}
int camcal_undistort_rectify_map(int model, int width, int height, const double* intr, const double* iR, int map_type,
                                 void* map1, void* map2, void*) {
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int64_t idx = (int64_t)i * width + j;
      double u, v;
      const bool ok = model == 0 ? rectify_map_px<PINHOLE>(i, j, iR, intr, u, v) : rectify_map_px<FISHEYE>(i, j, iR, intr, u, v);
      if (map_type == 0) {
        ((float*)map1)[idx] = (float)u;
        ((float*)map2)[idx] = (float)v;
      } else {
        short mx = -1, my = -1;
        unsigned short fr = 0;
        if (ok) encode_fixed_map(u, v, mx, my, fr);
        ((short*)map1)[2 * idx] = mx;
        ((short*)map1)[2 * idx + 1] = my;
        ((unsigned short*)map2)[idx] = fr;
      }
    }
  return 0;
}
int camcal_remap_u8(const unsigned char* src, int sw, int sh, int64_t sstride, int cn, const void* map1,
                    const void* map2, int map_type, int dw, int dh, unsigned char* dst, int64_t dstride, int cval,
                    void*) {
  for (int i = 0; i < dh; ++i)
    for (int j = 0; j < dw; ++j) {
      const int64_t idx = (int64_t)i * dw + j;
      int ix, iy, fx, fy;
      if (map_type == 0) {
        const int sx = remap_fixed_coord(((const float*)map1)[idx]), sy = remap_fixed_coord(((const float*)map2)[idx]);
        ix = sat_short(sx >> 5); iy = sat_short(sy >> 5); fx = sx & 31; fy = sy & 31;
      } else {
        const int a = ((const unsigned short*)map2)[idx] & 1023;
        ix = ((const short*)map1)[2 * idx]; iy = ((const short*)map1)[2 * idx + 1]; fx = a & 31; fy = a >> 5;
      }
      remap_pixel_u8(src, sw, sh, sstride, cn, ix, iy, fx, fy, (unsigned char)cval, dst + i * dstride + (int64_t)j * cn);
    }
  return 0;
}
// the emulator has no RCCL; a caller-provided communicator (the tests' gloo callback) works as in the HIP library
int camcal_rccl_unique_id(void*) { return CAMCAL_E_COMM; }
int camcal_comm_rccl_create(const void*, int, int, camcal_comm**) { return CAMCAL_E_COMM; }
int camcal_comm_destroy(camcal_comm* c) { return c ? CAMCAL_E_BADARG : 0; }
int camcal_comm_size(const camcal_comm* c) { return c ? CAMCAL_E_BADARG : 1; }
int camcal_comm_allreduce(const camcal_comm* comm, double* buf, int64_t n, void* stream) {
  if (!comm) return 0;
  if (!comm->allreduce_sum || !buf || n <= 0) return CAMCAL_E_BADARG;
  return comm->allreduce_sum(comm->ctx, buf, n, stream);
}
int camcal_peak_probe(int, void*, int64_t, int, void*, double* r) { if (r) *r = 0.0; return 0; }
}  // extern "C"

// Host-side solver loops: the same sequencing code as the HIP library, over this emulator's stages.
#include <cstdlib>
#include <cstring>
namespace {
struct camcal_reader {
  double landing[64];
  int begin(const double* src, int n, int offset, void*) {
    if (n <= 0 || offset < 0 || offset + n > 64 || !src) return CAMCAL_E_BADARG;
    std::memcpy(landing + offset, src, (size_t)n * sizeof(double));
    return 0;
  }
  int end(double* dst_host, int n_total) {
    if (n_total <= 0 || n_total > 64 || !dst_host) return CAMCAL_E_BADARG;
    std::memcpy(dst_host, landing, (size_t)n_total * sizeof(double));
    return 0;
  }
  double seq = 0.0;
  double* notify(double* seq_out) { seq += 1.0; *seq_out = seq; return landing; }
  int wait(double want, void*, double* dst_host, int n) {
    if (landing[CAMCAL_NOTIFY_SEQ] != want || n <= 0 || n > CAMCAL_NOTIFY_SEQ) return CAMCAL_E_BADARG;
    std::memcpy(dst_host, landing, (size_t)n * sizeof(double));
    return 0;
  }
};
int camcal_copy_doubles(double* dst, const double* src, int n, void*) {
  std::memmove(dst, src, (size_t)n * sizeof(double));
  return 0;
}
}  // namespace
namespace {
int rig_pair_sums(const camcal_rig* rg, double** sums_dev_out, void*) {
  const int nC = rg->n_cams;
  double* sums = rg->scratch;
  for (int i = 0; i < nC * nC * 13; ++i) sums[i] = 0.0;
  for (int c = 0; c < nC; ++c) {
    const camcal_rig_cam& k = rg->cam[c];
    if (!k.pose_slot || !k.pose) return CAMCAL_E_BADARG;
    for (int64_t p = 0; p < rg->P; ++p) k.pose_slot[p] = -1;
    for (int64_t s = 0; s < k.tiles.n_slots; ++s)
      if (k.slot_pose[s] >= 0) k.pose_slot[k.slot_pose[s]] = (int32_t)s;
  }
  for (int c = 0; c < nC; ++c)
    for (int k = 0; k < nC; ++k) {
      if (c == k) continue;
      const int64_t Sc = rg->cam[c].tiles.n_slots, Sk = rg->cam[k].tiles.n_slots;
      double* acc = sums + (c * nC + k) * 13;
      for (int64_t p = 0; p < rg->P; ++p) {
        const int sc = rg->cam[c].pose_slot[p], sk = rg->cam[k].pose_slot[p];
        if (sc < 0 || sk < 0) continue;
        double qc[6], qk[6], o[12];
        for (int i = 0; i < 6; ++i) { qc[i] = rg->cam[c].pose[i * Sc + sc]; qk[i] = rg->cam[k].pose[i * Sk + sk]; }
        rig_pair_terms(qc, qk, o);
        for (int i = 0; i < 12; ++i) acc[i] += o[i];
        acc[12] += 1.0;
      }
    }
  *sums_dev_out = sums;
  return 0;
}
int rig_init_poses(const camcal_rig* rg, void*) {
  for (int64_t p = 0; p < rg->P; ++p) {
    double out[6] = {0, 0, 0, 0, 0, 1};
    for (int c = 0; c < rg->n_cams; ++c) {
      const int s = rg->cam[c].pose_slot[p];
      if (s < 0) continue;
      const int64_t S = rg->cam[c].tiles.n_slots;
      double q[6];
      for (int i = 0; i < 6; ++i) q[i] = rg->cam[c].pose[i * S + s];
      rig_pose_from_cam(q, rg->ext[0] + c * 6, out);
      break;
    }
    for (int i = 0; i < 6; ++i) rg->rig_pose[0][i * rg->P + p] = out[i];
  }
  return 0;
}
int camcal_read_sync(double* dst_host, const double* src, int n, void*) {
  std::memcpy(dst_host, src, (size_t)n * sizeof(double));
  return 0;
}
int camcal_write_sync(double* dst, const double* src_host, int n, void*) {
  std::memcpy(dst, src_host, (size_t)n * sizeof(double));
  return 0;
}
int camcal_fork_streams(void* stream, int n, void** side_out) {   // no streams on the host: all the same
  for (int i = 0; i < n; ++i) side_out[i] = stream;
  return 0;
}
int camcal_join_streams(void*, int, void**) { return 0; }
int camcal_upload(void* dst, const void* src_host, size_t bytes, void*) { std::memcpy(dst, src_host, bytes); return 0; }
int camcal_upload_bulk(void* dst, const void* src_host, size_t bytes, void*) { std::memcpy(dst, src_host, bytes); return 0; }
int camcal_download(void* dst_host, const void* src, size_t bytes, void*) { std::memcpy(dst_host, src, bytes); return 0; }
int camcal_stream_sync(void*) { return 0; }
void* camcal_pinned_take(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void camcal_pinned_give(void* p) { std::free(p); }
int camcal_get_device(int* dev) { *dev = 0; return 0; }
int camcal_set_device(int) { return 0; }
}  // namespace
#include "../../cameracalibration_amd/csrc/camcal_driver.h"
#include "../../cameracalibration_amd/csrc/camcal_rig_driver.h"
#include "../../cameracalibration_amd/csrc/camcal_calibrate.h"
