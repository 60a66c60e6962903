// HIP kernels + C ABI of the calibration hot path for gfx950 (MI355X).
//
// Mapping: one lane == one view, one wavefront == one tile of 64 views, points
// packed so that corner k of the 64 views of a tile is one coalesced 256-byte
// row per component (see include/camcal.h).  All accumulators of a view live
// in that lane's registers; the only cross-lane traffic is the reduction of
// the tile's partial sums at the end of the reducing kernels (halving
// exchanges, WaveReduceScatter), followed by a fixed-order second stage
// (deterministic sums).  Small view sets give a view 16 or 64 lanes instead
// (group_lanes).  The single-camera kernels are FP64-VALU bound and use no MFMA (fp64 matrix
// rate = vector rate on this part, reduced system <= 9 x 9); the rig path's one dense
// contraction does run on v_mfma_f64_16x16x4 (camcal_rig_kernels.h, k_rig_gram).
//
// Replaces the arithmetic the reference delegates to cv2.calibrateCamera /
// cv2.fisheye.calibrate (server/aruco/camera_calib.py:74-77,153-156).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/camcal.h"
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "camcal_lane.h"
#include "camcal_rig.h"
#include "camcal_points.h"

using namespace camcal;

namespace {

constexpr int TPB = 256;                 // 4 wavefronts per workgroup
constexpr int WPB = TPB / LANES;
constexpr int MAX_WAVES = 4096;          // persistent waves of a reducing kernel (16 per CU)
constexpr int PARTIAL_STRIDE = 16384;    // row stride of the partial records in scratch: persistent kernels write <= MAX_WAVES
                                         // of them, grouped launches one per wavefront (16 lanes per view: tiles x 16)
constexpr int MAX_NV = 96;               // widest reduced record (rig assemble, pinhole: 54 + 21 + 6 = 81)

// Device functions that a fused kernel (k_refine_eval_grp, k_reduce_solve) shares with the separate stage
// kernels.  Inlined (the default) each kernel gets its own register allocation and schedule, and the compiler
// contracts different multiply-adds in the two contexts: the fused launch then agrees with the separate
// stages to ~1e-13 relative, not to the bit.  -DCAMCAL_SHARED_CALL='__attribute__((noinline))' makes them
// real calls, i.e. the SAME machine code in both (bit-identical results: checked once, tools/gpu_small_ab.py)
// at the price of the callee-saved register traffic of two calls per launch (C2 solve 0.985 vs 0.943 ms).
#ifndef CAMCAL_SHARED_CALL
#define CAMCAL_SHARED_CALL __forceinline__
#endif

#define DISPATCH_MODEL_PLANAR(model, planar, L)                       \
  do {                                                                \
    if ((model) == CAMCAL_PINHOLE) { if (planar) L(PINHOLE, true); else L(PINHOLE, false); } \
    else { if (planar) L(FISHEYE, true); else L(FISHEYE, false); }   \
  } while (0)

#define HIP_TRY(x)                                  \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) return (int)e_;           \
  } while (0)

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}


// Sum N per-lane values over the wavefront with N + N/2 + ... exchanges instead of the 6 N of
// a butterfly per value: at each step a lane keeps one value of every pair and hands the
// other to its partner (xor OFF), so the number of live values halves while the number of
// lanes summed doubles.  Afterwards lane l holds, in v[i], the wavefront total of value
// 64 i + bitreverse6(l); wave_scatter_store writes them out.
// One exchange step on the pair (a, b) between lanes l and l ^ OFF: afterwards a holds, in
// the lanes with bit OFF clear, what both lanes had in a, and b holds, in the lanes with the
// bit set, what both had in b -- arranged so that a + b is the step's result in every lane.
// OFF = 32 and 16 are single gfx950 instructions per dword (v_permlane32_swap: the upper
// half of the first operand <-> the lower half of the second; v_permlane16_swap: odd rows
// <-> even rows); the others go through the LDS crossbar (ds_bpermute).
// Partner of a lane for the steps inside a row of 16 lanes, as DPP moves (VALU, no LDS
// crossbar).  The partner only has to differ in bit OFF and the four pairings together
// have to connect the 16 lanes: 15 - i (row_mirror) for OFF = 8, 7 - i within each half
// row (row_half_mirror) for OFF = 4, i ^ 2 and i ^ 1 (quad_perm) for OFF = 2 and 1.
template <int OFF>
__device__ __forceinline__ double row_partner(double x) {
  static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "row-local steps only");
  constexpr int ctrl = OFF == 8 ? 0x140 : OFF == 4 ? 0x141 : OFF == 2 ? 0x4E : 0xB1;
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int plo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);
  const int phi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
  return __hiloint2double(phi, plo);
}

template <int OFF>
__device__ __forceinline__ double pair_exchange_sum(double a, double b, bool up) {
  if constexpr (OFF == 32 || OFF == 16) {
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (OFF == 32) {
      const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
      const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
  } else {
    const double send = up ? a : b, keep = up ? b : a;
    return keep + row_partner<OFF>(send);
  }
}

// the reverse step: `own` is the total of a in the lanes with bit OFF clear and of b in the
// others; returns both totals in every lane
template <int OFF>
__device__ __forceinline__ void pair_exchange_gather(double own, bool up, double& a, double& b) {
  if constexpr (OFF == 32 || OFF == 16) {
    const unsigned lo = (unsigned)__double2loint(own), hi = (unsigned)__double2hiint(own);
    if constexpr (OFF == 32) {
      const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      a = __hiloint2double((int)h[0], (int)l[0]);
      b = __hiloint2double((int)h[1], (int)l[1]);
    } else {
      const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      a = __hiloint2double((int)h[0], (int)l[0]);
      b = __hiloint2double((int)h[1], (int)l[1]);
    }
  } else {
    const double got = row_partner<OFF>(own);
    a = up ? got : own;
    b = up ? own : got;
  }
}

template <int N, int OFF>
struct WaveReduceScatter {
  static __device__ __forceinline__ void run(double* v, int lane) {
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double a = v[2 * i], b = (2 * i + 1 < N) ? v[2 * i + 1] : 0.0;
      v[i] = pair_exchange_sum<OFF>(a, b, up);
    }
    WaveReduceScatter<H, OFF / 2>::run(v, lane);
  }
};
template <int N>
struct WaveReduceScatter<N, 0> {
  static __device__ __forceinline__ void run(double*, int) {}
};

// All-reduce of N per-lane values over groups of 2 OFF lanes: the halving steps above
// down to one owner per value, then the same exchanges backwards so that every lane of
// the group ends with every total (bitwise the same in all of them).
template <int N, int OFF>
struct GroupAllReduce {
  static __device__ __forceinline__ void run(double* v, int lane) {
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double a = v[2 * i], b = (2 * i + 1 < N) ? v[2 * i + 1] : 0.0;
      v[i] = pair_exchange_sum<OFF>(a, b, up);
    }
    GroupAllReduce<H, OFF / 2>::run(v, lane);
#pragma unroll
    for (int i = H - 1; i >= 0; --i) {
      double a, b;
      pair_exchange_gather<OFF>(v[i], up, a, b);
      if (2 * i + 1 < N) v[2 * i + 1] = b;
      v[2 * i] = a;
    }
  }
};
template <int N>
struct GroupAllReduce<N, 0> {
  static __device__ __forceinline__ void run(double*, int) {}
};

template <int N>
__device__ __forceinline__ void wave_scatter_store(double* v, int lane, double* __restrict__ dst, size_t stride) {
  WaveReduceScatter<N, 32>::run(v, lane);
  const int r = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
  for (int i = 0; i < (N + 63) / 64; ++i) {
    const int idx = 64 * i + r;
    if (idx < N) dst[(size_t)idx * stride] = v[i];
  }
}

template <int NV>
__device__ __forceinline__ void wave_reduce_store(double* v, double* scratch, int wave_id) {
  wave_scatter_store<NV>(v, threadIdx.x & 63, scratch + wave_id, PARTIAL_STRIDE);
}

// this thread's share of a row of partials: elements tid, tid + TPB, ... in four independent
// chains (several loads in flight; the order of the additions is fixed)
__device__ __forceinline__ double strided_partial_sum(const double* __restrict__ row, int n) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int w = threadIdx.x;
  for (; w + 3 * TPB < n; w += 4 * TPB) {
    s0 += row[w]; s1 += row[w + TPB]; s2 += row[w + 2 * TPB]; s3 += row[w + 3 * TPB];
  }
  for (; w < n; w += TPB) s0 += row[w];
  return (s0 + s1) + (s2 + s3);
}

// second stage: out[i] = sum_w scratch[i][w], fixed order
__global__ __launch_bounds__(TPB) void k_finalize(const double* __restrict__ scratch, int n_waves,
                                                  double* __restrict__ out) {
  __shared__ double sh[TPB];
  const int i = blockIdx.x;
  const double s = strided_partial_sum(scratch + (size_t)i * PARTIAL_STRIDE, n_waves);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = TPB / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[i] = sh[0];
}

// The tile index is the same for the 64 lanes of a wavefront; saying so lets the
// tile's offset and loop bound be scalar loads and the corner loop run on SALU
// compares with SGPR-based addressing.
__device__ __forceinline__ LaneTile lane_tile(const camcal_tiles& t, int tile, int lane) {
  const int ut = __builtin_amdgcn_readfirstlane(tile);
  LaneTile tv;
  tv.base = t.pts + t.tile_off[ut] + lane;
  tv.n = t.lane_n[(int64_t)ut * LANES + lane];
  tv.nmax = t.tile_nmax[ut];
  return tv;
}

__device__ __forceinline__ void load_pose(const double* __restrict__ pose, int64_t S, int64_t slot,
                                          double p[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) p[c] = pose[c * S + slot];
}

template <int NI>
__device__ __forceinline__ void load_intr(const double* __restrict__ intr, double I[NI]) {
#pragma unroll
  for (int i = 0; i < NI; ++i) I[i] = intr[i];
}

// ---------------------------------------------------------------------------
template <int MODEL, bool PLANAR>
__global__ __launch_bounds__(TPB) void k_view_blocks(camcal_tiles t, const double* __restrict__ intr,
                                                     const double* __restrict__ pose,
                                                     double* __restrict__ Vb, double* __restrict__ Wb,
                                                     double* __restrict__ gb) {
  constexpr int NI = MT<MODEL>::NI;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot = (int64_t)tile * LANES + lane;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, slot, p);
  const LaneTile tv = lane_tile(t, tile, lane);
  ViewAcc<MODEL> acc;
  lane_view_blocks<MODEL, PLANAR>(tv, I, p, acc);
#pragma unroll
  for (int e = 0; e < 21; ++e) Vb[e * S + slot] = acc.V[e];
#pragma unroll
  for (int e = 0; e < NI * 6; ++e) Wb[e * S + slot] = acc.W[e];
#pragma unroll
  for (int e = 0; e < 6; ++e) gb[e * S + slot] = acc.g[e];
}

// Evaluation A (persistent): cost / U, b / + per-view V, g (see lane_intr_normal).
template <int MODEL, int MODE, bool PLANAR>
__global__ __launch_bounds__(TPB) void k_intr_normal(camcal_tiles t, const double* __restrict__ intr,
                                                     const double* __restrict__ pose,
                                                     double* __restrict__ view_cost,
                                                     double* __restrict__ Vb, double* __restrict__ gb,
                                                     double* __restrict__ scratch) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NR = IntrAcc<MODEL>::NR;
  constexpr int COST = IntrAcc<MODEL>::COST;
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double I[NI];
  load_intr<NI>(intr, I);
  IntrAcc<MODEL> acc;
  acc.zero();
  for (int tile = wave; tile < t.n_tiles; tile += n_waves) {
    const int64_t slot = (int64_t)tile * LANES + lane;
    double p[6];
    load_pose(pose, S, slot, p);
    const LaneTile tv = lane_tile(t, tile, lane);
    double V[21], g[6];
    const double vc = lane_intr_normal<MODEL, MODE, PLANAR>(tv, I, p, acc, V, g);
    if (view_cost) view_cost[slot] = vc;
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 21; ++e) Vb[e * S + slot] = V[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) gb[e * S + slot] = g[e];
    }
  }
  if (MODE > 0) {
    wave_reduce_store<NR>(acc.v, scratch, wave);
  } else {
    const double s = wave_sum(acc.v[COST]);
    if (lane == 0) scratch[(size_t)COST * PARTIAL_STRIDE + wave] = s;
  }
}

// Evaluation B: cross block W of every view + its Schur contribution
// C += W V'^-1 W^T, d += W V'^-1 g with V, g read back from evaluation A.
// One tile per wavefront; partial [C|d] of the tile -> scratch[i][tile].
template <int MODEL, bool STORE_W, bool PLANAR>
__global__ __launch_bounds__(TPB) void k_cross_schur(camcal_tiles t, const double* __restrict__ intr,
                                                     const double* __restrict__ pose,
                                                     const double* __restrict__ Vb, const double* __restrict__ gb,
                                                     double lam, double* __restrict__ Wb,
                                                     double* __restrict__ scratch, int scratch_stride) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NS = trisz(NI) + NI;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot = (int64_t)tile * LANES + lane;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, slot, p);
  const LaneTile tv = lane_tile(t, tile, lane);
  double W[NI * 6];
  lane_cross_block<MODEL, PLANAR>(tv, I, p, W);
  if (STORE_W) {
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Wb[e * S + slot] = W[e];
  }
  double sacc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) sacc[i] = 0.0;
  if (tv.n > 0) {
    double V[21], g[6];
#pragma unroll
    for (int e = 0; e < 21; ++e) V[e] = Vb[e * S + slot];
#pragma unroll
    for (int e = 0; e < 6; ++e) g[e] = gb[e * S + slot];
    lane_schur_regs<NI>(V, W, g, lam, sacc);
  }
  wave_scatter_store<NS>(sacc, lane, scratch + tile, scratch_stride);
}

// Fused normal-equation evaluation: U, b, cost + per-view blocks + Schur
// complement in one pass over the points.  One tile per wavefront; the wave's
// partial record [U|b|cost|C|d] goes to scratch[i][tile] (second stage sums).
// PRED: also leave the view's first-order pose response Y = V^-1 W^T (lane_pose_response;
// lam = 0) in Yb for k_pose_predict.
template <int MODEL, bool WRITE_BLOCKS, bool PLANAR, bool LOCAL = false, bool PRED = false>
__global__ __launch_bounds__(TPB) void k_normal_fused(camcal_tiles t, const double* __restrict__ intr,
                                                      const double* __restrict__ pose, double lam,
                                                      double* __restrict__ Vb, double* __restrict__ Wb,
                                                      double* __restrict__ gb, double* __restrict__ scratch,
                                                      int scratch_stride, double* __restrict__ Yb) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NG = IntrAcc<MODEL>::NR;
  constexpr int NS = trisz(NI) + NI;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot = (int64_t)tile * LANES + lane;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, slot, p);
  const LaneTile tv = lane_tile(t, tile, lane);
  ViewAcc<MODEL> acc;
  IntrAcc<MODEL> gacc;
  gacc.zero();
  lane_normal_fused<MODEL, PLANAR, true>(tv, I, p, acc, gacc);       // sums in the local tangent frame
  // C = W V^-1 W^T and d = W V^-1 g do not depend on the pose parametrisation, so the map to
  // Rodrigues coordinates is only needed for stored blocks and for the damping of diag(V)
  if (!LOCAL && (WRITE_BLOCKS || PRED || lam != 0.0)) {
    double Jl[9];
    left_jacobian(p, Jl);
    to_rodrigues_frame<NI>(Jl, acc.V, acc.W, acc.g);
  }
  if (WRITE_BLOCKS) {
#pragma unroll
    for (int e = 0; e < 21; ++e) Vb[e * S + slot] = acc.V[e];
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Wb[e * S + slot] = acc.W[e];
#pragma unroll
    for (int e = 0; e < 6; ++e) gb[e * S + slot] = acc.g[e];
  }
  wave_scatter_store<NG>(gacc.v, lane, scratch + tile, scratch_stride);
  if (LOCAL) return;                      // rig solver: elimination happens per board pose
  double sacc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) sacc[i] = 0.0;
  double id[6];
  if (tv.n > 0) lane_schur_regs<NI>(acc.V, acc.W, acc.g, lam, sacc, id);
  wave_scatter_store<NS>(sacc, lane, scratch + (size_t)NG * scratch_stride + tile, scratch_stride);
  if (PRED && tv.n > 0) {
    lane_pose_response<NI>(acc.V, id, acc.W);
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Yb[e * S + slot] = acc.W[e];
  }
}

__global__ __launch_bounds__(TPB) void k_finalize_strided(const double* __restrict__ scratch, int n_part, int stride,
                                                          double* __restrict__ out) {
  __shared__ double sh[TPB];
  const int i = blockIdx.x;
  const double s = strided_partial_sum(scratch + (size_t)i * stride, n_part);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = TPB / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[i] = sh[0];
}

// reduced accumulator values -> record [U | b | cost] (+ n_tail values copied after it)
template <int MODEL>
__global__ void k_expand(const double* __restrict__ red, double* __restrict__ out, int n_tail, int cost_only) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  constexpr int NR = IntrAcc<MODEL>::NR, NV = IntrAcc<MODEL>::NV;
  if (cost_only) { out[NV - 1] = red[IntrAcc<MODEL>::COST]; return; }
  double r[NR], o[NV];
  for (int i = 0; i < NR; ++i) r[i] = red[i];
  IntrAcc<MODEL>::expand(r, o);
  for (int i = 0; i < NV; ++i) out[i] = o[i];
  for (int i = 0; i < n_tail; ++i) out[NV + i] = red[NR + i];
}

// Per-view pose refinement (cv::internal::ComputeExtrinsicRefine's rule per view:
// Gauss-Newton until ||d|| / ||x + d|| <= tol, at most max_iter steps).  Views
// need very different numbers of steps (a few percent run to the cap), and a
// wavefront that walks one view per lane is as slow as its slowest view.  So a
// tile iterates lane-per-view only while at least STRAGGLER_LANES of its views
// are still moving.  The rest is finished cooperatively: the wavefront splits
// into 64 / GROUP groups, each group takes stragglers from the tile's list in
// turn, its GROUP lanes share the view's corners and butterfly-sum the 27 sums
// of V and g.  Poses travel through global memory between the phases (a lane
// of a finished group may be masked off, so no cross-lane reads are used).
// Measured on 1M views x 88 corners (first refinement at the guessed intrinsics):
// lane-per-view only 12.3 ms; whole-wavefront stragglers, threshold 8: 7.8 ms; groups
// of 16 lanes, threshold 16: 7.1 ms (groups of 4 / 8 and thresholds 24..40 within 5 %).
// One lane-per-view step of all views costs 0.75 ms and the mean view needs 5.8.
constexpr int STRAGGLER_LANES = 16;
constexpr int GROUP = 16;

// k_pose_predict's move applied to a pose just loaded (same arithmetic: lane_pose_predict), for the
// refinement kernels that start from the predicted pose without a launch of their own for it
template <int NI>
__device__ __forceinline__ void predicted_start_after(const double* __restrict__ Yb, const double* dK, double scale,
                                                      bool counted, int last, int cap, int64_t S, int64_t slot, int n,
                                                      double p[6]) {
  if (n <= 0) return;
  if (counted && (last == 0 || last >= cap)) return;
  double s[NI];
#pragma unroll
  for (int a = 0; a < NI; ++a) s[a] = -scale * dK[a];
  lane_pose_predict<NI>(Yb, S, slot, s, p);
}

template <int NI>
__device__ __forceinline__ void predicted_start(const double* __restrict__ Yb, const double* __restrict__ dK, double scale,
                                                const int32_t* __restrict__ iters, int cap, int64_t S, int64_t slot, int n,
                                                double p[6]) {
  predicted_start_after<NI>(Yb, dK, scale, iters != nullptr, iters ? iters[slot] : 0, cap, S, slot, n, p);
}

template <int G>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

template <int MODEL, bool PLANAR, int G>
__device__ __forceinline__ bool group_pose_gn_step(const float* base, int n, const double* intr, double pose[6],
                                                   double& change, int gl) {
  double R[9];
  rodrigues(pose, R);
  const double t[3] = {pose[3], pose[4], pose[5]};
  double V[21], g[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) V[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  for (int k = gl; k < n; k += G) {
    const Obs o = load_obs<PLANAR>(base, k);
    double Yr[3], Yc[3];
    transform_point<PLANAR>(R, t, o.X, o.Y, o.Z, Yr, Yc);
    PointEval<MODEL> p;
    eval_point<true>(Yc, intr, p);
    const double eu = p.pu - (double)o.u, ev = p.pv - (double)o.v;
    double ju[6], jv[6];
    pose_row(Yr, p.gu, ju);
    pose_row(Yr, p.gv, jv);
    accumulate_vg(V, g, ju, jv, eu, ev);
  }
  GroupAllReduce<21, G / 2>::run(V, threadIdx.x & 63);
  GroupAllReduce<6, G / 2>::run(g, threadIdx.x & 63);
  return pose_gn_update<MODEL == FISHEYE>(V, g, pose, change);       // every lane of the group holds the same sums
}

template <int MODEL, bool PLANAR>
__global__ __launch_bounds__(TPB, 2) void k_pose_refine(camcal_tiles t, const double* __restrict__ intr,
                                                        double* __restrict__ pose, int max_iter, double tol,
                                                        int32_t* __restrict__ iters, int straggler_lanes,
                                                        const double* __restrict__ Yb, const double* __restrict__ dK,
                                                        double scale, int cap) {
  constexpr int NI = MT<MODEL>::NI;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot0 = (int64_t)tile * LANES, slot = slot0 + lane;
  const LaneTile tv = lane_tile(t, tile, lane);
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, slot, p);
  if (Yb) predicted_start<NI>(Yb, dK, scale, iters, cap, S, slot, tv.n, p);   // camcal_pose_predict, in the same launch
  canonical_rvec(p);
  int it = 0;
  double change = 1.0;
  bool active = tv.n >= 4 && max_iter > 0 && change > tol;
  // phase 1: one view per lane while enough of the tile's views are still moving
  while (__popcll(__ballot(active)) >= straggler_lanes) {
    if (active) {
      const bool ok = lane_pose_gn_step<MODEL, PLANAR>(tv, I, p, change);
      if (ok) ++it;
      active = ok && change > tol && it < max_iter;
    }
  }
  const unsigned long long todo = __ballot(active);
  if (todo == 0) {
    if (tv.n >= 4) {
#pragma unroll
      for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
    }
    if (iters) iters[slot] = it;
    return;
  }
  // phase 2: groups of GROUP lanes finish the stragglers; poses and counts travel through LDS
  __shared__ int s_it[TPB];
  __shared__ double s_pose[6][TPB];
  const int w0 = threadIdx.x & ~63;
  s_it[threadIdx.x] = it;
#pragma unroll
  for (int c = 0; c < 6; ++c) s_pose[c][threadIdx.x] = p[c];
  __threadfence_block();                      // visible to the other lanes of the wave
  const int grp = lane / GROUP, gl = lane % GROUP;
  const int count = __popcll(todo);
  for (int s = grp; s < count; s += LANES / GROUP) {
    unsigned long long m = todo;
    for (int i = 0; i < s; ++i) m &= m - 1;
    const int L = __ffsll((long long)m) - 1;
    double q[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) q[c] = s_pose[c][w0 + L];
    const int n = t.lane_n[slot0 + L];
    int jt = s_it[w0 + L];
    double ch = 1.0;
    const float* base = tv.base - lane + L;
    bool go = true;
    while (go) {
      const bool ok = group_pose_gn_step<MODEL, PLANAR, GROUP>(base, n, I, q, ch, gl);
      if (ok) ++jt;
      go = ok && ch > tol && jt < max_iter;
    }
    if (gl == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) s_pose[c][w0 + L] = q[c];
      s_it[w0 + L] = jt;
    }
  }
  __threadfence_block();
  if (tv.n >= 4) {
#pragma unroll
    for (int c = 0; c < 6; ++c) pose[c * S + slot] = s_pose[c][threadIdx.x];
  }
  if (iters) iters[slot] = s_it[threadIdx.x];
}

// Small view sets (group_lanes() below): G lanes share one view from the start, so a
// production-sized call (20 ... a few thousand views) spreads over the whole chip
// instead of a handful of wavefronts that each walk their views' corners serially.
struct GroupView {
  int64_t slot;        // the view's slot
  const float* base;   // pts + tile_off + lane-of-the-view
  int n;               // its corners
  int gl;              // this lane's rank in the group
};

template <int G>
__device__ __forceinline__ bool group_view(const camcal_tiles& t, GroupView& gv) {
  gv.slot = ((int64_t)blockIdx.x * TPB + threadIdx.x) / G;
  gv.gl = threadIdx.x % G;
  if (gv.slot >= t.n_slots) return false;
  gv.base = t.pts + t.tile_off[gv.slot / LANES] + (gv.slot % LANES);
  gv.n = t.lane_n[gv.slot];
  return true;
}

// the Gauss-Newton loop of one view shared by the G lanes of its group (max_iter > 0); returns the steps taken
// (Shared by k_pose_refine_grp and k_refine_eval_grp -- inlined into each unless CAMCAL_SHARED_CALL says
// otherwise, see the note at its definition.  Likewise group_normal_fused.)
template <int MODEL, int G>
__device__ CAMCAL_SHARED_CALL int group_pose_refine(const GroupView& gvr, const double* Ip, double qp[6], int max_iter, double tol) {
  constexpr int NI = MT<MODEL>::NI;
  const GroupView gv = gvr;                  // arguments arrive through memory: work on register copies
  double I[NI], q[6];
#pragma unroll
  for (int i = 0; i < NI; ++i) I[i] = Ip[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) q[i] = qp[i];
  int it = 0;
  canonical_rvec(q);
  double ch = 1.0;
  bool go = ch > tol;
  while (go) {
    const bool ok = group_pose_gn_step<MODEL, true, G>(gv.base, gv.n, I, q, ch, gv.gl);
    if (ok) ++it;
    go = ok && ch > tol && it < max_iter;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) qp[i] = q[i];
  return it;
}

template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_pose_refine_grp(camcal_tiles t, const double* __restrict__ intr,
                                                         double* __restrict__ pose, int max_iter, double tol,
                                                         int32_t* __restrict__ iters, const double* __restrict__ Yb,
                                                         const double* __restrict__ dK, double scale, int cap) {
  constexpr int NI = MT<MODEL>::NI;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int64_t S = t.n_slots;
  int it = 0;
  if (gv.n >= 4 && max_iter > 0) {
    double I[NI], q[6];
    load_intr<NI>(intr, I);
    load_pose(pose, S, gv.slot, q);
    if (Yb) predicted_start<NI>(Yb, dK, scale, iters, cap, S, gv.slot, gv.n, q);   // every lane of the group alike
    it = group_pose_refine<MODEL, G>(gv, I, q, max_iter, tol);
    if (gv.gl == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) pose[c * S + gv.slot] = q[c];
    }
  }
  if (iters && gv.gl == 0) iters[gv.slot] = it;
}

template <int G>
struct GroupReduce {
  template <int M> __device__ __forceinline__ void sum(double* a) const {
    GroupAllReduce<M, G / 2>::run(a, threadIdx.x & 63);
  }
};

template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_pose_init_grp(camcal_tiles t, const double* __restrict__ intr,
                                                       double* __restrict__ pose, int32_t* __restrict__ status) {
  constexpr int NI = MT<MODEL>::NI;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int64_t S = t.n_slots;
  double I[NI];
  load_intr<NI>(intr, I);
  double p[6] = {0, 0, 0, 0, 0, 1};
  const int st = pose_init_walk<MODEL>(gv.base, gv.n, gv.gl, G, I, p, GroupReduce<G>());
  if (gv.gl == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) pose[c * S + gv.slot] = p[c];
    if (status) status[gv.slot] = gv.n == 0 ? 0 : st;
  }
}

// Fused normal-equation evaluation with G lanes per view, for one view group at intrinsics I and
// pose p: each lane accumulates its share of the view's corners, the view blocks are summed over
// the group, the group's first lane eliminates the pose; one partial record [U|b|cost|C|d] per wavefront.
template <int MODEL, int G>
__device__ CAMCAL_SHARED_CALL void group_normal_fused(const GroupView& gvr, int64_t S, const double* Ip, const double* pp, double lam,
                                                   double* __restrict__ Vb, double* __restrict__ Wb,
                                                   double* __restrict__ gb, double* __restrict__ scratch,
                                                   int scratch_stride, int local, double* __restrict__ Yb) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NG = IntrAcc<MODEL>::NR;
  constexpr int NS = trisz(NI) + NI;
  const GroupView gv = gvr;                  // arguments arrive through memory: work on register copies
  double I[NI], p[6];
#pragma unroll
  for (int i = 0; i < NI; ++i) I[i] = Ip[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) p[i] = pp[i];
  const int wave = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  double R[9];
  rodrigues(p, R);
  const double tr[3] = {p[3], p[4], p[5]};
  ViewAcc<MODEL> acc;
  IntrAcc<MODEL> gacc;
  acc.zero();
  gacc.zero();
  for (int k = gv.gl; k < gv.n; k += G) fused_obs<MODEL, true>(R, tr, I, load_obs<true>(gv.base, k), acc, gacc);
  if (gv.gl == 0) gacc.add_count(gv.n);
  const GroupReduce<G> red;
  red.template sum<21>(acc.V);
  red.template sum<NI * 6>(acc.W);
  red.template sum<6>(acc.g);
  if (!local && (Vb || Yb || lam != 0.0)) {           // see k_normal_fused
    double Jl[9];
    left_jacobian(p, Jl);
    to_rodrigues_frame<NI>(Jl, acc.V, acc.W, acc.g);
  }
  if (Vb && gv.gl == 0) {
#pragma unroll
    for (int e = 0; e < 21; ++e) Vb[e * S + gv.slot] = acc.V[e];
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Wb[e * S + gv.slot] = acc.W[e];
#pragma unroll
    for (int e = 0; e < 6; ++e) gb[e * S + gv.slot] = acc.g[e];
  }
  wave_scatter_store<NG>(gacc.v, lane, scratch + wave, scratch_stride);
  if (local) return;
  double sacc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) sacc[i] = 0.0;
  double id[6];
  if (gv.gl == 0 && gv.n > 0) lane_schur_regs<NI>(acc.V, acc.W, acc.g, lam, sacc, id);
  wave_scatter_store<NS>(sacc, lane, scratch + (size_t)NG * scratch_stride + wave, scratch_stride);
  if (Yb && gv.gl == 0 && gv.n > 0) {                  // first-order pose response (see k_normal_fused)
    lane_pose_response<NI>(acc.V, id, acc.W);
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Yb[e * S + gv.slot] = acc.W[e];
  }
}

template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_normal_fused_grp(camcal_tiles t, const double* __restrict__ intr,
                                                          const double* __restrict__ pose, double lam,
                                                          double* __restrict__ Vb, double* __restrict__ Wb,
                                                          double* __restrict__ gb, double* __restrict__ scratch,
                                                          int scratch_stride, int local, double* __restrict__ Yb) {
  constexpr int NI = MT<MODEL>::NI;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int64_t S = t.n_slots;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, gv.slot, p);
  group_normal_fused<MODEL, G>(gv, S, I, p, lam, Vb, Wb, gb, scratch, scratch_stride, local, Yb);
}

// Pose re-estimation and the next evaluation of a view in one launch (small sets, fisheye loop):
// k_pose_refine_grp's body, then k_normal_fused_grp's at the refined pose, which never leaves the
// registers.  The views need 1 ... 20 Gauss-Newton steps and the launch lasts as long as its
// slowest view (C2: one or two views run to 18 - 20 steps in each of the first four iterations
// while the mean is 3 - 4); evaluated in the same launch, the other views' records are done long
// before the slowest view has converged, and a kernel boundary and the evaluation's own ramp
// disappear from every iteration.  Same device functions as the two launches (equal to ~1e-13, see CAMCAL_SHARED_CALL).
// Leaves the partial records in scratch (k_reduce_solve sums them).
template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_refine_eval_grp(camcal_tiles t, const double* __restrict__ intr,
                                                         double* __restrict__ pose, int max_iter, double tol,
                                                         int32_t* __restrict__ iters, double* Yb, int predict,
                                                         const double* __restrict__ dK, double scale, int cap,
                                                         double* __restrict__ scratch, int scratch_stride,
                                                         unsigned* counter_reset) {
  constexpr int NI = MT<MODEL>::NI;
  if (counter_reset && blockIdx.x == 0 && threadIdx.x == 0) *counter_reset = 0u;   // k_reduce_solve's, first launch of a solve
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int64_t S = t.n_slots;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, gv.slot, p);
  if (max_iter > 0) {
    int it = 0;
    if (gv.n >= 4) {
      if (predict) predicted_start<NI>(Yb, dK, scale, iters, cap, S, gv.slot, gv.n, p);
      it = group_pose_refine<MODEL, G>(gv, I, p, max_iter, tol);
      if (gv.gl == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c * S + gv.slot] = p[c];
      }
    }
    if (iters && gv.gl == 0) iters[gv.slot] = it;
  }
  group_normal_fused<MODEL, G>(gv, S, I, p, 0.0, nullptr, nullptr, nullptr, scratch, scratch_stride, 0, Yb);
}

// Evaluations A and B with G lanes per view (the pinhole LM driver's stages).
template <int MODEL, int MODE, int G>
__device__ __forceinline__ void group_intr_normal(const GroupView& gv, int64_t S, const double* I, const double* p,
                                                  double* __restrict__ view_cost, double* __restrict__ Vb,
                                                  double* __restrict__ gb, double* __restrict__ scratch, int stride) {
  constexpr int NR = IntrAcc<MODEL>::NR;
  constexpr int COST = IntrAcc<MODEL>::COST;
  const int wave = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  double R[9];
  rodrigues(p, R);
  const double tr[3] = {p[3], p[4], p[5]};
  IntrAcc<MODEL> acc;
  acc.zero();
  double V[21], g[6], vc = 0.0;
#pragma unroll
  for (int i = 0; i < 21; ++i) V[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  for (int k = gv.gl; k < gv.n; k += G)
    intr_normal_obs<MODEL, MODE, true>(R, tr, I, load_obs<true>(gv.base, k), acc, V, g, vc);
  if (MODE > 0) { if (gv.gl == 0) acc.add_count(gv.n); }
  else acc.v[COST] += vc;
  if (view_cost) {
    const double s = group_sum<G>(vc);
    if (gv.gl == 0) view_cost[gv.slot] = s;
  }
  if (MODE == 2) {
    const GroupReduce<G> red;
    red.template sum<21>(V);
    red.template sum<6>(g);
    double Jl[9];
    left_jacobian(p, Jl);
    vg_to_rodrigues_frame(Jl, V, g);
    if (gv.gl == 0) {
#pragma unroll
      for (int e = 0; e < 21; ++e) Vb[e * S + gv.slot] = V[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) gb[e * S + gv.slot] = g[e];
    }
  }
  if (MODE > 0) {
    wave_scatter_store<NR>(acc.v, lane, scratch + wave, stride);
  } else {
    const double s = wave_sum(acc.v[COST]);
    if (lane == 0) scratch[(size_t)COST * stride + wave] = s;
  }
}

template <int MODEL, int MODE, int G>
__global__ __launch_bounds__(TPB) void k_intr_normal_grp(camcal_tiles t, const double* __restrict__ intr,
                                                         const double* __restrict__ pose,
                                                         double* __restrict__ view_cost, double* __restrict__ Vb,
                                                         double* __restrict__ gb, double* __restrict__ scratch) {
  constexpr int NI = MT<MODEL>::NI;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int64_t S = t.n_slots;
  double I[NI], p[6];
  load_intr<NI>(intr, I);
  load_pose(pose, S, gv.slot, p);
  group_intr_normal<MODEL, MODE, G>(gv, S, I, p, view_cost, Vb, gb, scratch, PARTIAL_STRIDE);
}

template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_cross_schur_grp(camcal_tiles t, const double* __restrict__ intr,
                                                         const double* __restrict__ pose,
                                                         const double* __restrict__ Vb, const double* __restrict__ gb,
                                                         double lam, double* __restrict__ Wb,
                                                         double* __restrict__ scratch, int scratch_stride) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NS = trisz(NI) + NI;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int wave = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double I[NI], p[6], R[9];
  load_intr<NI>(intr, I);
  load_pose(pose, S, gv.slot, p);
  rodrigues(p, R);
  const double tr[3] = {p[3], p[4], p[5]};
  double W[NI * 6];
#pragma unroll
  for (int i = 0; i < NI * 6; ++i) W[i] = 0.0;
  for (int k = gv.gl; k < gv.n; k += G) cross_block_obs<MODEL, true>(R, tr, I, load_obs<true>(gv.base, k), W);
  const GroupReduce<G> red;
  red.template sum<NI * 6>(W);
  double Jl[9];
  left_jacobian(p, Jl);
  w_to_rodrigues_frame<NI>(Jl, W);
  if (Wb && gv.gl == 0) {
#pragma unroll
    for (int e = 0; e < NI * 6; ++e) Wb[e * S + gv.slot] = W[e];
  }
  double sacc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) sacc[i] = 0.0;
  if (gv.gl == 0 && gv.n > 0) {
    double V[21], g[6];
#pragma unroll
    for (int e = 0; e < 21; ++e) V[e] = Vb[e * S + gv.slot];
#pragma unroll
    for (int e = 0; e < 6; ++e) g[e] = gb[e * S + gv.slot];
    lane_schur_regs<NI>(V, W, g, lam, sacc);
  }
  wave_scatter_store<NS>(sacc, lane, scratch + wave, scratch_stride);
}

template <int MODEL>
__global__ __launch_bounds__(TPB) void k_pose_init(camcal_tiles t, const double* __restrict__ intr,
                                                   double* __restrict__ pose, int32_t* __restrict__ status) {
  constexpr int NI = MT<MODEL>::NI;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot = (int64_t)tile * LANES + lane;
  const LaneTile tv = lane_tile(t, tile, lane);
  double I[NI];
  load_intr<NI>(intr, I);
  double p[6] = {0, 0, 0, 0, 0, 1};
  const int st = lane_pose_init<MODEL>(tv, I, p);
#pragma unroll
  for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
  if (status) status[slot] = tv.n == 0 ? 0 : st;
}

template <int NI>
__global__ __launch_bounds__(TPB) void k_schur(camcal_tiles t, const double* __restrict__ Vb,
                                               const double* __restrict__ Wb, const double* __restrict__ gb,
                                               double lam, double* __restrict__ scratch) {
  constexpr int NV = trisz(NI) + NI;
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  for (int tile = wave; tile < t.n_tiles; tile += n_waves) {
    const int64_t slot = (int64_t)tile * LANES + lane;
    if (t.lane_n[slot] > 0) lane_schur<NI>(Vb, Wb, gb, S, slot, lam, acc);
  }
  wave_reduce_store<NV>(acc, scratch, wave);
}

template <int NI>
__global__ __launch_bounds__(TPB) void k_backsub(camcal_tiles t, const double* __restrict__ Vb,
                                                 const double* __restrict__ Wb, const double* __restrict__ gb,
                                                 double lam, const double* __restrict__ dK,
                                                 const double* __restrict__ pose_in,
                                                 double* __restrict__ pose_out, double* __restrict__ scratch) {
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double dk[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) dk[i] = dK[i];
  double acc[2] = {0.0, 0.0};
  for (int tile = wave; tile < t.n_tiles; tile += n_waves) {
    const int64_t slot = (int64_t)tile * LANES + lane;
    double prev[6], out[6];
    load_pose(pose_in, S, slot, prev);
    if (t.lane_n[slot] > 0) {
      double dd, pp;
      lane_backsub<NI>(Vb, Wb, gb, S, slot, lam, dk, prev, out, dd, pp);
      acc[0] += dd;
      acc[1] += pp;
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) out[c] = prev[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) pose_out[c * S + slot] = out[c];
  }
  wave_reduce_store<2>(acc, scratch, wave);
}

// First-order move of every pose along the intrinsic step intr_out = intr_in + scale dK of
// k_solve_reduced (lane_pose_predict): the warm start of the re-estimation that follows.
// A view whose last refinement took no step (cv2's condition guard, a failed factorisation)
// stays where it is, as it does in the refinement itself; so does one that ran into the step
// cap: it is not at its optimum, where the response is derived.
template <int NI>
__global__ __launch_bounds__(TPB) void k_pose_predict(camcal_tiles t, const double* __restrict__ Yb,
                                                      const double* __restrict__ dK, double scale,
                                                      const int32_t* __restrict__ iters, int cap,
                                                      double* __restrict__ pose) {
  const int64_t S = t.n_slots, slot = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (slot >= S || t.lane_n[slot] <= 0) return;
  if (iters && (iters[slot] == 0 || iters[slot] >= cap)) return;
  double s[NI], p[6];
#pragma unroll
  for (int a = 0; a < NI; ++a) s[a] = -scale * dK[a];
  load_pose(pose, S, slot, p);
  lane_pose_predict<NI>(Yb, S, slot, s, p);
#pragma unroll
  for (int c = 0; c < 6; ++c) pose[c * S + slot] = p[c];
}

// Reduced solve (U' - C) dK = b - d on the free intrinsics, the arithmetic of
// camcal_lane.h: reduced_solve (Jacobi scaling, Gaussian elimination with partial
// pivoting, intr_out = intr_in + scale dK, step norms) spread over one wavefront: lane i
// holds row i of the scaled system in registers, the pivot row of every step is broadcast,
// rows are never moved (a lane remembers the step at which its row was the pivot).
__device__ __forceinline__ double lane_bcast(double x, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src),
                          __builtin_amdgcn_readlane(__double2loint(x), src));
}

// LDS traffic between the lanes of ONE wavefront (its ds instructions execute in order): only the
// compiler has to be kept from moving accesses across this point
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// executed by one whole wavefront; s_st / s_in / s_out: 9 doubles of LDS each, private to this wavefront;
// glob, schur, intr_in / intr_out (may be the same), dK, stats: global or LDS
__device__ CAMCAL_SHARED_CALL void solve_reduced_wave(int NI, const double* glob, const double* schur, unsigned mask,
                                                   double lam, double scale, const double* intr_in, double* intr_out,
                                                   double* dK, double* stats, double* s_st, double* s_in, double* s_out) {
  constexpr int MAXN = 9;
  const int lane = threadIdx.x & 63;
  const int NU = NI * (NI + 1) / 2;
  mask &= (1u << NI) - 1u;
  const int nf = __popc(mask);
  auto tr = [NI](int i, int j) { return i <= j ? i * NI - (i * (i - 1)) / 2 + (j - i) : j * NI - (j * (j - 1)) / 2 + (i - j); };
  // parameter of this lane's row: the lane-th set bit of the mask
  int a = 0;
  {
    int cnt = 0;
    for (int b = 0; b < NI; ++b)
      if ((mask >> b) & 1u) { if (cnt == lane) a = b; ++cnt; }
  }
  const bool active = lane < nf;
  double sc = 1.0;
  if (active) {
    const double dd = glob[tr(a, a)] * (1.0 + lam) - schur[tr(a, a)];
    sc = dd > 0.0 ? 1.0 / sqrt(dd) : 1.0;
  }
  double row[MAXN + 1];
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    const int b = __builtin_amdgcn_readlane(a, j);
    const double scj = lane_bcast(sc, j);
    double v = 0.0;
    if (active && j < nf) {
      double u = glob[tr(a, b)];
      if (a == b) u *= 1.0 + lam;
      v = (u - schur[tr(a, b)]) * sc * scj;
    }
    row[j] = v;
  }
  row[MAXN] = active ? (glob[NU + a] - schur[NU + a]) * sc : 0.0;
  int pos = -1;                      // elimination step at which this row was the pivot
  double okf = 1.0;
#pragma unroll
  for (int c = 0; c < MAXN; ++c) {
    if (c < nf) {                    // uniform
      const double val = (active && pos < 0) ? fabs(row[c]) : -1.0;
      double best = 0.0;
      int piv = -1;
#pragma unroll
      for (int r = 0; r < MAXN; ++r) {
        const double vr = lane_bcast(val, r);
        if (vr > best) { best = vr; piv = r; }
      }
      if (piv < 0) {                 // no usable pivot in this column
        okf = 0.0;
      } else {
        double prow[MAXN + 1];
#pragma unroll
        for (int k = 0; k <= MAXN; ++k) prow[k] = (k >= c) ? lane_bcast(row[k], piv) : 0.0;
        if (lane == piv) pos = c;
        const double ip = 1.0 / prow[c];
        if (active && pos < 0) {
          const double f = row[c] * ip;
          if (f != 0.0) {
#pragma unroll
            for (int k = 0; k < MAXN; ++k)
              if (k >= c) row[k] -= f * prow[k];
            row[MAXN] -= f * prow[MAXN];
          }
        }
      }
    }
  }
  // back substitution: x[c] comes from the row that was the pivot of step c
  double x[MAXN];
#pragma unroll
  for (int c = 0; c < MAXN; ++c) x[c] = 0.0;
#pragma unroll
  for (int c = MAXN - 1; c >= 0; --c) {
    if (c < nf) {                    // uniform
      double sacc = row[MAXN];
#pragma unroll
      for (int k = 0; k < MAXN; ++k)
        if (k > c) sacc -= row[k] * x[k];
      const double mine = row[c] != 0.0 ? sacc / row[c] : 0.0;
      const unsigned long long owner = __ballot(pos == c);
      x[c] = owner ? lane_bcast(mine, __ffsll((long long)owner) - 1) : 0.0;
    }
  }
  // this lane's own unknown (row i <-> column i of the scaled system), back to parameter units
  double xi = 0.0;
#pragma unroll
  for (int c = 0; c < MAXN; ++c)
    if (lane == c) xi = x[c];
  if (lane < NI) {                   // non-free parameters first, the free ones overwrite theirs
    const double in = intr_in[lane];
    s_st[lane] = 0.0; s_in[lane] = in; s_out[lane] = in;
  }
  wave_lds_sync();
  if (active) {
    const double d = xi * sc, in = s_in[a], st = scale * d;
    s_st[a] = st; s_out[a] = in + st;
    dK[a] = d;
  }
  wave_lds_sync();
  if (lane < NI) {
    intr_out[lane] = s_out[lane];
    if (!((mask >> lane) & 1u)) dK[lane] = 0.0;
  }
  if (lane == 0) {                   // the four step norms, summed in parameter order
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int p = 0; p < NI; ++p) {
      const double st = s_st[p], in = s_in[p], out = s_out[p];
      s0 += st * st;
      s1 += in * in;
      if (p < 4) { s2 += st * st; s3 += out * out; }
    }
    stats[0] = s0; stats[1] = s1; stats[2] = s2; stats[3] = s3; stats[4] = okf;
  }
}

__global__ __launch_bounds__(64) void k_solve_reduced(int NI, const double* __restrict__ glob,
                                                      const double* __restrict__ schur, unsigned mask, double lam,
                                                      double scale, const double* intr_in, double* intr_out,
                                                      double* __restrict__ dK, double* __restrict__ stats) {
  __shared__ double s_st[9], s_in[9], s_out[9];
  solve_reduced_wave(NI, glob, schur, mask, lam, scale, intr_in, intr_out, dK, stats, s_st, s_in, s_out);
}

// k_finalize_strided's sum of one row by the calling workgroup (every thread calls); thread 0 returns the sum
__device__ __forceinline__ double block_row_sum(const double* __restrict__ row, int n, double* sh) {
  const double v = strided_partial_sum(row, n);
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int st = TPB / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// Second stage of an evaluation and the reduced solve in one launch: workgroup r sums row r of the
// partial records (k_finalize_strided); the workgroup that finishes last (a counter in device memory,
// left at zero again) expands the sums into the record [U|b|cost|C|d] (k_expand) and, with do_solve,
// solves the reduced system (k_solve_reduced).  Nobody waits for anybody: no workgroup spins.
template <int MODEL>
__global__ __launch_bounds__(TPB) void k_reduce_solve(const double* __restrict__ scratch, int n_part, double* red,
                                                      unsigned* counter, double* __restrict__ rec, int do_solve,
                                                      unsigned mask, double lam, double scale, const double* intr_in,
                                                      double* intr_out, double* dK, double* stats, double* notify,
                                                      double seq) {
  constexpr int NI = MT<MODEL>::NI, NR = IntrAcc<MODEL>::NR, NV = IntrAcc<MODEL>::NV, NS = trisz(NI) + NI;
  __shared__ double s_sh[TPB], s_red[NR + NS], s_rec[NV + NS], s_st[9], s_in[9], s_out[9], s_stats[5];
  __shared__ unsigned s_last;
  const double v = block_row_sum(scratch + (size_t)blockIdx.x * n_part, n_part, s_sh);
  if (threadIdx.x == 0) {
    red[blockIdx.x] = v;
    __threadfence();
    s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x < NR + NS) s_red[threadIdx.x] = __hip_atomic_load(&red[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    *counter = 0u;                                     // for the next launch on this scratch
    IntrAcc<MODEL>::expand(s_red, s_rec);
    for (int i = 0; i < NS; ++i) s_rec[NV + i] = s_red[NR + i];
  }
  __syncthreads();
  if (threadIdx.x < NV + NS) rec[threadIdx.x] = s_rec[threadIdx.x];
  if (threadIdx.x >= 64) return;
  if (do_solve) {
    solve_reduced_wave(NI, s_rec, s_rec + NV, mask, lam, scale, intr_in, intr_out, dK, s_stats, s_st, s_in, s_out);
    wave_lds_sync();
    if (threadIdx.x < 5) stats[threadIdx.x] = s_stats[threadIdx.x];
  }
  // the scalars the host decides on, straight into its pinned memory: [stats (5) | cost | ... | seq], seq last
  if (notify && threadIdx.x == 0) {
    for (int i = 0; i < 5; ++i) notify[i] = do_solve ? s_stats[i] : 0.0;
    notify[5] = s_rec[NV - 1];
    __threadfence_system();
    __hip_atomic_store(&notify[CAMCAL_NOTIFY_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- the same folding for the pinhole LM step of a small set -------------------------------------------
// (1) sums of the Schur partial records + the reduced solve (k_finalize[_strided] + k_solve_reduced);
// (2) back-substitution + evaluation A at the trial point, per view (k_backsub + k_intr_normal_grp mode 2);
// (3) sums of the trial record and the step norms, expansion, and the eight scalars of the accept / reject
//     decision written to the host (k_finalize x 2 + k_expand + three copies).
template <int NI>
__global__ __launch_bounds__(TPB) void k_reduce_schur_solve(const double* __restrict__ scratch, int stride, int n_part,
                                                            double* red, unsigned* counter, const double* glob, double* schur,
                                                            unsigned mask, double lam, double scale, const double* intr_in,
                                                            double* intr_out, double* dK, double* stats) {
  constexpr int NS = trisz(NI) + NI, NG = NS + 1;
  __shared__ double s_sh[TPB], s_glob[NG], s_schur[NS], s_st[9], s_in[9], s_out[9], s_stats[5];
  __shared__ unsigned s_last;
  const double v = block_row_sum(scratch + (size_t)blockIdx.x * stride, n_part, s_sh);
  if (threadIdx.x == 0) {
    red[blockIdx.x] = v;
    __threadfence();
    s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x < NS) {
    const double x = __hip_atomic_load(&red[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_schur[threadIdx.x] = x;
    schur[threadIdx.x] = x;
  }
  if (threadIdx.x < NG) s_glob[threadIdx.x] = glob[threadIdx.x];
  if (threadIdx.x == 0) *counter = 0u;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  solve_reduced_wave(NI, s_glob, s_schur, mask, lam, scale, intr_in, intr_out, dK, s_stats, s_st, s_in, s_out);
  wave_lds_sync();
  if (threadIdx.x < 5) stats[threadIdx.x] = s_stats[threadIdx.x];
}

template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_backsub_eval_grp(camcal_tiles t, const double* __restrict__ Vb,
                                                          const double* __restrict__ Wb, const double* __restrict__ gb,
                                                          double lam, const double* __restrict__ dK,
                                                          const double* __restrict__ pose_in, double* __restrict__ pose_out,
                                                          const double* __restrict__ intr_trial, double* __restrict__ V_out,
                                                          double* __restrict__ g_out, double* __restrict__ scratch, int stride) {
  constexpr int NI = MT<MODEL>::NI, NR = IntrAcc<MODEL>::NR;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int wave = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double dk[NI], I[NI], prev[6], out[6];
#pragma unroll
  for (int i = 0; i < NI; ++i) dk[i] = dK[i];
  load_pose(pose_in, S, gv.slot, prev);
  double acc[2] = {0.0, 0.0};
  if (gv.n > 0) {                      // every lane of the group: the same 6 x 6 solve on the same blocks
    double dd, pp;
    lane_backsub<NI>(Vb, Wb, gb, S, gv.slot, lam, dk, prev, out, dd, pp);
    if (gv.gl == 0) { acc[0] = dd; acc[1] = pp; }
  } else {
#pragma unroll
    for (int c = 0; c < 6; ++c) out[c] = prev[c];
  }
  if (gv.gl == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) pose_out[c * S + gv.slot] = out[c];
  }
  load_intr<NI>(intr_trial, I);
  group_intr_normal<MODEL, 2, G>(gv, S, I, out, nullptr, V_out, g_out, scratch, stride);
  wave_scatter_store<2>(acc, lane, scratch + (size_t)NR * stride + wave, stride);
}

template <int MODEL>
__global__ __launch_bounds__(TPB) void k_reduce_trial(const double* __restrict__ scratch, int n_part, double* red,
                                                      unsigned* counter, double* __restrict__ glob_trial,
                                                      double* __restrict__ norms, const double* __restrict__ stats,
                                                      double* notify, double seq) {
  constexpr int NR = IntrAcc<MODEL>::NR, NV = IntrAcc<MODEL>::NV;
  __shared__ double s_sh[TPB], s_red[NR + 2], s_rec[NV];
  __shared__ unsigned s_last;
  const double v = block_row_sum(scratch + (size_t)blockIdx.x * n_part, n_part, s_sh);
  if (threadIdx.x == 0) {
    red[blockIdx.x] = v;
    __threadfence();
    s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x < NR + 2) s_red[threadIdx.x] = __hip_atomic_load(&red[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    *counter = 0u;
    IntrAcc<MODEL>::expand(s_red, s_rec);
  }
  __syncthreads();
  if (threadIdx.x < NV) glob_trial[threadIdx.x] = s_rec[threadIdx.x];
  if (threadIdx.x < 2) norms[threadIdx.x] = s_red[NR + threadIdx.x];
  if (notify && threadIdx.x == 0) {     // [stats (5) | step norms (2) | cost | ... | seq], seq last
    for (int i = 0; i < 5; ++i) notify[i] = stats[i];
    notify[5] = s_red[NR]; notify[6] = s_red[NR + 1]; notify[7] = s_rec[NV - 1];
    __threadfence_system();
    __hip_atomic_store(&notify[CAMCAL_NOTIFY_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// blockIdx.x = tile, the PACK_SPLIT * WPB wavefronts of a tile take its corners in turn:
// a lane reads one point of its view (12 + 8 bytes, strided by view) and the wavefront
// writes five coalesced 256-byte rows
constexpr int PACK_SPLIT = 8;
constexpr int PACK_RUN = 4;
__global__ __launch_bounds__(TPB) void k_pack_tiles(const float* __restrict__ obj, const float* __restrict__ img,
                                                    const int64_t* __restrict__ view_ptr,
                                                    const int32_t* __restrict__ slot_view, camcal_tiles t,
                                                    float* __restrict__ out) {
  const int tile = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t slot = (int64_t)tile * LANES + lane;
  const int v = slot_view[slot];
  const int n = t.lane_n[slot];
  const int nmax = t.tile_nmax[tile];
  const int64_t p0 = v >= 0 ? view_ptr[v] : 0;
  float* base = out + t.tile_off[tile] + lane;
  // a wavefront takes a RUN of corners (>= PACK_RUN = 4, i.e. 48 + 32 contiguous bytes of its lanes' views):
  // the cache line a lane touches is used up by the same wavefront's next loads instead of being
  // fetched again by another one.  Engine set-up at 1M x 88: k strided over the wavefronts 2.8 ms, runs of
  // 4 / 8 / 24 corners 2.40 / 2.52 / 3.4 ms (longer runs leave too few wavefronts in flight)
  const int slices = PACK_SPLIT * WPB;
  int run = (nmax + slices - 1) / slices;
  run = run < PACK_RUN ? PACK_RUN : run;
  const int k0 = (blockIdx.y * WPB + (threadIdx.x >> 6)) * run;
  const int k1 = k0 + run < nmax ? k0 + run : nmax;
  for (int k = k0; k < k1; ++k) {
    float X = 0, Y = 0, Z = 0, u = 0, w = 0;
    if (k < n) {
      const float* o = obj + (p0 + k) * 3;
      const float* m = img + (p0 + k) * 2;
      X = o[0]; Y = o[1]; Z = o[2]; u = m[0]; w = m[1];
    }
    float* p = base + (int64_t)k * (PT_COMP * LANES);
    p[0 * LANES] = X; p[1 * LANES] = Y; p[2 * LANES] = Z; p[3 * LANES] = u; p[4 * LANES] = w;
  }
}

// --- peak probes (bench only): dependent-free FP64 FMA stream, streaming read
__global__ __launch_bounds__(TPB) void k_peak_fma(double* __restrict__ out, int iters, double a, double b) {
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (double)(threadIdx.x + i) * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 123.456) out[0] = s;   // keeps the chain live, practically never taken
}

__global__ __launch_bounds__(TPB) void k_peak_read(const float4* __restrict__ in, size_t n4, float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

// one float per lane, 256-byte rows: the access shape of the point loads (PMC calibration)
__global__ __launch_bounds__(TPB) void k_peak_read4(const float* __restrict__ in, size_t n, float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
  if (acc == 123.456f) out[0] = acc;
}

inline int tiles_ok(const camcal_tiles* t) {
  if (!t || t->n_tiles <= 0 || t->n_slots != (int64_t)t->n_tiles * LANES) return 0;
  if (!t->tile_off || !t->tile_nmax || !t->lane_n) return 0;
  if (t->lanes_per_view != 0 && t->lanes_per_view != 1 && t->lanes_per_view != 16 && t->lanes_per_view != 64) return 0;
  return 1;
}
// ---------------------------------------------------------------------------
// Point-level kernels (camcal_points.h)
template <int MODEL>
__global__ __launch_bounds__(TPB) void k_project_points(int64_t n, const double* __restrict__ xyz,
                                                        const double* __restrict__ pose6,
                                                        const double* __restrict__ intr, double* __restrict__ uv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p[6], R[9], I[MT<MODEL>::NI];
#pragma unroll
  for (int c = 0; c < 6; ++c) p[c] = pose6[c];
#pragma unroll
  for (int c = 0; c < MT<MODEL>::NI; ++c) I[c] = intr[c];
  rodrigues(p, R);
  const double X[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
  double u, v;
  project_point<MODEL>(X, R, p + 3, I, u, v);
  uv[i * 2] = u;
  uv[i * 2 + 1] = v;
}

template <int MODEL, bool PLANAR, bool F32>
__global__ __launch_bounds__(TPB) void k_reproj_err(camcal_tiles tiles, const double* __restrict__ intr,
                                                    const double* __restrict__ pose, double* __restrict__ err,
                                                    double* __restrict__ scratch) {
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  double I[MT<MODEL>::NI];
#pragma unroll
  for (int c = 0; c < MT<MODEL>::NI; ++c) I[c] = intr[c];
  double acc[2] = {0.0, 0.0};
  for (int tile = wave; tile < tiles.n_tiles; tile += n_waves) {
    const int64_t slot = (int64_t)tile * LANES + lane;
    LaneTile tv;
    tv.base = tiles.pts + tiles.tile_off[tile] + lane;
    tv.n = tiles.lane_n[slot];
    tv.nmax = tiles.tile_nmax[tile];
    double p[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) p[c] = pose[c * tiles.n_slots + slot];
    const double e = lane_reproj_err<MODEL, PLANAR, F32>(tv, I, p);
    err[slot] = e;
    acc[0] += e;
    acc[1] += tv.n > 0 ? 1.0 : 0.0;
  }
  wave_reduce_store<2>(acc, scratch, wave);
}

template <int MODEL>
__global__ __launch_bounds__(TPB) void k_undistort_points(int64_t n, const double* __restrict__ uv,
                                                          const double* __restrict__ intr,
                                                          const double* __restrict__ M, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double I[MT<MODEL>::NI], Mm[9];
#pragma unroll
  for (int c = 0; c < MT<MODEL>::NI; ++c) I[c] = intr[c];
  if (M) {
#pragma unroll
    for (int c = 0; c < 9; ++c) Mm[c] = M[c];
  }
  double x, y;
  undistort_point<MODEL>(uv[i * 2], uv[i * 2 + 1], I, M ? Mm : nullptr, x, y);
  out[i * 2] = x;
  out[i * 2 + 1] = y;
}

template <int MODEL, int MAP_TYPE>
__global__ __launch_bounds__(TPB) void k_rectify_map(int width, int height, const double* __restrict__ intr,
                                                     const double* __restrict__ iR, void* __restrict__ map1,
                                                     void* __restrict__ map2) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)width * height) return;
  const int i = (int)(idx / width), j = (int)(idx - (int64_t)i * width);
  double I[MT<MODEL>::NI], Q[9];
#pragma unroll
  for (int c = 0; c < MT<MODEL>::NI; ++c) I[c] = intr[c];
#pragma unroll
  for (int c = 0; c < 9; ++c) Q[c] = iR[c];
  double u, v;
  const bool ok = rectify_map_px<MODEL>(i, j, Q, I, u, v);
  if (MAP_TYPE == 0) {
    ((float*)map1)[idx] = (float)u;
    ((float*)map2)[idx] = (float)v;
  } else {
    short mx = -1, my = -1;
    unsigned short fr = 0;
    if (ok) encode_fixed_map(u, v, mx, my, fr);
    ((short2*)map1)[idx] = make_short2(mx, my);
    ((unsigned short*)map2)[idx] = fr;
  }
}

template <int MAP_TYPE>
__global__ __launch_bounds__(TPB) void k_remap_u8(const unsigned char* __restrict__ src, int sw, int sh,
                                                  long long sstride, int cn, const void* __restrict__ map1,
                                                  const void* __restrict__ map2, int dw, int dh,
                                                  unsigned char* __restrict__ dst, long long dstride,
                                                  unsigned char cval) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)dw * dh) return;
  const int i = (int)(idx / dw), j = (int)(idx - (int64_t)i * dw);
  int ix, iy, fx, fy;
  if (MAP_TYPE == 0) {
    const int sx = remap_fixed_coord(((const float*)map1)[idx]), sy = remap_fixed_coord(((const float*)map2)[idx]);
    ix = sat_short(sx >> 5); iy = sat_short(sy >> 5); fx = sx & 31; fy = sy & 31;
  } else {
    const short2 m = ((const short2*)map1)[idx];
    const int a = ((const unsigned short*)map2)[idx] & 1023;
    ix = m.x; iy = m.y; fx = a & 31; fy = a >> 5;
  }
  remap_pixel_u8(src, sw, sh, sstride, cn, ix, iy, fx, fy, cval, dst + (long long)i * dstride + (long long)j * cn);
}

inline int grid_tiles(const camcal_tiles* t) { return (t->n_tiles + WPB - 1) / WPB; }
inline int grid_persistent(const camcal_tiles* t) {
  const int g = grid_tiles(t);
  return g < MAX_WAVES / WPB ? g : MAX_WAVES / WPB;
}

// Lanes per view for this set: 1 (large sets: a lane walks its view), or 16 / 64 for
// small planar sets.  The partial records of a grouped launch are one per wavefront
// (n_tiles * G of them) and must fit the scratch every caller holds (camcal_scratch_len).
static_assert(GROUP64_MAX_TILES * 64 <= PARTIAL_STRIDE && GROUP16_MAX_TILES * 16 <= PARTIAL_STRIDE,
              "grouped launches write one partial record per wavefront into PARTIAL_STRIDE-strided scratch");
inline int group_lanes(const camcal_tiles* t) { return lanes_per_view_of(t->n_tiles, t->planar, t->lanes_per_view); }
inline int grid_groups(const camcal_tiles* t, int G) { return (int)((t->n_slots * G + TPB - 1) / TPB); }

}  // namespace

#include "camcal_rig_kernels.h"

// Scratch of camcal_normal_eq_fused_static: the partial records of one single-pass evaluation,
// [NR + NS rows][partials] + the reduced values, for callers that cannot allocate.  The name
// is a host-side stand-in for a per-device array: host code only ever uses the address
// hipGetSymbolAddress returns for the current device (static_scratch_ptr).
constexpr int STATIC_ROWS = 109;                  // pinhole: 55 + 54 (fisheye 37 + 44)
constexpr size_t STATIC_SCRATCH_DOUBLES = (size_t)STATIC_ROWS * STATIC_MAX_PARTIALS + 2 * MAX_NV;
static __device__ __align__(256) double static_scratch[STATIC_SCRATCH_DOUBLES];

namespace {
constexpr int STATIC_MAX_DEVICES = 64;
std::mutex g_static_mu;
double* g_static_ptr[STATIC_MAX_DEVICES] = {};
// device address of `static_scratch` on the current device (looked up once per device)
int static_scratch_ptr(double** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= STATIC_MAX_DEVICES) return CAMCAL_E_BADARG;
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[dev]) {
    double* p = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(static_scratch)));
    if (!p) return CAMCAL_E_BADARG;
    g_static_ptr[dev] = p;
  }
  *out = g_static_ptr[dev];
  return 0;
}
}  // namespace

extern "C" {

int camcal_version(void) { return 100; }

int camcal_n_intr(int model) { return model == CAMCAL_PINHOLE ? 9 : model == CAMCAL_FISHEYE ? 8 : CAMCAL_E_BADMODEL; }
int camcal_glob_len(int model) {
  const int ni = camcal_n_intr(model);
  return ni < 0 ? ni : ni * (ni + 1) / 2 + ni + 1;
}
int camcal_schur_len(int model) {
  const int ni = camcal_n_intr(model);
  return ni < 0 ? ni : ni * (ni + 1) / 2 + ni;
}
int64_t camcal_scratch_len(int model, int n_tiles) {
  (void)model;
  // persistent / grouped launches: <= PARTIAL_STRIDE partial records of <= STATIC_ROWS values; one lane per view: one per tile
  const int64_t a = (int64_t)STATIC_ROWS * PARTIAL_STRIDE, b = (int64_t)2 * MAX_NV * (n_tiles > 0 ? n_tiles : 0);
  return (a > b ? a : b) + 2 * MAX_NV;   // + the reduced values ahead of their expansion
}

int camcal_layout(int64_t n_views, const int32_t* n_per_view, int32_t* n_tiles, int64_t* n_slots,
                  int64_t* total_floats, int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax, int64_t* tile_off) {
  if (n_views < 0 || (n_views > 0 && !n_per_view) || !n_tiles || !n_slots || !total_floats) return CAMCAL_E_BADARG;
  if (slot_view && (!lane_n || !tile_nmax || !tile_off)) return CAMCAL_E_BADARG;
  const int64_t T = n_views > 0 ? (n_views + LANES - 1) / LANES : 1;
  if (T > 2147483647) return CAMCAL_E_TOOBIG;
  *n_tiles = (int32_t)T;
  *n_slots = T * LANES;
  std::vector<int32_t> order((size_t)n_views);
  int32_t top = 0;
  for (int64_t i = 0; i < n_views; ++i) {
    if (n_per_view[i] < 0) return CAMCAL_E_BADARG;
    top = n_per_view[i] > top ? n_per_view[i] : top;
  }
  if (top <= (1 << 22)) {
    // counting sort by descending count, views of one count in their own order (1M ragged views: 200 ms of
    // std::stable_sort through the comparator, 6 ms this way -- and the solve on the device takes 37)
    std::vector<int64_t> first((size_t)top + 2, 0);
    for (int64_t i = 0; i < n_views; ++i) ++first[(size_t)n_per_view[i]];
    int64_t at = 0;
    for (int64_t c = top; c >= 0; --c) {
      const int64_t k = first[(size_t)c];
      first[(size_t)c] = at;
      at += k;
    }
    for (int64_t i = 0; i < n_views; ++i) order[(size_t)first[(size_t)n_per_view[i]]++] = (int32_t)i;
  } else {
    for (int64_t i = 0; i < n_views; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n_per_view[a] > n_per_view[b]; });
  }
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

int camcal_pack_tiles(const float* obj_xyz, const float* img_uv, const int64_t* view_ptr,
                      const int32_t* slot_view, const camcal_tiles* tiles, float* pts_out, void* stream) {
  if (!tiles_ok(tiles) || !obj_xyz || !img_uv || !view_ptr || !slot_view || !pts_out) return CAMCAL_E_BADARG;
  hipLaunchKernelGGL(k_pack_tiles, dim3(tiles->n_tiles, PACK_SPLIT), dim3(TPB), 0, (hipStream_t)stream, obj_xyz, img_uv,
                     view_ptr, slot_view, *tiles, pts_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_pose_init(int model, const camcal_tiles* tiles, const double* intr, double* pose, int32_t* status,
                     void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose) return CAMCAL_E_BADARG;
  const dim3 g(grid_tiles(tiles)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int G = group_lanes(tiles);
  if (G > 1) {
    const dim3 gg(grid_groups(tiles, G));
    if (model == CAMCAL_PINHOLE) {
      if (G == 64) hipLaunchKernelGGL((k_pose_init_grp<PINHOLE, 64>), gg, b, 0, s, *tiles, intr, pose, status);
      else hipLaunchKernelGGL((k_pose_init_grp<PINHOLE, 16>), gg, b, 0, s, *tiles, intr, pose, status);
    } else {
      if (G == 64) hipLaunchKernelGGL((k_pose_init_grp<FISHEYE, 64>), gg, b, 0, s, *tiles, intr, pose, status);
      else hipLaunchKernelGGL((k_pose_init_grp<FISHEYE, 16>), gg, b, 0, s, *tiles, intr, pose, status);
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_pose_init<PINHOLE>, g, b, 0, s, *tiles, intr, pose, status);
  else if (model == CAMCAL_FISHEYE) hipLaunchKernelGGL(k_pose_init<FISHEYE>, g, b, 0, s, *tiles, intr, pose, status);
  else return CAMCAL_E_BADMODEL;
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
namespace {
int pose_refine_impl(int model, const camcal_tiles* tiles, const double* intr, double* pose, int max_iter,
                     double tol, int32_t* iters, const double* Y, const double* dK, double scale, int cap, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || max_iter < 0) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const dim3 g(grid_tiles(tiles)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  const int sl = STRAGGLER_LANES;
  const int G = group_lanes(tiles);
  if (G > 1) {
    const dim3 gg(grid_groups(tiles, G));
    if (model == CAMCAL_PINHOLE) {
      if (G == 64) hipLaunchKernelGGL((k_pose_refine_grp<PINHOLE, 64>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, dK, scale, cap);
      else hipLaunchKernelGGL((k_pose_refine_grp<PINHOLE, 16>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, dK, scale, cap);
    } else {
      if (G == 64) hipLaunchKernelGGL((k_pose_refine_grp<FISHEYE, 64>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, dK, scale, cap);
      else hipLaunchKernelGGL((k_pose_refine_grp<FISHEYE, 16>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, dK, scale, cap);
    }
    HIP_TRY(hipGetLastError());
    return 0;
  }
#define LAUNCH(M, P) hipLaunchKernelGGL((k_pose_refine<M, P>), g, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, sl, Y, dK, scale, cap)
  DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace
extern "C" {
int camcal_pose_refine(int model, const camcal_tiles* tiles, const double* intr, double* pose, int max_iter,
                       double tol, int32_t* iters, void* stream) {
  return pose_refine_impl(model, tiles, intr, pose, max_iter, tol, iters, nullptr, nullptr, 0.0, 0, stream);
}
int camcal_pose_predict_refine(int model, const camcal_tiles* tiles, const double* intr, const double* Y,
                               const double* dK, double scale, int cap, double* pose, int max_iter, double tol,
                               int32_t* iters, void* stream) {
  if (!Y || !dK) return CAMCAL_E_BADARG;
  return pose_refine_impl(model, tiles, intr, pose, max_iter, tol, iters, Y, dK, scale, cap, stream);
}

}  // extern "C"
namespace {
int normal_eq_fused_launch(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double lam,
                           double* V, double* W, double* g, double* Y, double* scratch, void* stream);
}
extern "C" {
// which sets run the folded stages: camcal_lane.h, folded_set_of (with the measurements behind the limits)
static inline bool folded_set(const camcal_tiles* tiles, int model) {
  return folded_set_of(tiles->n_tiles, tiles->planar, tiles->lanes_per_view, model == CAMCAL_FISHEYE);
}

// where k_refine_eval_grp / k_reduce_solve keep their reduced values and their counter: behind the
// partial records, as camcal_normal_eq_fused does for a grouped set
static inline double* small_set_red(const camcal_tiles* tiles, int G, int rows, double* scratch) {
  return scratch + (size_t)rows * ((size_t)tiles->n_tiles * G);
}
static inline unsigned* small_set_counter(const camcal_tiles* tiles, int G, int rows, double* scratch) {
  return (unsigned*)(small_set_red(tiles, G, rows, scratch) + 2 * MAX_NV - 8);
}

int camcal_refine_normal_eq_partial(int model, const camcal_tiles* tiles, const double* intr, double* pose, int max_iter,
                                    double tol, int32_t* iters, double* Y, int predict, const double* dK, double scale,
                                    int cap, int first, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !scratch || max_iter < 0) return CAMCAL_E_BADARG;
  if (predict && (!Y || !dK)) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int G = group_lanes(tiles);
  if (model != CAMCAL_FISHEYE) return CAMCAL_E_UNSUPPORTED;
  const int np = tiles->n_tiles * G;
  const int rows = IntrAcc<FISHEYE>::NR + camcal_schur_len(CAMCAL_FISHEYE);
  unsigned* counter = first ? small_set_counter(tiles, G, rows, scratch) : nullptr;
  const dim3 gg(grid_groups(tiles, G)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (!folded_set(tiles, model)) {
    // large sets: the re-estimation and the evaluation as launches of their own (each with its own register
    // allocation and occupancy); what is shared with the small sets is that the evaluation stops at its partial
    // records and camcal_reduce_solve does the rest in one launch
    if (counter) HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned), s));
    if (max_iter > 0) {
      const int rc = pose_refine_impl(model, tiles, intr, pose, max_iter, tol, iters, predict ? Y : nullptr,
                                      predict ? dK : nullptr, scale, cap, stream);
      if (rc) return rc;
    }
    return normal_eq_fused_launch(model, tiles, intr, pose, 0.0, nullptr, nullptr, nullptr, Y, scratch, stream);
  }
  if (G == 64) hipLaunchKernelGGL((k_refine_eval_grp<FISHEYE, 64>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, predict, dK, scale, cap, scratch, np, counter);
  else hipLaunchKernelGGL((k_refine_eval_grp<FISHEYE, 16>), gg, b, 0, s, *tiles, intr, pose, max_iter, tol, iters, Y, predict, dK, scale, cap, scratch, np, counter);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_reduce_solve(int model, const camcal_tiles* tiles, double* scratch, double* rec, int do_solve,
                        const uint8_t* mask_host, double lam, double scale, const double* intr_in, double* intr_out,
                        double* dK, double* stats, double* notify, double seq, void* stream) {
  if (!tiles_ok(tiles) || !scratch || !rec) return CAMCAL_E_BADARG;
  if (do_solve && (!mask_host || !intr_in || !intr_out || !dK || !stats)) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int G = group_lanes(tiles);
  if (model != CAMCAL_FISHEYE) return CAMCAL_E_UNSUPPORTED;
  const int np = tiles->n_tiles * G;                   // partial records: wavefronts of a grouped launch, else tiles
  const int rows = IntrAcc<FISHEYE>::NR + camcal_schur_len(CAMCAL_FISHEYE);
  unsigned mask = 0;
  if (do_solve)
    for (int a = 0; a < 8; ++a)
      if (mask_host[a]) mask |= 1u << a;
  hipLaunchKernelGGL(k_reduce_solve<FISHEYE>, dim3(rows), dim3(TPB), 0, (hipStream_t)stream, scratch, np,
                     small_set_red(tiles, G, rows, scratch), small_set_counter(tiles, G, rows, scratch), rec, do_solve, mask,
                     lam, scale, intr_in, intr_out, dK, stats, notify, seq);
  HIP_TRY(hipGetLastError());
  return 0;
}

// pinhole LM step of a small set: reduced values and the last-workgroup counter at a fixed place behind
// every layout of partial records these launches use
static inline double* lm_red(double* scratch) { return scratch + (size_t)MAX_NV * PARTIAL_STRIDE; }
static inline unsigned* lm_counter(double* scratch) { return (unsigned*)(lm_red(scratch) + 2 * MAX_NV - 8); }
static inline unsigned mask_bits(const uint8_t* mask_host, int ni) {
  unsigned m = 0;
  for (int a = 0; a < ni; ++a)
    if (mask_host[a]) m |= 1u << a;
  return m;
}

int camcal_small_set_begin(int model, const camcal_tiles* tiles, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !scratch) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  if (model != CAMCAL_PINHOLE || !folded_set(tiles, model)) return CAMCAL_E_UNSUPPORTED;
  HIP_TRY(hipMemsetAsync(lm_counter(scratch), 0, sizeof(unsigned), (hipStream_t)stream));
  return 0;
}

int camcal_cross_schur_solve(int model, const camcal_tiles* tiles, const double* intr, const double* pose, const double* V,
                             const double* g, double lam, double* W_out, const double* glob, double* schur,
                             const uint8_t* mask_host, double scale, double* intr_out, double* dK, double* stats,
                             double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !V || !g || !glob || !schur || !mask_host || !intr_out || !dK ||
      !stats || !scratch)
    return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int G = group_lanes(tiles);
  if (model != CAMCAL_PINHOLE || !folded_set(tiles, model)) return CAMCAL_E_UNSUPPORTED;
  const int np = tiles->n_tiles * G, ns = camcal_schur_len(model);
  const dim3 gg(grid_groups(tiles, G)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (G == 64) hipLaunchKernelGGL((k_cross_schur_grp<PINHOLE, 64>), gg, b, 0, s, *tiles, intr, pose, V, g, lam, W_out, scratch, np);
  else hipLaunchKernelGGL((k_cross_schur_grp<PINHOLE, 16>), gg, b, 0, s, *tiles, intr, pose, V, g, lam, W_out, scratch, np);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_schur_solve<9>, dim3(ns), b, 0, s, scratch, np, np, lm_red(scratch), lm_counter(scratch), glob,
                     schur, mask_bits(mask_host, 9), lam, scale, intr, intr_out, dK, stats);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_schur_solve(int model, const camcal_tiles* tiles, const double* V, const double* W, const double* g, double lam,
                       const double* glob, double* schur, const uint8_t* mask_host, double scale, const double* intr_in,
                       double* intr_out, double* dK, double* stats, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !V || !W || !g || !glob || !schur || !mask_host || !intr_in || !intr_out || !dK || !stats || !scratch)
    return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  if (model != CAMCAL_PINHOLE || !folded_set(tiles, model)) return CAMCAL_E_UNSUPPORTED;
  const int gp = grid_persistent(tiles), ns = camcal_schur_len(model);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_schur<9>, dim3(gp), dim3(TPB), 0, s, *tiles, V, W, g, lam, scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_schur_solve<9>, dim3(ns), dim3(TPB), 0, s, scratch, PARTIAL_STRIDE, gp * WPB, lm_red(scratch),
                     lm_counter(scratch), glob, schur, mask_bits(mask_host, 9), lam, scale, intr_in, intr_out, dK, stats);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_backsub_intr_normal(int model, const camcal_tiles* tiles, const double* V, const double* W, const double* g,
                               double lam, const double* dK, const double* pose_in, double* pose_out,
                               const double* intr_trial, double* glob_trial, double* V_trial, double* g_trial, double* norms,
                               const double* stats, double* notify, double seq, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !V || !W || !g || !dK || !pose_in || !pose_out || !intr_trial || !glob_trial ||
      !V_trial || !g_trial || !norms || !stats || !scratch)
    return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int G = group_lanes(tiles);
  if (model != CAMCAL_PINHOLE || !folded_set(tiles, model)) return CAMCAL_E_UNSUPPORTED;
  const int np = tiles->n_tiles * G;
  const dim3 gg(grid_groups(tiles, G)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (G == 64) hipLaunchKernelGGL((k_backsub_eval_grp<PINHOLE, 64>), gg, b, 0, s, *tiles, V, W, g, lam, dK, pose_in, pose_out, intr_trial, V_trial, g_trial, scratch, np);
  else hipLaunchKernelGGL((k_backsub_eval_grp<PINHOLE, 16>), gg, b, 0, s, *tiles, V, W, g, lam, dK, pose_in, pose_out, intr_trial, V_trial, g_trial, scratch, np);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_trial<PINHOLE>, dim3(IntrAcc<PINHOLE>::NR + 2), b, 0, s, scratch, np, lm_red(scratch),
                     lm_counter(scratch), glob_trial, norms, stats, notify, seq);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_intr_normal(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                       int with_normal, double* glob, double* view_cost, double* V, double* g, double* scratch,
                       void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !glob || !scratch) return CAMCAL_E_BADARG;
  if ((V == nullptr) != (g == nullptr) || (V && !with_normal)) return CAMCAL_E_BADARG;
  const int len = camcal_glob_len(model);
  if (len < 0) return len;
  int gp = grid_persistent(tiles);
  const dim3 b(TPB);
  hipStream_t s = (hipStream_t)stream;
  double* red = scratch + (size_t)MAX_NV * PARTIAL_STRIDE; // reduced values ahead of their expansion
  const int nr = model == CAMCAL_PINHOLE ? IntrAcc<PINHOLE>::NR : IntrAcc<FISHEYE>::NR;
  const int ci = model == CAMCAL_PINHOLE ? IntrAcc<PINHOLE>::COST : IntrAcc<FISHEYE>::COST;
  const int G = group_lanes(tiles);
  if (G > 1) {
    gp = grid_groups(tiles, G);                             // one partial per wavefront, <= PARTIAL_STRIDE of them
    const dim3 gg(gp);
    const int mode = (with_normal && V) ? 2 : with_normal ? 1 : 0;
#define LAUNCHG(M, MODE, GG) hipLaunchKernelGGL((k_intr_normal_grp<M, MODE, GG>), gg, b, 0, s, *tiles, intr, pose, view_cost, V, g, scratch)
#define LAUNCHGM(M, GG) do { if (mode == 2) LAUNCHG(M, 2, GG); else if (mode == 1) LAUNCHG(M, 1, GG); else LAUNCHG(M, 0, GG); } while (0)
    if (model == CAMCAL_PINHOLE) { if (G == 64) LAUNCHGM(PINHOLE, 64); else LAUNCHGM(PINHOLE, 16); }
    else { if (G == 64) LAUNCHGM(FISHEYE, 64); else LAUNCHGM(FISHEYE, 16); }
#undef LAUNCHGM
#undef LAUNCHG
  } else {
  const dim3 gr(gp);
#define LAUNCH2(M, P) hipLaunchKernelGGL((k_intr_normal<M, 2, P>), gr, b, 0, s, *tiles, intr, pose, view_cost, V, g, scratch)
#define LAUNCH1(M, P) hipLaunchKernelGGL((k_intr_normal<M, 1, P>), gr, b, 0, s, *tiles, intr, pose, view_cost, V, g, scratch)
#define LAUNCH0(M, P) hipLaunchKernelGGL((k_intr_normal<M, 0, P>), gr, b, 0, s, *tiles, intr, pose, view_cost, V, g, scratch)
  if (with_normal && V) DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH2);
  else if (with_normal) DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH1);
  else DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH0);
#undef LAUNCH0
#undef LAUNCH1
#undef LAUNCH2
  }
  HIP_TRY(hipGetLastError());
  if (with_normal) hipLaunchKernelGGL(k_finalize, dim3(nr), dim3(TPB), 0, s, scratch, gp * WPB, red);
  else hipLaunchKernelGGL(k_finalize, dim3(1), dim3(TPB), 0, s, scratch + (size_t)ci * PARTIAL_STRIDE, gp * WPB, red + ci);
  HIP_TRY(hipGetLastError());
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_expand<PINHOLE>, dim3(1), dim3(64), 0, s, red, glob, 0, with_normal ? 0 : 1);
  else hipLaunchKernelGGL(k_expand<FISHEYE>, dim3(1), dim3(64), 0, s, red, glob, 0, with_normal ? 0 : 1);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_cross_schur(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                       const double* V, const double* g, double lam, double* W_out, double* schur,
                       double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !V || !g || !schur || !scratch) return CAMCAL_E_BADARG;
  const int ns = camcal_schur_len(model);
  if (ns < 0) return ns;
  const dim3 gr(grid_tiles(tiles)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  const int nt = tiles->n_tiles;
  const int G = group_lanes(tiles);
  if (G > 1) {
    const int np = nt * G;
    const dim3 gg(grid_groups(tiles, G));
#define LAUNCHG(M, GG) hipLaunchKernelGGL((k_cross_schur_grp<M, GG>), gg, b, 0, s, *tiles, intr, pose, V, g, lam, W_out, scratch, np)
    if (model == CAMCAL_PINHOLE) { if (G == 64) LAUNCHG(PINHOLE, 64); else LAUNCHG(PINHOLE, 16); }
    else { if (G == 64) LAUNCHG(FISHEYE, 64); else LAUNCHG(FISHEYE, 16); }
#undef LAUNCHG
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_strided, dim3(ns), dim3(TPB), 0, s, scratch, np, np, schur);
    HIP_TRY(hipGetLastError());
    return 0;
  }
#define LAUNCHS(M, P) hipLaunchKernelGGL((k_cross_schur<M, true, P>), gr, b, 0, s, *tiles, intr, pose, V, g, lam, W_out, scratch, nt)
#define LAUNCHN(M, P) hipLaunchKernelGGL((k_cross_schur<M, false, P>), gr, b, 0, s, *tiles, intr, pose, V, g, lam, W_out, scratch, nt)
  if (W_out) DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCHS);
  else DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCHN);
#undef LAUNCHS
#undef LAUNCHN
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize_strided, dim3(ns), dim3(TPB), 0, s, scratch, nt, nt, schur);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_view_blocks(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double* V,
                       double* W, double* g, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !V || !W || !g) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const dim3 gr(grid_tiles(tiles)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(M, P) hipLaunchKernelGGL((k_view_blocks<M, P>), gr, b, 0, s, *tiles, intr, pose, V, W, g)
  DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_schur(int model, const camcal_tiles* tiles, const double* V, const double* W, const double* g,
                 double lam, double* schur, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !V || !W || !g || !schur || !scratch) return CAMCAL_E_BADARG;
  const int len = camcal_schur_len(model);
  if (len < 0) return len;
  const int gp = grid_persistent(tiles);
  hipStream_t s = (hipStream_t)stream;
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_schur<9>, dim3(gp), dim3(TPB), 0, s, *tiles, V, W, g, lam, scratch);
  else hipLaunchKernelGGL(k_schur<8>, dim3(gp), dim3(TPB), 0, s, *tiles, V, W, g, lam, scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize, dim3(len), dim3(TPB), 0, s, scratch, gp * WPB, schur);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_solve_reduced(int model, const double* glob, const double* schur, const uint8_t* mask_host, double lam,
                         double scale, const double* intr_in, double* intr_out, double* dK, double* stats,
                         void* stream) {
  const int ni = camcal_n_intr(model);
  if (ni < 0) return ni;
  if (!glob || !schur || !mask_host || !intr_in || !intr_out || !dK || !stats) return CAMCAL_E_BADARG;
  unsigned mask = 0;
  for (int a = 0; a < ni; ++a)
    if (mask_host[a]) mask |= 1u << a;
  hipLaunchKernelGGL(k_solve_reduced, dim3(1), dim3(64), 0, (hipStream_t)stream, ni, glob, schur, mask, lam, scale,
                     intr_in, intr_out, dK, stats);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_backsub(int model, const camcal_tiles* tiles, const double* V, const double* W, const double* g,
                   double lam, const double* dK, const double* pose_in, double* pose_out, double* norms,
                   double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !V || !W || !g || !dK || !pose_in || !pose_out || !norms || !scratch) return CAMCAL_E_BADARG;
  const int ni = camcal_n_intr(model);
  if (ni < 0) return ni;
  const int gp = grid_persistent(tiles);
  hipStream_t s = (hipStream_t)stream;
  if (ni == 9) hipLaunchKernelGGL(k_backsub<9>, dim3(gp), dim3(TPB), 0, s, *tiles, V, W, g, lam, dK, pose_in, pose_out, scratch);
  else hipLaunchKernelGGL(k_backsub<8>, dim3(gp), dim3(TPB), 0, s, *tiles, V, W, g, lam, dK, pose_in, pose_out, scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize, dim3(2), dim3(TPB), 0, s, scratch, gp * WPB, norms);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_normal_eq(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double lam,
                     double* V, double* W, double* g, double* glob, double* schur, double* scratch, void* stream) {
  if (!V || !g) return CAMCAL_E_BADARG;
  int rc = camcal_intr_normal(model, tiles, intr, pose, 1, glob, nullptr, V, g, scratch, stream);
  if (rc) return rc;
  return camcal_cross_schur(model, tiles, intr, pose, V, g, lam, W, schur, scratch, stream);
}

}  // extern "C"

namespace {
// camcal_normal_eq_fused, optionally (Y != NULL; then no block stores and lam = 0) leaving the
// views' first-order pose response for camcal_pose_predict
// the first stage of camcal_normal_eq_fused alone: the partial records [U|b|cost|C|d], one per wavefront (several
// lanes per view) or per tile, row stride = their number
int normal_eq_fused_launch(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double lam,
                           double* V, double* W, double* g, double* Y, double* scratch, void* stream) {
  const bool wb = V && W && g;
  const dim3 gr(grid_tiles(tiles)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  const int nt = tiles->n_tiles;
  const int G = group_lanes(tiles);
  if (G > 1) {
    const int np = nt * G;                              // one partial record per wavefront
    const dim3 gg(grid_groups(tiles, G));
#define LAUNCHG(M, GG) hipLaunchKernelGGL((k_normal_fused_grp<M, GG>), gg, b, 0, s, *tiles, intr, pose, lam, V, W, g, scratch, np, 0, Y)
    if (model == CAMCAL_PINHOLE) { if (G == 64) LAUNCHG(PINHOLE, 64); else LAUNCHG(PINHOLE, 16); }
    else { if (G == 64) LAUNCHG(FISHEYE, 64); else LAUNCHG(FISHEYE, 16); }
#undef LAUNCHG
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (wb) {
#define LAUNCH(M, P) hipLaunchKernelGGL((k_normal_fused<M, true, P>), gr, b, 0, s, *tiles, intr, pose, lam, V, W, g, scratch, nt, nullptr)
    DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  } else if (Y) {
#define LAUNCH(M, P) hipLaunchKernelGGL((k_normal_fused<M, false, P, false, true>), gr, b, 0, s, *tiles, intr, pose, lam, V, W, g, scratch, nt, Y)
    DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  } else {
#define LAUNCH(M, P) hipLaunchKernelGGL((k_normal_fused<M, false, P>), gr, b, 0, s, *tiles, intr, pose, lam, V, W, g, scratch, nt, nullptr)
    DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int normal_eq_fused_impl(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double lam,
                         double* V, double* W, double* g, double* Y, double* glob, double* schur, double* scratch,
                         void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !glob || !schur || !scratch) return CAMCAL_E_BADARG;
  const int ng = camcal_glob_len(model), ns = camcal_schur_len(model);
  if (ng < 0) return ng;
  const bool wb = V && W && g;
  if (!wb && (V || W || g)) return CAMCAL_E_BADARG;
  if (Y && (wb || lam != 0.0)) return CAMCAL_E_BADARG;
  if (schur != glob + ng) return CAMCAL_E_BADARG;
  const int rc = normal_eq_fused_launch(model, tiles, intr, pose, lam, V, W, g, Y, scratch, stream);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int nr = model == CAMCAL_PINHOLE ? IntrAcc<PINHOLE>::NR : IntrAcc<FISHEYE>::NR;
  const int np = tiles->n_tiles * group_lanes(tiles);   // partial records: wavefronts of a grouped launch, else tiles
  double* red = scratch + (size_t)(nr + ns) * np;
  hipLaunchKernelGGL(k_finalize_strided, dim3(nr + ns), dim3(TPB), 0, s, scratch, np, np, red);
  HIP_TRY(hipGetLastError());
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_expand<PINHOLE>, dim3(1), dim3(64), 0, s, red, glob, ns, 0);
  else hipLaunchKernelGGL(k_expand<FISHEYE>, dim3(1), dim3(64), 0, s, red, glob, ns, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" {

int camcal_normal_eq_fused(int model, const camcal_tiles* tiles, const double* intr, const double* pose, double lam,
                           double* V, double* W, double* g, double* glob, double* schur, double* scratch, void* stream) {
  return normal_eq_fused_impl(model, tiles, intr, pose, lam, V, W, g, nullptr, glob, schur, scratch, stream);
}

int camcal_normal_eq_fused_response(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                                    double* Y, double* glob, double* schur, double* scratch, void* stream) {
  if (!Y) return CAMCAL_E_BADARG;
  return normal_eq_fused_impl(model, tiles, intr, pose, 0.0, nullptr, nullptr, nullptr, Y, glob, schur, scratch, stream);
}

int camcal_pose_predict(int model, const camcal_tiles* tiles, const double* Y, const double* dK, double scale,
                        const int32_t* iters, int cap, double* pose, void* stream) {
  if (!tiles_ok(tiles) || !Y || !dK || !pose) return CAMCAL_E_BADARG;
  const int ni = camcal_n_intr(model);
  if (ni < 0) return ni;
  const dim3 gr((unsigned)((tiles->n_slots + TPB - 1) / TPB)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (ni == 9) hipLaunchKernelGGL(k_pose_predict<9>, gr, b, 0, s, *tiles, Y, dK, scale, iters, cap, pose);
  else hipLaunchKernelGGL(k_pose_predict<8>, gr, b, 0, s, *tiles, Y, dK, scale, iters, cap, pose);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_normal_eq_fused_static(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                                  double lam, double* glob, double* schur, void* stream) {
  if (!tiles_ok(tiles)) return CAMCAL_E_BADARG;
  if (camcal_glob_len(model) < 0) return CAMCAL_E_BADMODEL;
  // partial records of the launch: one per tile, or one per wavefront for a grouped set
  const int64_t partials = (int64_t)tiles->n_tiles * group_lanes(tiles);
  if (partials > STATIC_MAX_PARTIALS) return CAMCAL_E_TOOBIG;
  double* ws = nullptr;
  const int rc = static_scratch_ptr(&ws);
  if (rc) return rc;
  return camcal_normal_eq_fused(model, tiles, intr, pose, lam, nullptr, nullptr, nullptr, glob, schur, ws, stream);
}

int camcal_time_stage(int stage, int reps, int model, const camcal_tiles* tiles, const double* intr,
                      const double* pose, double lam, double* V, double* W, double* g, double* glob, double* schur,
                      double* scratch, void* stream, float* ms_out_host) {
  if (reps <= 0 || !ms_out_host || stage < 0 || stage > 5) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  int rc = 0;
  HIP_TRY(hipEventRecord(e0, s));
  for (int r = 0; r < reps && rc == 0; ++r) {
    if (stage == 0) rc = camcal_intr_normal(model, tiles, intr, pose, 1, glob, nullptr, V, g, scratch, stream);
    else if (stage == 1) rc = camcal_view_blocks(model, tiles, intr, pose, V, W, g, stream);
    else if (stage == 2) rc = camcal_schur(model, tiles, V, W, g, lam, schur, scratch, stream);
    else if (stage == 3) rc = camcal_normal_eq(model, tiles, intr, pose, lam, V, W, g, glob, schur, scratch, stream);
    else if (stage == 4) rc = camcal_normal_eq_fused(model, tiles, intr, pose, lam, nullptr, nullptr, nullptr, glob, schur, scratch, stream);
    else rc = camcal_cross_schur(model, tiles, intr, pose, V, g, lam, nullptr, schur, scratch, stream);
  }
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_out_host = ms / reps;
  return rc;
}

int camcal_project_points(int model, int64_t n, const double* xyz, const double* pose6, const double* intr,
                          double* uv, void* stream) {
  if (n < 0 || (n > 0 && (!xyz || !uv)) || !pose6 || !intr) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  if (n == 0) return 0;
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_project_points<PINHOLE>, g, b, 0, s, n, xyz, pose6, intr, uv);
  else hipLaunchKernelGGL(k_project_points<FISHEYE>, g, b, 0, s, n, xyz, pose6, intr, uv);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_reproj_err(int model, const camcal_tiles* tiles, const double* intr, const double* pose, int f32_round,
                      double* err, double* sum_out, double* scratch, void* stream) {
  if (!tiles_ok(tiles) || !tiles->pts || !intr || !pose || !err || !sum_out || !scratch) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int gp = grid_persistent(tiles);
  const dim3 gr(gp), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (f32_round) {
#define LAUNCH(M, P) hipLaunchKernelGGL((k_reproj_err<M, P, true>), gr, b, 0, s, *tiles, intr, pose, err, scratch)
    DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  } else {
#define LAUNCH(M, P) hipLaunchKernelGGL((k_reproj_err<M, P, false>), gr, b, 0, s, *tiles, intr, pose, err, scratch)
    DISPATCH_MODEL_PLANAR(model, tiles->planar, LAUNCH);
#undef LAUNCH
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize, dim3(2), dim3(TPB), 0, s, scratch, gp * WPB, sum_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_undistort_points(int model, int64_t n, const double* uv, const double* intr, const double* M, double* out,
                            void* stream) {
  if (n < 0 || (n > 0 && (!uv || !out)) || !intr) return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  if (n == 0) return 0;
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_undistort_points<PINHOLE>, g, b, 0, s, n, uv, intr, M, out);
  else hipLaunchKernelGGL(k_undistort_points<FISHEYE>, g, b, 0, s, n, uv, intr, M, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_undistort_rectify_map(int model, int width, int height, const double* intr, const double* iR, int map_type,
                                 void* map1, void* map2, void* stream) {
  if (width <= 0 || height <= 0 || !intr || !iR || !map1 || !map2 || (map_type != 0 && map_type != 1))
    return CAMCAL_E_BADARG;
  if (model != CAMCAL_PINHOLE && model != CAMCAL_FISHEYE) return CAMCAL_E_BADMODEL;
  const int64_t n = (int64_t)width * height;
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (model == CAMCAL_PINHOLE) {
    if (map_type == 0) hipLaunchKernelGGL((k_rectify_map<PINHOLE, 0>), g, b, 0, s, width, height, intr, iR, map1, map2);
    else hipLaunchKernelGGL((k_rectify_map<PINHOLE, 1>), g, b, 0, s, width, height, intr, iR, map1, map2);
  } else {
    if (map_type == 0) hipLaunchKernelGGL((k_rectify_map<FISHEYE, 0>), g, b, 0, s, width, height, intr, iR, map1, map2);
    else hipLaunchKernelGGL((k_rectify_map<FISHEYE, 1>), g, b, 0, s, width, height, intr, iR, map1, map2);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_remap_u8(const unsigned char* src, int src_w, int src_h, int64_t src_stride, int channels,
                    const void* map1, const void* map2, int map_type, int dst_w, int dst_h, unsigned char* dst,
                    int64_t dst_stride, int border_value, void* stream) {
  if (!src || !dst || !map1 || !map2 || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 || channels <= 0 ||
      channels > 4 || src_stride < (int64_t)src_w * channels || dst_stride < (int64_t)dst_w * channels ||
      (map_type != 0 && map_type != 1) || border_value < 0 || border_value > 255)
    return CAMCAL_E_BADARG;
  const int64_t n = (int64_t)dst_w * dst_h;
  const dim3 g((unsigned)((n + TPB - 1) / TPB)), b(TPB);
  hipStream_t s = (hipStream_t)stream;
  if (map_type == 0) hipLaunchKernelGGL(k_remap_u8<0>, g, b, 0, s, src, src_w, src_h, (long long)src_stride, channels, map1, map2, dst_w, dst_h, dst, (long long)dst_stride, (unsigned char)border_value);
  else hipLaunchKernelGGL(k_remap_u8<1>, g, b, 0, s, src, src_w, src_h, (long long)src_stride, channels, map1, map2, dst_w, dst_h, dst, (long long)dst_stride, (unsigned char)border_value);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_peak_probe(int kind, void* buf, int64_t bytes, int reps, void* stream, double* result_host) {
  if (!buf || bytes < 4096 || reps <= 0 || !result_host) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const int iters = 4096;
  const int blocks = 256 * 8;
  for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) {
      if (kind == 0) hipLaunchKernelGGL(k_peak_fma, dim3(blocks), dim3(TPB), 0, s, (double*)buf, iters, 0.999999, 1e-7);
      else if (kind == 1) hipLaunchKernelGGL(k_peak_read, dim3(256 * 16), dim3(TPB), 0, s, (const float4*)buf, (size_t)bytes / 16, (float*)buf);
      else hipLaunchKernelGGL(k_peak_read4, dim3(256 * 16), dim3(TPB), 0, s, (const float*)buf, (size_t)bytes / 4, (float*)buf);
    }
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (kind == 0) *result_host = 2.0 * 16.0 * iters * (double)blocks * TPB * reps / (ms * 1e-3) * 1e-12;   // TFLOP/s
  else *result_host = (double)bytes * reps / (ms * 1e-3) * 1e-9;                                              // GB/s
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Communicator (include/camcal.h): RCCL opened at run time, one ncclAllReduce per record on
// the caller's stream.  The prototypes are RCCL's (rccl.h: ncclGetUniqueId, ncclCommInitRank,
// ncclAllReduce, ncclCommDestroy; ncclDouble = 8, ncclSum = 0, ncclSuccess = 0).
#include <dlfcn.h>
namespace {
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
  int (*get_unique_id)(RcclUniqueId*) = nullptr;
  int (*comm_init_rank)(void** comm, int nranks, RcclUniqueId id, int rank) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void* comm, hipStream_t) = nullptr;
  int (*comm_destroy)(void* comm) = nullptr;
  int (*comm_count)(const void* comm, int* count) = nullptr;
  bool ok = false;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
const RcclApi* rccl_api() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.ok) {
    // by soname: a process that already holds an RCCL (PyTorch's) gets that one
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return nullptr;
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(h, "ncclCommDestroy");
    g_rccl.comm_count = (decltype(g_rccl.comm_count))dlsym(h, "ncclCommCount");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy;
  }
  return g_rccl.ok ? &g_rccl : nullptr;
}
struct RcclComm {
  camcal_comm base;        // first member: the handle handed out
  void* nccl = nullptr;
  int nranks = 0;          // ncclCommCount at creation
};
int rccl_allreduce_sum(void* ctx, double* buf, int64_t n, void* stream) {
  const RcclComm* c = (const RcclComm*)ctx;
  const RcclApi* api = rccl_api();
  if (!c || !c->nccl || !api) return CAMCAL_E_COMM;
  return api->all_reduce(buf, buf, (size_t)n, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->nccl, (hipStream_t)stream) == 0
             ? 0 : CAMCAL_E_COMM;
}
}  // namespace

extern "C" {
int camcal_rccl_unique_id(void* id128_host) {
  if (!id128_host) return CAMCAL_E_BADARG;
  const RcclApi* api = rccl_api();
  if (!api) return CAMCAL_E_COMM;
  return api->get_unique_id((RcclUniqueId*)id128_host) == 0 ? 0 : CAMCAL_E_COMM;
}
int camcal_comm_rccl_create(const void* id128_host, int rank, int world_size, camcal_comm** comm_out) {
  if (!id128_host || !comm_out || world_size < 1 || rank < 0 || rank >= world_size) return CAMCAL_E_BADARG;
  const RcclApi* api = rccl_api();
  if (!api) return CAMCAL_E_COMM;
  RcclUniqueId id;
  std::memcpy(&id, id128_host, sizeof(id));
  RcclComm* c = new RcclComm();
  if (api->comm_init_rank(&c->nccl, world_size, id, rank) != 0 || !c->nccl) {
    delete c;
    return CAMCAL_E_COMM;
  }
  // what RCCL itself says the communicator spans (not what the caller asked for)
  if (!api->comm_count || api->comm_count(c->nccl, &c->nranks) != 0) c->nranks = -1;
  c->base.allreduce_sum = rccl_allreduce_sum;
  c->base.ctx = c;
  *comm_out = &c->base;
  return 0;
}
int camcal_comm_destroy(camcal_comm* comm) {
  if (!comm) return 0;
  if (comm->allreduce_sum != rccl_allreduce_sum) return CAMCAL_E_BADARG;   // not one of ours
  RcclComm* c = (RcclComm*)comm->ctx;
  const RcclApi* api = rccl_api();
  if (api && c->nccl) api->comm_destroy(c->nccl);
  delete c;
  return 0;
}
int camcal_comm_size(const camcal_comm* comm) {
  if (!comm) return 1;
  if (comm->allreduce_sum != rccl_allreduce_sum) return CAMCAL_E_BADARG;
  const int n = ((const RcclComm*)comm->ctx)->nranks;
  return n > 0 ? n : CAMCAL_E_COMM;
}
int camcal_comm_allreduce(const camcal_comm* comm, double* buf, int64_t n, void* stream) {
  if (!comm) return 0;
  if (!comm->allreduce_sum || !buf || n <= 0) return CAMCAL_E_BADARG;
  return comm->allreduce_sum(comm->ctx, buf, n, stream);
}
}  // extern "C"

// ---------------------------------------------------------------------------
// Host-side solver loops (camcal_driver.h) and their memory helpers.
namespace {
// Pinned landing buffers + events for the loops' read-backs: the copies are plain DMAs (no
// staging), and waiting on the event instead of the stream lets kernels launched after
// begin() run while the host reads.  One ReadBack per RUNNING solve, taken from a pool and
// returned to it (the pool grows to the peak number of concurrent solves per device and is
// kept until the process ends -- no per-thread state to leak).
constexpr int READ_CAP = 64;
struct ReadBack {
  double* pinned = nullptr;
  double* mapped = nullptr;        // the same buffer as the device sees it (NULL: not mapped)
  double seq = 0.0;                // last sequence number handed out for pinned[NOTIFY_SEQ]
  hipEvent_t ev = nullptr;
  int dev = -1;
};
constexpr int NOTIFY_SEQ = CAMCAL_NOTIFY_SEQ;
std::mutex g_rb_mu;
std::vector<ReadBack*> g_rb_free;

struct camcal_reader {
  ReadBack* rb = nullptr;
  bool pending = false;
  camcal_reader() = default;
  camcal_reader(const camcal_reader&) = delete;
  camcal_reader& operator=(const camcal_reader&) = delete;
  ~camcal_reader() {
    if (!rb) return;
    if (pending) (void)hipEventSynchronize(rb->ev);       // never hand out a buffer a DMA may still write
    std::lock_guard<std::mutex> lk(g_rb_mu);
    g_rb_free.push_back(rb);
  }
  int acquire() {
    if (rb) return 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lk(g_rb_mu);
      for (size_t i = 0; i < g_rb_free.size(); ++i)
        if (g_rb_free[i]->dev == dev) {
          rb = g_rb_free[i];
          g_rb_free.erase(g_rb_free.begin() + (long)i);
          return 0;
        }
    }
    ReadBack* n = new ReadBack();
    n->dev = dev;
    // coherent + mapped: kernels may write the landing buffer themselves (notify / wait below)
    hipError_t e = hipHostMalloc((void**)&n->pinned, READ_CAP * sizeof(double),
                                 hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, n->pinned, 0) == hipSuccess) n->mapped = (double*)d;
      else (void)hipGetLastError();
    } else {
      (void)hipGetLastError();
      e = hipHostMalloc((void**)&n->pinned, READ_CAP * sizeof(double), hipHostMallocPortable);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      if (n->pinned) (void)hipHostFree(n->pinned);
      delete n;
      return (int)e;
    }
    rb = n;
    return 0;
  }
  int begin(const double* src, int n, int offset, void* stream) {
    if (n <= 0 || offset < 0 || offset + n > READ_CAP || !src) return CAMCAL_E_BADARG;
    const int rc = acquire();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(rb->pinned + offset, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(rb->ev, s));
    pending = true;
    return 0;
  }
  int end(double* dst_host, int n_total) {
    if (n_total <= 0 || n_total > READ_CAP || !dst_host || !rb || !pending) return CAMCAL_E_BADARG;
    HIP_TRY(hipEventSynchronize(rb->ev));
    pending = false;
    for (int i = 0; i < n_total; ++i) dst_host[i] = rb->pinned[i];
    return 0;
  }
  // A kernel writes the first doubles of the landing buffer itself and the sequence number last
  // (camcal_reduce_solve): notify() = where (device view; NULL: this build / device cannot, use begin / end)
  // and which number; wait() = spin on the number, looking at the stream now and then so that a launch that
  // died does not leave the host spinning.
  double* notify(double* seq_out) {
    static const bool off = std::getenv("CAMCAL_NO_NOTIFY") != nullptr;      // tests: the copy + event path instead
    if (off || acquire() != 0 || !rb->mapped) return nullptr;
    rb->seq += 1.0;
    *seq_out = rb->seq;
    return rb->mapped;
  }
  int wait(double seq, void* stream, double* dst_host, int n) {
    if (!rb || !rb->mapped || n <= 0 || n > NOTIFY_SEQ || !dst_host) return CAMCAL_E_BADARG;
    volatile double* flag = rb->pinned + NOTIFY_SEQ;
    for (unsigned spins = 1;; ++spins) {
      if (*flag == seq) break;
      if ((spins & 1023u) == 0) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) {                       // everything queued has run: the number is there or never will be
          if (*flag == seq) break;
          return CAMCAL_E_NUMERIC;
        }
        if (e != hipErrorNotReady) return (int)e;
        (void)hipGetLastError();
      }
      __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < n; ++i) dst_host[i] = rb->pinned[i];
    return 0;
  }
};

// Side streams for stages that are independent per camera (camcal_rig_init: the pose
// initialisation of 4 cameras x 157 tiles each fills a sixth of the device when the cameras
// run one after another).  fork: the side streams wait for everything queued on `stream`;
// join: `stream` waits for the side streams.  One set per device, kept until the process
// ends; the mutex covers one fork or join (a few enqueues), so concurrent callers on a
// device serialise their side work but stay correct.
struct SideStreams {
  hipStream_t s[CAMCAL_RIG_MAX_CAMS - 1] = {};
  hipEvent_t fork = nullptr, join[CAMCAL_RIG_MAX_CAMS - 1] = {};
  int dev = -1;
};
std::mutex g_side_mu;
std::vector<SideStreams*> g_side;

int side_streams(SideStreams** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  for (SideStreams* k : g_side)
    if (k->dev == dev) { *out = k; return 0; }
  SideStreams* n = new SideStreams();
  n->dev = dev;
  hipError_t e = hipEventCreateWithFlags(&n->fork, hipEventDisableTiming);
  for (int i = 0; i < CAMCAL_RIG_MAX_CAMS - 1 && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&n->s[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&n->join[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) return (int)e;            // a failed set is not kept (and not freed: the process is in trouble)
  g_side.push_back(n);
  *out = n;
  return 0;
}
int camcal_fork_streams(void* stream, int n, void** side_out) {
  side_out[0] = stream;
  if (n <= 1) return 0;
  if (n > CAMCAL_RIG_MAX_CAMS) return CAMCAL_E_BADARG;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStreams* k = nullptr;
  const int rc = side_streams(&k);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(k->fork, (hipStream_t)stream));
  for (int i = 1; i < n; ++i) {
    HIP_TRY(hipStreamWaitEvent(k->s[i - 1], k->fork, 0));
    side_out[i] = k->s[i - 1];
  }
  return 0;
}
int camcal_join_streams(void* stream, int n, void** side) {
  if (n <= 1) return 0;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStreams* k = nullptr;
  const int rc = side_streams(&k);
  if (rc) return rc;
  for (int i = 1; i < n; ++i) {
    HIP_TRY(hipEventRecord(k->join[i - 1], (hipStream_t)side[i]));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, k->join[i - 1], 0));
  }
  return 0;
}

int camcal_copy_doubles(double* dst, const double* src, int n, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
// host <-> workspace copies of camcal_calibrate.h (asynchronous, ordered on the stream)
int camcal_upload(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int camcal_download(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int camcal_stream_sync(void* stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
// Page-locked staging for camcal_calibrate's one upload and one download: a free list, grown on demand and
// kept until the process ends (a production-sized call must not pay hipHostMalloc).
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
};
std::mutex g_pin_mu;
std::vector<PinBuf> g_pin_free, g_pin_used;
void* camcal_pinned_take(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  int best = -1;
  for (size_t i = 0; i < g_pin_free.size(); ++i)
    if (g_pin_free[i].cap >= bytes && (best < 0 || g_pin_free[i].cap < g_pin_free[(size_t)best].cap)) best = (int)i;
  PinBuf b;
  if (best >= 0) {
    b = g_pin_free[(size_t)best];
    g_pin_free.erase(g_pin_free.begin() + best);
  } else {
    const size_t cap = ((bytes + (bytes >> 2)) + 65535) & ~(size_t)65535;
    if (hipHostMalloc(&b.p, cap, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    b.cap = cap;
  }
  g_pin_used.push_back(b);
  return b.p;
}
void camcal_pinned_give(void* p) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  for (size_t i = 0; i < g_pin_used.size(); ++i)
    if (g_pin_used[i].p == p) {
      g_pin_free.push_back(g_pin_used[i]);
      g_pin_used.erase(g_pin_used.begin() + (long)i);
      break;
    }
  while (g_pin_free.size() > 16) {                 // keep the pool bounded: drop the smallest
    size_t k = 0;
    for (size_t i = 1; i < g_pin_free.size(); ++i)
      if (g_pin_free[i].cap < g_pin_free[k].cap) k = i;
    (void)hipHostFree(g_pin_free[k].p);
    g_pin_free.erase(g_pin_free.begin() + (long)k);
  }
}
// A large host array to the device through page-locked staging filled by several threads: the copy engine moves
// pinned memory at 57 GB/s on this part, one thread fills staging at 10 - 15, and hipMemcpyAsync from the caller's
// pageable array (one staging thread inside the runtime) reached 27 GB/s on the 1.77 GB of a 1M-view set.
// Thread t takes chunks t, t + T, ... through two buffers of its own: fill, enqueue the copy on `stream`, record an
// event, and wait for that event before the buffer is filled again.  Returns after the last copy has finished (the
// buffers go back to the pool); small arrays, or no staging to be had: the plain copy.
constexpr size_t BULK_CHUNK_DEFAULT = (size_t)8 << 20;
constexpr int BULK_THREADS = 4, BULK_BUFS = 2;
int camcal_upload_bulk(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
  // CAMCAL_BULK_CHUNK (bytes, read once): the chunk size (tests: a few KB send small arrays this way too)
  static const size_t BULK_CHUNK = [] {
    const char* e = std::getenv("CAMCAL_BULK_CHUNK");
    const size_t v = e ? (size_t)std::strtoull(e, nullptr, 10) : 0;
    return v >= 256 ? v : BULK_CHUNK_DEFAULT;
  }();
  if (bytes < 4 * BULK_CHUNK) return camcal_upload(dst_dev, src_host, bytes, stream);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  void* buf[BULK_THREADS][BULK_BUFS] = {};
  hipEvent_t ev[BULK_THREADS][BULK_BUFS] = {};
  bool ok = true;
  for (int t = 0; t < BULK_THREADS && ok; ++t)
    for (int b = 0; b < BULK_BUFS && ok; ++b) {
      buf[t][b] = camcal_pinned_take(BULK_CHUNK);
      ok = buf[t][b] != nullptr && hipEventCreateWithFlags(&ev[t][b], hipEventDisableTiming) == hipSuccess;
    }
  auto release = [&]() {
    for (int t = 0; t < BULK_THREADS; ++t)
      for (int b = 0; b < BULK_BUFS; ++b) {
        if (ev[t][b]) (void)hipEventDestroy(ev[t][b]);
        if (buf[t][b]) camcal_pinned_give(buf[t][b]);
      }
  };
  if (!ok) {
    (void)hipGetLastError();
    release();
    return camcal_upload(dst_dev, src_host, bytes, stream);
  }
  const size_t n_chunks = (bytes + BULK_CHUNK - 1) / BULK_CHUNK;
  int err[BULK_THREADS] = {};
  auto work = [&](int t, bool own_thread) {
    if (own_thread && hipSetDevice(dev) != hipSuccess) { err[t] = (int)hipErrorInvalidDevice; return; }
    int k = 0;
    for (size_t c = (size_t)t; c < n_chunks; c += BULK_THREADS, ++k) {
      const int b = k % BULK_BUFS;
      hipError_t e = k >= BULK_BUFS ? hipEventSynchronize(ev[t][b]) : hipSuccess;    // its last copy has left the buffer
      const size_t off = c * BULK_CHUNK, len = bytes - off < BULK_CHUNK ? bytes - off : BULK_CHUNK;
      if (e == hipSuccess) {
        std::memcpy(buf[t][b], (const char*)src_host + off, len);
        e = hipMemcpyAsync((char*)dst_dev + off, buf[t][b], len, hipMemcpyHostToDevice, (hipStream_t)stream);
      }
      if (e == hipSuccess) e = hipEventRecord(ev[t][b], (hipStream_t)stream);
      if (e != hipSuccess) { err[t] = (int)e; return; }
    }
    for (int b = 0; b < BULK_BUFS && b < k; ++b) {                                     // drain: the buffers are handed back
      const hipError_t e = hipEventSynchronize(ev[t][b]);
      if (e != hipSuccess) { err[t] = (int)e; return; }
    }
  };
  {
    std::vector<std::thread> th;
    int started = 1;
    try {
      for (; started < BULK_THREADS; ++started) th.emplace_back(work, started, true);
    } catch (...) {
    }
    work(0, false);
    for (int t = started; t < BULK_THREADS; ++t) work(t, false);
    for (auto& x : th) x.join();
  }
  int rc = 0;
  for (int t = 0; t < BULK_THREADS; ++t)
    if (err[t] && !rc) rc = err[t];
  if (rc) (void)hipStreamSynchronize((hipStream_t)stream);     // nothing may still read a buffer that goes back to the pool
  release();
  return rc;
}
int camcal_get_device(int* dev) {
  HIP_TRY(hipGetDevice(dev));
  return 0;
}
int camcal_set_device(int dev) {
  HIP_TRY(hipSetDevice(dev));
  return 0;
}
}  // namespace

#include "camcal_driver.h"
#include "camcal_rig_driver.h"
#include "camcal_calibrate.h"
