// Rig kernels + their C ABI (include/camcal.h, "Multi-camera rig").  Included by
// camcal_kernels.hip after its reduction helpers; per-lane bodies in camcal_rig.h.
//
// One launch covers all cameras (blockIdx.y = camera): an evaluation of the whole rig is
// five launches (blocks, their sums, record expansion, chain rule, its sums), a trial step
// of the LM loop eleven, all sequenced by camcal_rig_lm inside the library.
#pragma once

namespace {

// where each camera's partial records live in the scratch, and how many there are
struct RigParts {
  long long off[CAMCAL_RIG_MAX_CAMS];
  int np[CAMCAL_RIG_MAX_CAMS];
};

template <int NI>
__device__ __forceinline__ void rig_load_params(const camcal_rig& rg, int b, int cam, double* I, double e[6]) {
#pragma unroll
  for (int i = 0; i < NI; ++i) I[i] = rg.intr[b][cam * 16 + i];
#pragma unroll
  for (int c = 0; c < 6; ++c) e[c] = rg.ext[b][cam * 6 + c];
}

// composed pose of a slot (identity-like default for padding slots)
__device__ __forceinline__ void rig_slot_pose(const camcal_rig& rg, int b, int cam, long long slot, const double e[6],
                                              double p[6]) {
  const int idx = rg.cam[cam].slot_pose[slot];
  p[0] = p[1] = p[2] = p[3] = p[4] = 0.0;
  p[5] = 1.0;
  if (idx >= 0) {
    double pp[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) pp[c] = rg.rig_pose[b][c * rg.P + idx];
    rig_compose(pp, e, p);
  }
}

// Single-pass evaluation of every camera's views at the composed poses, blocks left in the
// local tangent frame; one tile per wavefront, partial [U|b|cost] sums per tile.
template <int MODEL, bool PLANAR>
__global__ __launch_bounds__(TPB) void k_rig_eval(camcal_rig rg, int b, RigParts parts) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NG = IntrAcc<MODEL>::NR;
  const int cam = blockIdx.y;
  const camcal_tiles& t = rg.cam[cam].tiles;
  const int tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= t.n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, slot = (int64_t)tile * LANES + lane;
  double I[NI], e[6], p[6];
  rig_load_params<NI>(rg, b, cam, I, e);
  rig_slot_pose(rg, b, cam, slot, e, p);
  const LaneTile tv = lane_tile(t, tile, lane);
  ViewAcc<MODEL> acc;
  IntrAcc<MODEL> gacc;
  gacc.zero();
  lane_normal_fused<MODEL, PLANAR, true>(tv, I, p, acc, gacc);
  double* Vb = rg.cam[cam].V;
  double* Wb = rg.cam[cam].W;
  double* gb = rg.cam[cam].g;
#pragma unroll
  for (int k = 0; k < 21; ++k) Vb[k * S + slot] = acc.V[k];
#pragma unroll
  for (int k = 0; k < NI * 6; ++k) Wb[k * S + slot] = acc.W[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) gb[k * S + slot] = acc.g[k];
  wave_scatter_store<NG>(gacc.v, lane, rg.scratch + parts.off[cam] + tile, parts.np[cam]);
}

// the same with G lanes per view (small planar rigs), one partial per wavefront
template <int MODEL, int G>
__global__ __launch_bounds__(TPB) void k_rig_eval_grp(camcal_rig rg, int b, RigParts parts) {
  constexpr int NI = MT<MODEL>::NI;
  constexpr int NG = IntrAcc<MODEL>::NR;
  const int cam = blockIdx.y;
  const camcal_tiles& t = rg.cam[cam].tiles;
  GroupView gv;
  if (!group_view<G>(t, gv)) return;
  const int wave = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots;
  double I[NI], e[6], p[6], R[9];
  rig_load_params<NI>(rg, b, cam, I, e);
  rig_slot_pose(rg, b, cam, gv.slot, e, p);
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
  if (gv.gl == 0) {
    double* Vb = rg.cam[cam].V;
    double* Wb = rg.cam[cam].W;
    double* gb = rg.cam[cam].g;
#pragma unroll
    for (int k = 0; k < 21; ++k) Vb[k * S + gv.slot] = acc.V[k];
#pragma unroll
    for (int k = 0; k < NI * 6; ++k) Wb[k * S + gv.slot] = acc.W[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) gb[k * S + gv.slot] = acc.g[k];
  }
  wave_scatter_store<NG>(gacc.v, lane, rg.scratch + parts.off[cam] + wave, parts.np[cam]);
}

// second stage, all cameras: out[cam * out_stride + row] = sum of that camera's partials of `row`
__global__ __launch_bounds__(TPB) void k_rig_finalize(const double* __restrict__ scratch, RigParts parts,
                                                      double* __restrict__ out, int out_stride) {
  __shared__ double sh[TPB];
  const int row = blockIdx.x, cam = blockIdx.y;
  const double s = strided_partial_sum(scratch + parts.off[cam] + (long long)row * parts.np[cam], parts.np[cam]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = TPB / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(long long)cam * out_stride + row] = sh[0];
}

// reduced accumulator values of every camera -> its [U|b|cost] record; then the rig's cost
template <int MODEL>
__global__ __launch_bounds__(64) void k_rig_expand(const double* __restrict__ red, int n_cams, double* rec,
                                                   int per, int o_cost, double* notify, double seq) {
  constexpr int NR = IntrAcc<MODEL>::NR, NV = IntrAcc<MODEL>::NV;
  __shared__ double cost[CAMCAL_RIG_MAX_CAMS];
  const int c = threadIdx.x;
  if (c < n_cams) {
    double r[NR], o[NV];
    for (int i = 0; i < NR; ++i) r[i] = red[c * NR + i];
    IntrAcc<MODEL>::expand(r, o);
    for (int i = 0; i < NV; ++i) rec[c * per + i] = o[i];
    cost[c] = o[NV - 1];
  }
  __syncthreads();
  if (c == 0) {
    double s = 0.0;
    for (int k = 0; k < n_cams; ++k) s += cost[k];
    rec[o_cost] = s;
    // the LM loop's five scalars [cost | pose-step norms (2) | intrinsic-step norms (2)] straight to the host's
    // pinned landing buffer, sequence number last (the norms were left in the record by the solve and the
    // back-substitution of this trial point, earlier launches on the stream)
    if (notify) {
      notify[0] = s;
      for (int k = 1; k < 5; ++k) notify[k] = rec[o_cost + k];
      __threadfence_system();
      __hip_atomic_store(&notify[CAMCAL_NOTIFY_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// chain rule to the rig parameters, all cameras (blockIdx.y); persistent waves per camera.
// blockIdx.z = 0: the pose half of a view (Vp, gp, extrinsic rows of Wp, sums U_ee, b_e),
// 1: the intrinsics half (rows 0 .. NI-1 of Wp, sum U_ie) -- camcal_rig.h.  As one body the
// two held 477 doubles per lane and spilled 932 bytes of it to scratch memory (43 us on C4).
template <int NI>
__global__ __launch_bounds__(TPB) void k_rig_assemble(camcal_rig rg, int b, int rows_stride) {
  constexpr int NACC = NI * 6 + 21 + 6;
  const int cam = blockIdx.y;
  const camcal_tiles& t = rg.cam[cam].tiles;
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  const int64_t S = t.n_slots, P = rg.P;
  double e[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) e[c] = rg.ext[b][cam * 6 + c];
  const int32_t* __restrict__ slot_pose = rg.cam[cam].slot_pose;
  const double* __restrict__ rig_pose = rg.rig_pose[b];
  double* __restrict__ Wp = rg.Wp[b] + (int64_t)cam * (NI + 6) * 6 * P;
  double* part = rg.scratch + (long long)cam * NACC * rows_stride + wave;
  if (blockIdx.z == 0) {
    const double* __restrict__ Vb = rg.cam[cam].V;
    const double* __restrict__ gb = rg.cam[cam].g;
    double* __restrict__ Vp = rg.Vp[b] + (int64_t)cam * 21 * P;
    double* __restrict__ gp = rg.gp[b] + (int64_t)cam * 6 * P;
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    for (int tile = wave; tile < t.n_tiles; tile += n_waves) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const int idx = slot_pose[slot];
      if (idx < 0) continue;
      double pp[6], Vl[21], gl[6], oV[21], og[6], oW[36];
#pragma unroll
      for (int c = 0; c < 6; ++c) pp[c] = rig_pose[c * P + idx];
#pragma unroll
      for (int i = 0; i < 21; ++i) Vl[i] = Vb[i * S + slot];
#pragma unroll
      for (int i = 0; i < 6; ++i) gl[i] = gb[i * S + slot];
      RigChain ch;
      rig_chain(pp, e, ch);
      rig_assemble_pose(Vl, gl, ch, oV, og, oW, acc);
#pragma unroll
      for (int i = 0; i < 21; ++i) Vp[i * P + idx] = oV[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) gp[i * P + idx] = og[i];
#pragma unroll
      for (int i = 0; i < 36; ++i) Wp[(NI * 6 + i) * P + idx] = oW[i];
    }
    wave_scatter_store<27>(acc, lane, part + (long long)NI * 6 * rows_stride, rows_stride);
  } else {
    const double* __restrict__ Wb = rg.cam[cam].W;
    double acc[NI * 6];
#pragma unroll
    for (int i = 0; i < NI * 6; ++i) acc[i] = 0.0;
    for (int tile = wave; tile < t.n_tiles; tile += n_waves) {
      const int64_t slot = (int64_t)tile * LANES + lane;
      const int idx = slot_pose[slot];
      if (idx < 0) continue;
      double pp[6], Wl[NI * 6], oW[NI * 6];
#pragma unroll
      for (int c = 0; c < 6; ++c) pp[c] = rig_pose[c * P + idx];
#pragma unroll
      for (int i = 0; i < NI * 6; ++i) Wl[i] = Wb[i * S + slot];
      RigChain ch;
      rig_chain(pp, e, ch);
      rig_assemble_intr<NI>(Wl, ch, oW, acc);
#pragma unroll
      for (int i = 0; i < NI * 6; ++i) Wp[i * P + idx] = oW[i];
    }
    wave_scatter_store<NI * 6>(acc, lane, part, rows_stride);
  }
}

// Pose elimination: per board pose V = sum_c Vp_c (damped) = U^T U, Yt = U^-T Wp^T, z = U^-T g.
// One lane per pose; blockIdx.y selects a chunk of RIG_SCHUR_CHUNK columns of Wp (the
// factorisation is cheap next to the latency of walking all n columns in one thread).
// Yt keeps the pose index fastest: lane-consecutive stores coalesce and k_rig_gram reads
// runs of poses.
constexpr int RIG_SCHUR_CHUNK = 8;
constexpr int RIG_FACTOR_ROWS = 27;                // U (21) | 1 / U_ii (6) per pose, kept for camcal_rig_backsub
template <int G>
__global__ __launch_bounds__(TPB) void k_rig_pose_schur(int n_cams, int64_t P, const double* __restrict__ Vp_all,
                                                        const double* __restrict__ gp_all,
                                                        const double* __restrict__ Wp_all, double lam,
                                                        double* __restrict__ Yt, double* __restrict__ z,
                                                        double* __restrict__ factor) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int ld = n_cams * G;
  // all columns of the chunk are loaded before the first is used: one memory latency per
  // chunk instead of one per column (the loads were most of this kernel)
  const int v0 = blockIdx.y * RIG_SCHUR_CHUNK;
  double y[RIG_SCHUR_CHUNK][6];
#pragma unroll
  for (int c = 0; c < RIG_SCHUR_CHUNK; ++c) {
    const int v = v0 + c < ld ? v0 + c : ld - 1;   // v = cam * G + a; clamped columns are not stored
#pragma unroll
    for (int k = 0; k < 6; ++k) y[c][k] = Wp_all[((int64_t)v * 6 + k) * P + p];
  }
  double U[21], g[6], id[6];
  const bool ok = rig_pose_factor<G>(n_cams, Vp_all, gp_all, P, p, lam, U, g, id);
  if (blockIdx.y == 0) {
    if (ok) chol6_fwd(U, id, g);
#pragma unroll
    for (int k = 0; k < 6; ++k) z[k * P + p] = ok ? g[k] : 0.0;
    // a pose no camera sees keeps a zero factor: its step comes out as zero
#pragma unroll
    for (int k = 0; k < 21; ++k) factor[k * P + p] = ok ? U[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) factor[(21 + k) * P + p] = ok ? id[k] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < RIG_SCHUR_CHUNK; ++c) {
    const int v = v0 + c;
    if (v >= ld) break;
    if (ok) chol6_fwd(U, id, y[c]);
#pragma unroll
    for (int k = 0; k < 6; ++k) Yt[((int64_t)v * 6 + k) * P + p] = ok ? y[c][k] : 0.0;
  }
}

// C = Y^T Y and d = Y^T z on the matrix cores: the one dense fp64 contraction of this path
// ((6P x n)^T (6P x n), n <= 63; z rides along as column n).  v_mfma_f64_16x16x4: A[i][k] in
// lane 16k + i, B[k][j] in lane 16k + j, D[(lane >> 4) + 4r][lane & 15] in register r.
// With A = Y^T and B = Y both fragments of a k-step are the SAME registers: lane l holds
// Y[row][16t + (l & 15)] for column tile t.  The order of the contraction index is free, so
// lane group g = l >> 4 takes poses 4g .. 4g+3 of a chunk of 16 (a contiguous 32-byte run
// of Yt) and uses them in four consecutive k-steps.  A wavefront walks chunks with a
// stride, keeps two of the 10 upper tiles of the 64 x 64 result in registers and writes
// its partial; k_rig_gram_finalize sums the partials in a fixed order (deterministic).
typedef double rig_f64x4 __attribute__((ext_vector_type(4)));
constexpr int RIG_GRAM_TILES = 10;                 // (ti <= tj) of a 4 x 4 tile grid
constexpr int RIG_GRAM_VALS = RIG_GRAM_TILES * 4 * 64;
constexpr int RIG_GRAM_MAX_WAVES = 204;                 // x 5 tile pairs = 1020 wavefronts, one per SIMD (128: C4 0.25 ms slower)

// One wavefront = (chunk group, tile pair): the 10 tiles are 5 pairs, so five times as many
// wavefronts share the MFMA work of a chunk (a 16x16x4 f64 MFMA occupies a SIMD for 64
// cycles; with whole chunks per wavefront C4 ran on an eighth of the device).
template <int TI0, int TJ0, int TI1, int TJ1>
__device__ __forceinline__ void rig_gram_pair(int n, int64_t P, const double* __restrict__ Yt,
                                              const double* __restrict__ z, int group, int n_groups, int lane,
                                              rig_f64x4& a0, rig_f64x4& a1) {
  const int col = lane & 15, grp = lane >> 4;
  const int64_t n_chunks = (P + 15) / 16;
  for (int64_t ch = group; ch < n_chunks; ch += n_groups) {
    const int64_t p0 = ch * 16 + grp * 4;
    double f[6][4][4];               // only the column tiles this pair needs are loaded
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t != TI0 && t != TJ0 && t != TI1 && t != TJ1) continue;
        const int v = 16 * t + col;
        const double* src = v < n ? Yt + ((int64_t)v * 6 + k) * P : (v == n ? z + (int64_t)k * P : nullptr);
#pragma unroll
        for (int s = 0; s < 4; ++s) f[k][t][s] = (src && p0 + s < P) ? src[p0 + s] : 0.0;
      }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f[k][TI0][s], f[k][TJ0][s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f[k][TI1][s], f[k][TJ1][s], a1, 0, 0, 0);
      }
  }
}

__global__ __launch_bounds__(TPB) void k_rig_gram(int n, int64_t P, const double* __restrict__ Yt,
                                                  const double* __restrict__ z, double* __restrict__ part, int n_groups) {
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (wave >= n_groups * 5) return;
  const int group = wave / 5, pair = __builtin_amdgcn_readfirstlane(wave % 5);
  const int lane = threadIdx.x & 63;
  rig_f64x4 a0 = rig_f64x4{0.0, 0.0, 0.0, 0.0}, a1 = a0;
  // tile order of the result: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
  if (pair == 0) rig_gram_pair<0, 0, 0, 1>(n, P, Yt, z, group, n_groups, lane, a0, a1);
  else if (pair == 1) rig_gram_pair<0, 2, 0, 3>(n, P, Yt, z, group, n_groups, lane, a0, a1);
  else if (pair == 2) rig_gram_pair<1, 1, 1, 2>(n, P, Yt, z, group, n_groups, lane, a0, a1);
  else if (pair == 3) rig_gram_pair<1, 3, 2, 2>(n, P, Yt, z, group, n_groups, lane, a0, a1);
  else rig_gram_pair<2, 3, 3, 3>(n, P, Yt, z, group, n_groups, lane, a0, a1);
  double* dst = part + (int64_t)group * RIG_GRAM_VALS + (int64_t)pair * 2 * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dst[r * 64 + lane] = a0[r];
    dst[256 + r * 64 + lane] = a1[r];
  }
}

constexpr int RIG_GF_WAVES = 16;                   // wavefronts of a k_rig_gram_finalize block
__global__ __launch_bounds__(RIG_GF_WAVES * 64) void k_rig_gram_finalize(int n, int n_waves, const double* __restrict__ part,
                                                                         double* __restrict__ Cm, double* __restrict__ d) {
  __shared__ double sh[RIG_GF_WAVES][64];
  const int row = blockIdx.x;                       // (tile index, register): 64 values, one per lane
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  // <= 13 partials per wavefront at RIG_GRAM_MAX_WAVES groups, eight loads in flight; fixed order
  double s = 0.0;
#pragma unroll 8
  for (int w = q; w < n_waves; w += RIG_GF_WAVES) s += part[(int64_t)w * RIG_GRAM_VALS + row * 64 + lane];
  sh[q][lane] = s;
  __syncthreads();
  if (q != 0) return;
  s = 0.0;
#pragma unroll
  for (int k = 0; k < RIG_GF_WAVES; ++k) s += sh[k][lane];
  const int r = row & 3, idx = row >> 2;
  int ti = 0, tj = 0, k = 0;
  for (int a = 0; a < 4; ++a)
    for (int bq = a; bq < 4; ++bq) {
      if (k == idx) { ti = a; tj = bq; }
      ++k;
    }
  const int i = 16 * ti + (lane >> 4) + 4 * r, j = 16 * tj + (lane & 15);
  if (i < n && j < n) {
    if (ti < tj || i <= j) {              // a diagonal tile holds both (i, j) and (j, i): keep the upper one
      Cm[i * n + j] = s;
      Cm[j * n + i] = s;
    }
  } else if (i < n && j == n) {
    d[i] = s;
  }
}

// d_p = U^-1 (z_p - Y_p dG) with the factor U, z = U^-T g and Y = U^-T Wp^T that the pose
// elimination left behind (no second factorisation, no walk over Vp / gp).  Four lanes share
// a pose: each takes a quarter of the n columns of Y, loaded RIG_BS_UNROLL at a time ahead
// of their use; the partial sums meet by two exchanges, the first lane of the quad solves.
constexpr int RIG_BS_UNROLL = 8;
__global__ __launch_bounds__(TPB) void k_rig_backsub(int n, int64_t P, const double* __restrict__ Yt,
                                                     const double* __restrict__ z, const double* __restrict__ factor,
                                                     const double* __restrict__ dG, const double* __restrict__ pose_in,
                                                     double* __restrict__ pose_out, double* __restrict__ scratch) {
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63, sub = lane >> 4, pl = lane & 15;      // sub: which quarter of the columns
  const int per = (n + 3) / 4;
  const int v0 = sub * per, v1 = v0 + per < n ? v0 + per : n;
  double acc[2] = {0.0, 0.0};
  const int64_t n_iter = (P + (int64_t)n_waves * 16 - 1) / ((int64_t)n_waves * 16);   // uniform trip count
  for (int64_t it = 0; it < n_iter; ++it) {
    const int64_t p = (it * n_waves + wave) * 16 + pl;
    const bool live = p < P;
    const int64_t pc = live ? p : P - 1;           // dead lanes read a valid pose and store nothing
    double U[21], id[6], rhs[6], prev[6];
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < 21; ++k) U[k] = factor[k * P + pc];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        id[k] = factor[(21 + k) * P + pc];
        rhs[k] = z[k * P + pc];
        prev[k] = pose_in[k * P + pc];
      }
    }
    double part[6] = {0, 0, 0, 0, 0, 0};
    for (int vb = v0; vb < v1; vb += RIG_BS_UNROLL) {
      double w[RIG_BS_UNROLL][6], dv[RIG_BS_UNROLL];
#pragma unroll
      for (int c = 0; c < RIG_BS_UNROLL; ++c) {
        const int v = vb + c < v1 ? vb + c : v1 - 1;
        dv[c] = vb + c < v1 ? dG[v] : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) w[c][k] = Yt[((int64_t)v * 6 + k) * P + pc];
      }
#pragma unroll
      for (int c = 0; c < RIG_BS_UNROLL; ++c)
#pragma unroll
        for (int k = 0; k < 6; ++k) part[k] = fma(w[c][k], dv[c], part[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      part[k] += __shfl_xor(part[k], 16, 64);
      part[k] += __shfl_xor(part[k], 32, 64);
    }
    if (live && sub == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) rhs[k] -= part[k];
      chol6_bwd(U, id, rhs);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        pose_out[k * P + p] = prev[k] - rhs[k];
        acc[0] += rhs[k] * rhs[k];
        acc[1] += prev[k] * prev[k];
      }
    }
  }
  wave_reduce_store<2>(acc, scratch, wave);
}

// Entry (i, j) of the augmented system as  fa * *pa + fc * *pc + id  with addresses that are
// always valid, so that a thread's 16 entries are 32 independent loads (one round trip to
// memory) instead of 16 branches with a load behind each.
struct RigEntry {
  const double* pa;
  const double* pc;
  double fa, fc, id;
};
__device__ __forceinline__ RigEntry rig_reduced_entry(int i, int j, int n, int NI, int n_cams,
                                                      const double* __restrict__ rec, unsigned long long mask, double lam) {
  RigEntry e = {rec, rec, 0.0, 0.0, 0.0};
  if (j >= n || i > n) return e;
  const int G = NI + 6;
  const int NU = NI * (NI + 1) / 2, NV = NU + NI + 1, NX = NI * 6 + 27;
  const double* Cm = rec + (long long)n_cams * (NV + NX) + 5;   // RigLayout::o_C
  const double* d = Cm + (long long)n * n;
  const int cj = j / G, b = j - cj * G;
  const bool fj = (mask >> j) & 1ull;
  if (i == n) {                                   // right-hand side
    const double* glob = rec + (long long)cj * (NV + NX);
    const double* rx = glob + NV;
    if (fj) {
      e.pa = b < NI ? glob + NU + b : rx + NI * 6 + 21 + (b - NI);
      e.fa = 1.0;
      e.pc = d + j;
      e.fc = -1.0;
    }
    return e;
  }
  if (!fj || !((mask >> i) & 1ull)) {
    e.id = (i == j) ? 1.0 : 0.0;
    return e;
  }
  const int c = i / G, a = i - c * G;
  if (c == cj) {
    const double* glob = rec + (long long)c * (NV + NX);
    const double* rx = glob + NV;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    if (hi < NI) e.pa = glob + tri(NI, lo, hi);
    else if (lo < NI) e.pa = rx + lo * 6 + (hi - NI);
    else e.pa = rx + NI * 6 + tri(6, lo - NI, hi - NI);
    e.fa = (i == j) ? 1.0 + lam : 1.0;
  }
  e.pc = Cm + i * n + j;
  e.fc = -1.0;
  return e;
}

__global__ __launch_bounds__(TPB) void k_rig_solve_reduced(int NI, int n_cams, const double* __restrict__ rec,
                                                           unsigned long long mask, double lam,
                                                           const double* __restrict__ intr_in,
                                                           const double* __restrict__ ext_in,
                                                           double* __restrict__ intr_out, double* __restrict__ ext_out,
                                                           double* __restrict__ dG, double* __restrict__ stats) {
  __shared__ double Lc[RIG_MAXN * RIG_LD];        // Lc[k * RIG_LD + i] = a_ik as published in step k
  __shared__ double sc[RIG_MAXN];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n = n_cams * (NI + 6);
  double v[4][4];
  {
    RigEntry en[4][4];
    double va[4][4], vc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) en[a][b] = rig_reduced_entry(ty + 16 * a, tx + 16 * b, n, NI, n_cams, rec, mask, lam);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) { va[a][b] = *en[a][b].pa; vc[a][b] = *en[a][b].pc; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // (the products are exact: fa is 1 except on the damped diagonal, fc is -1 or 0)
        const double t = en[a][b].fa * va[a][b];
        v[a][b] = (en[a][b].fa != 0.0 ? t : 0.0) + (en[a][b].fc != 0.0 ? -vc[a][b] : 0.0) + en[a][b].id;
      }
  }
  if (tx == ty) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = ty + 16 * a;
      const double dii = v[a][a];
      if (i < n) sc[i] = dii > 0.0 ? 1.0 / sqrt(dii) : 1.0;
    }
  }
  if (tid == 0) sc[n] = 1.0;
  __syncthreads();
  {
    double si[4], sj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      si[a] = ty + 16 * a <= n ? sc[ty + 16 * a] : 0.0;
      sj[a] = tx + 16 * a < n ? sc[tx + 16 * a] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) v[a][b] *= si[a] * sj[b];
  }
  bool ok = true;
  double rpl = 0.0;
  for (int k = 0; k < n; ++k) {
    const int kb = k >> 4;                           // uniform
    if (tx == (k & 15)) {
      double* dst = Lc + k * RIG_LD + ty;
      switch (kb) {                                  // uniform: one scalar branch, no selects
        case 0: dst[0] = v[0][0]; dst[16] = v[1][0]; dst[32] = v[2][0]; dst[48] = v[3][0]; break;
        case 1: dst[0] = v[0][1]; dst[16] = v[1][1]; dst[32] = v[2][1]; dst[48] = v[3][1]; break;
        case 2: dst[0] = v[0][2]; dst[16] = v[1][2]; dst[32] = v[2][2]; dst[48] = v[3][2]; break;
        default: dst[0] = v[0][3]; dst[16] = v[1][3]; dst[32] = v[2][3]; dst[48] = v[3][3]; break;
      }
    }
    __syncthreads();
    const double p = Lc[k * RIG_LD + k];
    const bool good = p > 0.0;
    // 1 / p from v_rcp_f64 (2^-23 or better): with e = 1 - p r0, r = r0 (1 + e + e^2) leaves e^3
    const double r0 = __builtin_amdgcn_rcp(good ? p : 1.0);
    const double er = fma(-p, r0, 1.0);
    const double rp = good ? fma(r0, fma(er, er, er), r0) : 0.0;
    ok = ok && good;
    rpl = tid == k ? rp : rpl;                       // lane k of the first wavefront keeps 1 / p_k
    double aik[4], ajk[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      aik[a] = Lc[k * RIG_LD + ty + 16 * a] * rp;
      ajk[a] = Lc[k * RIG_LD + tx + 16 * a];
    }
    // all ten blocks every step: finished ones take garbage that nothing reads, which is
    // cheaper than a scalar branch per block
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) v[a][b] = fma(-aik[a], ajk[b], v[a][b]);
  }
  __syncthreads();
  if (tid >= 64) return;
  // L^T x = y with the unscaled columns: x_k = (a_nk - sum_{i > k} a_ik x_i) / p_k.  Lane j keeps
  // z_j = y_j / p_j and its column pre-divided, so a step is one broadcast and one fma:
  //   x_k = z_k,   z_j -= (a_kj / p_j) x_k   (j < k).
  const int lane = tid;
  double col[RIG_MAXN];
#pragma unroll
  for (int k = 0; k < RIG_MAXN; ++k) col[k] = (k < n && lane < k) ? Lc[lane * RIG_LD + k] * rpl : 0.0;
  double y = lane < n ? Lc[lane * RIG_LD + n] * rpl : 0.0;
#pragma unroll
  for (int k = RIG_MAXN - 1; k >= 1; --k) {
    if (k < n) {                                   // uniform
      const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), k),
                                         __builtin_amdgcn_readlane(__double2loint(y), k));
      y = fma(-col[k], xk, y);                     // col[k] = 0 for lanes >= k
    }
  }
  // rig_solve_finish, one parameter per lane: out = in - dG, stats = [||dG||^2, ||globals_in||^2];
  // a failed factorisation gives NaN so that the LM driver rejects the step
  double s0 = 0.0, s1 = 0.0;
  if (lane < n) {
    const int G = NI + 6, c = lane / G, a = lane - c * G;
    const double bad = ok ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    const double dx = ((mask >> lane) & 1ull) ? y * sc[lane] + bad : 0.0;
    dG[lane] = dx;
    const double prev = a < NI ? intr_in[c * 16 + a] : ext_in[c * 6 + (a - NI)];
    if (a < NI) intr_out[c * 16 + a] = prev - dx;
    else ext_out[c * 6 + (a - NI)] = prev - dx;
    s0 = dx * dx;
    s1 = prev * prev;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) { stats[0] = s0; stats[1] = s1; }
}


// ---- initialisation ---------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_rig_pose_slot(int64_t S, const int32_t* __restrict__ slot_pose,
                                                       int32_t* __restrict__ pose_slot) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= S) return;
  const int idx = slot_pose[slot];
  if (idx >= 0) pose_slot[idx] = (int32_t)slot;
}

// blockIdx.y = ordered camera pair (c, k), c != k: sums of (R_ck, t_ck, 1) over the poses both see
__global__ __launch_bounds__(TPB) void k_rig_pair_sums(camcal_rig rg, double* __restrict__ scratch) {
  const int nC = rg.n_cams;
  const int c = blockIdx.y / nC, k = blockIdx.y % nC;
  const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * WPB;
  const int lane = threadIdx.x & 63;
  double acc[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] = 0.0;
  if (c != k) {
    const int64_t Sc = rg.cam[c].tiles.n_slots, Sk = rg.cam[k].tiles.n_slots;
    for (int64_t p = (int64_t)wave * LANES + lane; p < rg.P; p += (int64_t)n_waves * LANES) {
      const int sc = rg.cam[c].pose_slot[p], sk = rg.cam[k].pose_slot[p];
      if (sc < 0 || sk < 0) continue;
      double qc[6], qk[6], o[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) { qc[i] = rg.cam[c].pose[i * Sc + sc]; qk[i] = rg.cam[k].pose[i * Sk + sk]; }
      rig_pair_terms(qc, qk, o);
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] += o[i];
      acc[12] += 1.0;
    }
  }
  wave_scatter_store<13>(acc, lane, scratch + (long long)blockIdx.y * 13 * n_waves + wave, n_waves);
}

// board poses in the rig frame from the lowest-numbered camera that sees them (buffer 0)
__global__ __launch_bounds__(TPB) void k_rig_init_poses(camcal_rig rg) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rg.P) return;
  double out[6] = {0, 0, 0, 0, 0, 1};
  for (int c = 0; c < rg.n_cams; ++c) {
    const int s = rg.cam[c].pose_slot[p];
    if (s < 0) continue;
    const int64_t S = rg.cam[c].tiles.n_slots;
    double q[6], e[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { q[i] = rg.cam[c].pose[i * S + s]; e[i] = rg.ext[0][c * 6 + i]; }
    rig_pose_from_cam(q, e, out);
    break;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) rg.rig_pose[0][i * rg.P + p] = out[i];
}

// ---- host helpers -----------------------------------------------------------------------
inline int rig_ok(const camcal_rig* r) {
  if (!r || r->n_cams < 1 || r->n_cams > CAMCAL_RIG_MAX_CAMS || r->P <= 0) return 0;
  const int ni = camcal_n_intr(r->model);
  if (ni < 0 || r->n_cams * (ni + 6) > RIG_MAXN - 1) return 0;
  for (int c = 0; c < r->n_cams; ++c) {
    const camcal_rig_cam& k = r->cam[c];
    if (!tiles_ok(&k.tiles) || !k.tiles.pts || !k.slot_pose || !k.V || !k.W || !k.g) return 0;
  }
  for (int b = 0; b < 2; ++b)
    if (!r->intr[b] || !r->ext[b] || !r->rig_pose[b] || !r->Vp[b] || !r->gp[b] || !r->Wp[b] || !r->rec[b]) return 0;
  return r->Yt && r->z && r->dG && r->scratch;
}

// lanes per view of the rig's evaluation, by the views of all cameras together: 64 / 16 as
// for a single camera, 4 up to 2048 tiles (131072 views: C4 is 500 tiles -- one lane per view
// would leave half the SIMDs idle and walk 88 corners serially), 1 above.  A camera's
// lanes_per_view field forces the mapping (tests).
constexpr int RIG_GROUP16_MAX_TILES = 128;     // measured on C4-shaped rigs: 4 lanes per view are ahead of 16 from here
constexpr int RIG_GROUP4_MAX_TILES = 2048;
inline int rig_group_lanes(const camcal_rig* r) {
  int total = 0, forced = 0;
  for (int c = 0; c < r->n_cams; ++c) {
    total += r->cam[c].tiles.n_tiles;
    if (r->cam[c].tiles.lanes_per_view) forced = r->cam[c].tiles.lanes_per_view;
  }
  const int by_size = total <= GROUP64_MAX_TILES ? 64 : total <= RIG_GROUP16_MAX_TILES ? 16 : total <= RIG_GROUP4_MAX_TILES ? 4 : 1;
  if (forced == 1) return 1;
  if (forced == 64) return total <= GROUP64_MAX_TILES ? 64 : by_size;
  if (forced == 16) return total <= RIG_GROUP16_MAX_TILES ? 16 : by_size;
  return by_size;
}
inline bool rig_planar(const camcal_rig* r) {
  for (int c = 0; c < r->n_cams; ++c)
    if (!r->cam[c].tiles.planar) return false;
  return true;
}
inline int rig_max_tiles(const camcal_rig* r) {
  int m = 0;
  for (int c = 0; c < r->n_cams; ++c) m = r->cam[c].tiles.n_tiles > m ? r->cam[c].tiles.n_tiles : m;
  return m;
}
inline int rig_persistent_blocks(const camcal_rig* r) {
  const int g = (rig_max_tiles(r) + WPB - 1) / WPB;
  return g < MAX_WAVES / WPB ? g : MAX_WAVES / WPB;
}
// scratch = [ work area shared by the stages | pose factors (RIG_FACTOR_ROWS x P), written by
// camcal_rig_eliminate and read by camcal_rig_backsub ]
inline int64_t rig_work_len(int n_cams, const int32_t* n_tiles, int64_t P) {
  (void)P;
  int64_t ev = 0;                                   // evaluation partials: up to 64 per tile (grouped small sets)
  for (int c = 0; c < n_cams; ++c) {
    const int64_t nt = n_tiles[c] > 0 ? n_tiles[c] : 0;
    ev += (int64_t)MAX_NV * (nt <= GROUP64_MAX_TILES ? nt * 64 : nt <= RIG_GROUP16_MAX_TILES ? nt * 16
                                                                 : nt <= RIG_GROUP4_MAX_TILES ? nt * 4 : nt);
  }
  const int64_t as = (int64_t)n_cams * MAX_NV * MAX_WAVES;                   // chain-rule partials
  const int64_t gr = (int64_t)RIG_GRAM_MAX_WAVES * RIG_GRAM_VALS;            // gram partials
  const int64_t in = (int64_t)n_cams * n_cams * 13 * (MAX_WAVES + 1);        // pair sums
  int64_t m = ev > as ? ev : as;
  m = m > gr ? m : gr;
  m = m > in ? m : in;
  return m + (int64_t)CAMCAL_RIG_MAX_CAMS * MAX_NV + 256;                    // + reduced values / small records
}
inline double* rig_factor_ptr(const camcal_rig* r) {
  int32_t nt[CAMCAL_RIG_MAX_CAMS];
  for (int c = 0; c < r->n_cams; ++c) nt[c] = r->cam[c].tiles.n_tiles;
  return r->scratch + rig_work_len(r->n_cams, nt, r->P);
}
inline int rig_gram_waves(int64_t P) {
  const int64_t ch = (P + 15) / 16;
  return (int)(ch < RIG_GRAM_MAX_WAVES ? ch : RIG_GRAM_MAX_WAVES);
}

}  // namespace

extern "C" {

int64_t camcal_rig_rec_len(int model, int n_cams) {
  const int ni = camcal_n_intr(model);
  if (ni < 0 || n_cams < 1 || n_cams > CAMCAL_RIG_MAX_CAMS) return CAMCAL_E_BADARG;
  return RigLayout(ni, n_cams).total;
}

int64_t camcal_rig_scratch_len(int model, int n_cams, const int32_t* n_tiles_host, int64_t P) {
  const int ni = camcal_n_intr(model);
  if (ni < 0 || n_cams < 1 || n_cams > CAMCAL_RIG_MAX_CAMS || !n_tiles_host || P <= 0) return CAMCAL_E_BADARG;
  return rig_work_len(n_cams, n_tiles_host, P) + (int64_t)RIG_FACTOR_ROWS * P;
}

int camcal_rig_residuals(const camcal_rig* rig, int b, void* stream) {
  return camcal_rig_residuals_notify(rig, b, nullptr, 0.0, stream);
}

int camcal_rig_residuals_notify(const camcal_rig* rig, int b, double* notify, double seq, void* stream) {
  if (!rig_ok(rig) || (b != 0 && b != 1)) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int nC = rig->n_cams, model = rig->model;
  const RigLayout L(camcal_n_intr(model), nC);
  const int nr = model == CAMCAL_PINHOLE ? IntrAcc<PINHOLE>::NR : IntrAcc<FISHEYE>::NR;
  const int G = rig_planar(rig) ? rig_group_lanes(rig) : 1;
  RigParts parts;
  long long off = 0;
  int max_blocks = 1;
  for (int c = 0; c < CAMCAL_RIG_MAX_CAMS; ++c) {
    parts.off[c] = off;
    parts.np[c] = 0;
    if (c >= nC) continue;
    const camcal_tiles* t = &rig->cam[c].tiles;
    parts.np[c] = t->n_tiles * G;
    off += (long long)nr * parts.np[c];
    const int blocks = G > 1 ? grid_groups(t, G) : grid_tiles(t);
    max_blocks = blocks > max_blocks ? blocks : max_blocks;
  }
  double* red = rig->scratch + off;                 // [n_cams][nr] reduced values ahead of their expansion
  const dim3 gr(max_blocks, nC), bl(TPB);
  if (G > 1) {
#define LAUNCHG(M, GG) hipLaunchKernelGGL((k_rig_eval_grp<M, GG>), gr, bl, 0, s, *rig, b, parts)
    if (model == CAMCAL_PINHOLE) { if (G == 64) LAUNCHG(PINHOLE, 64); else if (G == 16) LAUNCHG(PINHOLE, 16); else LAUNCHG(PINHOLE, 4); }
    else { if (G == 64) LAUNCHG(FISHEYE, 64); else if (G == 16) LAUNCHG(FISHEYE, 16); else LAUNCHG(FISHEYE, 4); }
#undef LAUNCHG
  } else {
#define LAUNCH(M, PL) hipLaunchKernelGGL((k_rig_eval<M, PL>), gr, bl, 0, s, *rig, b, parts)
    DISPATCH_MODEL_PLANAR(model, rig_planar(rig), LAUNCH);
#undef LAUNCH
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_rig_finalize, dim3(nr, nC), bl, 0, s, rig->scratch, parts, red, nr);
  HIP_TRY(hipGetLastError());
  if (model == CAMCAL_PINHOLE) hipLaunchKernelGGL(k_rig_expand<PINHOLE>, dim3(1), dim3(64), 0, s, red, nC, rig->rec[b], L.per, L.o_cost, notify, seq);
  else hipLaunchKernelGGL(k_rig_expand<FISHEYE>, dim3(1), dim3(64), 0, s, red, nC, rig->rec[b], L.per, L.o_cost, notify, seq);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_rig_chain(const camcal_rig* rig, int b, void* stream) {
  if (!rig_ok(rig) || (b != 0 && b != 1)) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int nC = rig->n_cams;
  const RigLayout L(camcal_n_intr(rig->model), nC);
  const dim3 bl(TPB);
  // chain rule: per-camera sums [U_ie | U_ee | b_e] straight into the records
  const int pb = rig_persistent_blocks(rig), nw = pb * WPB;
  const int nacc = L.NX;
  RigParts ap;
  for (int c = 0; c < CAMCAL_RIG_MAX_CAMS; ++c) { ap.off[c] = (long long)c * nacc * nw; ap.np[c] = nw; }
  if (L.NI == 9) hipLaunchKernelGGL(k_rig_assemble<9>, dim3(pb, nC, 2), bl, 0, s, *rig, b, nw);
  else hipLaunchKernelGGL(k_rig_assemble<8>, dim3(pb, nC, 2), bl, 0, s, *rig, b, nw);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_rig_finalize, dim3(nacc, nC), bl, 0, s, rig->scratch, ap, rig->rec[b] + L.NV, L.per);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_rig_evaluate(const camcal_rig* rig, int b, void* stream) {
  const int rc = camcal_rig_residuals(rig, b, stream);
  return rc ? rc : camcal_rig_chain(rig, b, stream);
}

int camcal_rig_eliminate(const camcal_rig* rig, int b, double lam, void* stream) {
  if (!rig_ok(rig) || (b != 0 && b != 1)) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int nC = rig->n_cams;
  const RigLayout L(camcal_n_intr(rig->model), nC);
  const int64_t P = rig->P;
  const dim3 gr((unsigned)((P + TPB - 1) / TPB), (unsigned)((L.n + RIG_SCHUR_CHUNK - 1) / RIG_SCHUR_CHUNK)), bl(TPB);
  if (L.NI == 9) hipLaunchKernelGGL(k_rig_pose_schur<15>, gr, bl, 0, s, nC, P, rig->Vp[b], rig->gp[b], rig->Wp[b], lam, rig->Yt, rig->z, rig_factor_ptr(rig));
  else hipLaunchKernelGGL(k_rig_pose_schur<14>, gr, bl, 0, s, nC, P, rig->Vp[b], rig->gp[b], rig->Wp[b], lam, rig->Yt, rig->z, rig_factor_ptr(rig));
  HIP_TRY(hipGetLastError());
  const int nw = rig_gram_waves(P);
  hipLaunchKernelGGL(k_rig_gram, dim3((nw * 5 + WPB - 1) / WPB), bl, 0, s, L.n, P, rig->Yt, rig->z, rig->scratch, nw);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_rig_gram_finalize, dim3(RIG_GRAM_TILES * 4), dim3(RIG_GF_WAVES * 64), 0, s, L.n, nw, rig->scratch,
                     rig->rec[b] + L.o_C, rig->rec[b] + L.o_d);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_rig_solve(const camcal_rig* rig, int b, const uint8_t* mask_host, double lam, int out, void* stream) {
  if (!rig_ok(rig) || !mask_host || (b != 0 && b != 1) || (out != 0 && out != 1)) return CAMCAL_E_BADARG;
  const int ni = camcal_n_intr(rig->model);
  const RigLayout L(ni, rig->n_cams);
  unsigned long long mask = 0;
  for (int i = 0; i < L.n; ++i)
    if (mask_host[i]) mask |= 1ull << i;
  hipLaunchKernelGGL(k_rig_solve_reduced, dim3(1), dim3(TPB), 0, (hipStream_t)stream, ni, rig->n_cams, rig->rec[b], mask,
                     lam, rig->intr[b], rig->ext[b], rig->intr[out], rig->ext[out], rig->dG, rig->rec[out] + L.o_gstats);
  HIP_TRY(hipGetLastError());
  return 0;
}

int camcal_rig_backsub(const camcal_rig* rig, int b, double lam, int out, void* stream) {
  if (!rig_ok(rig) || (b != 0 && b != 1) || (out != 0 && out != 1)) return CAMCAL_E_BADARG;
  const RigLayout L(camcal_n_intr(rig->model), rig->n_cams);
  const int64_t P = rig->P;
  int64_t nb = (P + 16 * WPB - 1) / (16 * WPB);         // 16 poses per wavefront
  if (nb > MAX_WAVES / WPB) nb = MAX_WAVES / WPB;
  hipStream_t s = (hipStream_t)stream;
  (void)lam;                                            // the factor of camcal_rig_eliminate(rig, b, lam) is reused
  hipLaunchKernelGGL(k_rig_backsub, dim3((unsigned)nb), dim3(TPB), 0, s, L.n, P, rig->Yt, rig->z, rig_factor_ptr(rig), rig->dG,
                     rig->rig_pose[b], rig->rig_pose[out], rig->scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize, dim3(2), dim3(TPB), 0, s, rig->scratch, (int)nb * WPB, rig->rec[out] + L.o_norms);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"

// internals of camcal_rig_init (camcal_rig_driver.h sequences them)
namespace {
// per-camera pose_slot tables, then sums [c][k][13] over shared poses (device, *sums_dev_out)
int rig_pair_sums(const camcal_rig* rig, double** sums_dev_out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nC = rig->n_cams;
  for (int c = 0; c < nC; ++c) {
    const camcal_rig_cam& k = rig->cam[c];
    if (!k.pose_slot || !k.pose) return CAMCAL_E_BADARG;
    HIP_TRY(hipMemsetAsync(k.pose_slot, 0xFF, (size_t)rig->P * sizeof(int32_t), s));
    const int64_t S = k.tiles.n_slots;
    hipLaunchKernelGGL(k_rig_pose_slot, dim3((unsigned)((S + TPB - 1) / TPB)), dim3(TPB), 0, s, S, k.slot_pose, k.pose_slot);
    HIP_TRY(hipGetLastError());
  }
  int64_t nb = (rig->P + TPB - 1) / TPB;
  if (nb > MAX_WAVES / WPB) nb = MAX_WAVES / WPB;
  const int nw = (int)nb * WPB;
  double* sums_dev = rig->scratch + (size_t)nC * nC * 13 * nw;
  *sums_dev_out = sums_dev;
  hipLaunchKernelGGL(k_rig_pair_sums, dim3((unsigned)nb, nC * nC), dim3(TPB), 0, s, *rig, rig->scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_finalize_strided, dim3(nC * nC * 13), dim3(TPB), 0, s, rig->scratch, nw, nw, sums_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}
int rig_init_poses(const camcal_rig* rig, void* stream) {
  hipLaunchKernelGGL(k_rig_init_poses, dim3((unsigned)((rig->P + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, *rig);
  HIP_TRY(hipGetLastError());
  return 0;
}
int camcal_read_sync(double* dst_host, const double* src_dev, int n, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int camcal_write_sync(double* dst_dev, const double* src_host, int n, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst_dev, src_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
}  // namespace
