/* camcal.h -- C ABI of the MI355X calibration hot path (libcamcal.so).
 *
 * The reference (huangqi2002/CameraCalibration) has no FFI for this path: its
 * boundary is the pair of Python calls
 *     cv2.calibrateCamera(...)    server/aruco/camera_calib.py:153-156
 *                                 server/internal/IntrinsicCalibration/intrinsicCalib.py:194-197,223-226
 *     cv2.fisheye.calibrate(...)  server/aruco/camera_calib.py:74-77
 *                                 server/internal/IntrinsicCalibration/intrinsicCalib.py:102-105,131-134
 * whose arithmetic lives inside OpenCV.  cameracalibration_amd re-exports those
 * two Python signatures and binds the entry points below
 * with ctypes; each entry point is one stage of the solver that OpenCV runs
 * inside those calls (stage <-> OpenCV routine given per function).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - model: 0 = pinhole [fx fy cx cy k1 k2 p1 p2 k3] (NI = 9)
 *            1 = Kannala-Brandt fisheye [fx fy cx cy k1 k2 k3 k4] (NI = 8)
 *   - residual e = projected - observed; doubles everywhere except the packed
 *     float32 observations
 *   - views are packed in tiles of 64 "slots" (one wavefront); slot s of tile T
 *     is lane s%64.  Point c (X Y Z u v = 0..4) of corner k of that lane lives
 *     at pts[tile_off[T] + (k*5 + c)*64 + lane].  lane_n[s] = corner count
 *     (0 for padding slots), tile_nmax[T] = max over the tile.
 *   - per-slot arrays are component-major: pose[c*n_slots + s], c = r0 r1 r2 t0 t1 t2;
 *     V[e*n_slots + s] (21 upper-triangle entries of the 6x6 pose block),
 *     W[(a*6 + c)*n_slots + s] (NI x 6), g[c*n_slots + s]
 *   - fisheye: a point within 1e-8 of the optical axis (where cv2 switches to
 *     cdist = 1) is evaluated by the general formulas at r = 1e-8, which differs
 *     from cv2's special case by less than 1e-15 relative
 *   - all functions return 0 on success, a hipError_t (> 0) if a HIP call
 *     failed, or a negative CAMCAL_E_* code for rejected arguments.
 */
#ifndef CAMCAL_H
#define CAMCAL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMCAL_PINHOLE 0
#define CAMCAL_FISHEYE 1

#define CAMCAL_E_BADARG  (-1)
#define CAMCAL_E_BADMODEL (-2)
#define CAMCAL_E_TOOBIG  (-3)
#define CAMCAL_E_NUMERIC (-4)   /* a solve met a singular system or a non-finite cost (degenerate views) */
#define CAMCAL_E_COMM    (-5)   /* RCCL missing or a collective failed */
#define CAMCAL_E_DISJOINT (-6)  /* camcal_rig_init: a camera shares no board pose with the rest of the rig */
#define CAMCAL_E_UNSUPPORTED (-8) /* camcal_calibrate: a flag combination / target the library does not implement */
#define CAMCAL_E_PEER    (-7)   /* multi-rank call: another rank failed before the collective; every rank returns this */

/* Packed observations of the views owned by this process (device pointers). */
typedef struct camcal_tiles {
  const float*   pts;        /* packed X Y Z u v, see above                    */
  const int64_t* tile_off;   /* [n_tiles] offset of each tile in floats       */
  const int32_t* tile_nmax;  /* [n_tiles]                                      */
  const int32_t* lane_n;     /* [n_slots] corners per slot, 0 = padding        */
  int32_t        n_tiles;
  int64_t        n_slots;    /* == 64 * n_tiles                                */
  int32_t        planar;     /* 1: every object point has Z == 0 (the Z rows are
                                then never read); 0: general 3-D points        */
  int32_t        lanes_per_view; /* how many lanes share one view in the per-view
                                kernels: 0 = chosen by the size of the set (one lane per
                                view for large sets -- the throughput mapping --, 16 or 64
                                lanes per view for small planar sets of <= 28672 / <= 2048
                                views, so that a production-sized call (20 ... 500 views,
                                intrinsicCalib.py:102-105) spreads over the whole device
                                instead of a handful of wavefronts); 1, 16 or 64 force a
                                mapping where the set allows it (tests, measurements).
                                Results differ between mappings only by the order of the
                                sums.  A field of the set, not a process-wide switch:
                                concurrent solves do not interfere.               */
} camcal_tiles;

/* Library version (major*10000 + minor*100 + patch). */
int camcal_version(void);

/* Number of doubles in the global normal-equation record of a model:
 *   [ U (NI(NI+1)/2) | b (NI) | cost ]  ->  camcal_glob_len
 *   [ C (NI(NI+1)/2) | d (NI) ]         ->  camcal_schur_len                  */
int camcal_n_intr(int model);
int camcal_glob_len(int model);
int camcal_schur_len(int model);

/* Doubles of scratch the reducing entry points need (partial sums of the
 * launched wavefronts) for a problem of n_tiles tiles: at least 109 x 16384 + 192
 * (14.3 MB), more above 9300 tiles. */
int64_t camcal_scratch_len(int model, int n_tiles);

/* Host-side layout of n_views views with the given corner counts (HOST pointers, no
 * device work): views sorted by descending count (stable) fill the slots, 64 slots per
 * tile.  Pass slot_view == NULL to query the sizes only.  Outputs: n_tiles, n_slots
 * (= 64 n_tiles), total_floats (length of pts), slot_view [n_slots] (view of each slot,
 * -1 = padding), lane_n [n_slots], tile_nmax [n_tiles], tile_off [n_tiles]; copy the four
 * arrays to the device to fill camcal_tiles.  No reference counterpart (cv2 takes the
 * per-view arrays as they are, camera_calib.py:74-77). */
int camcal_layout(int64_t n_views, const int32_t* n_per_view, int32_t* n_tiles, int64_t* n_slots,
                  int64_t* total_floats, int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax,
                  int64_t* tile_off);

/* camcal_layout on the device, for sets whose CSR offsets and points are already there (no host
 * pass over the views, nothing to upload): the same four tables, bit for bit, written to DEVICE arrays
 * slot_view [n_slots], lane_n [n_slots], tile_nmax [n_tiles], tile_off [n_tiles] with n_tiles =
 * max(ceil(n_views / 64), 1), n_slots = 64 n_tiles (known to the caller without asking).  view_ptr
 * [n_views + 1] i64 (device), n_obs = the number of points the caller's arrays hold, obj_xyz
 * [n_obs][3] f32 (device; NULL: no planarity test).  head: 4 x i64 in DEVICE memory, written by the
 * call -- read it back (one copy, after the stream has passed the call) before anything is packed:
 *   head[0] = view_ptr[n_views] - view_ptr[0]     head[1] = total_floats (length of pts)
 *   head[2] = flags: CAMCAL_LAYOUT_NOT_PLANAR (some Z != 0), CAMCAL_LAYOUT_BAD_COUNT (a view with a
 *             negative or >= 2^31 count), CAMCAL_LAYOUT_BAD_TOTAL (the offsets do not start at 0 or do not
 *             end at n_obs) -- with either BAD flag set the tables must not be used
 *   head[3] = the largest count of a view
 * temp: camcal_layout_device_temp_len(n_views) bytes of device memory (sort keys + the sort's own
 * storage), free again once the stream has passed the call.  Everything is stream-ordered; the call
 * does not synchronise and allocates nothing.  No reference counterpart (as camcal_layout). */
#define CAMCAL_LAYOUT_NOT_PLANAR 1
#define CAMCAL_LAYOUT_BAD_COUNT 2
#define CAMCAL_LAYOUT_BAD_TOTAL 4
int64_t camcal_layout_device_temp_len(int64_t n_views);
int camcal_layout_device(int64_t n_views, const int64_t* view_ptr, int64_t n_obs, const float* obj_xyz,
                         int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax, int64_t* tile_off,
                         int64_t* head, void* temp, int64_t temp_bytes, void* stream);

/* The planarity test of camcal_layout_device on its own: flag = 4 x i64 in DEVICE memory (the layout's head:
 * flag[2] = CAMCAL_LAYOUT_NOT_PLANAR if some Z of obj_xyz [n_obs][3] f32 (device) is not 0, else 0; flag[0], [1],
 * [3] are cleared).  Stream-ordered, no allocation.  camcal_calibrate uses it for sets of a million points and
 * more, where the same test over the host array costs more than the upload it precedes. */
int camcal_points_off_plane(int64_t n_obs, const float* obj_xyz, int64_t* flag, void* stream);

/* Gather cv2-style AoS points into tiles.  obj_xyz [n_obs][3] f32, img_uv
 * [n_obs][2] f32, view_ptr [n_views+1] i64 (CSR), slot_view [n_slots] i32 =
 * view index placed in each slot or -1.  Writes tiles->pts (caller-allocated,
 * size = sum over tiles of 5*64*nmax floats).  Host-side counterpart of the
 * float32 lists the reference hands to cv2 (camera_calib.py:70-75). */
int camcal_pack_tiles(const float* obj_xyz, const float* img_uv, const int64_t* view_ptr,
                      const int32_t* slot_view, const camcal_tiles* tiles, float* pts_out,
                      void* stream);

/* Per-view pose initialisation for planar Z = 0 boards (cv::internal::
 * InitExtrinsics / cvFindExtrinsicCameraParams2 planar branch).  status[s] = 0
 * ok, >0 degenerate view. */
int camcal_pose_init(int model, const camcal_tiles* tiles, const double* intr, double* pose,
                     int32_t* status, void* stream);

/* Per-view Gauss-Newton pose refinement at fixed intrinsics
 * (cv::internal::ComputeExtrinsicRefine; also used in place of the 6-dof
 * CvLevMarq of cvFindExtrinsicCameraParams2).  iters[s] = steps taken. */
int camcal_pose_refine(int model, const camcal_tiles* tiles, const double* intr, double* pose,
                       int max_iter, double tol, int32_t* iters, void* stream);

/* Evaluation A -- cost, intrinsic normal block and pose blocks
 * (cvCalibrateCamera2Internal's JtJ(0:NI,0:NI), JtErr(0:NI), errNorm,
 * JtJ(ext,ext), JtErr(ext); fisheye ComputeJacobians' A A^T, A e, B B^T, B e).
 * with_normal = 0 computes the cost only (CvLevMarq CHECK_ERR pass).
 * glob receives camcal_glob_len doubles.  view_cost (optional) receives the
 * sum of squared residuals of every slot.  V, g (both or neither; need
 * with_normal = 1) receive the per-view 6x6 block and gradient in Rodrigues
 * coordinates. */
int camcal_intr_normal(int model, const camcal_tiles* tiles, const double* intr,
                       const double* pose, int with_normal, double* glob, double* view_cost,
                       double* V, double* g, double* scratch, void* stream);

/* Evaluation B -- cross blocks and their elimination: W = Ji^T Je per view
 * (JtJ(0:NI,ext) / A B^T), then C = sum W V'^-1 W^T, d = sum W V'^-1 g with the
 * V, g of evaluation A and V' = V, diag*(1+lam).  W_out (optional) keeps the
 * blocks for camcal_schur / camcal_backsub at other lam. */
int camcal_cross_schur(int model, const camcal_tiles* tiles, const double* intr,
                       const double* pose, const double* V, const double* g, double lam,
                       double* W_out, double* schur, double* scratch, void* stream);

/* Normal-equation evaluation, per-view part: V = Je^T Je, W = Ji^T Je,
 * g = Je^T e in Rodrigues coordinates (the B B^T, A B^T, B e blocks of
 * ComputeJacobians; the JtJ(ext,ext), JtJ(0:NI,ext), JtErr(ext) blocks of
 * cvCalibrateCamera2Internal). */
int camcal_view_blocks(int model, const camcal_tiles* tiles, const double* intr,
                       const double* pose, double* V, double* W, double* g, void* stream);

/* Schur complement of the (damped) pose blocks onto the intrinsics:
 * C = sum W V'^-1 W^T, d = sum W V'^-1 g with V' = V, diag*(1+lam).
 * schur receives camcal_schur_len doubles. */
int camcal_schur(int model, const camcal_tiles* tiles, const double* V, const double* W,
                 const double* g, double lam, double* schur, double* scratch, void* stream);

/* Reduced solve (U' - C) dK = b - d on the free intrinsics (mask[a] != 0),
 * U' = U with diag*(1+lam);  intr_out = intr_in + scale*dK.
 * (CvLevMarq::step on the eliminated system: scale = -1; fisheye::calibrate's
 * `finalParam + alpha_smooth2*G`: scale = -alpha_smooth2, lam = 0.)
 * stats[0] = ||scale*dK||^2, stats[1] = ||intr_in||^2,
 * stats[2] = ||d(fx fy cx cy)||^2, stats[3] = ||(fx fy cx cy)_out||^2,
 * stats[4] = 0 if the elimination met a non-positive pivot else 1. */
int camcal_solve_reduced(int model, const double* glob, const double* schur,
                         const uint8_t* mask_host, double lam, double scale,
                         const double* intr_in, double* intr_out, double* dK, double* stats,
                         void* stream);

/* Back-substitution d_i = V'^-1 (g_i - W_i^T dK); pose_out = pose_in - d_i.
 * norms[0] = sum ||d_i||^2, norms[1] = sum ||pose_in||^2 (the pose part of
 * cvNorm(param, prevParam, CV_RELATIVE_L2) in CvLevMarq::updateAlt). */
int camcal_backsub(int model, const camcal_tiles* tiles, const double* V, const double* W,
                   const double* g, double lam, const double* dK, const double* pose_in,
                   double* pose_out, double* norms, double* scratch, void* stream);

/* One full normal-equation evaluation (the unit the benchmark times) =
 * camcal_intr_normal(with V, g) + camcal_cross_schur: two passes over the
 * points, each holding half of the accumulators in registers.  V, g are
 * required (scratch for the hand-over), W may be NULL. */
int camcal_normal_eq(int model, const camcal_tiles* tiles, const double* intr,
                     const double* pose, double lam, double* V, double* W, double* g,
                     double* glob, double* schur, double* scratch, void* stream);

/* Same results from ONE pass over the points with every accumulator of a view
 * in one lane (what the fisheye driver and the benchmark use: the points are read
 * once).  Needs schur == glob + camcal_glob_len;
 * V, W, g all NULL or all set. */
int camcal_normal_eq_fused(int model, const camcal_tiles* tiles, const double* intr,
                           const double* pose, double lam, double* V, double* W, double* g,
                           double* glob, double* schur, double* scratch, void* stream);

/* camcal_normal_eq_fused (no block stores, lam = 0) that also leaves every view's first-order
 * pose response to an intrinsic step in Y [NI * 6][n_slots], component-major like W:
 *   Y[(a * 6 + c) * n_slots + s] = (V^-1 W_a^T)_c        (Rodrigues coordinates)
 * i.e. the back-substitution row d = V^-1 (g - W^T dK) of the joint normal equations at a pose
 * that is its view's optimum (g = 0), kept so that camcal_pose_predict can apply it once the
 * reduced solve has produced dK.  No OpenCV
 * counterpart as a stage: cv::fisheye::calibrate restarts every pose from InitExtrinsics
 * after its intrinsic step (camera_calib.py:74-77 always passes CALIB_RECOMPUTE_EXTRINSIC);
 * this library warm-starts the same ComputeExtrinsicRefine, and the response moves that warm
 * start to first order in the step.  The refinement's stopping rule is untouched, so the poses
 * it returns are the same per-view optima to its tolerance. */
int camcal_normal_eq_fused_response(int model, const camcal_tiles* tiles, const double* intr,
                                    const double* pose, double* Y, double* glob, double* schur,
                                    double* scratch, void* stream);

/* pose <- pose + V^-1 W^T s for the intrinsic step intr_out = intr_in + scale * dK that
 * camcal_solve_reduced took (s = -scale * dK), from the Y of camcal_normal_eq_fused_response
 * at intr_in.  iters (optional): the step counts of the last camcal_pose_refine; a view that
 * took none (cv2's cond(J) guard, a failed factorisation) or ran into the cap of `cap` steps
 * (it is not at its optimum) keeps its pose and is refined from there as before. */
int camcal_pose_predict(int model, const camcal_tiles* tiles, const double* Y, const double* dK,
                        double scale, const int32_t* iters, int cap, double* pose, void* stream);

/* camcal_pose_predict followed by camcal_pose_refine in ONE launch (the refinement kernels apply
 * the move to the pose they have just loaded): same arithmetic, same results bit for bit, one
 * kernel and one pass over the poses less per iteration of the fisheye loop.  iters: step
 * counts of the last refinement in (the skip rule of camcal_pose_predict), of this one out. */
int camcal_pose_predict_refine(int model, const camcal_tiles* tiles, const double* intr, const double* Y,
                               const double* dK, double scale, int cap, double* pose, int max_iter,
                               double tol, int32_t* iters, void* stream);

/* The evaluation of camcal_normal_eq_fused (no block stores: V = W = g = NULL) for callers
 * that cannot allocate device memory or carry a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.
 * The partial records of the launch go to a file-scope __device__ array inside the library
 * (109 x 16384 + 192 doubles = 14.3 MB, held by every GPU the library runs on until the
 * process ends) instead of a caller-provided scratch; the call allocates and frees nothing.
 * Limit: n_tiles (x 64 or x 16 for the small planar sets that run several lanes per view,
 * camcal_tiles.lanes_per_view) <= 16384 partial records, i.e. up to 1,048,576 views one lane per
 * view; larger sets return CAMCAL_E_TOOBIG.  glob / schur as in camcal_normal_eq_fused
 * (schur == glob + camcal_glob_len).  Nothing in the array has to start at zero (a launch
 * writes every entry it reads), but every call on one GPU shares the array: calls that
 * overlap on one GPU (two streams, two threads) race.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress); make it outside stream capture. */
int camcal_normal_eq_fused_static(int model, const camcal_tiles* tiles, const double* intr,
                                  const double* pose, double lam, double* glob, double* schur,
                                  void* stream);

/* ---- Small view sets (several lanes per view and at most 1600 wavefronts per launch: <= 1600 views at 64 lanes
 * per view, 2049 ... 6400 at 16 -- the reference's own calls hold 20 ... 500), fisheye loop: two stages that fold
 * the five launches of an iteration into two.  A
 * production-sized iteration is a chain of ~5 ... 60 us kernels whose length is set by the one or two views
 * that need 18 - 20 Gauss-Newton steps; every launch boundary and every separate small kernel on that
 * chain is wall clock.  Larger sets are accepted too: there the re-estimation and the evaluation remain
 * launches of their own (three launches per iteration instead of five and a copy).  CAMCAL_E_UNSUPPORTED for
 * the pinhole model.  NO COUNTERPART IN THE REFERENCE (cv2 runs the loop inside cv::fisheye::calibrate,
 * camera_calib.py:74-77).
 *
 * camcal_refine_normal_eq_partial: camcal_pose_predict_refine (predict != 0; needs Y, dK) or
 * camcal_pose_refine (predict == 0), skipped altogether with max_iter == 0, followed IN THE SAME LAUNCH
 * by the evaluation of camcal_normal_eq_fused_response (Y != NULL: the new response replaces the one
 * just used) or camcal_normal_eq_fused (Y == NULL, lam = 0, no stored blocks) at the refined poses --
 * a view's record is complete while slower views still iterate.  The evaluation is left as partial
 * records in scratch for camcal_reduce_solve.  first != 0: first launch of a solve on this scratch
 * (clears camcal_reduce_solve's counter).  Same device code as the separate stages: same bits. */
int camcal_refine_normal_eq_partial(int model, const camcal_tiles* tiles, const double* intr, double* pose,
                                    int max_iter, double tol, int32_t* iters, double* Y, int predict,
                                    const double* dK, double scale, int cap, int first, double* scratch,
                                    void* stream);
/* camcal_reduce_solve: the second stage of that evaluation (fixed-order sums, rec = [U|b|cost|C|d] as
 * camcal_normal_eq_fused leaves glob | schur) and, with do_solve, camcal_solve_reduced on it, in one
 * launch: the workgroup that finishes its row last (a counter in scratch) expands and solves.
 * notify (or NULL): the device-visible address of 16 doubles of pinned, coherent host memory
 * (hipHostMalloc + hipHostGetDevicePointer).  The launch itself writes [stats (5) | cost | ... | seq at
 * CAMCAL_NOTIFY_SEQ] there, seq last, so the host can act on the scalars as soon as they exist -- no copy on the stream
 * between this launch and the next, which is what a production-sized iteration would otherwise wait for. */
int camcal_reduce_solve(int model, const camcal_tiles* tiles, double* scratch, double* rec, int do_solve,
                        const uint8_t* mask_host, double lam, double scale, const double* intr_in,
                        double* intr_out, double* dK, double* stats, double* notify, double seq, void* stream);

/* The same folding for the pinhole LM step (CvLevMarq on the eliminated system, camera_calib.py:153-156) of a
 * set in a several-lanes-per-view mapping (planar, <= 28 672 views): 8 launches + 4 copies per step become 4
 * launches.  NO COUNTERPART IN THE REFERENCE.
 * CAMCAL_E_UNSUPPORTED for larger sets and for the fisheye model (camcal_pinhole_lm then runs the stages).
 *   camcal_small_set_begin        once per solve on a scratch, before the calls below (clears their counter)
 *   camcal_cross_schur_solve      camcal_cross_schur at (intr, pose) + camcal_solve_reduced(glob, schur, lam, scale,
 *                                 intr -> intr_out); two launches (the second sums the partial records and solves)
 *   camcal_schur_solve            camcal_schur (stored blocks at another lambda) + camcal_solve_reduced likewise
 *   camcal_backsub_intr_normal    camcal_backsub (pose_in -> pose_out) + camcal_intr_normal with blocks at
 *                                 (intr_trial, pose_out) -> glob_trial, V_trial, g_trial; norms, and with notify the
 *                                 decision's scalars [stats (5, as left by the solve) | norms (2) | cost | ... | seq]
 *                                 straight to the host; two launches
 * Results equal the separate stages' to the order of the sums (1e-13 relative). */
int camcal_small_set_begin(int model, const camcal_tiles* tiles, double* scratch, void* stream);
int camcal_cross_schur_solve(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                             const double* V, const double* g, double lam, double* W_out, const double* glob,
                             double* schur, const uint8_t* mask_host, double scale, double* intr_out, double* dK,
                             double* stats, double* scratch, void* stream);
int camcal_schur_solve(int model, const camcal_tiles* tiles, const double* V, const double* W, const double* g,
                       double lam, const double* glob, double* schur, const uint8_t* mask_host, double scale,
                       const double* intr_in, double* intr_out, double* dK, double* stats, double* scratch,
                       void* stream);
int camcal_backsub_intr_normal(int model, const camcal_tiles* tiles, const double* V, const double* W,
                               const double* g, double lam, const double* dK, const double* pose_in,
                               double* pose_out, const double* intr_trial, double* glob_trial, double* V_trial,
                               double* g_trial, double* norms, const double* stats, double* notify, double seq,
                               double* scratch, void* stream);

/* ---- Multi-GPU: views shard across ranks (one process per GPU); the only data exchanged are
 * the small global records, summed over the ranks between the stages of a solve.  The
 * reference has no counterpart (single process).  A camcal_comm tells the solver loops how
 * to sum: allreduce_sum(ctx, buf, n, stream) adds the n doubles at buf (device memory) over
 * all ranks in place, ordered on `stream` (it must not block the host on other ranks'
 * progress beyond what the collective itself needs), and returns 0 on success. */
/* Slot of the sequence number in a notify buffer (camcal_reduce_solve, camcal_backsub_intr_normal): 16 doubles
 * of pinned, coherent host memory; the launch writes its scalars to the first slots and the number last. */
#define CAMCAL_NOTIFY_SEQ 15

typedef struct camcal_comm {
  int (*allreduce_sum)(void* ctx, double* buf, int64_t n, void* stream);
  void* ctx;
} camcal_comm;

/* RCCL-backed communicator, created by the library itself (librccl.so.1 is opened on first
 * use; a single-GPU process never loads it): rank 0 obtains 128 bytes with
 * camcal_rccl_unique_id and hands them to every rank by any out-of-band means
 * (torch.distributed broadcast, MPI, a file); every rank then calls camcal_comm_rccl_create
 * with its device current.  allreduce_sum of such a communicator is ONE ncclAllReduce
 * (ncclDouble, ncclSum, in place) enqueued on the caller's stream: no host synchronisation,
 * no staging copy. */
int camcal_rccl_unique_id(void* id128_host);
int camcal_comm_rccl_create(const void* id128_host, int rank, int world_size, camcal_comm** comm_out);
int camcal_comm_destroy(camcal_comm* comm);
/* Ranks the communicator sums over: ncclCommCount for a communicator made by
 * camcal_comm_rccl_create, 1 for comm == NULL, CAMCAL_E_BADARG for a caller-provided callback
 * communicator (the library cannot know).  For benchmarks and logs that want to state how many
 * ranks a collective really spanned. */
int camcal_comm_size(const camcal_comm* comm);
/* comm->allreduce_sum(...) for hosts without function-pointer calls (ctypes); comm == NULL is
 * the identity (one rank) */
int camcal_comm_allreduce(const camcal_comm* comm, double* buf, int64_t n, void* stream);

/* ---- Host-side sequencing of whole solves: the loops that run inside OpenCV's C++ behind
 * the two reference calls.  They only call the stage entry points above and read a few
 * scalars back per iteration (waiting on an event, not on the stream).  comm == NULL: one
 * rank.  With a communicator every rank calls the loop with its own shard of the views and
 * the same starting intrinsics; the records are summed over the ranks inside the loop
 * (fisheye: one all-reduce of [U|b|cost|C|d] per iteration; pinhole: [C|d] before the
 * reduced solve and [U|b|cost | step norms] of the trial point after it), every rank solves
 * the reduced system redundantly and keeps its own poses.
 * _host pointers are host memory, everything else is device memory. */

/* cv::fisheye::calibrate's loop (camera_calib.py:74-77): until the criteria stop it
 * (crit_type 1 = COUNT, 2 = EPS, 3 = both; `change` = relative step of fx fy cx cy):
 * single-pass evaluation, reduced solve with the step damped by 1 - 0.6^(it+1), and --
 * with recompute_extrinsic (CALIB_RECOMPUTE_EXTRINSIC) -- re-estimation of every pose.
 * intr, pose: starting point in, result out.  rec: camcal_glob_len + camcal_schur_len
 * doubles, dK 16, stats 8, iters [n_slots] (may be NULL), scratch camcal_scratch_len.
 * Y: 48 * n_slots doubles of workspace (camcal_normal_eq_fused_response) or NULL; with it
 * every re-estimation starts from the first-order predicted pose (camcal_pose_predict)
 * instead of the previous one -- same results to the refinement's tolerance, about a third
 * fewer Gauss-Newton steps per view.
 * cost_out_host = sum of squared residuals at the result.
 * Single-rank calls run each iteration as camcal_reduce_solve + camcal_refine_normal_eq_partial (two
 * launches instead of five on small sets, three on large ones; the cost at the result comes from the
 * last evaluation instead of a pass of its own, so it can differ from the staged loop's in the last
 * bits of the sum); rec then holds the evaluation AT the result. */
int camcal_fisheye_gn(const camcal_tiles* tiles, double* intr, double* pose, const uint8_t* mask_host,
                      int crit_type, int max_count, double eps, int recompute_extrinsic, double* rec,
                      double* dK, double* stats, int32_t* iters, double* Y, double* scratch,
                      const camcal_comm* comm, void* stream, double* cost_out_host,
                      int32_t* iterations_out_host);

/* CvLevMarq::updateAlt as cvCalibrateCamera2Internal drives it (camera_calib.py:153-156) on
 * the eliminated system: lambda starts at 1e-3, x10 on a rejected step (only the stored-
 * block Schur kernel is repeated), /10 on an accepted one; stops after max_iter accepted
 * steps or when the relative parameter change < eps.  intr2 / pose2 / V2 / g2: the two
 * buffers of each double-buffered quantity (HOST arrays of two device pointers; [0] holds
 * the starting point), W NI*6*n_slots, rec as above, glob_trial camcal_glob_len, norms 2
 * (with a communicator, norms == glob_trial + camcal_glob_len makes the trial's records one
 * all-reduce instead of two).  current_out_host = index (0 / 1) of the buffers that hold the
 * result. */
int camcal_pinhole_lm(const camcal_tiles* tiles, double* const intr2[2], double* const pose2[2],
                      double* const V2[2], double* const g2[2], double* W, double* rec,
                      double* glob_trial, double* dK, double* stats, double* norms, double* scratch,
                      const uint8_t* mask_host, int max_iter, double eps, const camcal_comm* comm,
                      void* stream, double* err_out_host, int32_t* iterations_out_host,
                      int32_t* current_out_host);

/* ---- One call per calibration: host arrays in, host results out, every device byte in a
 * workspace the caller owns.  These are the two reference calls themselves,
 *     cv2.calibrateCamera(obj, img, size, K, D, flags=..., criteria=...)    camera_calib.py:153-156,
 *                                                                            intrinsicCalib.py:194-197,223-226
 *     cv2.fisheye.calibrate(obj, img, size, K, D, flags=..., criteria=...)  camera_calib.py:74-77,
 *                                                                            intrinsicCalib.py:102-105,131-134
 * as a C ABI: argument checks, view layout, upload, packing, per-view pose initialisation and
 * refinement, the solver loop, results in the caller's view order.  `flags` are cv2's flag
 * bits of the model's call (calib3d.hpp: CALIB_USE_INTRINSIC_GUESS = 1 ...; fisheye::CALIB_* for
 * the fisheye model), crit_type / max_iter / eps cv2's TermCriteria (COUNT = 1, EPS = 2).
 * Supported flag sets as in the Python drop-in; others return
 * CAMCAL_E_UNSUPPORTED, malformed input CAMCAL_E_BADARG, a degenerate view or a singular
 * system CAMCAL_E_NUMERIC (view_status then names the views: > 0 = pose initialisation failed).
 * All pointers of a problem are HOST memory except `workspace` (device, 256-byte aligned,
 * at least camcal_workspace_len bytes; nothing in it has to be initialised and nothing is
 * allocated or freed by the call). */
typedef struct camcal_problem {
  int32_t model;              /* CAMCAL_PINHOLE / CAMCAL_FISHEYE                                  */
  int32_t flags;              /* cv2 flag bits of that model's call                               */
  int32_t width, height;      /* imageSize                                                        */
  int32_t crit_type, max_iter;/* cv2 TermCriteria                                                 */
  double  eps;
  int64_t n_views;
  const int64_t* view_ptr;    /* [n_views + 1] CSR offsets of the views in obj / img              */
  const float* obj;           /* [n_obs][3] board points (Z = 0), the reference's float32 lists   */
  const float* img;           /* [n_obs][2] detected corners                                      */
  double* K;                  /* [9] row-major 3x3: guess in, result out                          */
  double* D;                  /* pinhole: 5 doubles (n_dist = 4 or 5 valid on entry, k1 k2 p1 p2 k3
                                 out); fisheye: 4 (k1..k4): guess in, result out                  */
  int32_t n_dist;
  int32_t lanes_per_view;     /* camcal_tiles.lanes_per_view: 0 = chosen by the size of the set   */
  double* rt;                 /* [n_views][6] out: (rvec, tvec) of every view                     */
  int32_t* view_status;       /* [n_views] out, may be NULL                                       */
  void* workspace;            /* DEVICE                                                           */
  int64_t workspace_bytes;
  double  rms;                /* out: sqrt(sum of squared residuals / observations)               */
  int32_t iterations;         /* out                                                              */
  int32_t status;             /* out (camcal_calibrate_many): 0 or this problem's error           */
} camcal_problem;

/* Bytes of device workspace a problem of these views needs (HOST view_ptr; no device work). */
int64_t camcal_workspace_len(int model, int64_t n_views, const int64_t* view_ptr_host);

/* One calibration.  With a communicator every rank calls it with its own shard of the views
 * and the same guess: the records are summed inside the loop, rms is over all ranks, every
 * rank receives the same K, D and the poses of its own views; a shard that is rejected (bad
 * arguments, workspace too small) or a degenerate view on any rank fails the call on every rank
 * (its own code where it happened, CAMCAL_E_PEER elsewhere) -- provided every rank passes at
 * least 256 bytes of aligned device workspace to exchange that verdict through.
 * Large sets (single rank): from 2^20 points on the planarity of the target is tested on the device behind the
 * upload, so CAMCAL_E_UNSUPPORTED for a non-planar target is returned at the end of the call (nothing is handed
 * out) rather than before any device work; host arrays of 32 MB and more are uploaded through page-locked
 * staging filled by a few threads of the library's own. */
int camcal_calibrate(int model, int64_t n_views, const int64_t* view_ptr_host, const float* obj_host,
                     const float* img_host, int width, int height, double* K_host, double* D_host,
                     int n_dist, double* rt_host, int32_t* view_status_host, int flags, int crit_type,
                     int max_iter, double eps, void* workspace, int64_t workspace_bytes,
                     const camcal_comm* comm, void* stream, double* rms_out_host,
                     int32_t* iterations_out_host);

/* n independent calibrations in one call (the reference solves its four cameras L, ML, MR, R
 * one after another, controller/controller_internal_calibration.py:375-399): each problem runs
 * on a stream of its own and the steps of their loops are interleaved -- every camera's next
 * iteration is enqueued before the host waits for any of them -- so the small launches of a
 * production-sized solve (25 views x 57 corners) overlap instead of queueing behind each
 * other's host round trips.  Same kernels in the same order per problem: every result equals
 * the one camcal_calibrate gives for that problem alone, bit for bit.  Each problem needs its
 * own workspace.  Returns 0 or the first failing problem's status (problems[i].status holds
 * each one's; the others are still solved). */
int camcal_calibrate_many(camcal_problem* problems, int n, void* stream);

/* Timing helper for benchmarks: average duration in milliseconds of `reps`
 * back-to-back launches of one stage (0 = intr_normal, 1 = view_blocks,
 * 2 = schur, 3 = camcal_normal_eq, 4 = camcal_normal_eq_fused without block stores, 5 = cross_schur),
 * measured with HIP events on `stream`. */
int camcal_time_stage(int stage, int reps, int model, const camcal_tiles* tiles,
                      const double* intr, const double* pose, double lam, double* V, double* W,
                      double* g, double* glob, double* schur, double* scratch, void* stream,
                      float* ms_out_host);

/* ---- Multi-camera rig (BASELINE.json config 4; no reference counterpart: the tool
 * calibrates its cameras one after another, controller_internal_calibration.py:375-399).
 * Residual proj_c(R_c (R_p X + t_p) + t_c) - obs, camera 0 = rig frame.  Unknowns ("globals",
 * n = n_cams * (NI + 6) <= 63, camera-major: intrinsics then the camera-from-rig (r_c, t_c)) and
 * one 6-dof pose per board placement.  Each camera is an ordinary packed view set; a view's
 * pose is the composed transform.  Everything below is reachable from this header alone:
 * fill a camcal_rig with device buffers, camcal_rig_init, camcal_rig_lm.
 *
 * Parameter and block buffers come in pairs (index b = 0 / 1: current / trial point).
 * rec[b] (camcal_rig_rec_len doubles) =
 *   [ per camera: glob (camcal_glob_len: U|b|cost) | U_ie (NI*6) | U_ee (21) | b_e (6) ]
 *   [ cost (sum over the cameras) | norms (2: ||d poses||^2, ||poses||^2) ]      <- summed over ranks with the records
 *   [ gstats (2: ||dG||^2, ||globals||^2) ]                                     <- identical on every rank
 *   [ C (n x n) | d (n) ]                                                        <- summed over ranks on their own */
#define CAMCAL_RIG_MAX_CAMS 4

typedef struct camcal_rig_cam {
  camcal_tiles   tiles;      /* this camera's views, packed as for a single camera          */
  const int32_t* slot_pose;  /* [n_slots] board pose seen in each slot, -1 = padding        */
  int32_t*       pose_slot;  /* [P] workspace of camcal_rig_init: slot of a pose, -1 unseen */
  double*        pose;       /* [6][n_slots] workspace of camcal_rig_init (per-view poses)  */
  double*        V;          /* [21][n_slots]   workspace: blocks of a view in the local    */
  double*        W;          /* [NI*6][n_slots]   tangent frame of its composed pose        */
  double*        g;          /* [6][n_slots]                                                */
} camcal_rig_cam;

typedef struct camcal_rig {
  int32_t model, n_cams;
  int64_t P;                 /* board poses held by this process                            */
  camcal_rig_cam cam[CAMCAL_RIG_MAX_CAMS];
  double* intr[2];           /* [n_cams][16]                                                */
  double* ext[2];            /* [n_cams][6]  (r_c, t_c); camera 0 stays zero                */
  double* rig_pose[2];       /* [6][P]                                                      */
  double* Vp[2];             /* [n_cams][21][P]        pose blocks per camera               */
  double* gp[2];             /* [n_cams][6][P]                                              */
  double* Wp[2];             /* [n_cams][(NI+6)*6][P]  globals x pose                       */
  double* rec[2];            /* camcal_rig_rec_len                                          */
  double* Yt;                /* [n*6][P]  U^-T Wp^T, pose index fastest                     */
  double* z;                 /* [6][P]    U^-T g                                            */
  double* dG;                /* [64]                                                        */
  double* scratch;           /* camcal_rig_scratch_len doubles                              */
  int32_t* status;           /* [sum of n_slots] pose-initialisation status (may be NULL)   */
} camcal_rig;

int64_t camcal_rig_rec_len(int model, int n_cams);
/* n_tiles_host: HOST array [n_cams] of the cameras' tile counts.  The length covers the work
 * area the stages share and, behind it, 27 doubles per board pose that camcal_rig_eliminate
 * leaves for camcal_rig_backsub (the pose factors). */
int64_t camcal_rig_scratch_len(int model, int n_cams, const int32_t* n_tiles_host, int64_t P);

/* Blocks of the whole rig at parameter buffer b: per camera the single-pass evaluation of
 * its views at the composed poses (local tangent frame), then the point-independent 6x6
 * chain rule to the rig parameters: rec[b] records + cost, Vp/gp/Wp[b].  The caller zeroes
 * Vp, gp, Wp ONCE (which poses a camera sees never changes; unseen entries stay zero). */
int camcal_rig_evaluate(const camcal_rig* rig, int b, void* stream);
/* The two halves of camcal_rig_evaluate, for a caller that wants the cost early (the LM
 * loop reads it back while the chain rule runs): residuals = per-camera evaluation ->
 * V/W/g of every camera, the [U|b|cost] records and the cost of rec[b]; chain = chain rule
 * -> Vp/gp/Wp[b] and the extrinsic records of rec[b].  evaluate = residuals, then chain. */
int camcal_rig_residuals(const camcal_rig* rig, int b, void* stream);
int camcal_rig_chain(const camcal_rig* rig, int b, void* stream);
/* camcal_rig_residuals that also hands the LM loop's five scalars of rec[b] -- [cost | pose-step norms (2) |
 * intrinsic-step norms (2)], the last four as the solve and the back-substitution of this trial point left them
 * -- to the host without a copy on the stream: notify = device-visible address of 16 pinned, coherent doubles
 * (see camcal_reduce_solve), sequence number written last at CAMCAL_NOTIFY_SEQ. */
int camcal_rig_residuals_notify(const camcal_rig* rig, int b, double* notify, double seq, void* stream);

/* Pose elimination at damping lam: per board pose V = sum_c Vp_c (diag*(1+lam)) = U^T U,
 * Yt = U^-T Wp^T, z = U^-T g; then C = Y^T Y and d = Y^T z on the matrix cores
 * (v_mfma_f64_16x16x4: the one dense fp64 contraction of this path, (6P x n)^T (6P x n)),
 * summed in a fixed order -> [C | d] of rec[b]. */
int camcal_rig_eliminate(const camcal_rig* rig, int b, double lam, void* stream);

/* Reduced solve over the globals from rec[b]: diag damped by (1+lam), parameters with
 * mask_host[i] == 0 held, Jacobi-scaled Cholesky on one wavefront; intr/ext[out] =
 * intr/ext[b] - dG; gstats of rec[out]; NaN step if the matrix is not positive. */
int camcal_rig_solve(const camcal_rig* rig, int b, const uint8_t* mask_host, double lam, int out, void* stream);

/* d_p = V'^-1 (g_p - Wp^T dG) = U^-1 (z_p - Y_p dG), rig_pose[out] = rig_pose[b] - d_p; norms of
 * rec[out].  Uses Yt, z and the pose factors that camcal_rig_eliminate(rig, b, lam) left in
 * rig->scratch: call it after that eliminate (and the solve), with nothing but
 * camcal_rig_solve in between. */
int camcal_rig_backsub(const camcal_rig* rig, int b, double lam, int out, void* stream);

/* Starting point into buffer 0 from intr[0] (the guesses): per-camera pose initialisation
 * and refinement (the single-camera kernels), camera-from-rig transforms from the chordal
 * mean of T_pc o T_pk^-1 over shared poses (spanning tree from camera 0 by shared-pose
 * count; the pair sums are summed over the ranks), board poses from the lowest-numbered
 * camera that sees them.  CAMCAL_E_DISJOINT if a camera shares no pose with the rest.  With a
 * communicator the failure of any rank's per-camera stage is summed with the pair sums, so
 * every rank returns an error (the failing rank its own code, the others CAMCAL_E_PEER) and
 * none is left waiting in a collective. */
int camcal_rig_init(const camcal_rig* rig, const camcal_comm* comm, void* stream);

/* CvLevMarq::updateAlt over the rig (the control flow of camcal_pinhole_lm) from buffer 0;
 * current_out_host = buffer that holds the result.  err_out_host = sum of squared
 * residuals over all ranks. */
int camcal_rig_lm(const camcal_rig* rig, const uint8_t* mask_host, int max_iter, double eps,
                  const camcal_comm* comm, void* stream, double* err_out_host,
                  int32_t* iterations_out_host, int32_t* current_out_host);

/* ---- Point-level operations either side of the solver ---- */

/* cv2.projectPoints / cv2.fisheye.projectPoints for one pose (camera_calib.py:97,175;
 * intrinsicCalib.py:143,235): xyz [n][3] -> uv [n][2], all double, all on the device;
 * pose6 = (rvec, tvec). */
int camcal_project_points(int model, int64_t n, const double* xyz, const double* pose6,
                          const double* intr, double* uv, void* stream);

/* The tool's acceptance number for every packed view at once: err[slot] =
 * ||proj - corners||_2 / n  (camera_calib.py:96-104,174-181; intrinsicCalib.py:137-146,
 * 229-238); sum_out = [sum of err over views, number of views] so that the caller's
 * np.mean is sum_out[0] / sum_out[1].  f32_round != 0 rounds the projection to float32
 * and subtracts in float32 as cv2 does for float32 corner arrays. */
int camcal_reproj_err(int model, const camcal_tiles* tiles, const double* intr, const double* pose,
                      int f32_round, double* err, double* sum_out, double* scratch, void* stream);

/* cv2.undistortPoints / cv2.fisheye.undistortPoints (camera_calib.py:241,252,261,281,292,
 * 301; controller_internal_calibration.py:553,570): uv [n][2] pixels -> out [n][2];
 * M = P[:, :3] * R row-major on the device, or NULL for normalised coordinates. */
int camcal_undistort_points(int model, int64_t n, const double* uv, const double* intr,
                            const double* M, double* out, void* stream);

/* cv2.initUndistortRectifyMap / cv2.fisheye.initUndistortRectifyMap
 * (video_server.py:473,497; intrinsicCalib.py:159,251): iR = (newK * R)^-1 row-major on
 * the device; map_type 0: map1, map2 float32 [h][w] (CV_32FC1), 1: map1 int16 [h][w][2],
 * map2 uint16 [h][w] (CV_16SC2 + CV_16UC1 fixed point, INTER_BITS 5). */
int camcal_undistort_rectify_map(int model, int width, int height, const double* intr,
                                 const double* iR, int map_type, void* map1, void* map2,
                                 void* stream);

/* cv2.remap(src, map1, map2, INTER_LINEAR) with BORDER_CONSTANT for 8-bit images of 1..4
 * channels (video_server.py:504; intrinsicCalib.py:369; ex_Calib.py:204,210): maps as
 * written by camcal_undistort_rectify_map (map_type 0 float32 pair, 1 fixed point);
 * strides in bytes; arithmetic is cv2's 1/32-pixel fixed point with 15-bit weights. */
int camcal_remap_u8(const unsigned char* src, int src_w, int src_h, int64_t src_stride, int channels,
                    const void* map1, const void* map2, int map_type, int dst_w, int dst_h,
                    unsigned char* dst, int64_t dst_stride, int border_value, void* stream);

/* Peak probes for the roofline report: kind 0 = dependent-free FP64 FMA stream
 * (result TFLOP/s), kind 1 = streaming read of `bytes` from buf with 16-byte loads
 * (result GB/s), kind 2 = the same with 4-byte loads (PMC calibration of the point-load shape).
 * buf: device scratch of at least `bytes` (>= 4096).  No reference counterpart. */
int camcal_peak_probe(int kind, void* buf, int64_t bytes, int reps, void* stream,
                      double* result_host);

#ifdef __cplusplus
}
#endif
#endif /* CAMCAL_H */
