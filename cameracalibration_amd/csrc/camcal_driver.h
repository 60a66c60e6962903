// Host-side sequencing of the two solver loops over the stage entry points of
// include/camcal.h -- the loops that run inside OpenCV's C++ for the reference:
//   cv::fisheye::calibrate's damped Gauss-Newton   (camcal_fisheye_gn)
//   CvLevMarq::updateAlt inside cvCalibrateCamera2 (camcal_pinhole_lm)
// One process per device; with a camcal_comm the global records are summed over
// the ranks between the stages (include/camcal.h), on the same stream, without a
// host round trip of their own.
//
// Each loop is a small state machine with two halves per step: launch() enqueues
// the next piece of work on the solve's stream up to (and including) the copy of
// the few scalars the host decides on, finish() waits for that copy and takes the
// decision.  A single solve alternates the two; camcal_calibrate_many
// (camcal_calibrate.h) launches the step of every camera before it waits for any of
// them, so that independent solves overlap on the device -- same kernels, same
// order per solve, hence the same bits as solves made one after another.
//
// Included by camcal_kernels.hip and by the tests' CPU lane emulator; the
// includer provides, with internal linkage,
//   struct camcal_reader {            // one per running solve (RAII)
//     int begin(const double* src, int n, int offset, void* stream);
//     int end(double* dst_host, int n_total);
//     double* notify(double* seq_out);  // where a kernel may write up to 7 doubles + a sequence number itself
//     int wait(double seq, void* stream, double* dst_host, int n);     (NULL: not on this build / device)
//   };
//   int camcal_copy_doubles(double* dst, const double* src, int n, void* stream)
// (begin: asynchronous copy of n doubles to position `offset` of the reader's landing
// buffer, ordered on the stream; end: wait for the LAST begin and hand out the first
// n_total doubles -- work enqueued after the begin keeps running meanwhile;
// copy_doubles: device -> device on the stream).
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/camcal.h"

#define CAMCAL_DRV_TRY(x)      \
  do {                         \
    const int rc_ = (x);       \
    if (rc_ != 0) return rc_;  \
  } while (0)

namespace {          // internal linkage, like the includer's camcal_reader
namespace camcal_drv {

// cv::fisheye::calibrate's loop (camera_calib.py:74-77).
struct FisheyeGN {
  // arguments of camcal_fisheye_gn
  const camcal_tiles* tiles = nullptr;
  double *intr = nullptr, *pose = nullptr;
  const uint8_t* mask_host = nullptr;
  int crit_type = 3, max_count = 100;
  double eps = 0.0;
  int recompute_extrinsic = 1;
  double *rec = nullptr, *dK = nullptr, *stats = nullptr;
  int32_t* iters = nullptr;
  double *Y = nullptr, *scratch = nullptr;
  const camcal_comm* comm = nullptr;
  void* stream = nullptr;
  // state
  camcal_reader rd;
  double change = 1.0, cost = 0.0;
  int it = 0;
  bool refine_first = false;            // the caller's poses come straight from the initialisation: refine them first
  int fused = -1;                       // one rank: two / three launches per iteration (-1: not tried yet, 0: no)
  bool have_partials = false;           // an evaluation at the current point waits in scratch
  double seq = 0.0;                     // > 0: the scalars of this step come through the reader's notify buffer
  enum { READY, ITERATION, FINAL_COST, DONE } pending = READY;

  bool args_ok() const {
    return tiles && intr && pose && mask_host && rec && dK && stats && scratch && crit_type >= 1 && crit_type <= 3;
  }
  bool done() const { return pending == DONE; }

  int launch() {
    if (pending != READY) return CAMCAL_E_BADARG;
    const int ng = camcal_glob_len(CAMCAL_FISHEYE), ns = camcal_schur_len(CAMCAL_FISHEYE);
    const bool stop = (crit_type == 1 && it >= max_count) || (crit_type == 2 && change <= eps) ||
                      (crit_type == 3 && (change <= eps || it >= max_count));
    if (fused != 0 && !comm) {
      // One rank: [sum + solve] is one launch (camcal_reduce_solve) and [re-estimation + the NEXT evaluation]
      // one (small sets) or two (camcal_refine_normal_eq_partial); the scalars come through the notify buffer.
      // The evaluation at the point the loop stops at has then been made already: its cost is the result's,
      // no pass of its own.
      const bool respond = Y && recompute_extrinsic;
      if (!have_partials) {              // (with refine_first: the first refinement rides along)
        const int rc = camcal_refine_normal_eq_partial(CAMCAL_FISHEYE, tiles, intr, pose, refine_first ? 20 : 0, 1e-10, iters,
                                                       respond ? Y : nullptr, 0, dK, 0.0, 20, 1, scratch, stream);
        if (rc == CAMCAL_E_UNSUPPORTED && fused < 0) fused = 0;
        else {
          CAMCAL_DRV_TRY(rc);
          fused = 1;
          have_partials = true;
          refine_first = false;
        }
      }
      if (fused == 1) {
        seq = 0.0;
        double* note = rd.notify(&seq);
        if (!note) seq = 0.0;
        if (stop) {
          CAMCAL_DRV_TRY(camcal_reduce_solve(CAMCAL_FISHEYE, tiles, scratch, rec, 0, nullptr, 0.0, 0.0, nullptr, nullptr,
                                             nullptr, nullptr, note, seq, stream));
          if (!note) CAMCAL_DRV_TRY(rd.begin(rec + ng - 1, 1, 0, stream));
          pending = FINAL_COST;
          return 0;
        }
        const double alpha2 = 1.0 - std::pow(0.6, it + 1.0);
        CAMCAL_DRV_TRY(camcal_reduce_solve(CAMCAL_FISHEYE, tiles, scratch, rec, 1, mask_host, 0.0, -alpha2, intr, intr, dK,
                                           stats, note, seq, stream));
        if (!note) CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
        CAMCAL_DRV_TRY(camcal_refine_normal_eq_partial(CAMCAL_FISHEYE, tiles, intr, pose, recompute_extrinsic ? 20 : 0, 1e-10,
                                                       iters, respond ? Y : nullptr, respond ? 1 : 0, dK, -alpha2, 20, 0,
                                                       scratch, stream));
        pending = ITERATION;
        return 0;
      }
    }
    if (refine_first) {
      CAMCAL_DRV_TRY(camcal_pose_refine(CAMCAL_FISHEYE, tiles, intr, pose, 20, 1e-10, iters, stream));
      refine_first = false;
    }
    if (stop) {
      CAMCAL_DRV_TRY(camcal_intr_normal(CAMCAL_FISHEYE, tiles, intr, pose, 0, rec, nullptr, nullptr, nullptr, scratch,
                                        stream));
      CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rec + ng - 1, 1, stream));
      CAMCAL_DRV_TRY(rd.begin(rec + ng - 1, 1, 0, stream));
      pending = FINAL_COST;
      return 0;
    }
    const double alpha2 = 1.0 - std::pow(0.6, it + 1.0);
    // with Y (and poses that are re-estimated) the evaluation also leaves every view's first-order
    // response to the intrinsic step, which moves the warm start of the re-estimation below
    const bool predict = Y && recompute_extrinsic;
    if (predict) CAMCAL_DRV_TRY(camcal_normal_eq_fused_response(CAMCAL_FISHEYE, tiles, intr, pose, Y, rec, rec + ng, scratch, stream));
    else CAMCAL_DRV_TRY(camcal_normal_eq_fused(CAMCAL_FISHEYE, tiles, intr, pose, 0.0, nullptr, nullptr, nullptr, rec,
                                               rec + ng, scratch, stream));
    CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rec, ng + ns, stream));      // [U|b|cost|C|d] in one message
    CAMCAL_DRV_TRY(camcal_solve_reduced(CAMCAL_FISHEYE, rec, rec + ng, mask_host, 0.0, -alpha2, intr, intr, dK,
                                        stats, stream));
    CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
    // cv2 re-estimates the poses before it looks at `change`: the launch does not wait for the host
    if (predict && iters) CAMCAL_DRV_TRY(camcal_pose_predict_refine(CAMCAL_FISHEYE, tiles, intr, Y, dK, -alpha2, 20, pose, 20, 1e-10, iters, stream));
    else {
      if (predict) CAMCAL_DRV_TRY(camcal_pose_predict(CAMCAL_FISHEYE, tiles, Y, dK, -alpha2, iters, 20, pose, stream));
      if (recompute_extrinsic) CAMCAL_DRV_TRY(camcal_pose_refine(CAMCAL_FISHEYE, tiles, intr, pose, 20, 1e-10, iters, stream));
    }
    pending = ITERATION;
    return 0;
  }

  int finish() {
    if (pending == ITERATION) {
      double st[5];
      if (seq > 0.0) CAMCAL_DRV_TRY(rd.wait(seq, stream, st, 5));
      else CAMCAL_DRV_TRY(rd.end(st, 5));
      // a view whose 6x6 block is not positive definite poisons C, d with NaN; a singular reduced
      // system leaves no step (cv2 would throw from its solve / produce NaN for the caller's checkRange)
      if (st[4] == 0.0 || !std::isfinite(st[2]) || !std::isfinite(st[3])) return CAMCAL_E_NUMERIC;
      change = std::sqrt(st[2]) / std::sqrt(st[3]);
      ++it;
      pending = READY;
      return 0;
    }
    if (pending == FINAL_COST) {
      if (seq > 0.0) {
        double st[6];
        CAMCAL_DRV_TRY(rd.wait(seq, stream, st, 6));
        cost = st[5];
      } else {
        CAMCAL_DRV_TRY(rd.end(&cost, 1));
      }
      pending = DONE;
      return std::isfinite(cost) ? 0 : CAMCAL_E_NUMERIC;
    }
    return CAMCAL_E_BADARG;
  }
};

// CvLevMarq::updateAlt as cvCalibrateCamera2Internal drives it (camera_calib.py:153-156) on the
// eliminated system: lambda starts at 1e-3, x10 on a rejected step (only the stored-block Schur
// kernel is repeated), /10 on an accepted one; stops after max_iter accepted steps or when the
// relative parameter change < eps.
struct PinholeLM {
  const camcal_tiles* tiles = nullptr;
  double* intr2[2] = {nullptr, nullptr};
  double* pose2[2] = {nullptr, nullptr};
  double* V2[2] = {nullptr, nullptr};
  double* g2[2] = {nullptr, nullptr};
  double *W = nullptr, *rec = nullptr, *glob_trial = nullptr, *dK = nullptr, *stats = nullptr, *norms = nullptr,
         *scratch = nullptr;
  const uint8_t* mask_host = nullptr;
  int max_iter = 30;
  double eps = 0.0;
  const camcal_comm* comm = nullptr;
  void* stream = nullptr;
  // state
  camcal_reader rd;
  int cur = 0;                         // which physical buffer holds the current point
  int fused = -1;                      // small sets: four launches per step (-1: not tried yet, 0: no)
  double* glob_cur = nullptr;          // fused: [U|b|cost] of the current point (rec or glob_trial, swapped on accept)
  double* glob_tri = nullptr;
  double seq = 0.0;                    // > 0: the step's scalars come through the reader's notify buffer
  bool first_cost_pending = false;     // the starting cost arrives with the first step's scalars
  // fused, with W_alt and intr_spare given: the first half of the NEXT step (evaluation B + reduced solve at the
  // trial point, lambda / 10) is enqueued behind the current step before the host waits for its verdict, so that
  // an accepted step -- the usual case -- finds it done; a rejected one ignores it (everything it wrote is spare)
  double* W_alt = nullptr;
  double* intr_spare = nullptr;
  double *W_c = nullptr, *intr_c = nullptr, *intr_t = nullptr, *intr_s = nullptr;
  bool spec_enqueued = false, spec_done = false;
  int spec_lam_lg10 = 0;
  int lam_lg10 = -3, iters = 0;
  double err = 0.0, prev_err = 0.0;
  bool fresh = true;
  enum { START, FIRST_COST, READY, TRIAL, DONE } pending = START;

  bool args_ok() const {
    if (!tiles || !W || !rec || !glob_trial || !dK || !stats || !norms || !scratch || !mask_host) return false;
    for (int k = 0; k < 2; ++k)
      if (!intr2[k] || !pose2[k] || !V2[k] || !g2[k]) return false;
    return true;
  }
  bool done() const { return pending == DONE; }

  int launch() {
    const int M = CAMCAL_PINHOLE;
    const int ng = camcal_glob_len(M), ns = camcal_schur_len(M);
    double* glob = rec;                  // [U|b|cost] of the current point
    double* schur = rec + ng;            // [C|d]
    if (pending == START) {
      max_iter = max_iter < 1 ? 1 : (max_iter > 1000 ? 1000 : max_iter);
      if (!(eps > 2.220446049250313e-16)) eps = 2.220446049250313e-16;
      if (!comm && fused < 0) {          // small set, one rank: the folded step (camcal_cross_schur_solve ...)
        const int rc = camcal_small_set_begin(M, tiles, scratch, stream);
        if (rc != 0 && rc != CAMCAL_E_UNSUPPORTED) return rc;
        fused = rc == 0 ? 1 : 0;
      } else if (fused < 0) {
        fused = 0;
      }
      glob_cur = rec;
      glob_tri = glob_trial;
      W_c = W; intr_c = intr2[0]; intr_t = intr2[1]; intr_s = intr_spare;
      CAMCAL_DRV_TRY(camcal_intr_normal(M, tiles, intr2[cur], pose2[cur], 1, glob, nullptr, V2[cur], g2[cur], scratch,
                                        stream));
      CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, glob, ng, stream));
      if (fused == 1) {
        // the first step does not depend on the starting cost, only its verdict does: launch it now and read the
        // cost with the step's scalars (slot 8 of the landing buffer) -- one host round trip less per solve
        CAMCAL_DRV_TRY(rd.begin(glob + ng - 1, 1, 8, stream));
        first_cost_pending = true;
        fresh = true;
        pending = READY;
      } else {
        CAMCAL_DRV_TRY(rd.begin(glob + ng - 1, 1, 0, stream));
        pending = FIRST_COST;
        return 0;
      }
    }
    if (pending != READY) return CAMCAL_E_BADARG;
    const int tri = 1 - cur;
    const double lam = std::pow(10.0, (double)lam_lg10);
    if (fused == 1) {
      if (spec_done) {
        spec_done = false;               // the previous launch() already did this step's first half
      } else if (fresh) {
        CAMCAL_DRV_TRY(camcal_cross_schur_solve(M, tiles, intr_c, pose2[cur], V2[cur], g2[cur], lam, W_c, glob_cur, schur,
                                                mask_host, -1.0, intr_t, dK, stats, scratch, stream));
      } else {
        CAMCAL_DRV_TRY(camcal_schur_solve(M, tiles, V2[cur], W_c, g2[cur], lam, glob_cur, schur, mask_host, -1.0, intr_c,
                                          intr_t, dK, stats, scratch, stream));
      }
      seq = 0.0;
      double* note = rd.notify(&seq);
      if (!note) seq = 0.0;
      CAMCAL_DRV_TRY(camcal_backsub_intr_normal(M, tiles, V2[cur], W_c, g2[cur], lam, dK, pose2[cur], pose2[tri], intr_t,
                                                glob_tri, V2[tri], g2[tri], norms, stats, note, seq, scratch, stream));
      if (!note) {
        CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
        CAMCAL_DRV_TRY(rd.begin(norms, 2, 5, stream));
        CAMCAL_DRV_TRY(rd.begin(glob_tri + ng - 1, 1, 7, stream));
      }
      spec_enqueued = false;
      if (W_alt && intr_s && iters + 1 < max_iter) {       // the next step's first half, should this one be accepted
        spec_lam_lg10 = lam_lg10 - 1 < -16 ? -16 : lam_lg10 - 1;
        CAMCAL_DRV_TRY(camcal_cross_schur_solve(M, tiles, intr_t, pose2[tri], V2[tri], g2[tri],
                                                std::pow(10.0, (double)spec_lam_lg10), W_alt, glob_tri, schur, mask_host, -1.0,
                                                intr_s, dK, stats, scratch, stream));
        spec_enqueued = true;
      }
      pending = TRIAL;
      return 0;
    }
    // evaluation B at the current point for a fresh step, the stored blocks at another lambda otherwise
    if (fresh) CAMCAL_DRV_TRY(camcal_cross_schur(M, tiles, intr2[cur], pose2[cur], V2[cur], g2[cur], lam, W, schur, scratch, stream));
    else CAMCAL_DRV_TRY(camcal_schur(M, tiles, V2[cur], W, g2[cur], lam, schur, scratch, stream));
    CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, schur, ns, stream));
    CAMCAL_DRV_TRY(camcal_solve_reduced(M, glob, schur, mask_host, lam, -1.0, intr2[cur], intr2[tri], dK, stats, stream));
    CAMCAL_DRV_TRY(camcal_backsub(M, tiles, V2[cur], W, g2[cur], lam, dK, pose2[cur], pose2[tri], norms, scratch, stream));
    CAMCAL_DRV_TRY(camcal_intr_normal(M, tiles, intr2[tri], pose2[tri], 1, glob_trial, nullptr, V2[tri], g2[tri],
                                      scratch, stream));
    if (comm) {                                  // the trial point's records: one message when they are adjacent
      if (norms == glob_trial + ng) CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, glob_trial, ng + 2, stream));
      else {
        CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, glob_trial, ng, stream));
        CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, norms, 2, stream));
      }
    }
    // one wait for the three small records
    CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
    CAMCAL_DRV_TRY(rd.begin(norms, 2, 5, stream));
    CAMCAL_DRV_TRY(rd.begin(glob_trial + ng - 1, 1, 7, stream));
    pending = TRIAL;
    return 0;
  }

  int finish() {
    if (pending == FIRST_COST) {
      CAMCAL_DRV_TRY(rd.end(&err, 1));
      prev_err = err;
      fresh = true;
      pending = READY;
      return 0;
    }
    if (pending != TRIAL) return CAMCAL_E_BADARG;
    const int ng = camcal_glob_len(CAMCAL_PINHOLE);
    double back[9];
    const int nb = first_cost_pending ? 9 : 8;
    if (seq > 0.0) CAMCAL_DRV_TRY(rd.wait(seq, stream, back, nb));
    else CAMCAL_DRV_TRY(rd.end(back, nb));
    if (first_cost_pending) {
      prev_err = back[8];
      first_cost_pending = false;
    }
    err = back[7];
    if (back[4] == 0.0) return CAMCAL_E_NUMERIC;   // singular reduced system
    if (err > prev_err) {                          // rejected: the same blocks at ten times the damping
      ++lam_lg10;
      if (lam_lg10 <= 16) {
        fresh = false;
        spec_enqueued = false;                     // what ran ahead assumed acceptance: ignored
        pending = READY;
        return 0;
      }
    }
    lam_lg10 = lam_lg10 - 1 < -16 ? -16 : lam_lg10 - 1;
    ++iters;
    const double rel = std::sqrt(back[0] + back[5]) / std::sqrt(back[1] + back[6]);
    // accept: the trial point becomes the current one
    if (fused == 1) {                    // its record by reference; rec itself only when the loop ends
      double* t = glob_cur; glob_cur = glob_tri; glob_tri = t;
      if (spec_enqueued && spec_lam_lg10 == lam_lg10) {      // ... and its evaluation B + solve have run already
        t = W_c; W_c = W_alt; W_alt = t;
        t = intr_c; intr_c = intr_t; intr_t = intr_s; intr_s = t;
        spec_done = true;
      } else {
        t = intr_c; intr_c = intr_t; intr_t = t;
      }
      spec_enqueued = false;
    } else {
      CAMCAL_DRV_TRY(camcal_copy_doubles(rec, glob_trial, ng, stream));
    }
    cur = 1 - cur;
    if (iters >= max_iter || rel < eps) {
      if (fused == 1) {                  // hand the result over where the caller expects it
        if (glob_cur != rec) CAMCAL_DRV_TRY(camcal_copy_doubles(rec, glob_cur, ng, stream));
        if (intr_c != intr2[cur]) CAMCAL_DRV_TRY(camcal_copy_doubles(intr2[cur], intr_c, 9, stream));
      }
      pending = DONE;
      return std::isfinite(err) ? 0 : CAMCAL_E_NUMERIC;
    }
    prev_err = err;
    fresh = true;
    pending = READY;
    return 0;
  }
};

}  // namespace camcal_drv
}  // namespace

extern "C" {

int camcal_fisheye_gn(const camcal_tiles* tiles, double* intr, double* pose, const uint8_t* mask_host,
                      int crit_type, int max_count, double eps, int recompute_extrinsic, double* rec, double* dK,
                      double* stats, int32_t* iters, double* Y, double* scratch, const camcal_comm* comm, void* stream,
                      double* cost_out_host, int32_t* iterations_out_host) {
  camcal_drv::FisheyeGN s;
  s.tiles = tiles; s.intr = intr; s.pose = pose; s.mask_host = mask_host; s.crit_type = crit_type;
  s.max_count = max_count; s.eps = eps; s.recompute_extrinsic = recompute_extrinsic; s.rec = rec; s.dK = dK;
  s.stats = stats; s.iters = iters; s.Y = Y; s.scratch = scratch; s.comm = comm; s.stream = stream;
  if (!s.args_ok() || !cost_out_host || !iterations_out_host) return CAMCAL_E_BADARG;
  while (!s.done()) {
    CAMCAL_DRV_TRY(s.launch());
    CAMCAL_DRV_TRY(s.finish());
  }
  *cost_out_host = s.cost;
  *iterations_out_host = s.it;
  return 0;
}

int camcal_pinhole_lm(const camcal_tiles* tiles, double* const intr2[2], double* const pose2[2],
                      double* const V2[2], double* const g2[2], double* W, double* rec, double* glob_trial,
                      double* dK, double* stats, double* norms, double* scratch, const uint8_t* mask_host,
                      int max_iter, double eps, const camcal_comm* comm, void* stream, double* err_out_host,
                      int32_t* iterations_out_host, int32_t* current_out_host) {
  if (!intr2 || !pose2 || !V2 || !g2 || !err_out_host || !iterations_out_host || !current_out_host)
    return CAMCAL_E_BADARG;
  camcal_drv::PinholeLM s;
  s.tiles = tiles;
  for (int k = 0; k < 2; ++k) { s.intr2[k] = intr2[k]; s.pose2[k] = pose2[k]; s.V2[k] = V2[k]; s.g2[k] = g2[k]; }
  s.W = W; s.rec = rec; s.glob_trial = glob_trial; s.dK = dK; s.stats = stats; s.norms = norms; s.scratch = scratch;
  s.mask_host = mask_host; s.max_iter = max_iter; s.eps = eps; s.comm = comm; s.stream = stream;
  if (!s.args_ok()) return CAMCAL_E_BADARG;
  while (!s.done()) {
    CAMCAL_DRV_TRY(s.launch());
    CAMCAL_DRV_TRY(s.finish());
  }
  *err_out_host = s.err;
  *iterations_out_host = s.iters;
  *current_out_host = s.cur;
  return 0;
}

}  // extern "C"
