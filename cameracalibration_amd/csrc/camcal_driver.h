// Host-side sequencing of the two solver loops over the stage entry points of
// include/camcal.h -- the loops that run inside OpenCV's C++ for the reference:
//   cv::fisheye::calibrate's damped Gauss-Newton   (camcal_fisheye_gn)
//   CvLevMarq::updateAlt inside cvCalibrateCamera2 (camcal_pinhole_lm)
// One process per device; with a camcal_comm the global records are summed over
// the ranks between the stages (include/camcal.h), on the same stream, without a
// host round trip of their own.
//
// Included by camcal_kernels.hip and by the tests' CPU lane emulator; the
// includer provides, with internal linkage,
//   struct camcal_reader {            // one per running solve (RAII)
//     int begin(const double* src, int n, int offset, void* stream);
//     int end(double* dst_host, int n_total);
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

extern "C" {

int camcal_fisheye_gn(const camcal_tiles* tiles, double* intr, double* pose, const uint8_t* mask_host,
                      int crit_type, int max_count, double eps, int recompute_extrinsic, double* rec, double* dK,
                      double* stats, int32_t* iters, double* Y, double* scratch, const camcal_comm* comm, void* stream,
                      double* cost_out_host, int32_t* iterations_out_host) {
  if (!tiles || !intr || !pose || !mask_host || !rec || !dK || !stats || !scratch || !cost_out_host ||
      !iterations_out_host || crit_type < 1 || crit_type > 3)
    return CAMCAL_E_BADARG;
  const int ng = camcal_glob_len(CAMCAL_FISHEYE), ns = camcal_schur_len(CAMCAL_FISHEYE);
  camcal_reader rd;
  double change = 1.0;
  int it = 0;
  for (;;) {
    if ((crit_type == 1 && it >= max_count) || (crit_type == 2 && change <= eps) ||
        (crit_type == 3 && (change <= eps || it >= max_count)))
      break;
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
    double st[5];
    CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
    // cv2 re-estimates the poses before it looks at `change`: the launch does not wait for the host
    if (predict) CAMCAL_DRV_TRY(camcal_pose_predict(CAMCAL_FISHEYE, tiles, Y, dK, -alpha2, iters, 20, pose, stream));
    if (recompute_extrinsic) CAMCAL_DRV_TRY(camcal_pose_refine(CAMCAL_FISHEYE, tiles, intr, pose, 20, 1e-10, iters, stream));
    CAMCAL_DRV_TRY(rd.end(st, 5));
    // a view whose 6x6 block is not positive definite poisons C, d with NaN; a singular reduced
    // system leaves no step (cv2 would throw from its solve / produce NaN for the caller's checkRange)
    if (st[4] == 0.0 || !std::isfinite(st[2]) || !std::isfinite(st[3])) return CAMCAL_E_NUMERIC;
    change = std::sqrt(st[2]) / std::sqrt(st[3]);
    ++it;
  }
  CAMCAL_DRV_TRY(camcal_intr_normal(CAMCAL_FISHEYE, tiles, intr, pose, 0, rec, nullptr, nullptr, nullptr, scratch,
                                    stream));
  CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rec + ng - 1, 1, stream));
  CAMCAL_DRV_TRY(rd.begin(rec + ng - 1, 1, 0, stream));
  CAMCAL_DRV_TRY(rd.end(cost_out_host, 1));
  *iterations_out_host = it;
  return std::isfinite(*cost_out_host) ? 0 : CAMCAL_E_NUMERIC;
}

int camcal_pinhole_lm(const camcal_tiles* tiles, double* const intr2[2], double* const pose2[2],
                      double* const V2[2], double* const g2[2], double* W, double* rec, double* glob_trial,
                      double* dK, double* stats, double* norms, double* scratch, const uint8_t* mask_host,
                      int max_iter, double eps, const camcal_comm* comm, void* stream, double* err_out_host,
                      int32_t* iterations_out_host, int32_t* current_out_host) {
  if (!tiles || !intr2 || !pose2 || !V2 || !g2 || !W || !rec || !glob_trial || !dK || !stats || !norms || !scratch ||
      !mask_host || !err_out_host || !iterations_out_host || !current_out_host)
    return CAMCAL_E_BADARG;
  for (int k = 0; k < 2; ++k)
    if (!intr2[k] || !pose2[k] || !V2[k] || !g2[k]) return CAMCAL_E_BADARG;
  const int M = CAMCAL_PINHOLE;
  const int ng = camcal_glob_len(M), ns = camcal_schur_len(M);
  camcal_reader rd;
  double* glob = rec;                  // [U|b|cost] of the current point
  double* schur = rec + ng;            // [C|d]
  max_iter = max_iter < 1 ? 1 : (max_iter > 1000 ? 1000 : max_iter);
  if (!(eps > 2.220446049250313e-16)) eps = 2.220446049250313e-16;
  int cur = 0;                         // which physical buffer holds the current point
  int lam_lg10 = -3, iters = 0;
  CAMCAL_DRV_TRY(camcal_intr_normal(M, tiles, intr2[cur], pose2[cur], 1, glob, nullptr, V2[cur], g2[cur], scratch,
                                    stream));
  CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, glob, ng, stream));
  double err = 0.0;
  CAMCAL_DRV_TRY(rd.begin(glob + ng - 1, 1, 0, stream));
  CAMCAL_DRV_TRY(rd.end(&err, 1));
  for (;;) {
    const int tri = 1 - cur;
    const double prev_err = err;
    double st[5] = {0, 0, 0, 0, 0}, nr[2] = {0, 0};
    bool fresh = true;
    for (;;) {
      const double lam = std::pow(10.0, (double)lam_lg10);
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
      double back[8];                              // one wait for the three small records
      CAMCAL_DRV_TRY(rd.begin(stats, 5, 0, stream));
      CAMCAL_DRV_TRY(rd.begin(norms, 2, 5, stream));
      CAMCAL_DRV_TRY(rd.begin(glob_trial + ng - 1, 1, 7, stream));
      CAMCAL_DRV_TRY(rd.end(back, 8));
      for (int i = 0; i < 5; ++i) st[i] = back[i];
      nr[0] = back[5]; nr[1] = back[6];
      err = back[7];
      if (st[4] == 0.0) return CAMCAL_E_NUMERIC;   // singular reduced system
      if (!(err > prev_err)) break;
      ++lam_lg10;
      if (lam_lg10 > 16) break;
      fresh = false;
    }
    lam_lg10 = lam_lg10 - 1 < -16 ? -16 : lam_lg10 - 1;
    ++iters;
    const double rel = std::sqrt(st[0] + nr[0]) / std::sqrt(st[1] + nr[1]);
    // accept: the trial point becomes the current one
    CAMCAL_DRV_TRY(camcal_copy_doubles(glob, glob_trial, ng, stream));
    cur = tri;
    if (iters >= max_iter || rel < eps) break;
  }
  *err_out_host = err;
  *iterations_out_host = iters;
  *current_out_host = cur;
  return std::isfinite(err) ? 0 : CAMCAL_E_NUMERIC;
}

}  // extern "C"
