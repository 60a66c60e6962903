// Host-side sequencing of the rig solve over the rig stage entry points of
// include/camcal.h: the starting point (camcal_rig_init) and CvLevMarq::updateAlt
// over the rig (camcal_rig_lm, the control flow of camcal_pinhole_lm).  No
// reference counterpart (the tool calibrates its cameras one after another,
// controller/controller_internal_calibration.py:375-399).
//
// Included by camcal_kernels.hip and by the tests' CPU lane emulator after
// camcal_driver.h; the includer provides, with internal linkage,
//   int rig_pair_sums(const camcal_rig*, double** sums_dev_out, void* stream)
//       pose_slot tables + the sums [c][k][13] of (R_ck, t_ck, 1) over shared poses
//   int rig_init_poses(const camcal_rig*, void* stream)
//       board poses of buffer 0 from the first camera that sees them and ext[0]
//   int camcal_read_sync(double* dst_host, const double* src_dev, int n, void* stream)
//   int camcal_write_sync(double* dst_dev, const double* src_host, int n, void* stream)
//   int camcal_fork_streams(void* stream, int n, void** side_out)   side_out[0] = stream; the others start
//       behind everything queued on stream (the emulator hands back `stream` n times)
//   int camcal_join_streams(void* stream, int n, void** side)       stream continues behind all of them
// and struct camcal_reader (camcal_driver.h).
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/camcal.h"
#include "camcal_rig.h"

extern "C" {

int camcal_rig_init(const camcal_rig* rig, const camcal_comm* comm, void* stream) {
  if (!rig || rig->n_cams < 1 || rig->n_cams > CAMCAL_RIG_MAX_CAMS || rig->P <= 0 || !rig->intr[0] || !rig->ext[0] ||
      !rig->rig_pose[0] || !rig->scratch)
    return CAMCAL_E_BADARG;
  const int nC = rig->n_cams, model = rig->model;
  if (camcal_n_intr(model) < 0) return CAMCAL_E_BADMODEL;
  // per-camera poses of every view at the guessed intrinsics (the single-camera kernels)
  // -- the cameras are independent here: one stream each, joined before the pair sums
  for (int c = 0; c < nC; ++c)
    if (!rig->cam[c].pose || !rig->cam[c].pose_slot) return CAMCAL_E_BADARG;
  void* side[CAMCAL_RIG_MAX_CAMS];
  CAMCAL_DRV_TRY(camcal_fork_streams(stream, nC, side));
  int64_t so = 0;
  int rc_cam = 0;
  for (int c = 0; c < nC && rc_cam == 0; ++c) {
    const camcal_rig_cam& k = rig->cam[c];
    int32_t* st = rig->status ? rig->status + so : nullptr;
    so += k.tiles.n_slots;
    rc_cam = camcal_pose_init(model, &k.tiles, rig->intr[0] + c * 16, k.pose, st, side[c]);
    if (rc_cam == 0) rc_cam = camcal_pose_refine(model, &k.tiles, rig->intr[0] + c * 16, k.pose, 20, 1e-10, nullptr, side[c]);
  }
  CAMCAL_DRV_TRY(camcal_join_streams(stream, nC, side));      // also after a failed launch: nothing stays forked
  if (rc_cam && !comm) return rc_cam;
  // With a communicator a rank-local failure must not leave the other ranks waiting in the
  // all-reduce below: the number of failed ranks travels with the pair sums (one more double)
  // and every rank takes the same decision afterwards.
  const int n_sums = nC * nC * 13;
  double* sums_dev = nullptr;
  int rc_local = rc_cam;
  if (rc_local == 0) rc_local = rig_pair_sums(rig, &sums_dev, stream);
  if (!sums_dev) sums_dev = rig->scratch;                     // where rig_pair_sums puts them
  const double failed = rc_local ? 1.0 : 0.0;
  if (comm) CAMCAL_DRV_TRY(camcal_write_sync(sums_dev + n_sums, &failed, 1, stream));
  else if (rc_local) return rc_local;
  CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, sums_dev, (int64_t)n_sums + (comm ? 1 : 0), stream));
  double sums[CAMCAL_RIG_MAX_CAMS * CAMCAL_RIG_MAX_CAMS * 13 + 1], ext[CAMCAL_RIG_MAX_CAMS * 6];
  sums[n_sums] = 0.0;
  CAMCAL_DRV_TRY(camcal_read_sync(sums, sums_dev, n_sums + (comm ? 1 : 0), stream));
  if (rc_local) return rc_local;
  if (sums[n_sums] != 0.0) return CAMCAL_E_PEER;
  if (!camcal::righost::ext_from_pair_sums(nC, sums, ext)) return CAMCAL_E_DISJOINT;   // a camera shares no pose (same sums on every rank)
  CAMCAL_DRV_TRY(camcal_write_sync(rig->ext[0], ext, nC * 6, stream));
  return rig_init_poses(rig, stream);
}

int camcal_rig_lm(const camcal_rig* rig, const uint8_t* mask_host, int max_iter, double eps, const camcal_comm* comm,
                  void* stream, double* err_out_host, int32_t* iterations_out_host, int32_t* current_out_host) {
  if (!rig || !mask_host || !err_out_host || !iterations_out_host || !current_out_host) return CAMCAL_E_BADARG;
  const int ni = camcal_n_intr(rig->model);
  if (ni < 0) return CAMCAL_E_BADMODEL;
  if (rig->n_cams < 1 || rig->n_cams > CAMCAL_RIG_MAX_CAMS) return CAMCAL_E_BADARG;
  const camcal::RigLayout L(ni, rig->n_cams);
  max_iter = max_iter < 1 ? 1 : (max_iter > 1000 ? 1000 : max_iter);
  if (!(eps > 2.220446049250313e-16)) eps = 2.220446049250313e-16;
  camcal_reader rd;
  int cur = 0, lam_lg10 = -3, iters = 0;
  CAMCAL_DRV_TRY(camcal_rig_evaluate(rig, cur, stream));
  CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rig->rec[cur], L.o_cost + 1, stream));
  double err = 0.0;
  CAMCAL_DRV_TRY(rd.begin(rig->rec[cur] + L.o_cost, 1, 0, stream));
  CAMCAL_DRV_TRY(rd.end(&err, 1));
  for (;;) {
    const int tri = 1 - cur;
    const double prev_err = err;
    double back[5] = {0, 0, 0, 0, 0};              // cost | norms (2) | gstats (2) of the trial point
    for (;;) {
      const double lam = std::pow(10.0, (double)lam_lg10);
      double* note = nullptr;
      double seq = 0.0;
      CAMCAL_DRV_TRY(camcal_rig_eliminate(rig, cur, lam, stream));
      CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rig->rec[cur] + L.o_C, (int64_t)L.n * L.n + L.n, stream));
      CAMCAL_DRV_TRY(camcal_rig_solve(rig, cur, mask_host, lam, tri, stream));
      CAMCAL_DRV_TRY(camcal_rig_backsub(rig, cur, lam, tri, stream));
      if (comm) {
        CAMCAL_DRV_TRY(camcal_rig_evaluate(rig, tri, stream));
        // records, cost and pose-step norms of the trial point: one message
        CAMCAL_DRV_TRY(camcal_comm_allreduce(comm, rig->rec[tri], L.o_gstats, stream));
        CAMCAL_DRV_TRY(rd.begin(rig->rec[tri] + L.o_cost, 5, 0, stream));
      } else {
        // one process: the cost travels to the host while the chain rule of the trial point runs -- written by
        // the residual stage's last kernel into the reader's pinned buffer where that is mapped (no copy between
        // the stage and the chain rule), else by a copy
        note = rd.notify(&seq);
        if (!note) seq = 0.0;
        CAMCAL_DRV_TRY(camcal_rig_residuals_notify(rig, tri, note, seq, stream));
        if (!note) CAMCAL_DRV_TRY(rd.begin(rig->rec[tri] + L.o_cost, 5, 0, stream));
        CAMCAL_DRV_TRY(camcal_rig_chain(rig, tri, stream));
      }
      if (note) CAMCAL_DRV_TRY(rd.wait(seq, stream, back, 5));
      else CAMCAL_DRV_TRY(rd.end(back, 5));
      err = back[0];
      if (err <= prev_err) break;                  // NaN (failed factorisation) rejects the step
      ++lam_lg10;
      if (lam_lg10 > 16) break;
    }
    lam_lg10 = lam_lg10 - 1 < -16 ? -16 : lam_lg10 - 1;
    ++iters;
    const double rel = std::sqrt(back[3] + back[1]) / std::sqrt(back[4] + back[2]);
    cur = tri;                                     // accept: the trial buffers become the current ones
    if (iters >= max_iter || rel < eps) break;
  }
  *err_out_host = err;
  *iterations_out_host = iters;
  *current_out_host = cur;
  return std::isfinite(err) ? 0 : CAMCAL_E_NUMERIC;
}

}  // extern "C"
