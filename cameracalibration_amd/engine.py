"""Device-resident state of one process's share of the views and the stage
calls on it (thin layer over the C ABI in include/camcal.h).

PyTorch is plumbing here: it owns device memory and the stream; every number is
produced by libcamcal.so.  One :class:`Engine` per process / per GPU.

Layout (see include/camcal.h): views are sorted by corner count, placed in
"slots", 64 slots form a tile (= one wavefront), points are packed so that
corner k of a tile's 64 views is contiguous per component.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

LANES = 64


def make_layout(n_per_view: np.ndarray):
    """Host-side slot/tile layout for views with the given corner counts.

    Returns dict(slot_view i32[S], lane_n i32[S], tile_nmax i32[T],
    tile_off i64[T], total_floats)."""
    n = np.asarray(n_per_view, dtype=np.int64)
    N = n.shape[0]
    T = max((N + LANES - 1) // LANES, 1)
    S = T * LANES
    order = np.argsort(-n, kind="stable")
    slot_view = np.full(S, -1, dtype=np.int32)
    slot_view[:N] = order
    lane_n = np.zeros(S, dtype=np.int32)
    lane_n[:N] = n[order]
    tile_nmax = lane_n.reshape(T, LANES).max(axis=1).astype(np.int32)
    sizes = tile_nmax.astype(np.int64) * (5 * LANES)
    tile_off = np.zeros(T, dtype=np.int64)
    tile_off[1:] = np.cumsum(sizes)[:-1]
    return dict(slot_view=slot_view, lane_n=lane_n, tile_nmax=tile_nmax, tile_off=tile_off,
                total_floats=int(sizes.sum()), n_tiles=T, n_slots=S, n_views=N)


class Engine:
    """Packed observations + solver buffers of one rank.

    Parameters
    ----------
    model : 0 pinhole / 1 fisheye
    view_ptr : (N+1,) int64 CSR offsets
    obj : (n_obs, 3) float32, img : (n_obs, 2) float32  (numpy or torch)
    device : torch device; ``lib``: CamcalLib (default: the HIP build)
    """

    #: from this many views on the layout tables are computed on the device
    DEVICE_LAYOUT_MIN_VIEWS = 8192

    @property
    def layout(self):
        """Host copy of the layout tables (make_layout's dict); made on first use for large sets."""
        if self._layout is None:
            T, S = self.tiles.n_tiles, self.S
            sizes = self.t_tile_nmax.to(torch.int64) * (5 * LANES)
            self._layout = dict(slot_view=self.t_slot_view.cpu().numpy(), lane_n=self.t_lane_n.cpu().numpy(),
                                tile_nmax=self.t_tile_nmax.cpu().numpy(), tile_off=self.t_tile_off.cpu().numpy(),
                                total_floats=int(sizes.sum().item()), n_tiles=T, n_slots=S, n_views=self.n_views)
        return self._layout

    @layout.setter
    def layout(self, value):
        self._layout = value

    #: the fisheye loop starts every pose re-estimation from the first-order predicted pose (camcal_pose_predict);
    #: False: from the previous pose (the oracle's warm-start path, step for step) -- tests and measurements
    PREDICT_POSES = True

    #: lanes per view of the engines created from now on (camcal_tiles.lanes_per_view: 0 = by size,
    #: 1 / 16 / 64 force a mapping); a plain class default so that tests and tools can pin a mapping
    LANES_PER_VIEW = 0

    def _setup(self, model, device, lib):
        self.lib = lib if lib is not None else _lib.load()
        self.model = int(model)
        self.device = torch.device(device)
        if self.device.type != "cuda" and not self.lib.is_emulator:
            raise _lib.CamcalError("the HIP library needs a cuda (ROCm) device; there is no CPU path")
        self.NI = self.lib.camcal_n_intr(self.model)
        if self.NI < 0:
            raise _lib.CamcalError("unknown camera model")
        self.glob_len = self.lib.camcal_glob_len(self.model)
        self.schur_len = self.lib.camcal_schur_len(self.model)

    def __init__(self, model, view_ptr, obj, img, device, lib=None, lanes_per_view=None):
        self._setup(model, device, lib)
        dev = self.device
        lpv = int(self.LANES_PER_VIEW if lanes_per_view is None else lanes_per_view)
        n_views = int(len(view_ptr)) - 1
        t_obj = t_img = None
        if dev.type == "cuda" and n_views >= self.DEVICE_LAYOUT_MIN_VIEWS:
            # large sets: camcal_layout_device -- the sort and the prefix sums of the layout run on the device (the
            # same tables as make_layout; 1M views cost 10-15 ms in numpy, more than a third of a whole C5 solve),
            # offsets and points are checked there too, and one read-back brings the sizes and the flags; the
            # host copy of the tables is only made when a caller asks for it
            t_vp = torch.as_tensor(view_ptr).to(device=dev, dtype=torch.int64).contiguous()
            t_obj = torch.as_tensor(obj, dtype=torch.float32).reshape(-1, 3).contiguous().to(dev)
            t_img = torch.as_tensor(img, dtype=torch.float32).reshape(-1, 2).contiguous().to(dev)
            self.n_views = n_views
            T = max((n_views + LANES - 1) // LANES, 1)
            S = T * LANES
            tab = torch.empty(2 * T + 2 * 4 + 2 * S + T, dtype=torch.int32, device=dev)    # int64 tables first: alignment
            self.t_tile_off = tab[:2 * T].view(torch.int64)
            head = tab[2 * T:2 * T + 8].view(torch.int64)
            o = 2 * T + 8
            self.t_slot_view, self.t_lane_n, self.t_tile_nmax = tab[o:o + S], tab[o + S:o + 2 * S], tab[o + 2 * S:]
            tmp_len = int(self.lib.camcal_layout_device_temp_len(n_views))
            if tmp_len < 0:
                self.lib.check(tmp_len, "camcal_layout_device_temp_len")
            tmp = torch.empty(tmp_len, dtype=torch.uint8, device=dev)
            self._call("camcal_layout_device", n_views, t_vp.data_ptr(), int(t_obj.shape[0]), t_obj.data_ptr(),
                       self.t_slot_view.data_ptr(), self.t_lane_n.data_ptr(), self.t_tile_nmax.data_ptr(),
                       self.t_tile_off.data_ptr(), head.data_ptr(), tmp.data_ptr(), tmp_len, self._stream())
            h = head.cpu()                                          # the one read-back: n_obs, packed floats, flags
            del tmp
            self.n_obs, total_floats, flags = int(h[0]), int(h[1]), int(h[2])
            if flags & (_lib.LAYOUT_BAD_COUNT | _lib.LAYOUT_BAD_TOTAL) or t_img.shape[0] != t_obj.shape[0]:
                raise ValueError("view_ptr does not match the point arrays")
            planar = not flags & _lib.LAYOUT_NOT_PLANAR
            self._layout = None
            lay = dict(n_tiles=T, n_slots=S, total_floats=total_floats)
        else:
            view_ptr = np.ascontiguousarray(torch.as_tensor(view_ptr).cpu().numpy(), dtype=np.int64)
            n = np.diff(view_ptr)
            self.n_views = int(n.shape[0])
            self.n_obs = int(view_ptr[-1])
            lay = make_layout(n)
            self._layout = lay
            # layout tables + CSR offsets: one upload, carved up (int64 tables first: alignment)
            T, S = lay["n_tiles"], lay["n_slots"]
            tab = np.concatenate([lay["tile_off"].view(np.int32), view_ptr.view(np.int32), lay["slot_view"],
                                  lay["lane_n"], lay["tile_nmax"]])
            t_tab = torch.from_numpy(tab).to(dev)
            o = 0
            self.t_tile_off = t_tab[o:o + 2 * T].view(torch.int64); o += 2 * T
            t_vp = t_tab[o:o + 2 * (self.n_views + 1)].view(torch.int64); o += 2 * (self.n_views + 1)
            self.t_slot_view = t_tab[o:o + S]; o += S
            self.t_lane_n = t_tab[o:o + S]; o += S
            self.t_tile_nmax = t_tab[o:o + T]
        self.t_pts = torch.empty(max(lay["total_floats"], 1), dtype=torch.float32, device=dev)   # pack writes all of it
        self.S = S
        self.tiles = _lib.Tiles(self.t_pts.data_ptr(), self.t_tile_off.data_ptr(),
                                self.t_tile_nmax.data_ptr(), self.t_lane_n.data_ptr(),
                                lay["n_tiles"], S, 0, lpv)
        # pack on the device
        if t_obj is None:
            if isinstance(obj, np.ndarray):                     # host arrays: no device round trip for the flag
                planar = not np.asarray(obj).reshape(-1, 3)[:, 2].any()
                t_obj = torch.as_tensor(obj, dtype=torch.float32).reshape(-1, 3).contiguous().to(dev)
            else:
                t_obj = torch.as_tensor(obj, dtype=torch.float32).reshape(-1, 3).contiguous().to(dev)
                planar = bool((t_obj[:, 2] == 0).all().item())
            t_img = torch.as_tensor(img, dtype=torch.float32).reshape(-1, 2).contiguous().to(dev)
            if t_obj.shape[0] != self.n_obs or t_img.shape[0] != self.n_obs:
                raise ValueError("view_ptr does not match the point arrays")
        self.tiles.planar = 1 if planar else 0
        self._call("camcal_pack_tiles", t_obj.data_ptr(), t_img.data_ptr(), t_vp.data_ptr(),
                   self.t_slot_view.data_ptr(), C.byref(self.tiles), self.t_pts.data_ptr(), self._stream())
        del t_obj, t_img, t_vp        # stream-ordered: the allocator reuses them only behind the pack
        self._alloc_solver_buffers(S, lay["n_tiles"])

    def _alloc_solver_buffers(self, S, n_tiles):
        """Solver state for S slots: one zeroed allocation, carved up (every piece a multiple of 8 bytes)."""
        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        NI = self.NI
        # norms right behind glob1: one all-reduce for a trial point
        sizes = dict(intr0=16, intr1=16, pose0=6 * S, pose1=6 * S, V0=21 * S, V1=21 * S, g0=6 * S, g1=6 * S,
                     W=NI * 6 * S, rec=self.glob_len + self.schur_len, glob1=self.glob_len, norms=2, dK=16,
                     stats=8, view_cost=S, ints=S)
        pool = torch.zeros(sum(sizes.values()), **f64)
        part, off = {}, 0
        for k, n_k in sizes.items():
            part[k] = pool[off:off + n_k]
            off += n_k
        self._pool = pool
        self.intr = [part["intr0"], part["intr1"]]
        self.pose = [part["pose0"].view(6, S), part["pose1"].view(6, S)]
        pool[32:32 + 12 * S].view(2, 6, S)[:, 5] = 1.0      # both pose buffers: identity-like default, one fill
        # pose blocks are double-buffered (current / trial point), W is not
        self.Vs = [part["V0"].view(21, S), part["V1"].view(21, S)]
        self.gs = [part["g0"].view(6, S), part["g1"].view(6, S)]
        self.W = part["W"].view(NI * 6, S)
        self.Y = self.W          # the fisheye loop's pose response V^-1 W^T: that loop stores no blocks
        # [U|b|cost] and [C|d] of buffer 0 share one allocation so a full
        # normal-equation evaluation needs a single all-reduce (self.rec)
        self.rec = part["rec"]
        self.glob = [self.rec[:self.glob_len], part["glob1"]]
        self.schur = self.rec[self.glob_len:]
        self.dK = part["dK"]
        self.stats = part["stats"]
        self.norms = part["norms"]
        self.view_cost = part["view_cost"]
        # partial sums: every entry a launch reads was written by that launch
        self.scratch = torch.empty(int(self.lib.camcal_scratch_len(self.model, n_tiles)), **f64)
        ints = part["ints"].view(torch.int32)                # 2 S int32 in the same allocation
        self.status, self.iters = ints[:S], ints[S:]

    # -- plumbing ------------------------------------------------------------
    def _comm_pools(self):
        """Tensors a torch.distributed callback communicator may be asked to reduce parts of."""
        return [self._pool]

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _call(self, name, *args):
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                rc = getattr(self.lib, name)(*args)
        else:
            rc = getattr(self.lib, name)(*args)
        self.lib.check(rc, name)

    def _tp(self):
        return C.byref(self.tiles)

    # -- parameters ------------------------------------------------------------
    def set_intr(self, which, values):
        v = np.zeros(16)
        v[:self.NI] = np.asarray(values, dtype=np.float64)[:self.NI]
        self.intr[which].copy_(torch.from_numpy(v))          # one host-to-device copy

    def get_intr(self, which):
        return self.intr[which][:self.NI].cpu().numpy().copy()

    def set_poses(self, which, poses_view_order):
        """poses (N, 6) in caller's view order -> slot order on the device."""
        p = np.zeros((self.S, 6))
        p[:, 5] = 1.0
        sv = self.layout["slot_view"]
        valid = sv >= 0
        p[valid] = np.asarray(poses_view_order, dtype=np.float64)[sv[valid]]
        self.pose[which].copy_(torch.from_numpy(np.ascontiguousarray(p.T)).to(self.device))

    def get_poses(self, which):
        """(N, 6) poses in the caller's view order."""
        p = self.pose[which].cpu().numpy().T
        sv = self.layout["slot_view"]
        valid = sv >= 0
        out = np.empty((self.n_views, 6))
        out[sv[valid]] = p[valid]
        return out

    def read_solution(self):
        """(intrinsics (NI,), poses (N, 6) in the caller's view order, indices of the views whose pose
        initialisation failed) of buffer 0 -- for small sets one device-to-host copy instead of three."""
        S = self.S
        pool = self._pool
        head = 32 + 12 * S                     # intr0 | intr1 | pose0 | pose1 at the head of the pool
        if S <= 8192:
            h = pool[:head].cpu().numpy()
            base = pool.data_ptr()
            io = (self.intr[0].data_ptr() - base) // 8
            po = (self.pose[0].data_ptr() - base) // 8
            intr = h[io:io + self.NI].copy()
            p = h[po:po + 6 * S].reshape(6, S).T
        else:
            intr = self.get_intr(0)
            p = self.pose[0].cpu().numpy().T
        sv = self.layout["slot_view"]
        valid = sv >= 0
        poses = np.empty((self.n_views, 6))
        poses[sv[valid]] = p[valid]
        st = self.status.cpu().numpy()
        bad = np.sort(sv[valid][st[valid] != 0])
        return intr, poses, bad

    def slots_to_views(self, per_slot):
        a = per_slot.cpu().numpy()
        sv = self.layout["slot_view"]
        valid = sv >= 0
        out = np.empty((self.n_views,) + a.shape[1:], dtype=a.dtype)
        out[sv[valid]] = a[valid]
        return out

    # -- stages ----------------------------------------------------------------
    def pose_init(self, intr_idx, pose_idx):
        self._call("camcal_pose_init", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), self.status.data_ptr(), self._stream())

    def pose_refine(self, intr_idx, pose_idx, max_iter=20, tol=1e-10):
        self._call("camcal_pose_refine", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), int(max_iter), float(tol), self.iters.data_ptr(),
                   self._stream())

    @property
    def V(self):
        return self.Vs[0]

    @property
    def g(self):
        return self.gs[0]

    def swap_blocks(self):
        """Trial pose blocks become the current ones."""
        self.Vs[0], self.Vs[1] = self.Vs[1], self.Vs[0]
        self.gs[0], self.gs[1] = self.gs[1], self.gs[0]

    def intr_normal(self, intr_idx, pose_idx, glob_idx, with_normal=True, view_cost=False, blocks=None):
        """Evaluation A.  ``blocks`` = 0/1 also stores V, g into that buffer."""
        self._call("camcal_intr_normal", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), 1 if with_normal else 0,
                   self.glob[glob_idx].data_ptr(),
                   self.view_cost.data_ptr() if view_cost else None,
                   self.Vs[blocks].data_ptr() if blocks is not None else None,
                   self.gs[blocks].data_ptr() if blocks is not None else None,
                   self.scratch.data_ptr(), self._stream())

    def cross_schur(self, intr_idx, pose_idx, lam, store_w=True):
        """Evaluation B at the current point: W (optionally stored) and [C|d]."""
        self._call("camcal_cross_schur", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), self.Vs[0].data_ptr(), self.gs[0].data_ptr(), float(lam),
                   self.W.data_ptr() if store_w else None, self.schur.data_ptr(),
                   self.scratch.data_ptr(), self._stream())

    def view_blocks(self, intr_idx, pose_idx):
        self._call("camcal_view_blocks", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), self.V.data_ptr(), self.W.data_ptr(),
                   self.g.data_ptr(), self._stream())

    def schur_complement(self, lam):
        self._call("camcal_schur", self.model, self._tp(), self.V.data_ptr(), self.W.data_ptr(),
                   self.g.data_ptr(), float(lam), self.schur.data_ptr(), self.scratch.data_ptr(),
                   self._stream())

    def solve_reduced(self, glob_idx, mask, lam, scale, intr_in, intr_out):
        m = bytes(bytearray(int(bool(x)) for x in mask))
        self._call("camcal_solve_reduced", self.model, self.glob[glob_idx].data_ptr(),
                   self.schur.data_ptr(), m, float(lam), float(scale), self.intr[intr_in].data_ptr(),
                   self.intr[intr_out].data_ptr(), self.dK.data_ptr(), self.stats.data_ptr(),
                   self._stream())

    def backsub(self, lam, pose_in, pose_out):
        self._call("camcal_backsub", self.model, self._tp(), self.V.data_ptr(), self.W.data_ptr(),
                   self.g.data_ptr(), float(lam), self.dK.data_ptr(), self.pose[pose_in].data_ptr(),
                   self.pose[pose_out].data_ptr(), self.norms.data_ptr(), self.scratch.data_ptr(),
                   self._stream())

    # -- whole solves sequenced by the library (single rank) ---------------------------
    @staticmethod
    def _mask_bytes(mask):
        return bytes(bytearray(int(bool(x)) for x in mask))

    def solve_fisheye(self, mask, criteria, recompute_extrinsic=True, comm=None):
        """cv::fisheye::calibrate's loop from ``intr[0]`` / ``pose[0]`` (camcal_fisheye_gn).
        ``comm``: pointer to a camcal_comm (multi-rank solve) or None.
        Returns (sum of squared residuals over all ranks, iterations)."""
        ctype, maxc, eps = criteria
        cost, its = C.c_double(0.0), C.c_int32(0)
        self._call("camcal_fisheye_gn", self._tp(), self.intr[0].data_ptr(), self.pose[0].data_ptr(),
                   self._mask_bytes(mask), int(ctype), int(maxc), float(eps), 1 if recompute_extrinsic else 0,
                   self.rec.data_ptr(), self.dK.data_ptr(), self.stats.data_ptr(), self.iters.data_ptr(),
                   self.Y.data_ptr() if self.PREDICT_POSES else None,
                   self.scratch.data_ptr(), comm, self._stream(), C.byref(cost), C.byref(its))
        return float(cost.value), int(its.value)

    def solve_pinhole(self, mask, max_iter, eps, comm=None):
        """CvLevMarq loop from ``intr[0]`` / ``pose[0]`` (camcal_pinhole_lm); on return
        buffer 0 holds the result.  Returns (errNorm2, iterations)."""
        two = C.c_void_p * 2
        err, its, cur = C.c_double(0.0), C.c_int32(0), C.c_int32(0)
        self._call("camcal_pinhole_lm", self._tp(),
                   two(self.intr[0].data_ptr(), self.intr[1].data_ptr()),
                   two(self.pose[0].data_ptr(), self.pose[1].data_ptr()),
                   two(self.Vs[0].data_ptr(), self.Vs[1].data_ptr()),
                   two(self.gs[0].data_ptr(), self.gs[1].data_ptr()),
                   self.W.data_ptr(), self.rec.data_ptr(), self.glob[1].data_ptr(), self.dK.data_ptr(),
                   self.stats.data_ptr(), self.norms.data_ptr(), self.scratch.data_ptr(), self._mask_bytes(mask),
                   int(max_iter), float(eps), comm, self._stream(), C.byref(err), C.byref(its), C.byref(cur))
        if cur.value == 1:                   # the result sits in the second buffers: make them buffer 0
            self.intr.reverse()
            self.pose.reverse()
            self.Vs.reverse()
            self.gs.reverse()
        return float(err.value), int(its.value)

    def normal_eq(self, intr_idx, pose_idx, lam, store_blocks=True, fused=False):
        """One full normal-equation evaluation (the benchmarked unit) into
        ``self.rec`` = [U|b|cost|C|d].  ``fused``: single pass over the points
        (camcal_normal_eq_fused), else evaluation A + evaluation B."""
        if fused:
            sb = bool(store_blocks)
            self._call("camcal_normal_eq_fused", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                       self.pose[pose_idx].data_ptr(), float(lam),
                       self.V.data_ptr() if sb else None, self.W.data_ptr() if sb else None,
                       self.g.data_ptr() if sb else None,
                       self.glob[0].data_ptr(), self.schur.data_ptr(), self.scratch.data_ptr(), self._stream())
        else:
            self._call("camcal_normal_eq", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                       self.pose[pose_idx].data_ptr(), float(lam), self.V.data_ptr(),
                       self.W.data_ptr() if store_blocks else None, self.g.data_ptr(),
                       self.glob[0].data_ptr(), self.schur.data_ptr(), self.scratch.data_ptr(), self._stream())

    def normal_eq_response(self, intr_idx, pose_idx):
        """The single-pass evaluation (no block stores, lambda = 0) that also leaves every view's first-order
        pose response in ``self.Y`` (camcal_normal_eq_fused_response; Y shares the W buffer)."""
        self._call("camcal_normal_eq_fused_response", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), self.Y.data_ptr(), self.glob[0].data_ptr(),
                   self.schur.data_ptr(), self.scratch.data_ptr(), self._stream())

    def pose_predict(self, pose_idx, scale):
        """Move the poses along the step ``intr += scale * dK`` the last reduced solve took (camcal_pose_predict)."""
        self._call("camcal_pose_predict", self.model, self._tp(), self.Y.data_ptr(), self.dK.data_ptr(),
                   float(scale), self.iters.data_ptr(), 20, self.pose[pose_idx].data_ptr(), self._stream())

    def refine_normal_eq_partial(self, intr_idx, pose_idx, max_iter=20, predict=False, scale=0.0, first=False,
                                 respond=True):
        """Small sets, fisheye: the re-estimation (from the predicted poses with ``predict``; skipped with
        ``max_iter`` 0) and the evaluation at the refined poses in one launch; the evaluation stays in
        ``self.scratch`` as partial records for ``reduce_solve`` (camcal_refine_normal_eq_partial)."""
        self._call("camcal_refine_normal_eq_partial", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), int(max_iter), 1e-10, self.iters.data_ptr(),
                   self.Y.data_ptr() if respond else None, 1 if predict else 0, self.dK.data_ptr(), float(scale), 20,
                   1 if first else 0, self.scratch.data_ptr(), self._stream())

    def reduce_solve(self, mask, lam, scale, intr_in, intr_out, solve=True):
        """Sum of those partial records into ``self.rec`` and, with ``solve``, the reduced solve on it in the
        same launch (camcal_reduce_solve)."""
        self._call("camcal_reduce_solve", self.model, self._tp(), self.scratch.data_ptr(), self.rec.data_ptr(),
                   1 if solve else 0, self._mask_bytes(mask), float(lam), float(scale), self.intr[intr_in].data_ptr(),
                   self.intr[intr_out].data_ptr(), self.dK.data_ptr(), self.stats.data_ptr(), None, 0.0, self._stream())

    # -- the folded pinhole LM step of small sets (camcal_pinhole_lm runs these) ---------
    def small_set_begin(self):
        self._call("camcal_small_set_begin", self.model, self._tp(), self.scratch.data_ptr(), self._stream())

    def cross_schur_solve(self, intr_in, pose_idx, lam, mask, scale, intr_out):
        """camcal_cross_schur + camcal_solve_reduced(glob[0], schur) in two launches."""
        self._call("camcal_cross_schur_solve", self.model, self._tp(), self.intr[intr_in].data_ptr(),
                   self.pose[pose_idx].data_ptr(), self.V.data_ptr(), self.g.data_ptr(), float(lam), self.W.data_ptr(),
                   self.glob[0].data_ptr(), self.schur.data_ptr(), self._mask_bytes(mask), float(scale),
                   self.intr[intr_out].data_ptr(), self.dK.data_ptr(), self.stats.data_ptr(), self.scratch.data_ptr(),
                   self._stream())

    def schur_solve(self, lam, mask, scale, intr_in, intr_out):
        """camcal_schur (stored blocks) + camcal_solve_reduced in two launches."""
        self._call("camcal_schur_solve", self.model, self._tp(), self.V.data_ptr(), self.W.data_ptr(), self.g.data_ptr(),
                   float(lam), self.glob[0].data_ptr(), self.schur.data_ptr(), self._mask_bytes(mask), float(scale),
                   self.intr[intr_in].data_ptr(), self.intr[intr_out].data_ptr(), self.dK.data_ptr(),
                   self.stats.data_ptr(), self.scratch.data_ptr(), self._stream())

    def backsub_intr_normal(self, lam, pose_in, pose_out, intr_trial):
        """camcal_backsub + camcal_intr_normal (with blocks, into the other V / g buffers and glob[1]) in two launches."""
        self._call("camcal_backsub_intr_normal", self.model, self._tp(), self.V.data_ptr(), self.W.data_ptr(),
                   self.g.data_ptr(), float(lam), self.dK.data_ptr(), self.pose[pose_in].data_ptr(),
                   self.pose[pose_out].data_ptr(), self.intr[intr_trial].data_ptr(), self.glob[1].data_ptr(),
                   self.Vs[1].data_ptr(), self.gs[1].data_ptr(), self.norms.data_ptr(), self.stats.data_ptr(), None, 0.0,
                   self.scratch.data_ptr(), self._stream())

    def reproj_err(self, intr_idx, pose_idx, f32_round=True):
        """Per-view ||proj - corners||_2 / n (caller's view order) and their mean."""
        out = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._call("camcal_reproj_err", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[pose_idx].data_ptr(), 1 if f32_round else 0, self.view_cost.data_ptr(),
                   out.data_ptr(), self.scratch.data_ptr(), self._stream())
        o = out.cpu().numpy()
        return self.slots_to_views(self.view_cost), float(o[0] / max(o[1], 1.0))

    def time_stage(self, stage, reps, lam=0.0):
        """Average milliseconds per launch of one stage, HIP events on the
        current stream (0 evaluation A, 1 view_blocks, 2 schur from stored blocks,
        3 normal_eq = A + B, 4 fused single-pass normal_eq, 5 evaluation B)."""
        ms = C.c_float(0.0)
        self._call("camcal_time_stage", int(stage), int(reps), self.model, self._tp(),
                   self.intr[0].data_ptr(), self.pose[0].data_ptr(), float(lam), self.V.data_ptr(),
                   self.W.data_ptr(), self.g.data_ptr(), self.glob[0].data_ptr(),
                   self.schur.data_ptr(), self.scratch.data_ptr(), self._stream(), C.byref(ms))
        return float(ms.value)


class IncrementalEngine(Engine):
    """Engine for the growing-list caller (the reference calls the solver once per captured
    frame with a longer list, intrinsicCalib.py:538-564 -> :86-94,129-135): views are
    APPENDED, only the tiles they touch are packed, the solver buffers and the poses of the
    views already held stay on the device (the next solve warm-starts from them).

    Views keep their arrival order (slot = view index): tile t holds views 64 t .. 64 t + 63,
    only the last tile is partly filled and only it can grow, its rows being the end of the
    packed buffer.  Capacity (tiles, packed floats) doubles when exceeded; a rebuild re-packs
    from the host copies and carries the poses over."""

    def __init__(self, model, device, lib=None, capacity_views=256, lanes_per_view=None):
        self._setup(model, device, lib)
        self._lpv = int(self.LANES_PER_VIEW if lanes_per_view is None else lanes_per_view)
        self._views = []                       # host copies (obj (n,3) f32, img (n,2) f32)
        self._alloc(max(int(capacity_views), LANES), 0)

    # -- storage -----------------------------------------------------------------
    def _alloc(self, capacity_views, capacity_floats):
        T = (capacity_views + LANES - 1) // LANES
        S = T * LANES
        dev = self.device
        self.S, self._T = S, T
        self.n_views, self.n_obs = 0, 0
        self._lane_n = np.zeros(S, dtype=np.int32)
        self._tile_nmax = np.zeros(T, dtype=np.int32)
        self._tile_off = np.zeros(T, dtype=np.int64)
        self._floats = 0                       # packed floats in use
        self._cap_floats = max(int(capacity_floats), 5 * LANES * 128)
        self.t_pts = torch.zeros(self._cap_floats, dtype=torch.float32, device=dev)
        self.t_tile_off = torch.zeros(T, dtype=torch.int64, device=dev)
        self.t_tile_nmax = torch.zeros(T, dtype=torch.int32, device=dev)
        self.t_lane_n = torch.zeros(S, dtype=torch.int32, device=dev)
        self.tiles = _lib.Tiles(self.t_pts.data_ptr(), self.t_tile_off.data_ptr(), self.t_tile_nmax.data_ptr(),
                                self.t_lane_n.data_ptr(), T, S, 1, self._lpv)
        sv = np.full(S, -1, dtype=np.int32)
        self.layout = dict(slot_view=sv, lane_n=self._lane_n, tile_nmax=self._tile_nmax, tile_off=self._tile_off,
                           n_tiles=T, n_slots=S, n_views=0, total_floats=0)
        self._alloc_solver_buffers(S, T)

    def _grow(self, need_views, need_floats):
        """Double the capacity, re-pack every view from the host copies, keep the poses."""
        views, old_n = self._views, self.n_views
        old_pose = self.pose[0][:, :old_n].clone() if old_n else None
        old_intr = self.intr[0].clone()
        cap_v = self.S
        while cap_v < need_views:
            cap_v *= 2
        cap_f = self._cap_floats
        while cap_f < need_floats:
            cap_f *= 2
        self._views = []
        self._alloc(cap_v, cap_f)
        if views:
            self.append([v[0] for v in views], [v[1] for v in views])
        if old_pose is not None:
            self.pose[0][:, :old_n] = old_pose
        self.intr[0].copy_(old_intr)

    # -- growing -------------------------------------------------------------------
    def append(self, objs, imgs):
        """Add views (lists of (n,3) / (n,2) float32 arrays, Z = 0).  Returns the slots they took."""
        k = len(objs)
        if k == 0:
            return np.zeros(0, dtype=np.int64)
        counts = np.array([o.shape[0] for o in objs], dtype=np.int32)
        n0, n1 = self.n_views, self.n_views + k
        T0 = n0 // LANES                                     # first tile touched
        T1 = (n1 + LANES - 1) // LANES                       # one past the last tile in use
        if n1 > self.S:
            need_f = self._floats + int(counts.astype(np.int64).max()) * 5 * LANES * ((k + LANES - 1) // LANES + 1)
            self._grow(n1, need_f)
            return self.append(objs, imgs)
        # sizes after the append: the touched tiles are re-laid out from tile T0 on
        lane_n = self._lane_n.copy()
        lane_n[n0:n1] = counts
        off0 = int(self._tile_off[T0])
        nmax = lane_n[T0 * LANES:T1 * LANES].reshape(-1, LANES).max(axis=1).astype(np.int32)
        sizes = nmax.astype(np.int64) * (5 * LANES)
        end = off0 + int(sizes.sum())
        if end > self._cap_floats:
            self._grow(n1, end)
            return self.append(objs, imgs)
        offs = off0 + np.concatenate([[0], np.cumsum(sizes)[:-1]])
        self._lane_n = lane_n
        self._tile_nmax[T0:T1] = nmax
        self._tile_off[T0:T1] = offs
        if T1 < self._T:
            self._tile_off[T1:] = end                        # empty tiles point at valid memory
        self._floats = end
        for o, i in zip(objs, imgs):
            self._views.append((o, i))
        self.n_views = n1
        self.n_obs += int(counts.sum())
        self.layout["slot_view"][n0:n1] = np.arange(n0, n1, dtype=np.int32)
        self.layout.update(lane_n=self._lane_n, n_views=n1, total_floats=end)
        # device tables of the touched range
        dev = self.device
        self.t_lane_n[T0 * LANES:T1 * LANES].copy_(torch.from_numpy(lane_n[T0 * LANES:T1 * LANES]).to(dev))
        self.t_tile_nmax[T0:T1].copy_(torch.from_numpy(self._tile_nmax[T0:T1]).to(dev))
        self.t_tile_off[T0:].copy_(torch.from_numpy(self._tile_off[T0:]).to(dev))
        # pack the touched tiles (the partly filled one again in full: its rows may have grown)
        v0 = T0 * LANES
        sub = self._views[v0:n1]
        vp = np.zeros(len(sub) + 1, dtype=np.int64)
        vp[1:] = np.cumsum([o.shape[0] for o, _ in sub])
        t_obj = torch.from_numpy(np.concatenate([o for o, _ in sub])).to(dev)
        t_img = torch.from_numpy(np.concatenate([i for _, i in sub])).to(dev)
        t_vp = torch.from_numpy(vp).to(dev)
        nsub = (T1 - T0) * LANES
        sv = np.full(nsub, -1, dtype=np.int32)
        sv[:len(sub)] = np.arange(len(sub), dtype=np.int32)
        t_sv = torch.from_numpy(sv).to(dev)
        part = _lib.Tiles(self.t_pts.data_ptr(), self.t_tile_off[T0:].data_ptr(), self.t_tile_nmax[T0:].data_ptr(),
                          self.t_lane_n[T0 * LANES:].data_ptr(), T1 - T0, nsub, 1, self._lpv)
        self._call("camcal_pack_tiles", t_obj.data_ptr(), t_img.data_ptr(), t_vp.data_ptr(), t_sv.data_ptr(),
                   C.byref(part), self.t_pts.data_ptr(), self._stream())
        return np.arange(n0, n1, dtype=np.int64)

    def init_new_poses(self, slots, intr_idx=0):
        """Pose initialisation of the appended views only: the launch covers every slot (small sets: tens of
        microseconds) into the spare pose buffer, the new columns are copied over; the views already held keep
        their poses as the warm start of the next refinement."""
        if len(slots) == 0:
            return
        self._call("camcal_pose_init", self.model, self._tp(), self.intr[intr_idx].data_ptr(),
                   self.pose[1].data_ptr(), self.status.data_ptr(), self._stream())
        lo, hi = int(slots[0]), int(slots[-1]) + 1
        self.pose[0][:, lo:hi] = self.pose[1][:, lo:hi]
