"""The allocation-free entry point camcal_normal_eq_fused_static (include/camcal.h; no
reference counterpart -- it exists for callers that cannot allocate) against the entry
point it is a variant of, camcal_normal_eq_fused, through the C ABI: two calls in a row, the
second with a smaller input, bit for bit (same kernels, same order of the sums), under the
one-lane-per-view mapping and the grouped mapping of small sets; sizes beyond the limit are
rejected."""
import ctypes as C

import numpy as np
import pytest
import torch

from cameracalibration_amd import _lib, synth
from cameracalibration_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _both(E, lam):
    lib = E.lib
    E.rec.fill_(float("nan"))
    E.normal_eq(0, 0, lam, store_blocks=False, fused=True)          # caller-provided scratch
    want = E.rec.cpu().numpy().copy()
    E.rec.fill_(float("nan"))
    rc = lib.camcal_normal_eq_fused_static(E.model, C.byref(E.tiles), E.intr[0].data_ptr(), E.pose[0].data_ptr(),
                                           float(lam), E.glob[0].data_ptr(), E.schur.data_ptr(), E._stream())
    assert rc == 0, rc
    got = E.rec.cpu().numpy().copy()
    return want, got


def test_static_scratch_entry_point_equals_the_workspace_one():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert not lib.is_emulator
    free0 = None
    # first a large set, then smaller ones (stale partials of the earlier call stay in the array)
    for model, name, nv, lam in ((1, "C5", 20000, 0.0), (1, "C3", 3000, 1e-3), (0, "C1", 20, 1e-3), (1, "C2", 500, 0.0)):
        P = synth.config(name, n_views=nv)
        E = Engine(model, P.view_ptr, P.obj, P.img, dev)
        E.set_intr(0, P.intr_guess())
        E.set_poses(0, P.poses_true.numpy())
        want, got = _both(E, lam)
        assert np.isfinite(want).all()
        assert np.array_equal(want, got), (name, nv, np.abs(want - got).max())
        if free0 is None:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(dev)[0]
            for _ in range(3):                                       # the call itself allocates nothing
                _both(E, lam)
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info(dev)[0] == free0


def test_static_scratch_entry_point_rejects_sizes_beyond_its_limit():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    P = synth.config("C3", n_views=64)
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    t = _lib.Tiles.from_buffer_copy(E.tiles)
    t.n_tiles, t.n_slots = 16385, 16385 * 64
    rc = lib.camcal_normal_eq_fused_static(1, C.byref(t), E.intr[0].data_ptr(), E.pose[0].data_ptr(), 0.0,
                                           E.glob[0].data_ptr(), E.schur.data_ptr(), E._stream())
    assert rc == -3                                                   # CAMCAL_E_TOOBIG, nothing launched
    rc = lib.camcal_normal_eq_fused_static(7, C.byref(E.tiles), E.intr[0].data_ptr(), E.pose[0].data_ptr(), 0.0,
                                           E.glob[0].data_ptr(), E.schur.data_ptr(), E._stream())
    assert rc == -2


def test_static_scratch_entry_point_replays_from_a_hip_graph():
    """What the entry point is for: a caller under stream capture cannot allocate.  The evaluation is
    captured once into a HIP graph and replayed after the intrinsics and the poses changed in place;
    every replay equals the direct call bit for bit."""
    dev = torch.device("cuda:0")
    lib = _lib.load()
    P = synth.config("C2", n_views=3000)
    E = Engine(1, P.view_ptr, P.obj, P.img, dev)
    E.set_intr(0, P.intr_guess())
    E.set_poses(0, P.poses_true.numpy())

    def static_call(stream):
        rc = lib.camcal_normal_eq_fused_static(E.model, C.byref(E.tiles), E.intr[0].data_ptr(), E.pose[0].data_ptr(), 0.0,
                                               E.glob[0].data_ptr(), E.schur.data_ptr(), C.c_void_p(stream.cuda_stream))
        assert rc == 0, rc

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        static_call(side)                       # first call outside the capture: the library looks its array up once
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_call(torch.cuda.current_stream(dev))
    for k in range(3):
        E.intr[0][:4] *= 1.0 + 1e-3 * (k + 1)   # parameters change in place between replays
        E.pose[0][3] += 0.5
        E.rec.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        got = E.rec.cpu().numpy().copy()
        E.rec.fill_(float("nan"))
        E.normal_eq(0, 0, 0.0, store_blocks=False, fused=True)
        want = E.rec.cpu().numpy().copy()
        assert np.isfinite(got).all() and np.array_equal(got, want)
