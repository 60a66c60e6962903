"""Oracle pinning, part 1: the camera models' analytic Jacobians against central
differences, Rodrigues round trips, undistortion round trips."""
import numpy as np
import pytest

from oracle import cvmodel as M

RNG = np.random.default_rng(7)
PIN_K = np.array([1365.77, 1367.35, 959.51, 556.16])
PIN_D = np.array([-0.40958, 0.23505, -1.70e-4, -9.53e-4, -0.08694])
FE_K = np.array([1066.58, 1066.31, 1478.65, 791.19])
FE_D = np.array([-0.02334, 0.007086, -0.005574, 0.000994])


def _points(n=12):
    X = np.zeros((n, 3))
    X[:, :2] = RNG.uniform(-120, 120, (n, 2))
    return X


def _num_jac(f, p, h):
    J = []
    for i in range(len(p)):
        d = np.zeros_like(p)
        d[i] = h[i]
        J.append((f(p + d) - f(p - d)) / (2 * h[i]))
    return np.stack(J, axis=-1)


@pytest.mark.parametrize("model", ["pinhole", "fisheye"])
def test_projection_jacobian_matches_central_differences(model):
    X = _points()
    r = np.array([0.21, -0.35, 1.3])
    t = np.array([40.0, -25.0, 900.0])
    if model == "pinhole":
        intr = np.concatenate([PIN_K, PIN_D])
        f = lambda p: M.project_pinhole(X, p[9:12], p[12:15], p[:4], p[4:9])
        _, J = M.project_pinhole(X, r, t, PIN_K, PIN_D, jac=True)
    else:
        intr = np.concatenate([FE_K, FE_D])
        f = lambda p: M.project_fisheye(X, p[8:11], p[11:14], p[:4], p[4:8])
        _, J, _ = M.project_fisheye(X, r, t, FE_K, FE_D, jac=True)
    p = np.concatenate([intr, r, t])
    h = 1e-6 * np.maximum(np.abs(p), 1e-2)
    ni = 9 if model == "pinhole" else 8
    h[4:ni] = 1e-2      # the projection is linear in the distortion coefficients
    Jn = _num_jac(f, p, h)
    scale = np.abs(Jn).max(axis=(0, 1))
    assert np.all(np.abs(J - Jn).max(axis=(0, 1)) <= 1e-7 * scale + 1e-9)


def test_rodrigues_derivative_and_inverse():
    for r in (np.array([0.3, -0.2, 0.9]), np.array([2.5, 1.0, -1.2]), np.array([1e-9, 0, 0]), np.zeros(3)):
        R, J = M.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        if np.linalg.norm(r) > 1e-6:
            Jn = _num_jac(lambda p: M.rodrigues(p)[0].ravel(), r, np.full(3, 1e-6))
            assert np.allclose(J.T, Jn, atol=1e-8)
            assert np.allclose(M.rodrigues_inv(R), r, atol=1e-12)


def test_left_jacobian_identity():
    """d(R X)/dr = -[R X]x Jl(r): the identity the HIP kernels rely on, checked
    against cvRodrigues2's explicit 3x9 derivative."""
    for r in (np.array([0.3, -0.2, 0.9]), np.array([2.9, 0.4, -0.7]), np.array([0.01, 0.02, -0.005])):
        R, J = M.rodrigues(r)
        th = np.linalg.norm(r)
        K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        Jl = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
        X = np.array([13.0, -7.0, 2.0])
        Y = R @ X
        Yx = np.array([[0, -Y[2], Y[1]], [Y[2], 0, -Y[0]], [-Y[1], Y[0], 0]])
        dY = np.stack([J[i].reshape(3, 3) @ X for i in range(3)], axis=1)
        assert np.allclose(dY, -Yx @ Jl, rtol=1e-9, atol=1e-9)


def test_undistort_round_trips():
    X = _points(30)
    r, t = np.array([0.1, 0.2, -0.4]), np.array([10.0, 20.0, 700.0])
    R, _ = M.rodrigues(r)
    Y = X @ R.T + t
    xn = Y[:, :2] / Y[:, 2:]
    uv = M.project_fisheye(X, r, t, FE_K, FE_D)
    assert np.allclose(M.undistort_fisheye(uv, FE_K, FE_D), xn, atol=1e-7)
    uv = M.project_pinhole(X, r, t, PIN_K, PIN_D)
    assert np.allclose(M.undistort_pinhole(uv, PIN_K, PIN_D, iters=40), xn, atol=1e-9)
    assert np.allclose(M.undistort_pinhole(uv, PIN_K, PIN_D), xn, atol=5e-3)   # cv2's 5 sweeps
