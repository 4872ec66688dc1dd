"""iLQG backward pass: the reference's golden vectors (mjpc/test/ilqg_planner/backward_pass_test.cc:29-140,
LQR problem of mjpc/test/lqr.cc:24-105) pin the oracle's Riccati step + box-QP; plus properties."""
import numpy as np

from oracle import pyoracle


def lqr_problem(T=3):
    n, m = 2, 1
    A = np.tile(np.array([[1.0, 1.0], [0.0, 1.0]]), (T, 1, 1))      # lqr.cc:32-37
    B = np.tile(np.array([[0.0], [1.0]]), (T, 1, 1))                  # lqr.cc:40-43
    u = np.full((T - 1, m), 0.5)                                      # backward_pass_test.cc:47-48
    x = np.zeros((T, n))
    for t in range(T - 1):                                            # RolloutReturn, lqr.cc:88-103
        x[t + 1] = [x[t, 0] + x[t, 1], x[t, 1] + u[t, 0]]
    cx, cxx = x.copy(), np.tile(np.eye(n), (T, 1, 1))                 # cost 0.5 x'x + 0.5 u'u
    cu, cuu, cxu = np.zeros((T, m)), np.zeros((T, m, m)), np.zeros((T, n, m))
    cu[:T - 1], cuu[:T - 1] = u, 1.0
    return n, m, A, B, cx, cu, cxx, cxu, cuu, np.vstack([u, u[-1:]]), x


def test_backward_pass_golden_vectors():
    T = 3
    n, m, A, B, cx, cu, cxx, cxu, cuu, actions, x = lqr_problem(T)
    out = pyoracle.riccati(n, m, T, 0.0, 0, 1, A, B, cx, cu, cxx, cxu, cuu, actions, [-1.0, 1.0])
    assert out["ok"]
    Vx = np.array([[0.0, 0.0], [0.5, 1.25], [0.5, 1.0]])                                    # :102
    Vxx = np.array([[[2.71428571, 2.0], [2.0, 4.0]], [[2.0, 1.0], [1.0, 2.5]], [[1.0, 0.0], [0.0, 1.0]]])  # :104-105
    K = np.array([[[-0.285714285, -1.0]], [[0.0, -0.5]]])                                   # :107
    du = np.array([[-0.5], [-0.75]])                                                        # :109
    assert np.allclose(out["Vx"], Vx, atol=1e-5) and np.allclose(out["Vxx"], Vxx, atol=1e-5)
    assert np.allclose(out["K"][:T - 1], K, atol=1e-5) and np.allclose(out["du"][:T - 1], du, atol=1e-5)
    # the unconstrained branch (settings.action_limits = 0) agrees on this interior problem
    out2 = pyoracle.riccati(n, m, T, 0.0, 0, 0, A, B, cx, cu, cxx, cxu, cuu, actions, [-1.0, 1.0])
    for k in ("Vx", "Vxx", "K", "du", "dV"):
        assert np.allclose(out[k], out2[k], atol=1e-12)
    assert out["dV"][0] < 0 < out["dV"][1]


def test_riccati_matches_discrete_lqr_fixed_point():
    """Long horizon, zero nominal: K converges to the DARE gain of (A, B, Q=I, R=1)."""
    import scipy.linalg
    T = 60
    n, m, A, B, cx, cu, cxx, cxu, cuu, actions, x = lqr_problem(T)
    cx[:] = 0; cu[:] = 0; actions[:] = 0
    out = pyoracle.riccati(n, m, T, 0.0, 0, 0, A, B, cx, cu, cxx, cxu, cuu, actions, [-1e9, 1e9])
    P = scipy.linalg.solve_discrete_are(A[0], B[0], np.eye(2), np.eye(1))
    Kinf = -np.linalg.solve(np.eye(1) + B[0].T @ P @ B[0], B[0].T @ P @ A[0])
    assert np.allclose(out["K"][0], Kinf, atol=1e-9) and np.allclose(out["Vxx"][0], P, atol=1e-8)


def test_boxqp_against_bruteforce():
    rng = np.random.default_rng(0)
    for trial in range(50):
        n = int(rng.integers(1, 5))
        M = rng.normal(size=(n, n))
        H = M @ M.T + 0.1 * np.eye(n)
        g = rng.normal(size=n) * 2
        lo, hi = -rng.uniform(0.1, 1, n), rng.uniform(0.1, 1, n)
        nfree, x, idx = pyoracle.boxqp(H, g, lo, hi)
        assert nfree >= 0 and np.all(x >= lo - 1e-12) and np.all(x <= hi + 1e-12)
        # KKT: free coordinates have zero gradient, clamped ones push outward
        grad = H @ x + g
        for i in range(n):
            if lo[i] + 1e-9 < x[i] < hi[i] - 1e-9:
                assert abs(grad[i]) < 1e-7
            elif x[i] <= lo[i] + 1e-9:
                assert grad[i] > -1e-7
            else:
                assert grad[i] < 1e-7


def test_clamped_controls_get_zero_gain():
    """backward_pass.cc:176-192: rows of K for clamped controls stay zero."""
    T = 3
    n, m, A, B, cx, cu, cxx, cxu, cuu, actions, x = lqr_problem(T)
    actions[:] = 1.0
    cu[:T - 1] = -5.0          # strong pull upward, but u is already at the +1 limit
    out = pyoracle.riccati(n, m, T, 0.0, 0, 1, A, B, cx, cu, cxx, cxu, cuu, actions, [-1.0, 1.0])
    assert out["ok"] and np.all(out["du"][:T - 1] == 0.0) and np.all(out["K"][:T - 1] == 0.0)


def test_regularisation_types_and_failure():
    T = 4
    n, m, A, B, cx, cu, cxx, cxu, cuu, actions, x = lqr_problem(T)
    base = pyoracle.riccati(n, m, T, 0.0, 0, 0, A, B, cx, cu, cxx, cxu, cuu, actions, [-1, 1])
    for reg in (0, 1, 2):
        out = pyoracle.riccati(n, m, T, 10.0, reg, 0, A, B, cx, cu, cxx, cxu, cuu, actions, [-1, 1])
        assert out["ok"] and np.all(np.abs(out["du"][:T - 1]) <= np.abs(base["du"][:T - 1]) + 1e-12)
    cuu[:] = -10.0             # indefinite Quu -> failure is reported, not hidden
    assert not pyoracle.riccati(n, m, T, 0.0, 0, 0, A, B, cx, cu, cxx, cxu, cuu, actions, [-1, 1])["ok"]
