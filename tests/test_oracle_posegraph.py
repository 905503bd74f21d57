"""CPU checks of the pose-graph oracle (oracle/oracle_posegraph.c, contract G1-G6): known answers, a numpy restatement of
the edge error with finite-difference Jacobians, and convergence on loops with drift."""
import numpy as np
import pytest

import oracle_lib as ol
import synth_pg as sp


def np_error(Xi, Xj, Z, oracle):
    E = sp.mul(sp.mul(sp.inv(Z), sp.inv(Xi)), Xj)
    return np.concatenate([E.reshape(3, 4)[:, 3], oracle.so3_log(E)])


def test_so3_log_known_answers(oracle):
    for w in ([0, 0, 0], [0.3, 0, 0], [0, -1.2, 0], [0.1, 0.2, -0.3], [1e-9, 0, 2e-9], [2.0, 1.0, -1.5]):
        assert np.allclose(oracle.so3_log(sp.pose(sp.rotvec(w), [1, 2, 3])), w, atol=1e-12)


def test_cost_and_zero_residual_graph(oracle):
    g = sp.make_loop(1, n=12, noise_t=0, noise_r=0)
    assert oracle.pg_cost(g["gt"], g["edge_ij"], g["edge_z"]) < 1e-20
    X, r = oracle.pg_optimize(g["gt"], g["edge_ij"], g["edge_z"])
    assert r.stop_reason in (1, 4) and r.cost_final < 1e-20 and np.abs(X - g["gt"]).max() < 1e-9
    e = g["edge_ij"][3]
    moved = g["gt"].copy()
    moved[5] = sp.mul(sp.pose(np.eye(3), [0.1, 0, 0]), moved[5])                 # shift node 5 by 10 cm: its two chain edges see it
    c = oracle.pg_cost(moved, g["edge_ij"], g["edge_z"])
    exp = sum(np_error(moved[i], moved[j], z, oracle) @ np_error(moved[i], moved[j], z, oracle) for (i, j), z in zip(g["edge_ij"], g["edge_z"]))
    assert abs(c - exp) < 1e-12 and 0.015 < c < 0.025 and e[0] == 3


def test_gauss_newton_model_is_first_order_exact(oracle):
    """One LM attempt with a tiny lambda on a slightly perturbed graph lands (almost) on the optimum: the Jacobians of G4 are the
    derivatives of the error of G2 under the update of G3."""
    g = sp.make_loop(2, n=10, noise_t=0, noise_r=0, n_closures=2)
    rng = np.random.default_rng(3)
    X0 = g["gt"].copy()
    for k in range(1, 10):
        X0[k] = sp.mul(sp.pose(sp.rotvec(rng.normal(0, 1e-3, 3)), rng.normal(0, 1e-3, 3)), X0[k])
    c0 = oracle.pg_cost(X0, g["edge_ij"], g["edge_z"])
    X, r = oracle.pg_optimize(X0, g["edge_ij"], g["edge_z"], prm=ol.PgParams(max_iters=1, tau=1e-12))
    assert r.accepted == 1 and r.cost_final < 1e-4 * c0          # what is left is second order in the 1e-3 perturbation


@pytest.mark.parametrize("seed,n,fwd", [(5, 60, True), (6, 90, False), (7, 25, False)])
def test_loop_with_drift_closes(oracle, seed, n, fwd):
    g = sp.make_loop(seed, n=n, edge_forward=fwd)
    c0 = oracle.pg_cost(g["init"], g["edge_ij"], g["edge_z"])
    X, r = oracle.pg_optimize(g["init"], g["edge_ij"], g["edge_z"])
    assert r.cost_init == c0 and r.stop_reason == 4 and r.cost_final < 0.05 * c0 and r.accepted >= 2
    assert np.array_equal(X[0], g["init"][0])                                     # the root stays where it is
    def end_error(P):
        return np.linalg.norm(P[-1].reshape(3, 4)[:, 3] - g["gt"][-1].reshape(3, 4)[:, 3])
    assert end_error(X) < 0.35 * end_error(g["init"])                             # the loop closure pulls the drifted end back
    # optimum: the gradient vanishes -- a second run from the result does not move
    X2, r2 = oracle.pg_optimize(X, g["edge_ij"], g["edge_z"])
    assert abs(r2.cost_final - r.cost_final) <= 1e-6 * r.cost_final and np.abs(X2 - X).max() < 1e-4


def test_degenerate_inputs(oracle):
    X, r = oracle.pg_optimize(np.array([sp.pose(np.eye(3), [0, 0, 0])]), np.zeros((0, 2), np.int32), np.zeros((0, 12)))
    assert r.stop_reason == 3 and r.iters == 0
    g = sp.make_loop(4, n=8)
    X, r = oracle.pg_optimize(g["init"], g["edge_ij"], g["edge_z"], prm=ol.PgParams(max_iters=2))
    assert r.iters == 2 and r.stop_reason == 0


def test_golden_fixture(oracle):
    """tests/golden/posegraph_case0.npz (made by tests/golden/make_golden.py from this oracle): a 48-keyframe loop."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "posegraph_case0.npz"))
    X, r = oracle.pg_optimize(g["init"], g["edge_ij"], g["edge_z"])
    assert [r.iters, r.accepted, r.stop_reason, r.pcg_iters_total] == g["exp_counts"].tolist()
    assert np.array_equal(X, g["exp_poses"]) and [r.cost_init, r.cost_final] == g["exp_cost"].tolist()
