"""CPU check of the algebra behind the Gram-Schmidt sweeps that exchange several links at once (csrc/nsx_blas.hip: k_mgs_blk for M
links per grid-wide exchange, mgs_lowsync for a whole sweep with two collectives).  SolverGMRES' add_and_dot chain computes
h_j = v_j . w_j, w_{j+1} = w_j - h_j v_j.  By linearity h_j = v_j . w_{j0} - sum_{j0 <= i < j} (v_i . v_j) h_i for any j0 <= j:
no orthogonality of the basis is needed, so the blocked evaluation must reproduce the chain's coefficients to rounding even for a
basis that has lost orthogonality."""
import numpy as np
import pytest


def chain(w, V):
    w = w.copy()
    h = np.zeros(len(V))
    for j, v in enumerate(V):
        h[j] = v @ w
        w -= h[j] * v
    return h, w


def blocked(w, V, M):
    w = w.copy()
    h = np.zeros(len(V))
    for j0 in range(0, len(V), M):
        blk = V[j0:j0 + M]
        r = np.array([v @ w for v in blk])                      # one exchange: all r_j of the block ...
        G = np.array([[vi @ vj for vi in blk] for vj in blk])    # ... and the pairwise products of its vectors
        for j in range(len(blk)):
            h[j0 + j] = r[j] - sum(G[j, i] * h[j0 + i] for i in range(j))
        for j, v in enumerate(blk):                             # the chain's updates, in the chain's order
            w -= h[j0 + j] * v
    return h, w


@pytest.mark.parametrize("loss", [0.0, 1e-8, 1e-2, 0.3])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 30])
def test_blocked_coefficients_equal_the_chain(loss, M):
    rng = np.random.default_rng(11)
    n, dim = 4000, 13
    Q, _ = np.linalg.qr(rng.standard_normal((n, dim)))
    V = [q.copy() for q in Q.T]
    for j in range(1, dim):                                     # a basis that is NOT orthogonal: every vector leans on its predecessors
        V[j] = V[j] + loss * sum(V[:j]) / j
        V[j] /= np.linalg.norm(V[j])
    w = rng.standard_normal(n)
    h0, w0 = chain(w, V)
    h1, w1 = blocked(w, V, M)
    scale = np.abs(h0).max()
    assert np.abs(h1 - h0).max() < 5e-13 * scale, (loss, M)
    assert np.abs(w1 - w0).max() < 5e-13 * np.abs(w).max()
