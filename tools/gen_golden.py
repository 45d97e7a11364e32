"""Closed-form element matrices of the P2/P1 Taylor-Hood pair (exact rationals, SymPy) -> tests/golden/.

These are the only "golden vectors" this path can have: the reference ships no tests or fixtures,
so the oracle is pinned against exact integrals of the weak forms the reference assembles
(reference Navier-Stokes/src/NavierStokes3D.cpp:246-264,456-459):
   mass      int N_a N_b                      stiffness  int grad N_a . grad N_b
   div       int psi_v d_c N_a                pmass      int psi_v psi_u
   conv      int (w . grad N_b) N_a           temam      int (div w) N_a N_b        with w in (P2)^dim prescribed
on the reference simplex and on one affine image with rational vertices.  Local ordering = deal.II's
(vertices, then lines {01,12,20} / {01,12,20,03,13,23}).
Run:  python tools/gen_golden.py   (writes tests/golden/p2p1_element_dim{2,3}.json)
"""
import json
import os
from fractions import Fraction

import sympy as sp

LINES = {2: [(0, 1), (1, 2), (2, 0)], 3: [(0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3)]}


def simplex_integral(expr, xs):
    """Exact integral of a polynomial over the reference simplex."""
    dim = len(xs)
    e = sp.expand(expr)
    if dim == 2:
        x, y = xs
        return sp.integrate(sp.integrate(e, (y, 0, 1 - x)), (x, 0, 1))
    x, y, z = xs
    return sp.integrate(sp.integrate(sp.integrate(e, (z, 0, 1 - x - y)), (y, 0, 1 - x)), (x, 0, 1))


def build(dim, verts, wcoef):
    xs = sp.symbols("x y z")[:dim]
    lam = [1 - sum(xs)] + list(xs)
    N1 = lam[:]
    N2 = [l * (2 * l - 1) for l in lam] + [4 * lam[a] * lam[b] for a, b in LINES[dim]]
    # affine map: X = V0 + J xhat
    V = [sp.Matrix([sp.Rational(c.numerator, c.denominator) for c in v]) for v in verts]
    J = sp.Matrix.hstack(*[V[k + 1] - V[0] for k in range(dim)])
    Jinv = J.inv()
    detJ = sp.Abs(J.det())

    def grad(f):  # physical gradient: J^{-T} grad_hat
        gh = sp.Matrix([sp.diff(f, x) for x in xs])
        return Jinv.T * gh

    G2 = [grad(f) for f in N2]
    w = [sum(wcoef[a][c] * N2[a] for a in range(len(N2))) for c in range(dim)]
    divw = sum(sum(wcoef[a][c] * G2[a][c] for a in range(len(N2))) for c in range(dim))
    n2, n1 = len(N2), len(N1)
    I = lambda e: simplex_integral(e, xs) * detJ
    out = {
        "dim": dim, "vertices": [[str(c) for c in v] for v in verts], "w": [[str(c) for c in r] for r in wcoef],
        "mass": [[str(I(N2[a] * N2[b])) for b in range(n2)] for a in range(n2)],
        "stiffness": [[str(I((G2[a].T * G2[b])[0])) for b in range(n2)] for a in range(n2)],
        "div": [[[str(I(N1[v] * G2[a][c])) for c in range(dim)] for v in range(n1)] for a in range(n2)],
        "pmass": [[str(I(N1[v] * N1[u])) for u in range(n1)] for v in range(n1)],
        "conv": [[str(I(sum(w[c] * G2[b][c] for c in range(dim)) * N2[a])) for b in range(n2)] for a in range(n2)],
        "temam": [[str(I(divw * N2[a] * N2[b])) for b in range(n2)] for a in range(n2)],
    }
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    F = Fraction
    for dim in (2, 3):
        n2 = 6 if dim == 2 else 10
        # prescribed quadratic field: nodal coefficients (small rationals, not symmetric)
        wcoef = [[F((3 * a + 2 * c + 1) % 7 - 3, 1 + (a + c) % 3) for c in range(dim)] for a in range(n2)]
        ref = [[F(0)] * dim] + [[F(1) if k == d else F(0) for k in range(dim)] for d in range(dim)]
        if dim == 2:
            aff = [[F(1, 2), F(1, 3)], [F(2), F(1)], [F(1), F(5, 2)]]
        else:
            aff = [[F(1, 2), F(1, 3), F(0)], [F(2), F(1), F(1, 4)], [F(1), F(5, 2), F(1, 2)], [F(3, 4), F(1), F(2)]]
        cases = {"reference": build(dim, ref, wcoef), "affine": build(dim, aff, wcoef)}
        path = os.path.join(here, "tests", "golden", "p2p1_element_dim%d.json" % dim)
        with open(path, "w") as f:
            json.dump(cases, f, indent=0)
        print("wrote", path)


if __name__ == "__main__":
    main()
