"""Generate / verify simplex quadrature constants (degree-5, positive weights).

2D: Radon 7-point (closed form). 3D: 14-point rule with orbits
(a,a,a,1-3a) x2 and (b,b,1/2-b,1/2-b); constants polished with mpmath Newton
against the exact monomial moments of the unit tetrahedron.
Prints C initialisers used by host/tables.cpp.
"""
import itertools
from mpmath import mp, mpf, findroot, factorial, sqrt
mp.dps = 40

def tet_moment(a, b, c):
    return factorial(a) * factorial(b) * factorial(c) / factorial(a + b + c + 3)

def orbit4(a):
    r = 1 - 3 * a
    return [(a, a, a), (r, a, a), (a, r, a), (a, a, r)]

def orbit6(b):
    c = mpf(1) / 2 - b
    pts = set()
    for perm in itertools.permutations([b, b, c, c]):
        pts.add(perm[:3])
    return sorted(pts)

def rule(x):
    a1, w1, a2, w2, b, w3 = x
    pts = [(p, w1) for p in orbit4(a1)] + [(p, w2) for p in orbit4(a2)] + [(p, w3) for p in orbit6(b)]
    return pts

def integ(x, a, b, c):
    return sum(w * p[0] ** a * p[1] ** b * p[2] ** c for p, w in rule(x))

# symmetric rule: enough to match a set of independent moments
mons = [(0, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (2, 2, 0), (5, 0, 0)]
def F(*x):
    return [integ(x, *m) - tet_moment(*m) for m in mons]

x0 = [mpf('0.31088591926330060980'), mpf('0.11268792571801585080') / 6,
      mpf('0.092735250310891226402'), mpf('0.073493043116361949544') / 6,
      mpf('0.045503704125649649492'), mpf('0.042546020777081466438') / 6]
x = findroot(F, x0, tol=1e-35, maxsteps=50)
x = [v for v in x]
# verify all monomials up to degree 5
worst = 0
for a in range(6):
    for b in range(6 - a):
        for c in range(6 - a - b):
            worst = max(worst, abs(integ(x, a, b, c) - tet_moment(a, b, c)))
print("// worst degree<=5 moment error:", mp.nstr(worst, 5))
names = ["a1", "w1", "a2", "w2", "b", "w3"]
for n, v in zip(names, x):
    print("static const double T14_%s = %s;" % (n, mp.nstr(v, 20)))
s15 = sqrt(15)
print("// Radon 7: a1=(6-sqrt15)/21 w1=(155-sqrt15)/2400 a2=(6+sqrt15)/21 w2=(155+sqrt15)/2400 wc=9/80")
for n, v in [("a1", (6 - s15) / 21), ("w1", (155 - s15) / 2400), ("a2", (6 + s15) / 21), ("w2", (155 + s15) / 2400), ("wc", mpf(9) / 80)]:
    print("static const double R7_%s = %s;" % (n, mp.nstr(v, 20)))
