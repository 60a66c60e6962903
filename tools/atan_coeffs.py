#!/usr/bin/env python3
"""Derives the coefficients of camcal_math.h::atan_pos (needs mpmath):
Chebyshev-node interpolation of (atan(sqrt z)/sqrt z - 1)/z on [0, tan(pi/8)^2]."""
import mpmath as mp
mp.mp.dps = 60
z1 = mp.tan(mp.pi / 8) ** 2
f = lambda z: mp.mpf(-1) / 3 if z == 0 else (mp.atan(mp.sqrt(z)) / mp.sqrt(z) - 1) / z
n = 11
nodes = [(z1 / 2) * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
A = mp.matrix(n, n); b = mp.matrix(n, 1)
for i, z in enumerate(nodes):
    for j in range(n):
        A[i, j] = z ** j
    b[i] = f(z)
c = mp.lu_solve(A, b)
print([mp.nstr(c[j], 20) for j in range(n)])
