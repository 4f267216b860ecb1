"""Pure-Python restatement of the "rmh-rng v1" stream (oracle/rmh_oracle.c
section 1; DESIGN.md "Random stream"): Python integers for xoshiro256++ and
exact rational arithmetic for every fma, so that it shares no code and no
floating-point library with the C oracle or the HIP kernels.  Slow: tests only.
"""
import math
import struct
from fractions import Fraction

import numpy as np

M64 = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


def stream_init(seed, sid):
    x = mix64((mix64((seed + GAMMA) & M64) + sid) & M64)
    s = []
    for _ in range(4):
        x = (x + GAMMA) & M64
        s.append(mix64(x))
    return s


def nxt(s):
    r = (rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
    t = (s[1] << 17) & M64
    s[2] ^= s[0]
    s[3] ^= s[1]
    s[1] ^= s[2]
    s[0] ^= s[3]
    s[2] ^= t
    s[3] = rotl(s[3], 45)
    return r


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def frombits(b):
    return struct.unpack("<d", struct.pack("<Q", b))[0]


LN2_HI, LN2_LO = 6.93147180369123816490e-01, 1.90821492927058770002e-10
LG = [6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
      2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
      1.479819860511658591e-01]


def spec_log(x):
    """positive normal doubles only (all the stream needs)"""
    ix = bits(x)
    k = (ix >> 52) - 1023
    mant = ix & 0x000FFFFFFFFFFFFF
    if mant >= 0x0006A09E667F3BCD:
        m = frombits(mant | 0x3FE0000000000000)
        k += 1
    else:
        m = frombits(mant | 0x3FF0000000000000)
    f = m - 1.0
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * fma(w, fma(w, LG[5], LG[3]), LG[1])
    t2 = z * fma(w, fma(w, fma(w, LG[6], LG[4]), LG[2]), LG[0])
    R = t2 + t1
    hfsq = 0.5 * f * f
    dk = float(k)
    return dk * LN2_HI - ((hfsq - fma(s, hfsq + R, dk * LN2_LO)) - f)


TWO_PI = 6.283185307179586476925
S = [-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
     2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10]
Cc = [4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
      -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11]


def spec_sincos2pi(u):
    q = math.floor(fma(4.0, u, 0.5))
    r = fma(-0.25, float(q), u)
    x = r * TWO_PI
    z = x * x
    ps = fma(z, fma(z, fma(z, fma(z, S[5], S[4]), S[3]), S[2]), S[1])
    sx = fma(x * z, fma(z, ps, S[0]), x)
    pc = fma(z, fma(z, fma(z, fma(z, fma(z, Cc[5], Cc[4]), Cc[3]), Cc[2]), Cc[1]), Cc[0])
    zz = z * z
    cx = 1.0 - fma(-zz, pc, 0.5 * z)
    return [(sx, cx), (cx, -sx), (-sx, -cx), (-cx, sx)][q & 3]


def pair(s):
    a, b = nxt(s), nxt(s)
    u1 = float((a >> 11) + 1) * 2.0 ** -53
    u2 = float(b >> 11) * 2.0 ** -53
    r = math.sqrt(-2.0 * spec_log(u1))
    sn, cs = spec_sincos2pi(u2)
    return r * cs, r * sn


def stream_normals_uniform(seed, chain, d, T):
    """xi (T,d), u (T,) of GLOBAL chain id `chain`."""
    st = [stream_init(seed, 4 * chain + g) for g in range(4)]
    P = ((d + 3) // 4 + 1) // 2
    xi = np.zeros((T, d))
    u = np.zeros(T)
    for t in range(T):
        for g in range(4):
            for p in range(P):
                z0, z1 = pair(st[g])
                j0, j1 = 4 * (2 * p) + g, 4 * (2 * p + 1) + g
                if j0 < d:
                    xi[t, j0] = z0
                if j1 < d:
                    xi[t, j1] = z1
        u[t] = float(nxt(st[0]) >> 11) * 2.0 ** -53
    return xi, u
