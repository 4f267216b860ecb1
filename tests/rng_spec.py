"""Pure-Python restatement of the "rmh-rng v4" stream (oracle/rmh_oracle.c
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


_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        import os
        _TABLE = np.load(os.path.join(os.path.dirname(__file__), "golden", "icdf_table.npy"))
    return _TABLE


def normal_from_half(h):
    """rmh-rng v4: one standard normal per 32-bit half of a stream word
    (tools/gen_icdf_table.py; rows 0..31 of the table)."""
    tab = table()
    q = h | 1
    r = min(32 - q.bit_length(), 31)
    N = ((q << r) & 0xFFFFFFFF) | (1 << 31)
    j = (N >> 28) & 7
    m27 = (N >> 1) & 0x7FFFFFF
    t = float(m27)            # the table's polynomials are in m27 itself (t = m27 2^-26 - 1 folded in)
    c = tab[r, j]
    z = float(c[-1])
    for k in range(len(c) - 2, -1, -1):
        z = fma(z, t, float(c[k]))
    return -z if (h & 1) else z


_TABLE64 = None


def table64():
    global _TABLE64
    if _TABLE64 is None:
        import os
        _TABLE64 = np.load(os.path.join(os.path.dirname(__file__), "golden", "icdf64_table.npy"))
    return _TABLE64


def normal_from_word(x):
    """the 64-bit-resolution normal mode: one standard normal per stream word
    (tools/gen_icdf_table.py --f64; all 56 rows, degree 10 in t)."""
    tab = table64()
    q = x | 1
    r = min(64 - q.bit_length(), 55)
    N = ((q << r) & M64) | (1 << 63)
    j = (N >> 60) & 7
    mant = (N >> 8) & ((1 << 52) - 1)
    v = float(Fraction(1) + Fraction(mant, 1 << 52))      # exact
    t = fma(2.0, v, -3.0)
    c = tab[r, j]
    z = float(c[-1])
    for k in range(len(c) - 2, -1, -1):
        z = fma(z, t, float(c[k]))
    return -z if (x & 1) else z


def stream_normals_uniform(seed, chain, d, T, mode="f32x2"):
    """xi (T,d), u (T,) of GLOBAL chain id `chain`."""
    st = [stream_init(seed, 4 * chain + g) for g in range(4)]
    if mode == "f64":                 # word s of sub-stream g is coordinate 4s+g, ceil(d/4) words each
        xi = np.zeros((T, d))
        u = np.zeros(T)
        for t in range(T):
            for g in range(4):
                for s in range((d + 3) // 4):
                    z = normal_from_word(nxt(st[g]))
                    if 4 * s + g < d:
                        xi[t, 4 * s + g] = z
            u[t] = float(nxt(st[0]) >> 11) * 2.0 ** -53
        return xi, u
    M = (d + 7) // 8
    xi = np.zeros((T, d))
    u = np.zeros(T)
    for t in range(T):
        for g in range(4):
            for m in range(M):
                w = nxt(st[g])
                zh, zl = normal_from_half(w >> 32), normal_from_half(w & 0xFFFFFFFF)
                if 8 * m + g < d:
                    xi[t, 8 * m + g] = zh
                if 8 * m + 4 + g < d:
                    xi[t, 8 * m + 4 + g] = zl
        u[t] = float(nxt(st[0]) >> 11) * 2.0 ** -53
    return xi, u
