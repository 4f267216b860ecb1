"""Host-side seeding of the "rmh-rng v4" stream (Random stream):
integer-only (splitmix64 mixing), used to create the (N,16) uint64 state that
the stand-alone ``propose`` entry point advances on the device.  The chains of
a Sampler are seeded on the device by ``rmh_state_create``."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def rng_seed_host(seed, N, chain0=0):
    """(N,16) uint64: word w of sub-stream g of chain c at [c, 4g+w]."""
    with np.errstate(over="ignore"):
        sid = (np.uint64(4) * (np.uint64(chain0) + np.arange(N, dtype=np.uint64))[:, None]
               + np.arange(4, dtype=np.uint64)[None, :])
        x = _mix64(_mix64(np.uint64(seed) + _GAMMA) + sid)
        st = np.empty((N, 4, 4), dtype=np.uint64)
        for w in range(4):
            x = x + _GAMMA
            st[:, :, w] = _mix64(x)
    return np.ascontiguousarray(st.reshape(N, 16))
