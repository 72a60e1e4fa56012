"""Synthetic workloads of SURVEY.md section 8d / BASELINE.json `configs`.

Generator: SplitMix64, seed = 0xE70100 + 0x100*config + instance.  The numbers
are produced on the host with numpy and uploaded; they only feed the evaluator.
"""
import numpy as np

from . import _lib as L

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def uniform(self, n, lo=0.0, hi=1.0):
        u = np.array([self.next_u64() >> 11 for _ in range(n)], dtype=np.float64) * (1.0 / (1 << 53))
        return lo + (hi - lo) * u


def _fast_uniform(seed, shape, lo, hi):
    """Vectorised SplitMix64 stream (same sequence as SplitMix64.uniform)."""
    n = int(np.prod(shape))
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = np.uint64(seed & MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


QUAD_PARAMS = np.array([1.0, 0.01, 9.81, 1.0, 1.0])  # m, I, g, w_thrust, w_torque
FW_PARAMS = np.array([10.0, 0.8, 1.1, 1.8, 9.81, 120.0, 0.3, 4.5, 0.03, 0.05,
                      0.08, -0.6, 0.06, 25.0, 0.9, 1.0])
TF = 16.0  # mirrors resource/configs/ocp_2d_ex1.xml: nsteps*dt = 32*0.5


def quadrotor_batch(config, B, M, n_obstacles, first_instance=0):
    """C2/C3/C4: X[B][6][M], U[B][2][M], disc records [B][np][8]."""
    X = np.empty((B, 6, M))
    U = np.empty((B, 2, M))
    recs = np.zeros((B, n_obstacles, L.PATH_REC))
    lo = np.array([0, 0, -np.pi / 4, -2, -2, -1.0])
    hi = np.array([10, 10, np.pi / 4, 2, 2, 1.0])
    mg = QUAD_PARAMS[0] * QUAD_PARAMS[2]
    for b in range(B):
        seed = 0xE70100 + 0x100 * config + (first_instance + b)
        u01 = _fast_uniform(seed, (8, M), 0.0, 1.0)
        X[b] = lo[:, None] + (hi - lo)[:, None] * u01[:6]
        U[b, 0] = (0.5 + u01[6]) * mg
        U[b, 1] = -0.1 + 0.2 * u01[7]
        if n_obstacles:
            o = _fast_uniform(seed ^ 0x0B57AC1E, (n_obstacles, 3), 0.0, 1.0)
            recs[b, :, 0] = L.PATH_DISC
            recs[b, :, 1] = 1.0 + 8.0 * o[:, 0]
            recs[b, :, 2] = 1.0 + 8.0 * o[:, 1]
            recs[b, :, 3] = (0.2 + 0.4 * o[:, 2]) ** 2
    return X, U, recs


def fixedwing_batch(config, B, M, first_instance=0):
    """C5: X[B][12][M], U[B][4][M] ~ U(-1,1) scaled per channel."""
    sx = np.array([100, 100, 50, 0.4, 0.3, 3.0, 25, 2, 2, 0.5, 0.5, 0.5])
    ox = np.array([0, 0, -100, 0, 0, 0, 25, 0, 0, 0, 0, 0.0])
    su = np.array([20, 0.3, 0.3, 0.3])
    ou = np.array([30, 0, 0, 0.0])
    X = np.empty((B, 12, M))
    U = np.empty((B, 4, M))
    for b in range(B):
        seed = 0xE70100 + 0x100 * config + (first_instance + b)
        u = _fast_uniform(seed, (16, M), -1.0, 1.0)
        X[b] = ox[:, None] + sx[:, None] * u[:12] * np.where(np.arange(12)[:, None] == 6, 0.2, 1.0)
        U[b] = ou[:, None] + su[:, None] * u[12:]
    return X, U


def pointmass_batch(config, B, M, first_instance=0):
    X = np.empty((B, 2, M))
    U = np.empty((B, 2, M))
    for b in range(B):
        seed = 0xE70100 + 0x100 * config + (first_instance + b)
        u = _fast_uniform(seed, (4, M), 0.0, 1.0)
        X[b] = 7.0 * u[:2]
        U[b] = -0.5 + u[2:]
    return X, U
