"""Vectorised synthetic track generator for benchmarks and large-size tests.

The reference's ``simulate_tracks.sim_FOV`` is a per-track Python loop (extrack/simulate_tracks.py:177-223) and is
unusable at 1e6+ tracks; this generator draws the same kind of data (frame-level Markov chain of diffusive
states, Brownian displacements, Gaussian localisation error) in bulk.  It is NOT a restatement of sim_FOV
(no field-of-view exits / bleaching) - the likelihood code does not care where tracks come from.
"""
import numpy as np


def markov_states(rng, n, length, TrMat, Fs):
    TrMat = np.asarray(TrMat, float)
    S = len(TrMat)
    cumF = np.cumsum(Fs)
    cumT = np.cumsum(TrMat, axis=1)
    st = np.empty((n, length), dtype=np.int8)
    st[:, 0] = np.minimum((rng.random(n)[:, None] > cumF[None, :-1]).sum(1), S - 1)
    for k in range(1, length):
        u = rng.random(n)
        st[:, k] = np.minimum((u[:, None] > cumT[st[:, k - 1]][:, :-1]).sum(1), S - 1)
    return st


def brownian_tracks(n, length, Ds, TrMat, Fs, LocErr=0.02, dt=0.02, dims=2, seed=0, dtype=np.float64, chunk=250000):
    """Returns ndarray[n, length, dims] (float64): positions = cumsum of N(0, sqrt(2 D_s dt)) + N(0, LocErr)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, length, dims), dtype=dtype)
    sd = np.sqrt(2 * np.asarray(Ds, float) * dt)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        st = markov_states(rng, m, length, TrMat, Fs)
        steps = rng.standard_normal((m, length, dims)) * sd[st][:, :, None]
        steps[:, 0, :] = rng.random((m, dims))  # start anywhere in a unit cell
        pos = np.cumsum(steps, axis=1)
        pos += rng.standard_normal((m, length, dims)) * LocErr
        out[a:a + m] = pos
    return out


def bucket_sizes_geometric(total, lengths, ratio=0.9):
    w = ratio ** np.arange(len(lengths))
    n = np.floor(total * w / w.sum()).astype(int)
    n[0] += total - n.sum()
    return {int(l): int(k) for l, k in zip(lengths, n)}
