#!/usr/bin/env python3
"""Offline statistical look at candidate counter-based contact generators (numpy, CPU).

Not a parity tool: the synthetic contact source is build-defined.  For each candidate this prints
what a physics stand-in needs to get right: the per-pair Bernoulli rate, correlation between pairs
at the same step, correlation of a pair with itself / other pairs at later steps, and bit-level
statistics of the underlying words (per-bit frequency, lag correlations).

    python tools/synth_quality.py [n_envs] [T]
"""
import sys

import numpy as np

M32 = np.uint64(0xFFFFFFFF)
M21 = np.uint32((1 << 21) - 1)


def u32(x):
    return (np.asarray(x, dtype=np.uint64) & M32).astype(np.uint32)


def mul32(a, b):
    return u32(a.astype(np.uint64) * np.uint64(b))


def lowbias32(x):
    x = x ^ (x >> np.uint32(16)); x = mul32(x, 0x7feb352d)
    x = x ^ (x >> np.uint32(15)); x = mul32(x, 0x846ca68b)
    return x ^ (x >> np.uint32(16))


def hashB(x):
    x = x ^ (x >> np.uint32(16)); x = mul32(x, 0x21f0aaad)
    x = x ^ (x >> np.uint32(15)); x = mul32(x, 0x735a2d97)
    return x ^ (x >> np.uint32(15))


def hash1(x, m):
    x = x ^ (x >> np.uint32(16)); x = mul32(x, m)
    return x ^ (x >> np.uint32(15))


def fold(x, k):
    p = x.astype(np.uint64) * np.uint64(k)
    return u32(p >> np.uint64(32)) ^ u32(p)


def rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def env_key(seed, env):
    """key(seed + 1000 * env, 0) of docs/HISTORY.md 5a (the reference's per-env seed arithmetic, rollout.py:206-210)"""
    es = (np.uint64(seed) + np.uint64(1000) * env.astype(np.uint64))
    return lowbias32(u32(es) ^ lowbias32(u32((es >> np.uint64(32)) + np.uint64(0x9E3779B9))))


def keys(seed, env):
    ka = env_key(seed, env)
    kb = lowbias32(ka ^ np.uint32(0x85EBCA6B))
    return ka, kb


def words(kind, seed, env, t):
    ka, kb = keys(seed, env)
    u = u32(ka.astype(np.uint64) + np.uint64((t + 1) * 0x9E3779B9))
    if kind == 'lb2':
        return lowbias32(u), hashB(u ^ kb)
    if kind == 'h1':
        return hash1(u, 0x7feb352d), hash1(u ^ kb, 0x21f0aaad)
    if kind == 'mum2':
        a = fold(fold(u, 0xD6E8FEB9), 0x9FB21C65)
        b = fold(fold(u ^ kb, 0xCA5A8263), 0x1D8E4E27)
        return a, b
    raise KeyError(kind)


def mask_tail(a, b):
    C = (a >> np.uint32(21)) | ((b >> np.uint32(21)) << np.uint32(11))
    D = rotl(a, 7) & rotl(b, 13)
    return a & b & (C | D) & M21


def mask_cur(seed, env, t):
    key = env_key(seed, env)
    w = [lowbias32(u32(key.astype(np.uint64) + np.uint64((4 * t + i) * 0x9E3779B9 + 0x85EBCA6B))) for i in range(4)]
    acc = ((w[0] >> np.uint32(21)) | ((w[1] >> np.uint32(21)) << np.uint32(11))) & M21
    acc &= w[3] & M21
    acc |= w[2] & M21
    acc &= w[1] & M21
    acc &= w[0] & M21
    return acc


def bits(x, n):
    return ((x[..., None] >> np.arange(n, dtype=np.uint32)) & np.uint32(1)).astype(np.float64)


def report(name, masks, wa=None, wb=None):
    T, B = masks.shape
    mb = bits(masks, 21)                         # (T, B, 21)
    p = mb.mean(axis=(0, 1))
    print('== %s' % name)
    print('  rate per pair: mean %.5f  min %.5f  max %.5f   (5/32 = %.5f; 3-sigma of one pair = %.5f)' % (
        p.mean(), p.min(), p.max(), 5 / 32, 3 * np.sqrt(5 / 32 * 27 / 32 / (T * B))))
    z = (mb - 5 / 32)
    sd = np.sqrt(5 / 32 * 27 / 32)
    flat = z.reshape(T * B, 21)
    cc = flat.T @ flat / (T * B) / sd ** 2        # same-step pair x pair correlation
    off = cc - np.diag(np.diag(cc))
    print('  same-step pair-pair correlation: max |r| %.4f  (noise 3-sigma %.4f)' % (np.abs(off).max(), 3 / np.sqrt(T * B)))
    for lag in (1, 2, 3, 16, 32):
        if lag >= T:
            continue
        a, b = z[:-lag].reshape(-1, 21), z[lag:].reshape(-1, 21)
        c2 = a.T @ b / a.shape[0] / sd ** 2
        print('  lag %2d: max |r| same pair %.4f, any pair %.4f  (noise 3-sigma %.4f)' % (
            lag, np.abs(np.diag(c2)).max(), np.abs(c2).max(), 3 / np.sqrt(a.shape[0])))
    # time-pattern per env: number of steps a pair touches over the episode vs Binomial(T, 5/32)
    k = mb.sum(axis=0)                           # (B, 21)
    print('  touches per episode: mean %.4f var %.4f   (binomial: %.4f, %.4f)' % (
        k.mean(), k.var(), T * 5 / 32, T * 5 / 32 * 27 / 32))
    if wa is not None:
        for nm, w in (('a', wa), ('b', wb)):
            wbits = bits(w, 32)
            f = wbits.mean(axis=(0, 1))
            zz = wbits - 0.5
            l1 = (zz[:-1] * zz[1:]).mean(axis=(0, 1)) * 4
            adj = (zz[..., :-1] * zz[..., 1:]).mean(axis=(0, 1)) * 4
            print('  word %s: bit freq in [%.4f, %.4f]; lag-1 autocorr max |r| %.4f; adjacent-bit max |r| %.4f (noise %.4f)' % (
                nm, f.min(), f.max(), np.abs(l1).max(), np.abs(adj).max(), 3 / np.sqrt(T * B)))
        ab = ((bits(wa, 32) - 0.5)[..., :, None] * (bits(wb, 32) - 0.5)[..., None, :]).mean(axis=(0, 1)) * 4
        print('  word a x word b bit correlation: max |r| %.4f' % np.abs(ab).max())


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    env = np.arange(B, dtype=np.uint32)
    seed = 12345
    cur = np.stack([mask_cur(seed, env, t) for t in range(T)])
    report('round-1 source: 4 x lowbias32, five fresh fields', cur)
    for kind in ('lb2', 'mum2', 'h1'):
        ws = [words(kind, seed, env, t) for t in range(T)]
        wa, wb = np.stack([w[0] for w in ws]), np.stack([w[1] for w in ws])
        report('%s + tail' % kind, mask_tail(wa, wb), wa, wb)


if __name__ == '__main__':
    main()
