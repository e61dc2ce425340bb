#!/usr/bin/env python3
"""Which stage of an all-float32 iteration limits parity on ill-conditioned input?  (CPU study, not a test.)

Runs the oracle's staged form on the golden mixture fixtures with float32 rounding injected at one stage at a
time (power pass, activation weights, covariance accumulation, per-bin update) and prints the distance of W
from the reference's complex128 result in units of the reference's own complex64 floor.  The outcome decides
which stages the library's precise mode computes in float64.

    python tests/precision_study.py
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import overiva_oracle as orc  # noqa: E402


def power32(X, W):
    """demix + |y|^2 + sum over bins, all float32 (like a cgemm + float32 norm)"""
    Y = np.einsum("tfm,fmk->tfk", X.astype(np.complex64), np.conj(W.astype(np.complex64)))
    p = (Y.real.astype(np.float32) ** 2 + Y.imag.astype(np.float32) ** 2)
    return np.sum(p, axis=1, dtype=np.float32).astype(np.float64)


def weights32(p, F, model):
    p = p.astype(np.float32)
    r = (np.float32(2) * np.sqrt(p)) if model == "laplace" else p / np.float32(F)
    gamma = np.float32(np.mean(r.astype(np.float64), axis=0))
    rn = r * (np.float32(1) / gamma)
    rn = np.maximum(rn, np.float32(1e-15))
    ws = gamma if model == "laplace" else np.sqrt(gamma)
    return (np.float32(1) / rn).astype(np.float64), ws.astype(np.float64)


def cov32_chain(X, rinv, chain):
    """float32 products and float32 running sums over chains of `chain` frames, chains added in float64"""
    T, F, M = X.shape
    K = rinv.shape[1]
    Xc = X.astype(np.complex64)
    V = np.zeros((K, F, M, M), np.complex128)
    w = rinv.astype(np.float32)
    for t0 in range(0, T, chain):
        acc_r = np.zeros((K, F, M, M), np.float32)
        acc_i = np.zeros((K, F, M, M), np.float32)
        for t in range(t0, min(T, t0 + chain)):
            x = Xc[t]                                   # (F, M)
            pr = (x.real[:, :, None] * x.real[:, None, :] + x.imag[:, :, None] * x.imag[:, None, :]).astype(np.float32)
            pi = (x.imag[:, :, None] * x.real[:, None, :] - x.real[:, :, None] * x.imag[:, None, :]).astype(np.float32)
            for k in range(K):
                acc_r[k] = (acc_r[k] + w[t, k] * pr).astype(np.float32)
                acc_i[k] = (acc_i[k] + w[t, k] * pi).astype(np.float32)
        V += acc_r.astype(np.float64) + 1j * acc_i.astype(np.float64)
    return V / T


def update32(W_hat, V, Cx, K):
    """the per-bin chain in complex64 LAPACK (what the reference's complex64 run does)"""
    W_hat = W_hat.astype(np.complex64).copy()
    V = V.astype(np.complex64)
    Cx = Cx.astype(np.complex64)
    F, M, _ = W_hat.shape
    for s in range(K):
        e = np.zeros((F, M), np.complex64)
        e[:, s] = 1
        w = np.linalg.solve(np.conj(np.swapaxes(W_hat, 1, 2)) @ V[s], e[..., None])[..., 0]
        d = np.einsum("fc,fcd,fd->f", np.conj(w), V[s], w)
        W_hat[:, :, s] = (w / np.sqrt(d)[:, None]).astype(np.complex64)
        if K < M:
            W_hat[:, :K, K:] = orc.orth_constraint_J(W_hat[:, :, :K], Cx, K)
    return W_hat.astype(np.complex128)


def run(X, K, model, n_iter, f32_power=False, f32_weights=False, cov=None, f32_update=False, round_v=False):
    T, F, M = X.shape
    Xc = X.astype(np.complex128)
    Cx = orc.input_covariance(Xc)
    W_hat = orc.init_demixing(Cx, K)
    for _ in range(n_iter):
        p = power32(X, W_hat[:, :, :K]) if f32_power else orc.demix_power(Xc, W_hat[:, :, :K])
        rinv, ws = weights32(p, F, model) if f32_weights else orc.finalize_activation(p, F, model)
        W_hat[:, :, :K] /= ws[None, None, :]
        V = orc.weighted_cov_all(Xc, rinv) if cov is None else cov32_chain(X, rinv, cov)
        if round_v:
            V = V.astype(np.complex64).astype(np.complex128)
        W_hat = update32(W_hat, V, Cx, K) if f32_update else orc.ip_update_bin(W_hat, V, Cx, K)
    return W_hat[:, :, :K]


def main():
    variants = [
        ("all f64 (staged oracle)", dict()),
        ("f32 power pass only", dict(f32_power=True)),
        ("f32 weights only", dict(f32_weights=True)),
        ("f32 cov, chains of 64", dict(cov=64)),
        ("f32 cov, chains of 8", dict(cov=8)),
        ("exact cov rounded to f32", dict(round_v=True)),
        ("f32 update only (LAPACK c64)", dict(f32_update=True)),
        ("f32 power+weights+update, f64 cov", dict(f32_power=True, f32_weights=True, f32_update=True)),
        ("f32 everything, chains 64", dict(f32_power=True, f32_weights=True, cov=64, f32_update=True)),
    ]
    files = sorted(glob.glob(os.path.join(HERE, "golden", "overiva_*_mix.npz")))
    for path in files:
        with np.load(path) as d:
            g = {k: d[k] for k in d.files}
        X, K = g["X"], int(g["K"])
        name = os.path.basename(path)[8:-4]
        for model in ("laplace",):
            n_iter = 20
            ref = g[f"W_c128_{model}_{n_iter}"]
            k64 = f"W_c64_{model}_{n_iter}"
            if k64 not in g:
                continue
            floor = orc.rel_err(g[k64], ref)
            print(f"== {name} {X.shape} K={K} {model}/{n_iter}: reference c64 floor {floor:.2e}, amp {float(g[f'amp_{model}_{n_iter}']):.1f}")
            for label, kw in variants:
                W = run(X, K, model, n_iter, **kw)
                e128 = orc.rel_err(W, ref)
                e64 = orc.rel_err(W, g[k64])
                print(f"   {label:38s} vs c128 {e128:.2e} ({e128 / floor:5.2f} floors)   vs ref-c64 {e64:.2e} ({e64 / floor:5.2f} floors)")


if __name__ == "__main__":
    main()
