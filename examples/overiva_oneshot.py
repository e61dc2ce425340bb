#!/usr/bin/env python3
"""
One-shot separation demo with the reference's command line (overiva_oneshot.py:72-116), on a
synthetic convolutive mixture instead of a simulated room (the reference needs pyroomacoustics,
mir_eval and a dataset download for that part, which are outside this repository's scope):

    python examples/overiva_oneshot.py -a overiva -m 4 -s 2 -n 20 [-d laplace|gauss] [-i eye|eig] [--no_cb]

The flow is the reference's (overiva_oneshot.py:293-379): time-domain microphone signals -> STFT (frame 4096, hop
2048, Hann; on the GPU, overiva_amd.stft) -> separation -> inverse STFT of the separated channels; `--domain stft`
skips the transforms and draws the mixture directly in the STFT domain.

What it keeps from the reference driver: the algorithm choices and their dispatch
(overiva_oneshot.py:301-330: 'auxiva' = all channels, 'auxiva_pca' = PCA + determined, 'overiva' = n_src
< n_mics), the flag names and defaults (-m 5 -s 2 -n 51, overiva_oneshot.py:103-105), the STFT shape of
the reference (frame 4096 -> 2049 bins, complex128), the convergence callback every 10 iterations and the
timing printout (overiva_oneshot.py:298,366-368).  Separation quality is reported as the
signal-to-interference ratio of the demixed bins computed from the known mixing (no mir_eval here).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

algo_choices = ["auxiva", "auxiva_pca", "overiva"]
model_choices = ["laplace", "gauss"]
init_choices = ["eye", "eig"]


def synthetic_scene(n_mics, n_targets, n_frames, n_freq, seed, sinr_db=10.0):
    """n_targets super-Gaussian sources + diffuse-like interference, mixed per bin by random matrices."""
    rng = np.random.default_rng(seed)
    act = rng.gamma(0.3, 1.0, (n_frames, 1, n_targets))
    S = act * (rng.standard_normal((n_frames, n_freq, n_targets)) + 1j * rng.standard_normal((n_frames, n_freq, n_targets)))
    A = rng.standard_normal((n_freq, n_mics, n_targets)) + 1j * rng.standard_normal((n_freq, n_mics, n_targets))
    images = np.einsum("fmk,tfk->tfmk", A, S)                      # (T, F, M, K) source images at the mics
    mix = images.sum(axis=-1)
    noise = rng.standard_normal(mix.shape) + 1j * rng.standard_normal(mix.shape)
    noise *= np.sqrt(np.mean(np.abs(mix) ** 2) / np.mean(np.abs(noise) ** 2) / 10 ** (sinr_db / 10))
    return (mix + noise).astype(np.complex128), images


def synthetic_audio_scene(n_mics, n_targets, n_frames, framesize, seed, sinr_db=10.0, taps=48):
    """time-domain scene: n_targets sources with slowly varying activity, each reaching every microphone through
    its own short random filter (convolutive mixture), plus white noise.  Returns the microphone signals
    (n_samples, n_mics) and the source images at the microphones (n_samples, n_mics, n_targets)."""
    rng = np.random.default_rng(seed)
    hop = framesize // 2
    n = n_frames * hop
    seg = hop // 2
    env = np.repeat(rng.gamma(0.3, 1.0, (n // seg + 1, n_targets)), seg, axis=0)[:n]
    src = env * rng.standard_normal((n, n_targets))
    h = rng.standard_normal((n_mics, n_targets, taps)) * np.exp(-np.arange(taps) / 8.0)
    images = np.empty((n, n_mics, n_targets))
    for m in range(n_mics):
        for k in range(n_targets):
            images[:, m, k] = np.convolve(src[:, k], h[m, k])[:n]
    mix = images.sum(axis=-1)
    noise = rng.standard_normal(mix.shape)
    noise *= np.sqrt(np.mean(mix ** 2) / np.mean(noise ** 2) / 10 ** (sinr_db / 10))
    return mix + noise, images


def sir_db(W, images):
    """mean over sources of best-permutation SIR of y_k = w_k^H x, from the known source images"""
    Y = np.einsum("fmk,tfmj->tfkj", np.conj(W), images)            # output k due to source j
    P = np.sum(np.abs(Y) ** 2, axis=(0, 1))                        # (K_out, K_src)
    K = P.shape[0]
    import itertools

    best = -np.inf
    for perm in itertools.permutations(range(P.shape[1]), K):
        s = np.mean([10 * np.log10(P[k, perm[k]] / max(P[k].sum() - P[k, perm[k]], 1e-30)) for k in range(K)])
        best = max(best, s)
    return best


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Demonstration of blind source extraction using overdetermined IVA.")
    ap.add_argument("--no_cb", action="store_true", help="Removes callback function")
    ap.add_argument("-a", "--algo", type=str, default=algo_choices[0], choices=algo_choices, help="Chooses BSS method to run")
    ap.add_argument("-d", "--dist", type=str, default=model_choices[0], choices=model_choices, help="IVA model distribution")
    ap.add_argument("-i", "--init", type=str, default=init_choices[0], choices=init_choices, help="Initialization, eye: identity, eig: principal eigenvectors")
    ap.add_argument("-m", "--mics", type=int, default=5, help="Number of mics")
    ap.add_argument("-s", "--srcs", type=int, default=2, help="Number of sources")
    ap.add_argument("-n", "--n_iter", type=int, default=51, help="Number of iterations")
    ap.add_argument("--frames", type=int, default=160, help="STFT frames of the synthetic scene")
    ap.add_argument("--domain", choices=["audio", "stft"], default="audio",
                    help="audio: time-domain scene, STFT and inverse STFT on the GPU (the reference's flow); stft: scene drawn in the STFT domain")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    assert args.srcs <= args.mics, "More sources than microphones is not supported"      # overiva_oneshot.py:118
    return args


def separate(args, X_mics, overiva, auxiva_pca, callback):
    """the dispatch of overiva_oneshot.py:301-330, argument for argument (auxiva: all channels, no n_src and no
    init_eig; auxiva_pca: n_src, no init_eig; overiva: n_src and init_eig)"""
    if args.algo == "auxiva":                          # overiva_oneshot.py:303-309
        return overiva(X_mics, n_iter=args.n_iter, proj_back=True, model=args.dist, callback=callback)
    if args.algo == "auxiva_pca":                      # overiva_oneshot.py:312-319
        return auxiva_pca(X_mics, n_src=args.srcs, n_iter=args.n_iter, proj_back=True, model=args.dist, callback=callback)
    return overiva(X_mics, n_src=args.srcs, n_iter=args.n_iter, proj_back=True, model=args.dist,   # :322-330
                   init_eig=(args.init == init_choices[1]), callback=callback)


def output_sir(Y, X_mics, images, n_keep):
    """SIR of the n_keep strongest outputs (the reference re-orders by power, overiva_oneshot.py:387-389).  The
    outputs are linear in the mixture per bin: the demixing of each bin is recovered by least squares from (X, Y),
    then applied to the known source images."""
    order = np.argsort(np.sum(np.abs(Y) ** 2, axis=(0, 1)))[::-1][:n_keep]
    T, F, M = X_mics.shape
    W = np.empty((F, M, n_keep), dtype=np.complex128)
    for f in range(F):
        G, *_ = np.linalg.lstsq(X_mics[:, f, :], Y[:, f, order], rcond=None)     # Y = X G, G = conj(W)
        W[f] = np.conj(G)
    return sir_db(W, images)


def run(argv=None, verbose=True):
    """run the driver; returns what a test needs (arguments, scene, outputs, callback trace, timings, SIRs)"""
    args = parse_args(argv)
    from overiva_amd import auxiva_pca, overiva

    from overiva_amd import stft as transform

    framesize = 4096                                  # overiva_oneshot.py:156
    n_freq = framesize // 2 + 1
    mics_signals = None
    if args.domain == "audio":
        win_a = transform.hann(framesize)             # overiva_oneshot.py:157-158
        win_s = transform.compute_synthesis_window(win_a, framesize // 2)
        mics_signals, images_t = synthetic_audio_scene(args.mics, args.srcs, args.frames, framesize, args.seed)
        # overiva_oneshot.py:293-296: analysis of all microphones, complex128, (n_frames, n_freq, n_mics)
        X_mics = transform.analysis(mics_signals, framesize, framesize // 2, win=win_a).astype(np.complex128)
        images = np.stack([transform.analysis(images_t[:, :, k], framesize, framesize // 2, win=win_a)
                           for k in range(args.srcs)], axis=-1).astype(np.complex128)
    else:
        X_mics, images = synthetic_scene(args.mics, args.srcs, args.frames, n_freq, args.seed)
    if verbose:
        print(f"scene ({args.domain}): {X_mics.shape[0]} frames x {n_freq} bins x {args.mics} mics, {args.srcs} targets, dtype {X_mics.dtype}")
    trace = []

    def convergence_callback(Y):                      # overiva_oneshot.py:263-284 (metric instead of bss_eval)
        trace.append(float(np.mean(np.abs(Y) ** 2)))

    cb = None if args.no_cb else convergence_callback
    t_begin = time.perf_counter()                     # overiva_oneshot.py:298
    Y = separate(args, X_mics, overiva, auxiva_pca, cb)
    t_end = time.perf_counter()
    y = None
    if args.domain == "audio":                        # overiva_oneshot.py:371-379: back to the time domain
        y = transform.synthesis(Y, framesize, framesize // 2, win=win_s)
    sir_out = output_sir(Y, X_mics, images, args.srcs)
    W0 = np.zeros((n_freq, args.mics, args.srcs), dtype=np.complex128)
    W0[:, : args.srcs, :] = np.eye(args.srcs)
    sir_in = sir_db(W0, images)
    if verbose:
        print("Time for BSS: {:.2f} s".format(t_end - t_begin))   # overiva_oneshot.py:366-368
        print(f"output {Y.shape} {Y.dtype}; callback fired {len(trace)} times")
        print(f"SIR of the {args.srcs} strongest outputs: {sir_out:.1f} dB (first {args.srcs} microphones: {sir_in:.1f} dB)")
        if y is not None:
            print(f"separated audio {y.shape} {y.dtype} ({mics_signals.shape[0]} samples in)")
    return {"args": args, "X": X_mics, "images": images, "Y": Y, "trace": trace, "seconds": t_end - t_begin,
            "sir_out": sir_out, "sir_in": sir_in, "audio_in": mics_signals, "audio_out": y}


if __name__ == "__main__":
    run()
