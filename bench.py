#!/usr/bin/env python3
"""Benchmark of the AuxIVA / OverIVA iteration hot path on MI355X.

Metric (BASELINE.json): AuxIVA iterations/sec at 2048 bins x 4000 frames x 8 mics / 2 sources
(OverIVA, laplace), synthetic complex64 STFT, at 1/2/4/8 GPUs.  One "step" = one iteration of the
loop body reference overiva.py:138-190; prologue and epilogue are outside the timed region and X is
resident in HBM when it starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 shards the 2048 bins over the ranks (strong scaling: the problem is fixed) with one RCCL
all-gather of the (T, K) partial source powers per iteration.  Started WITHOUT a launcher (no
WORLD_SIZE in the environment) `--gpus N` starts its N ranks itself, as child processes, before this
process has touched a GPU, and relays rank 0's line; a host watchdog ends a run that does not finish
(a dead peer, a transport that does not work on this node) instead of hanging, and an attempt with
the opt-in push exchange that fails is repeated, in fresh processes, with the collective.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- weighted-covariance pass: algorithmic bytes (8TFM + 4TK + 8FKM^2) /
                  average kernel duration measured with HIP events on the kernel's own stream
  cpu_baseline -- the oracle's reference-faithful NumPy restatement of overiva.py timed on this
                  box's host cores on the full workload (N = 1 only)
  other_modes  -- the other arithmetic modes on the same workload, same protocol
  configs      -- (N = 1) BASELINE configs[1], configs[4] and one rank's shard of configs[3], measured in
                  the same process after the headline
"""
import argparse
import json
import os

# The host driver of this platform shares device memory between processes through dmabuf only (RCCL, the library's own
# exchanges): the variable has to be in the environment before this process's first GPU call, whatever launched it
# (torch.distributed.run, the driver, a shell).  Nothing above this line touches a GPU.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import signal
import subprocess
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MODEL = "laplace"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (= the fp32 vector peak)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # fp64 vector peak of AMD's MI355X data sheet (the guide lists no fp64 figure; measured with tools/pkbench.hip: 60)
# --config: the headline workload (default; the one BASELINE.json's metric is quoted on) and the other shapes
CONFIGS = {
    "headline": dict(T=4000, F=2048, M=8, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (BASELINE.json configs[2])"),
    "cfg5": dict(T=4000, F=2048, M=16, K=16, name="determined AuxIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (BASELINE.json configs[4])"),
    "cfg2": dict(T=1000, F=513, M=4, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (BASELINE.json configs[1])"),
    "shard8": dict(T=4000, F=256, M=8, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64: one rank's shard of the headline shape at 8 GPUs (BASELINE.json configs[3])"),
    "shard2": dict(T=4000, F=1024, M=8, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64: one rank's shard of the headline shape at 2 GPUs"),
    "shard4": dict(T=4000, F=512, M=8, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64: one rank's shard of the headline shape at 4 GPUs"),
    # BASELINE configs[0] is the reference's own one-shot call (overiva_oneshot.py -a overiva -m 4 -s 2: STFT of 4096-point frames
    # = 2049 bins x ~160 frames x 4 mics, complex128): the iteration of that shape, in the arithmetic of complex128 input
    "cfg0": dict(T=160, F=2049, M=4, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, the shape and arithmetic (complex128 input) of the reference's one-shot call (BASELINE.json configs[0])"),
    # not a BASELINE config: 16 channels with few sources, the shape the four-lanes-per-(bin, frame) covariance kernel
    # (csrc/kernels_cov_quad.hip) exists for; timed in its default arithmetic (`mixed`)
    "m16k2": dict(T=4000, F=2048, M=16, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (16 channels, few sources)"),
    # test-only: small enough for the X-resident kernels of two ranks to be resident side by side on ONE GPU
    # (tests/test_sharded_2proc_gpu.py); not a BASELINE config
    "tiny": dict(T=600, F=128, M=4, K=2, name="test shape {F} bins x {T} frames x {M} mics / {K} src"),
}
T, F, M, K = 4000, 2048, 8, 2
WORKLOAD = CONFIGS["headline"]["name"].format(T=T, F=F, M=M, K=K)
MODES = ("fast", "mixed", "precise")
MODE_TEXT = {
    "fast": "fast (float32 products, lane chains and per-bin algebra; float64 sums across lanes and frame splits)",
    "mixed": "mixed (float32 products and lane chains of the covariance pass, float64 sums across lanes / splits, float64 per-bin "
             "algebra with W_hat in complex128: what overiva() runs for complex64 input)",
    "precise": "precise (covariance as float64 sums of exact float64 products + float64 per-bin algebra: what overiva() runs "
               "for complex128 input and for 9/11/13/15 channels)",
}


def select_config(name):
    global T, F, M, K, WORKLOAD
    c = CONFIGS[name]
    T, F, M, K = c["T"], c["F"], c["M"], c["K"]
    WORKLOAD = c["name"].format(T=T, F=F, M=M, K=K)


def cov_algorithmic_bytes(t, f, m, k):
    """read X once + read r_inv + write V"""
    return 8 * t * f * m + 4 * t * k + 8 * f * k * m * m


def measured_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of this same workload
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction; see
    profiles/r05_pmc_hbm_traffic.json).  PMC counters need rocprofv3 around the process, so they cannot
    be sampled from inside a plain bench run; None when the profile is absent."""
    for name in ("r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json"):
        path = os.path.join(REPO, "profiles", name)
        try:
            with open(path) as f:
                return float(json.load(f)["kernels"][kernel_key]["hbm_bytes_per_launch"]), os.path.relpath(path, REPO)
        except Exception:
            continue
    return None, None


def synth_x_device(torch, device, f0, f1, shape=None):
    """The iid complex64 workload, generated on the device (seeded per bin so that any sharding of
    the bins sees the same tensor)."""
    t, f, m = shape or (T, F, M)
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    x = torch.randn((t, f, m, 2), generator=g, device=device, dtype=torch.float32)
    x = x[:, f0:f1].contiguous()
    return torch.view_as_complex(x)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(X):
    """seconds per iteration of the reference-faithful oracle: (t(4 its) - t(1 it)) / 3, so the prologue (input
    covariance, allocation) cancels; after a warm-up call"""
    from oracle import overiva_oracle as orc

    orc.overiva_faithful(X, n_src=K, n_iter=1, proj_back=False, model=MODEL)      # warm-up (page faults, BLAS threads)
    t0 = time.perf_counter()
    orc.overiva_faithful(X, n_src=K, n_iter=1, proj_back=False, model=MODEL)
    t1 = time.perf_counter()
    orc.overiva_faithful(X, n_src=K, n_iter=4, proj_back=False, model=MODEL)
    t2 = time.perf_counter()
    return max(((t2 - t1) - (t1 - t0)) / 3.0, 1e-6)


def cpu_baseline():
    """Reference-faithful NumPy restatement (oracle/overiva_oracle.py::overiva_faithful: same statements, temporaries and
    dtypes as overiva.py:80-204; profiles/r03_cpu_side_by_side.json shows it next to the real reference in the build
    container) on the FULL workload -- all 2048 bins x 4000 frames x 8 mics / 2 src, about 10 s of CPU work per threading --
    with the BLAS threading the process starts with (`cores` = the threads threadpoolctl reports for that pool) and, as the
    reference's own sweep pinned BLAS to one thread (overiva_sim.py:85-91), once more with one thread."""
    from oracle import overiva_oracle as orc

    X = orc.synth_iid(T, F, M, seed=0)
    per_iter = _time_oracle(X)
    threads, pools, one = os.cpu_count(), None, None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        info = threadpool_info()
        pools = [{k: i.get(k) for k in ("internal_api", "num_threads", "version")} for i in info]
        blas = [i.get("num_threads", 1) for i in info if i.get("user_api") == "blas"]
        threads = max(blas or [i.get("num_threads", 1) for i in info] or [1])
        with threadpool_limits(limits=1):
            per1 = _time_oracle(X)
        one = {"value": 1.0 / per1, "cores": 1, "sample": f"the same workload with one BLAS thread, {per1:.3f} s per iteration"}
    except Exception:
        pass
    return {"value": 1.0 / per_iter, "unit": "iterations/s", "cores": threads, "kind": "port",
            "threadpools": pools, "cpu": _cpu_model(), "host_cpus": os.cpu_count(), "single_thread": one,
            "note": "the batched (M x T)(T x M) products of overiva.py:179 hardly thread in OpenBLAS: compare single_thread",
            "sample": f"oracle.overiva_faithful (NumPy, complex64 in / float64 r like overiva.py) on the full {F} bins x "
                      f"{T} frames x {M} mics / {K} src, iterations 2-4 of a 4-iteration call minus a 1-iteration call; "
                      f"{per_iter:.3f} s per iteration"}


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


_PRE_PASS_STEPS = 0


def _time_plan(plan, args, repeats=0):
    """W untimed warm-up iterations, then exactly K timed ones bracketed by synchronisation (the contract measurement);
    then `repeats` more measurements of the same K iterations (for the median); then the same K again with every kernel
    bracketed by HIP events on the plan's stream"""
    import torch

    # (an untimed pass first: the first replay of the captured graphs and the clock ramp of a GPU that was idle -- the box has
    #  just imported torch -- cost 2-4 % of a 4-10 ms measurement: round 5 saw 4786 it/s in the first K = 20 steps and 4900-4960 in
    #  each of the five that followed.  The pass is the same K steps repeated until >= 50 ms of iterations have run
    #  (`untimed_pre_pass_steps` in the output); the W warm-up steps of the contract follow, then exactly K timed ones)
    global _PRE_PASS_STEPS
    t0 = time.perf_counter()
    plan.iterate(args.steps)
    plan.sync()
    n_pre = args.steps
    while time.perf_counter() - t0 < 0.05 and n_pre < 100000:
        plan.iterate(args.steps)
        plan.sync()
        n_pre += args.steps
    _PRE_PASS_STEPS = n_pre
    plan.iterate(args.warmup)
    plan.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.iterate(args.steps)
    plan.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    more = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        plan.iterate(args.steps)
        plan.sync()
        torch.cuda.synchronize()
        more.append(time.perf_counter() - t0)
    total_ms, stages = plan.iterate_timed(args.steps, per_kernel=True)
    return dt, total_ms, stages, more


def _make_plan(oa, X, shape, mode, graph, resident=False):
    t, f, m, k = shape
    plan = oa.Plan(t, f, m, k, MODEL, device=0)
    plan.set_precision(mode)
    plan.set_x_device(X.data_ptr(), keepalive=X)
    plan.covariance()
    plan.set_w(None)
    if resident:
        plan.set_resident(True)
    elif graph:
        plan.use_graph(True)
    return plan


# ---- roofline of the dominant (covariance) kernel ------------------------------------------------------------------------
# One row per kernel family of csrc/kernels_cov*.hip: (applies(t, f, m, k, mode, fused), builder).  The first row that applies
# names the kernel the plan dispatches for the shape (csrc/plan.hip::choose_cov_geom) and prices its launch: HBM-bound families
# by SURVEY.md 8(d)'s algorithmic bytes, arithmetic-bound ones by the flops they ISSUE against the peak of the unit they issue to.
_EVENT_NOTE = "avg_launch_ms is the event-bracketed stage: the weights pre-pass (~4-6 us) + the kernel + the gaps of the bracketing events"


def _hbm_roof(kname, what, bytes_cov, cov_ms, note=None):
    achieved = bytes_cov / (cov_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": f"{kname} ({what})", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_cov, "avg_launch_ms": cov_ms}
    if note:
        roof["note"] = note
    return kname, roof


def _flop_roof(kname, what, bound, peak, issued, cov_ms, shape, note, useful=None, parts=None, bytes_cov=None):
    """arithmetic-bound family: achieved / frac = ISSUED flops per second against `peak`; beside them SURVEY 8d's naive complex
    count (8 K M^2 T F; 268.4 GF at 16 x 16) and, where given, the useful count (Hermitian half, each product formed once)"""
    t, f, m, k = shape
    sec = cov_ms * 1e-3
    naive = 8.0 * k * m * m * t * f
    roof = {"bound": bound, "kernel": f"{kname} ({what})", "achieved": issued / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": issued / sec / 1e12 / peak, "issued_flops_per_launch": issued}
    if parts:
        roof["issued_matrix_flops_per_launch"], roof["issued_vector_flops_per_launch"] = parts
    if useful is not None:
        roof.update({"useful_flops_per_launch": useful, "useful_tflops": useful / sec / 1e12, "frac_useful": useful / sec / 1e12 / peak})
    roof.update({"naive_complex_flops_per_launch": naive, "naive_complex_tflops": naive / sec / 1e12})
    if useful is not None:
        roof["frac_naive"] = naive / sec / 1e12 / peak
    if bytes_cov is not None:
        roof.update({"algorithmic_bytes_per_launch": bytes_cov, "hbm_gbs": bytes_cov / sec / 1e9})
    roof.update({"avg_launch_ms": cov_ms, "traffic": None, "note": note})
    return kname, roof


def _useful_flops(t, f, m, k):
    return ((m * (m - 1) // 2) * (6.0 + 4.0 * k) + m * (3.0 + 2.0 * k)) * t * f     # Hermitian half, products formed once


def _roof_cov_update(t, f, m, k, mode, cov_ms):
    # covariance + per-bin update of the same bins in ONE launch (csrc/kernels_cov_update.hip): the float64 partials stay in LDS --
    # the 8 F K M^2 bytes SURVEY 8d counts for V are not written -- and W_hat (complex64 + complex128) and Cx are read, W_hat
    # written: 8 T F M + 4 T K + F M^2 (8 + 16 + 8 + 8 + 16)
    kname = "cov_update_kernel<double>" if mode != "fast" else "cov_update_kernel<float>"
    return _hbm_roof(kname, "weighted spatial covariance of all sources in one pass AND the per-bin IP1 solve / normalisation / "
                            "orthogonal-constraint update of the same bins in one launch, overiva.py:158-190",
                     8 * t * f * m + 4 * t * k + f * m * m * 56, cov_ms,
                     "the launch ends with the latency-bound per-bin chain of its four bins per workgroup (5-6 us during which nothing streams); "
                     "the covariance kernel it replaces (cov_dma_kernel<8, 2>, $OIVA_COV_UPDATE=0) takes 97-99 us for 526.4 MB = 0.67 of the "
                     "peak and is followed by update_bg_kernel (12.8 us)")


def _roof_stream8(t, f, m, k, mode, cov_ms):
    # up to 8 channels: one pass over X for all sources (two per pass of cov_dma), HBM-bound
    if mode == "precise":
        kname = f"cov_pair64_kernel<{min(k, 2)}, false>" if m == 8 else f"cov_kernel<{m}, {min(k, 2)}, false, double>"
    else:
        kname = "cov_pair32_kernel<false>" if m == 8 and k >= 3 else f"cov_dma_kernel<{m}, {min(k, 2)}>"
    return _hbm_roof(kname, "weighted spatial covariance of all sources in one pass, overiva.py:179", cov_algorithmic_bytes(t, f, m, k), cov_ms,
                     _EVENT_NOTE if kname.startswith(("cov_pair64", "cov_pair32")) else None)


def _roof_quad(t, f, m, k, mode, cov_ms):
    # 9..16 channels, few sources: the Hermitian half on the vector ALU, four lanes per (bin, frame), one pass over X per two sources
    # (9 / 11 / 13 / 15 channels: the kernels of the next even count on a zero-padded copy of X; bytes stay the algorithmic ones)
    kname = f"cov_quad_kernel<{min(k, 2)}, false>"
    return _hbm_roof(kname, "weighted spatial covariance, two sources per pass over X, overiva.py:179",
                     cov_algorithmic_bytes(t, f, m, k) + (-(-k // 2) - 1) * 8 * t * f * m, cov_ms,
                     "co-limited by the vector ALU: 512 real FMAs per (bin, frame, source pair) at 16 channels against 128 at 8")


def _roof_hmfma(t, f, m, k, mode, cov_ms):
    # the sources on the fp32 matrix cores (csrc/kernels_cov_hmfma.hip): per bin and 4 frames mc + 1 v_mfma_f32_16x16x4_f32 (16 at 16
    # channels: the 256 real numbers of the Hermitian half exactly), 2048 flops each whatever k, and the vector instructions forming the
    # Hermitian products; bound by fp32 arithmetic -- matrix and vector instructions share the 157.3 TFLOP/s ALUs
    mc = m + m % 2
    mfma = (16 if mc == 16 else mc + 1) * 2048.0 / 4 * t * f
    pk = mc == 16 and os.environ.get("OIVA_HMFMA_PK", "1") != "0"
    # (16 channels: the partners from LDS, a product pair = two PACKED instructions of 256 flop slots -- 2 + 4 plain instructions for
    #  the diagonal and the shared group of distance 8, 14 packed ones -- instead of 34 plain ones of 128)
    valu = ((6 * 128.0 + 14 * 256.0) if pk else (2 + 4 * (mc // 2)) * 128.0) / 4 * t * f
    kname = f"cov_hmfma_kernel<{'true' if mc == 16 else 'false'}, true, {'true' if pk else 'false'}>"
    return _flop_roof(kname, "weighted spatial covariance of all sources in one pass, the sources on the fp32 matrix cores, overiva.py:179", "fp32",
                      MFMA_F32_PEAK_TFLOPS, mfma + valu, cov_ms, (t, f, m, k),
                      "achieved / frac count ISSUED flops (matrix instructions 2048 each, vector instructions 128) against the 157.3 TFLOP/s spec "
                      "peak, which on CDNA4 the fp32 matrix and vector instructions SHARE (measured: their times add); useful = Hermitian half "
                      "with each product formed once; naive = SURVEY 8d's 8 K M^2 T F.  " + _EVENT_NOTE,
                      useful=_useful_flops(t, f, m, k), parts=(mfma, valu), bytes_cov=cov_algorithmic_bytes(t, f, m, k))


def _roof_half16(t, f, m, k, mode, cov_ms):
    # the Hermitian half on the vector ALU, 32 lanes per (bin, frame), every source in one pass: bound by fp32 arithmetic
    np_ = 4 if k <= 8 else (6 if k <= 12 else 8)            # source PAIRS per lane of the instantiation
    issued = 160.0 * (2 + 2 * np_) * 4.0 * t * f            # 160 complex entry slots per (bin, frame) x packed instructions x 4 flop slots
    floor_ms = 32.0 * (10 + 10 * np_) * t * f / 64 / 1024 * 2.0e-6      # 1024 SIMDs, 2.0 ns per packed instruction and SIMD (tools/pkbench.hip)
    kname = f"cov_half16_kernel<{np_}, false>"
    return _flop_roof(kname, "weighted spatial covariance of all sources in one pass on the packed-fp32 vector ALU, overiva.py:179", "fp32",
                      MFMA_F32_PEAK_TFLOPS, issued, cov_ms, (t, f, m, k),
                      "achieved / frac count ISSUED flop slots of the packed vector ALU against the 157.3 TFLOP/s spec peak (vector = matrix peak on "
                      "CDNA4; the planar matrix-core form needs 2.8x the multiply-adds and measured 1.69 ms at 16 x 16); useful = Hermitian half with "
                      "each product formed once; naive = SURVEY 8d's 8 K M^2 T F.  Self-measured issue ceiling, for orientation only: 2.0 ns per "
                      f"packed instruction and SIMD (tools/pkbench.hip) = {floor_ms:.3f} ms for this launch",
                      useful=_useful_flops(t, f, m, k), bytes_cov=cov_algorithmic_bytes(t, f, m, k))


def _roof_hmfma64(t, f, m, k, mode, cov_ms):
    # the sources on the fp64 matrix cores (cov_hmfma64_kernel): 17 v_mfma_f64_16x16x4_f64 per bin and 4 frames + 2 + 8 x 6 float64 vector
    # instructions per lane (conversions and moves not counted); fp64 matrix peak = fp64 vector peak = 78.6 TFLOP/s
    mfma = 17 * 2048.0 / 4 * t * f
    valu = (2 + 4 * 8) * 128.0 / 4 * t * f
    return _flop_roof("cov_hmfma64_kernel", "weighted spatial covariance of all sources in one pass, float64, the sources on the fp64 matrix cores, "
                      "overiva.py:179", "fp64", FP64_VECTOR_PEAK_TFLOPS, mfma + valu, cov_ms, (t, f, m, k),
                      "issued float64 flops (matrix instructions 2048 each, vector multiplies / FMAs 128) against the 78.6 TFLOP/s fp64 peak (matrix = "
                      "vector on MI355X); the float64 vector-ALU kernel it replaces at 9..16 sources measured 2.04 ms at 16 x 16",
                      parts=(mfma, valu), bytes_cov=cov_algorithmic_bytes(t, f, m, k))


def _roof_half16_f64(t, f, m, k, mode, cov_ms):
    # the 32-lane form with float64 sums of exact products, four or eight sources per pass: bound by the float64 rate of the vector ALU
    # (14 conversions + 20 + 10 * sources float64 instructions per lane and frame)
    ns = 4 if k <= 4 else 8
    passes = -(-k // ns)
    floor_ms = 32.0 * (14 + 20 + 10 * ns) * passes * t * f / 64 / 1024 * 2.2e-6      # 2.2 ns per float64 instruction and SIMD (tools/pkbench.hip)
    issued = 32.0 * (20 + 10 * ns) * 2.0 * passes * t * f  # float64 FMA slots x 2 flops (conversions not counted)
    kname = f"cov_half16f64_kernel<{ns}>"
    return _flop_roof(kname, f"weighted spatial covariance, {ns} sources per pass over X, float64 on the vector ALU, overiva.py:179", "fp64",
                      FP64_VECTOR_PEAK_TFLOPS, issued, cov_ms, (t, f, m, k),
                      "float64 FMA slots of the vector ALU against its 78.6 TFLOP/s spec peak; the fp64 matrix-core form it replaces measured 3.1 ms at "
                      f"16 x 16.  Self-measured issue ceiling, for orientation only: 2.2 ns per float64 instruction and SIMD = {floor_ms:.3f} ms",
                      bytes_cov=cov_algorithmic_bytes(t, f, m, k) + (passes - 1) * 8 * t * f * m)


def _roof_planar(t, f, m, k, mode, cov_ms):
    # the planar matrix-core form: 3 MFMAs of 16x16x4 per 4 frames and source
    kname = "cov_mfma16_kernel<double, 16>" if mode == "precise" else "cov_mfma16_kernel<float, 16>"
    return _flop_roof(kname, "planar v_mfma_f32_16x16x4_f32, overiva.py:179", "mfma", MFMA_F32_PEAK_TFLOPS, 6.0 * k * m * m * t * f, cov_ms, (t, f, m, k),
                      "achieved/frac count ISSUED matrix-core flops; the naive-complex figure is algorithmic speed only; " + _EVENT_NOTE)


def _hmfma_on():
    return os.environ.get("OIVA_COV_HMFMA", "1") != "0"


_COV_FAMILIES = (
    (lambda t, f, m, k, mode, fused: m <= 8 and fused, _roof_cov_update),
    (lambda t, f, m, k, mode, fused: m <= 8, _roof_stream8),
    (lambda t, f, m, k, mode, fused: mode != "precise" and (k <= 2 or (k <= 4 and mode == "mixed")), _roof_quad),
    (lambda t, f, m, k, mode, fused: mode != "precise" and k >= 9 and m + m % 2 >= 10 and _hmfma_on(), _roof_hmfma),
    (lambda t, f, m, k, mode, fused: mode != "precise" and k > 4, _roof_half16),
    (lambda t, f, m, k, mode, fused: mode == "precise" and k >= 9 and m + m % 2 == 16 and _hmfma_on(), _roof_hmfma64),
    (lambda t, f, m, k, mode, fused: mode == "precise" and k >= 3, _roof_half16_f64),
    (lambda t, f, m, k, mode, fused: True, _roof_planar),
)


def _cov_roofline(shape, mode, cov_ms, fused=False):
    t, f, m, k = shape
    for applies, build in _COV_FAMILIES:
        if applies(t, f, m, k, mode, fused):
            return build(t, f, m, k, mode, cov_ms)


def _rates(steps, dts):
    r = [steps / d for d in dts]
    return {"value_median": _median(r), "value_min": min(r), "value_max": max(r), "repeats": len(r)}


def _time_resident(plan, args, repeats):
    """the X-resident kernel: W warm-up iterations, then K timed ones = ONE persistent launch, `repeats` more of the same"""
    plan.iterate(args.warmup)
    plan.sync()
    dts = []
    for _ in range(1 + repeats):
        t0 = time.perf_counter()
        plan.iterate(args.steps)
        plan.sync()
        dts.append(time.perf_counter() - t0)
    return dts


def _secondary_config(torch, oa, dev, name, args):
    """one of the other BASELINE configs, same protocol as the headline (and `--repeats` more measurements of the same K steps
    for a median): the four-launch path (graph replay) and, where the shape qualifies, the X-resident kernel; the faster one
    is `value`.  shard8 also runs the kernel's multi-GPU exchange in loop-back (this GPU plays all 8 ranks)."""
    c = CONFIGS[name]
    shape = (c["T"], c["F"], c["M"], c["K"])
    mode = "precise" if name == "cfg0" else ("mixed" if c["M"] <= 8 or name == "m16k2" else args.cfg5_precision)
    X = synth_x_device(torch, dev, 0, c["F"], shape[:3])
    torch.cuda.synchronize()
    out = {"workload": c["name"].format(**c), "precision": mode, "steps": args.steps, "warmup": args.warmup}
    plan = _make_plan(oa, X, shape, mode, True)
    dt, total_ms, stages, more = _time_plan(plan, args, repeats=args.repeats)
    info = plan.resident_info()
    one_launch = plan.set_fuse_cov_update(None)
    plan.close()
    cov_ms = stages["weighted_cov"] / args.steps
    _, roof = _cov_roofline(shape, mode, cov_ms, one_launch)
    if name in ("cfg2", "cfg0"):
        roof["note"] = "16 MB of X: resident in L2 / Infinity Cache, the pass is launch- and latency-bound, not a roofline claim"
    out["four_launch"] = {"value": args.steps / dt, "unit": "iterations/s", "ms_per_step": dt / args.steps * 1e3, **_rates(args.steps, [dt] + more),
                          "stage_ms_per_step": {k: v / args.steps for k, v in stages.items()}, "roofline": roof}
    out["value"], out["path"] = args.steps / dt, "four launches per iteration (hipGraph replay)"
    best = out["four_launch"]
    if info["qualifies"]:
        note = {"bound": "latency", "frac": None,
                "note": "X stays in registers + LDS for the whole launch: per iteration the kernel moves only the exchange words (parts, "
                        "partial covariances, demixing vectors) through L2; what bounds it is the dependency chain power -> r -> V -> W "
                        "across workgroups (phase_us_workgroup0)"}

        def resident_entry(loopback):
            plan = _make_plan(oa, X, shape, mode, False)
            if loopback:
                plan.resident_loopback(loopback)
            plan.set_resident(True)
            dts = _time_resident(plan, args, args.repeats)
            phases, nit = plan.resident_phases()
            inf = plan.resident_info()
            plan.close()
            return {"value": args.steps / dts[0], "unit": "iterations/s", "ms_per_step": dts[0] / args.steps * 1e3, **_rates(args.steps, dts),
                    "phase_us_workgroup0": phases, "grid": [inf["bin_groups"], inf["frame_splits"]],
                    "frames_per_lane": inf["frames_per_lane"], "frames_in_registers": inf["frames_in_registers"],
                    "lds_bytes": inf["lds_bytes"], "x_bytes_per_cu": inf["x_bytes_per_cu"], "fallbacks": inf["fallbacks"], "roofline": note}

        out["resident"] = resident_entry(0)
        if out["resident"]["fallbacks"] == 0 and out["resident"]["value"] > out["value"]:
            out["value"], out["path"] = out["resident"]["value"], "X-resident persistent launch (one launch for all the timed iterations)"
            best = out["resident"]
        if name == "shard8":
            try:
                out["resident_loopback8"] = resident_entry(8)
                out["resident_loopback8"]["what"] = ("the kernel's MULTI-GPU exchange on one GPU: per frame split a leader workgroup gathers the "
                                                     "rank's parts and stores the sums of 8 slots (its own, zeros for the 7 phantom ranks) into a "
                                                     "fine-grained gather buffer; every workgroup polls its words of all 8 slots and adds them in rank "
                                                     "order -- what every rank of `bench.py --gpus 8` runs, minus the flight over xGMI")
            except Exception as e:
                out["resident_loopback8"] = {"error": f"{type(e).__name__}: {e}"}
    if name in ("shard2", "shard4"):
        # what every rank of `bench.py --gpus 2 / 4` runs: the four-launch iteration replayed from graphs with the exchange of the
        # ranks' partial powers inside the activation kernel -- here in loop-back (this GPU plays all ranks: everything but the
        # flight of the peer stores over xGMI)
        world = 2 if name == "shard2" else 4
        try:
            plan = _make_plan(oa, X, shape, mode, False)
            plan.fused_loopback(world)
            plan.use_graph(True)
            dtf, totf, stagesf, moref = _time_plan(plan, args, repeats=args.repeats)
            plan.close()
            out["fused_loopback"] = {"world": world, "value": args.steps / dtf, "unit": "iterations/s", "ms_per_step": dtf / args.steps * 1e3,
                                     **_rates(args.steps, [dtf] + moref), "stage_ms_per_step": {k: v / args.steps for k, v in stagesf.items()},
                                     "what": "four kernels per iteration replayed from hipGraphs, no host call and no collective in the loop: the "
                                             "activation kernel adds the rank's parts, stores the sums into the other ranks' slots, polls its own "
                                             f"buffer and adds the {world} ranks' sums in rank order"}
            out["value"], out["path"] = out["fused_loopback"]["value"], f"four launches per iteration (hipGraph replay) with the in-kernel exchange, loop-back world {world}"
            best = out["fused_loopback"]
        except Exception as e:
            out["fused_loopback"] = {"error": f"{type(e).__name__}: {e}"}
    if name == "cfg5" and mode == "mixed" and os.environ.get("OIVA_HMFMA_PART32") is None:
        # the opt-in form of this config: the covariance kernel's partial blocks cross HBM as float32 (half the bytes that bound the
        # per-bin update; one more rounding per block: DESIGN.md 3) -- reported beside `value`, never as it
        os.environ["OIVA_HMFMA_PART32"] = "1"
        try:
            plan = _make_plan(oa, X, shape, mode, True)
            dt32, _, stages32, more32 = _time_plan(plan, args, repeats=args.repeats)
            plan.close()
            out["float32_partial_blocks"] = {"value": args.steps / dt32, "unit": "iterations/s", "ms_per_step": dt32 / args.steps * 1e3,
                                             **_rates(args.steps, [dt32] + more32), "stage_ms_per_step": {k: v / args.steps for k, v in stages32.items()},
                                             "what": "$OIVA_HMFMA_PART32=1: not the default arithmetic of overiva()"}
        except Exception as e:
            out["float32_partial_blocks"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            del os.environ["OIVA_HMFMA_PART32"]
    out.update({k: best[k] for k in ("value_median", "value_min", "value_max", "repeats")})
    out["unit"] = "iterations/s"
    return out


def run_single(args):
    import torch

    import overiva_amd as oa

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    shape = (T, F, M, K)
    X = synth_x_device(torch, dev, 0, F)
    torch.cuda.synchronize()

    other_modes = []
    if not args.no_other_mode:
        # the other arithmetic modes on the same workload, same timing protocol.  Measured before the metric's own mode:
        # a process's first tens of milliseconds of GPU work run 2-3 % slower (clock ramp), and that should not land in
        # `value`.
        for other in MODES:
            if other == args.precision:
                continue
            plan = _make_plan(oa, X, shape, other, args.graph)
            dt2, total2, stages2, more2 = _time_plan(plan, args, repeats=args.repeats)
            one_launch2 = plan.set_fuse_cov_update(None)
            plan.close()
            cov2 = stages2["weighted_cov"] / args.steps
            _, roof2 = _cov_roofline(shape, other, cov2, one_launch2)
            other_modes.append({"precision": other, "value": args.steps / dt2, "unit": "iterations/s", "ms_per_step": dt2 / args.steps * 1e3,
                                **_rates(args.steps, [dt2] + more2),
                                "stage_ms_per_step": {k: v / args.steps for k, v in stages2.items()},
                                "roofline": {k: roof2[k] for k in ("bound", "kernel", "achieved", "unit", "frac")}})
    plan = _make_plan(oa, X, shape, args.precision, args.graph)
    dt, total_ms, stages, more = _time_plan(plan, args, repeats=args.repeats)
    pre_pass = _PRE_PASS_STEPS
    W = plan.get_w()
    assert np.all(np.isfinite(W))
    splits = plan.cov_splits()
    one_launch = plan.set_fuse_cov_update(None)
    digests = None
    if args.w_digest > 0:
        import hashlib

        from overiva_amd.sharded import shard_bounds
        plan.set_w(None)
        plan.iterate(args.w_digest)
        Wd = plan.get_w(np.complex64 if args.precision == "fast" else np.complex128)
        b = shard_bounds(F, args.digest_shards)
        digests = [hashlib.sha256(np.ascontiguousarray(Wd[b[g]:b[g + 1]]).tobytes()).hexdigest()[:16] for g in range(args.digest_shards)]
        if args.w_dump:
            np.save(os.path.join(args.w_dump, "w_rank0of1.npy"), Wd)
    plan.close()
    cov_ms = stages["weighted_cov"] / args.steps
    per_step = {k: v / args.steps for k, v in stages.items()}
    kname, roofline = _cov_roofline(shape, args.precision, cov_ms, one_launch)
    if M <= 8:
        traffic, traffic_src = measured_traffic(kname)
        roofline.update({"traffic": traffic, "traffic_source": traffic_src})
        # the whole iteration against the same roofline: X is read twice (demix/power pass, covariance pass)
        bytes_iter = roofline["algorithmic_bytes_per_launch"] + 8 * T * F * M + 4 * T * K * (1 + F // 64)
        roofline["iteration"] = {"algorithmic_bytes": bytes_iter, "achieved": bytes_iter / (total_ms / args.steps * 1e-3) / 1e9,
                                 "frac": bytes_iter / (total_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
    roofline.update({"stage_ms_per_step": per_step, "event_timed_ms_per_step": total_ms / args.steps, "cov_splits": splits})
    out = result_line(args, 1, dt)
    out["untimed_pre_pass_steps"] = pre_pass        # (see _time_plan: >= 50 ms of untimed iterations in front of the W warm-up steps)
    if digests is not None:
        out["w_digest_shards"] = {"iterations": args.w_digest, "shards": args.digest_shards, "sha256_16": digests}
    if more:
        out.update(_rates(args.steps, [dt] + more))
    out["roofline"] = roofline
    # (the driver's record keeps the scalars of `config`: the other arithmetic modes' rates on the same workload go there too)
    for om in other_modes:
        out["config"][f"iterations_per_s_{om['precision']}"] = om["value"]
    if other_modes:
        out["other_modes"] = other_modes
    del X
    torch.cuda.empty_cache()
    if not args.no_configs and args.config == "headline":
        out["configs"] = {}
        for name in ("cfg0", "cfg2", "shard2", "shard4", "shard8", "cfg5", "m16k2"):
            try:
                out["configs"][name] = _secondary_config(torch, oa, dev, name, args)
                out["config"][f"iterations_per_s_{name}"] = out["configs"][name]["value"]
                lb = out["configs"][name].get("resident_loopback8", {}).get("value")
                if lb:
                    out["config"][f"iterations_per_s_{name}_loopback8"] = lb
                f32 = out["configs"][name].get("float32_partial_blocks", {}).get("value")
                if f32:
                    out["config"][f"iterations_per_s_{name}_float32_partial_blocks_optin"] = f32
            except Exception as e:  # a secondary shape must not cost the headline line
                out["configs"][name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    out["cpu_baseline"] = cpu_baseline() if not args.no_cpu else None
    if out["cpu_baseline"]:
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def _watchdog(seconds, what):
    """a run that does not finish (a rank waiting for a dead peer inside a collective, a stream wait that is never
    satisfied) must end with a non-zero exit code, not hang: the timer thread ends the process"""
    def fire():
        sys.stderr.write(f"[bench] watchdog: {what} did not finish within {seconds} s; exiting with code 3\n")
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def run_sharded(args):
    import torch
    import torch.distributed as dist

    from overiva_amd.sharded import HipEngine, fused_blocks, shard_bounds

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    rank = int(os.environ.setdefault("RANK", "0"))
    world = int(os.environ.setdefault("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dog = _watchdog(args.launch_timeout, f"rank {rank} of the sharded run")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    b = shard_bounds(F, world)
    f0, f1 = b[rank], b[rank + 1]
    X = synth_x_device(torch, dev, f0, f1)
    torch.cuda.synchronize()
    eng = HipEngine(T, f1 - f0, M, K, MODEL, F, local, precision=args.precision)
    stream = eng.stream
    with eng.stream_ctx():
        eng.set_x_device(X.data_ptr(), keepalive=X)
        eng.covariance()
        eng.set_w(None)
        ppr = max(eng.power_parts(b[r + 1] - b[r]) for r in range(world))
        p_local = eng.exchange_buffer(ppr)
        p_all = eng.new_gather_buffer(world)

        # transport of the per-iteration all-gather: RCCL through torch.distributed (default), the library's push exchange
        # (opt-in; validated -- including a stream wait that really blocks -- before it is used, exchange.py), or none at
        # all: the X-resident kernel with the exchange inside it (opt-in; every rank's shard must fit on chip)
        from overiva_amd.exchange import make_exchange

        want_resident = args.exchange in ("resident", "auto")
        resident_refused = None
        if want_resident:
            resident_refused = eng.setup_resident(dist, None, rank, world) if args.precision != "precise" else "precise arithmetic"
        # shards that do not fit on chip (2 and 4 GPUs): the exchange inside the activation kernel of the four-launch iteration
        want_fused = args.exchange == "fused" or (args.exchange == "auto" and resident_refused is not None)
        fused_refused = eng.setup_fused(dist, None, rank, world, fused_blocks(b)) if want_fused else None
        fused = want_fused and fused_refused is None
        xchg = make_exchange(eng, dist, None, rank, world, p_local, p_all, prefer="collective" if (want_resident or want_fused) else args.exchange)
        nparts = world * ppr

        def step():
            eng.power()
            eng.update_ptr(xchg.gather(), nparts)

        if want_resident and resident_refused is None:
            # Validate the kernel's own exchange on THIS platform before it is timed: one iteration through it against one
            # through the collective, from the same start, on every rank.  (A launch whose waits run into their time-outs
            # returns an error without having written W; whatever happens, W is re-initialised afterwards.)
            why = None
            try:
                eng.plan.set_resident(False)
                step()
                stream.synchronize()
                w_coll = eng.get_w()
                eng.set_w(None)
                eng.plan.set_resident(True)
                eng.plan.iterate(1)
                w_res = eng.get_w()
                err = float(np.linalg.norm(w_res - w_coll) / max(np.linalg.norm(w_coll), 1e-30))
                if not err < 1e-4:
                    why = f"one iteration differs from the collective path by {err:.1e}"
            except Exception as e:
                why = f"{type(e).__name__}: {e}"
            notes = [None] * world
            dist.all_gather_object(notes, why)
            if any(n is not None for n in notes):
                resident_refused = "validation failed: " + "; ".join(f"rank {r}: {n}" for r, n in enumerate(notes) if n is not None)
                eng.plan.set_resident(False)
            eng.set_w(None)
            stream.synchronize()
        if want_resident and resident_refused is not None and rank == 0:
            print(f"[bench] X-resident exchange not used: {resident_refused}", file=sys.stderr)
        resident = want_resident and resident_refused is None
        if fused:
            # the same validation for the exchange inside the activation kernel: one iteration through it against one through the
            # collective, from the same start, on every rank (rank-by-rank association of the sum: agreement to rounding)
            why = None
            try:
                step()                  # (power pass + all-gather + update on the gathered parts: the plain activation kernel)
                stream.synchronize()
                w_coll = eng.get_w()
                eng.set_w(None)
                eng.plan.iterate(1)
                eng.plan.sync()
                w_f = eng.get_w()
                err = float(np.linalg.norm(w_f - w_coll) / max(np.linalg.norm(w_coll), 1e-30))
                if not err < 1e-4:
                    why = f"one iteration differs from the collective path by {err:.1e}"
            except Exception as e:
                why = f"{type(e).__name__}: {e}"
            notes = [None] * world
            dist.all_gather_object(notes, why)
            if any(n is not None for n in notes):
                fused_refused = "validation failed: " + "; ".join(f"rank {r}: {n}" for r, n in enumerate(notes) if n is not None)
                fused = False
                try:
                    eng.plan.fused_connect(None)
                except Exception:
                    pass
            eng.set_w(None)
            stream.synchronize()
        if want_fused and not fused and rank == 0:
            print(f"[bench] exchange inside the activation kernel not used: {fused_refused}; collective", file=sys.stderr)

        graph = None
        if resident or fused:
            # resident: one persistent launch per call; fused: captured graphs of four kernels per iteration, no host call in
            # between.  W warm-up iterations, then exactly K timed ones
            if fused:
                eng.plan.use_graph(True)
            # the protocol of _time_plan, so that N = 1 through this leg and the single-GPU line agree: an untimed pass of the same K
            # steps until >= 50 ms of iterations have run (the same number on every rank: rank 0's), then W, then K timed
            global _PRE_PASS_STEPS
            t_pre = time.perf_counter()
            n_pre = 0
            while True:
                eng.plan.iterate(args.steps)
                stream.synchronize()
                n_pre += args.steps
                go_on = [time.perf_counter() - t_pre < 0.05 and n_pre < 100000]
                if world > 1:
                    dist.broadcast_object_list(go_on, src=0)
                if not go_on[0]:
                    break
            _PRE_PASS_STEPS = n_pre
            eng.plan.iterate(args.warmup)
            stream.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.plan.iterate(args.steps)
            stream.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if resident:
                phases, _ = eng.plan.resident_phases()
                breakdown = {"resident_phase_us_workgroup0": phases}
                cov_ms = None
            else:
                breakdown = {}
                cov_ms = eng.plan.t_time_stage("weighted_cov", 20)
                stream.synchronize()
        else:
            for _ in range(args.warmup):
                step()
            stream.synchronize()
            # per-rank stage breakdown (eager, events on the shared stream): power pass | all-gather | activation +
            # covariance + update
            # roofline of the dominant kernel on this rank's shard: the covariance kernel alone, HIP events on the plan's
            # stream (same definition as the single-GPU line, with this rank's number of bins).  Like the stage breakdown
            # it runs before the wall-clock measurement, which then does not start on a cold GPU.
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            acc = [0.0, 0.0, 0.0]
            nb = max(5, args.steps)
            for _ in range(nb):
                ev[0].record(stream)
                eng.power()
                ev[1].record(stream)
                gathered = xchg.gather()
                ev[2].record(stream)
                eng.update_ptr(gathered, nparts)
                ev[3].record(stream)
                stream.synchronize()
                for i in range(3):
                    acc[i] += ev[i].elapsed_time(ev[i + 1]) / nb
            breakdown = {"power_ms": acc[0], "all_gather_ms": acc[1], "activation_cov_update_ms": acc[2]}
            cov_ms = eng.plan.t_time_stage("weighted_cov", 20)
            stream.synchronize()
            spg = 8            # iterations per captured graph (kernels + the RCCL all-gather)
            # The eager loop costs the host ~45 us per step against 30-50 us of device time at 8 GPUs, so the collective path
            # replays captured graphs (kernels + all-gather) -- after the capture has proved itself on THIS platform with THESE
            # ranks: spg steps eagerly, then from the same start one replay; every rank must see the same bits.  Anything
            # else (capture refused, a rank that differs) leaves the run eager, with the reason in ranks.graph_refused.
            # (--graph 0: never)
            graph_refused = None
            if args.graph >= 1 and args.steps >= spg and xchg.name == "collective" and args.backend == "nccl":
                why = None
                try:
                    eng.set_w(None)
                    for _ in range(spg):
                        step()
                    stream.synchronize()
                    w_eager = eng.get_w()
                    eng.set_w(None)
                    stream.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=stream):
                        for _ in range(spg):
                            step()
                    graph.replay()
                    stream.synchronize()
                    w_graph = eng.get_w()
                    if not np.array_equal(w_eager, w_graph):
                        why = "a replayed graph of the collective path differs from the eager steps"
                except Exception as e:  # capture of the collective not supported: stay eager
                    why = f"{type(e).__name__}: {e}"
                notes = [None] * world
                dist.all_gather_object(notes, why)
                if any(n is not None for n in notes):
                    graph_refused = "; ".join(f"rank {r}: {n}" for r, n in enumerate(notes) if n is not None)
                    graph = None
                    if rank == 0:
                        print(f"[bench] graph replay of the collective path not used ({graph_refused}); eager", file=sys.stderr)
                eng.set_w(None)
                for _ in range(args.warmup):
                    step()
                stream.synchronize()
            breakdown["graph_refused"] = graph_refused
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = 0
            if graph is not None:
                while done + spg <= args.steps:
                    graph.replay()
                    done += spg
            while done < args.steps:       # remainder (or everything, when eager): exactly K steps in total
                step()
                done += 1
            stream.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    dist.barrier()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    W = eng.get_w()
    assert np.all(np.isfinite(W))
    if args.w_digest > 0:
        # N iterations from the identity start through the transport that was timed; this rank's bins
        import hashlib
        with eng.stream_ctx():
            eng.set_w(None)
            stream.synchronize()
            dist.barrier()
            if resident or fused:
                eng.plan.iterate(args.w_digest)
            else:
                for _ in range(args.w_digest):
                    step()
            stream.synchronize()
        Wd = np.ascontiguousarray(eng.get_w())
        breakdown["w_digest"] = hashlib.sha256(Wd.tobytes()).hexdigest()[:16]
        if args.w_dump:
            np.save(os.path.join(args.w_dump, f"w_rank{rank}of{world}.npy"), Wd)
    one_launch = eng.plan.set_fuse_cov_update(None)
    exchange_name, exchange_fallback = xchg.name, getattr(xchg, "fallback_reason", None)
    xchg.close()
    eng.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "bins": [f0, f1], **breakdown})
    out = result_line(args, world, float(tmax.item()))
    out["config"]["graph"] = graph is not None
    fl = f1 - f0
    if resident:
        out["roofline"] = {"bound": "latency", "frac": None, "traffic": None, "per": "GPU",
                           "kernel": f"resident_kernel<{M}, {K}, ...> on rank 0's {fl} bins: the whole iteration in one persistent launch",
                           "note": "X stays in registers + LDS for the whole launch; what bounds the iteration is the dependency chain "
                                   "power -> (exchange) -> r -> V -> W across workgroups and GPUs (per_rank_stage_ms)"}
        out["config"]["parallelism"] = (f"bins sharded over {world} GPU(s); X-resident persistent kernel per rank, the partial source powers "
                                        "exchanged inside it by peer stores over xGMI (no collective in the loop)")
    else:
        _, roof = _cov_roofline((T, fl, M, K), args.precision, cov_ms, one_launch)
        roof["kernel"] += f" on rank 0's {fl} bins"
        roof["per"] = "GPU"
        roof.setdefault("traffic", None)
        out["roofline"] = roof
        if fused:
            out["config"]["graph"] = True
            out["config"]["parallelism"] = (f"bins sharded over {world} GPU(s); four kernels per iteration replayed from hipGraphs, the partial source "
                                            "powers exchanged inside the activation kernel by peer stores over xGMI (no collective, no host call in the loop)")
    out["cpu_baseline"] = None      # reported at N = 1 only
    if resident or fused:
        out["untimed_pre_pass_steps"] = _PRE_PASS_STEPS   # (as the single-GPU line: >= 50 ms of untimed iterations in front of the W warm-up steps)
    # a fast exchange that was asked for (or implied by `auto`) and refused is a DEGRADED run: said at the top level, not
    # only among the per-rank details
    degraded = None if (resident or fused or args.exchange in ("collective", "push")) else (fused_refused or resident_refused or "refused")
    if args.exchange == "push" and exchange_name != "push":
        degraded = exchange_fallback or "push exchange refused"
    out["exchange_degraded"] = degraded
    if degraded and rank == 0:
        print(f"[bench] DEGRADED: the in-kernel exchange was not used ({degraded}); this line measures the collective path", file=sys.stderr)
    out["ranks"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                    "exchange": "resident" if resident else ("fused" if fused else exchange_name),
                    "exchange_requested": args.exchange,
                    "fallback": None if (resident or fused) else (fused_refused or resident_refused or exchange_fallback),
                    "resident_refused": resident_refused, "fused_refused": fused_refused,
                    "per_rank_stage_ms": gathered, "message_bytes_per_rank": int(p_local.numel() * 4)}
    dist.destroy_process_group()
    dog.cancel()
    return out if rank == 0 else None


def result_line(args, n_gpus, seconds):
    return {
        "metric": f"AuxIVA iterations/sec ({F} bins x {T} frames x {M} mics)",
        "value": args.steps / seconds,
        "unit": "iterations/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": seconds / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        # the arithmetic type of the streaming passes' products and lane chains ("precise": float64 throughout)
        "dtype": "f64" if args.precision == "precise" else "f32",
        "data": "synthetic",
        "config": {"workload": WORKLOAD, "bins": F, "frames": T, "mics": M, "sources": K, "model": MODEL,
                   "precision": MODE_TEXT[args.precision],
                   "parallelism": f"bins sharded over {n_gpus} GPU(s), one RCCL all-gather of (T,K) f32 per iteration"
                   if n_gpus > 1 else "single GPU", "graph": bool(args.graph)},
    }


# ---- `--gpus N` without a launcher ------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _strip_option(argv, name):
    """argv without `name VALUE` / `name=VALUE`"""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == name:
            skip = True
            continue
        if a.startswith(name + "="):
            continue
        out.append(a)
    return out


def _stop_children(p):
    """end a launcher child and the ranks it started -- torch.distributed.run puts every rank into a session of its own, so
    the process group of the launcher does not reach them: SIGTERM to the launcher (its agent then stops its workers),
    and after a grace period SIGKILL to exactly the processes that descend from it.  Returns what it had written."""
    kids = []
    try:
        import psutil

        kids = psutil.Process(p.pid).children(recursive=True)
    except Exception:
        pass
    p.terminate()
    try:
        out, _ = p.communicate(timeout=15)
    except subprocess.TimeoutExpired:
        out = None
    for k in kids:
        try:
            k.kill()
        except Exception:
            pass
    if p.poll() is None:
        try:
            os.killpg(p.pid, signal.SIGKILL)      # exactly the process group started by launch_ranks
        except ProcessLookupError:
            pass
    if out is None:
        try:
            out, _ = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            out = ""
    return out


def launch_ranks(args, argv, script=None):
    """Start the N ranks as children (python -m torch.distributed.run) and relay rank 0's JSON line.  Runs before this
    process has made any GPU call; never replaces this process.  Every attempt is bounded by --launch-timeout: on expiry the
    children's process group is killed.  An attempt with the push exchange that fails or times out is repeated with the
    collective, in fresh processes."""
    attempts = []
    exchanges = [args.exchange] + (["collective"] if args.exchange != "collective" else [])      # second attempt: the plain collective
    base = _strip_option(argv, "--exchange")
    for ex in exchanges:
        port = _free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + base + ["--exchange", ex]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver supports dmabuf IPC only (RCCL, the push exchange)
        t0 = time.perf_counter()
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True, start_new_session=True)
        status = "ok"
        try:
            out, _ = p.communicate(timeout=args.launch_timeout + args.launch_grace)     # the ranks' own watchdogs fire first
        except subprocess.TimeoutExpired:
            status = "timeout"
            out = _stop_children(p)
        lines = [l for l in (out or "").splitlines() if l.startswith("{")]
        rec = {"exchange": ex, "status": status, "returncode": p.returncode, "seconds": round(time.perf_counter() - t0, 1)}
        attempts.append(rec)
        if status == "ok" and p.returncode == 0 and lines:
            d = json.loads(lines[-1])
            d["launcher"] = {"spawned_ranks": args.gpus, "attempts": attempts}
            print(json.dumps(d), flush=True)
            return 0
        sys.stderr.write(f"[bench] attempt with exchange={ex} failed: {rec}\n")
    sys.stderr.write(f"[bench] no attempt produced a result: {attempts}\n")
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", type=int, default=1,
                    help="0: eager; 1 (default): hipGraph replay on a single GPU, eager when sharded; 2: graph also when sharded")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary configs (cfg0, cfg2, shard8, cfg5, m16k2) of the N = 1 line")
    ap.add_argument("--repeats", type=int, default=4, help="extra measurements of the same K steps for value_median (N = 1)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline",
                    help="headline: BASELINE.json configs[2] (the metric's workload); cfg5: configs[4], 16 mics / 16 sources; "
                         "cfg2: configs[1]; shard8: one rank's shard of configs[3]; m16k2: 16 mics / 2 sources; tiny: test-only")
    ap.add_argument("--precision", choices=list(MODES), default=None,
                    help="arithmetic of the timed run (default: mixed, what overiva() runs on complex64 input); the other modes "
                         "are timed too")
    ap.add_argument("--cfg5-precision", choices=list(MODES), default="mixed",
                    help="arithmetic of the configs[4] entry of the N = 1 line (default: what overiva() runs on it)")
    ap.add_argument("--no-other-mode", action="store_true", help="do not also time the other arithmetic modes")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (exercises RCCL + graph capture on 1 GPU)")
    ap.add_argument("--exchange", choices=["auto", "collective", "push", "resident", "fused"], default=os.environ.get("OIVA_EXCHANGE", "auto"),
                    help="exchange of the partial powers when sharded.  auto (default): inside the X-resident kernel where every "
                         "rank's shard fits on chip (the headline shape at 8 GPUs), else inside the activation kernel of the "
                         "four-launch iteration (2 and 4 GPUs) -- each only if one iteration through it reproduces the collective "
                         "path on this platform --, else torch.distributed's collective (RCCL); collective; push: the library's "
                         "push exchange (validated at start-up, falls back to the collective); resident / fused: only that one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the sharded path (nccl = RCCL; tests use gloo)")
    ap.add_argument("--single-device", action="store_true",
                    help="tests on a 1-GPU box: every rank uses GPU 0 (needs --backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--w-digest", type=int, default=0, metavar="N",
                    help="after the measurement: N iterations from the identity start and sha256 of the demixing matrices, per rank (N > 1 GPUs: "
                         "ranks.per_rank_stage_ms[*].w_digest) or per shard of --digest-shards equal bin ranges (N = 1: w_digest_shards) -- "
                         "equal shards through the four-launch path give the bits of one GPU")
    ap.add_argument("--digest-shards", type=int, default=1)
    ap.add_argument("--w-dump", default=None, metavar="DIR", help="with --w-digest: also save those matrices as DIR/w_rank<r>of<N>.npy (tests)")
    ap.add_argument("--launch-timeout", type=int, default=600,
                    help="seconds a sharded run may take before its watchdog ends it with a non-zero exit code")
    ap.add_argument("--launch-grace", type=int, default=60, help=argparse.SUPPRESS)
    args = ap.parse_args()
    select_config(args.config)
    if args.precision is None:
        args.precision = "mixed"      # overiva_amd.overiva.resolve_precision for complex64 input
    world_env = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and world_env == 0:
        # no launcher around us: be one.  Nothing above touched a GPU (no torch import, no HIP call).
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.gpus > 1 or world_env > 1 or args.force_sharded:
        out = run_sharded(args)
    else:
        out = run_single(args)
    if out is not None:
        # the JSON line must be the LAST thing on stdout: RCCL writes its version banner to C stdio, which would
        # otherwise be flushed at exit, i.e. after it
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
