#!/usr/bin/env python3
"""Benchmark of the AuxIVA / OverIVA iteration hot path on MI355X.

Metric (BASELINE.json): AuxIVA iterations/sec at 2048 bins x 4000 frames x 8 mics / 2 sources
(OverIVA, laplace), synthetic complex64 STFT, at 1/2/4/8 GPUs.  One "step" = one iteration of the
loop body reference overiva.py:138-190; prologue and epilogue are outside the timed region and X is
resident in HBM when it starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 shards the 2048 bins over the ranks (strong scaling: the problem is fixed) with one RCCL
all-gather of the (T, K) partial source powers per iteration.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- weighted-covariance pass: algorithmic bytes (8TFM + 4TK + 8FKM^2) /
                  average kernel duration measured with HIP events on the kernel's own stream
  cpu_baseline -- the oracle's reference-faithful NumPy restatement of overiva.py timed on this
                  box's host cores on a bounded sample (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MODEL = "laplace"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
# --config: the headline workload (default; the one BASELINE.json's metric is quoted on) and configs[4]
CONFIGS = {
    "headline": dict(T=4000, F=2048, M=8, K=2, name="OverIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (BASELINE.json configs[2])"),
    "cfg5": dict(T=4000, F=2048, M=16, K=16, name="determined AuxIVA {F} bins x {T} frames x {M} mics / {K} src, laplace, complex64 (BASELINE.json configs[4])"),
}
T, F, M, K = 4000, 2048, 8, 2
WORKLOAD = CONFIGS["headline"]["name"].format(T=T, F=F, M=M, K=K)


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
    profiles/r01_pmc_hbm_traffic.json).  PMC counters need rocprofv3 around the process, so they cannot
    be sampled from inside a plain bench run; None when the profile is absent."""
    for name in ("r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        path = os.path.join(REPO, "profiles", name)
        try:
            with open(path) as f:
                return float(json.load(f)["kernels"][kernel_key]["hbm_bytes_per_launch"]), os.path.relpath(path, REPO)
        except Exception:
            continue
    return None, None


def synth_x_device(torch, device, f0, f1):
    """The iid complex64 workload, generated on the device (seeded per bin so that any sharding of
    the bins sees the same tensor)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    x = torch.randn((T, F, M, 2), generator=g, device=device, dtype=torch.float32)
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


def _time_oracle(fs):
    """seconds per iteration of the reference-faithful oracle on `fs` bins: (t(5 its) - t(1 it)) / 4, so the
    prologue (input covariance, allocation) cancels; best of two after a warm-up call"""
    from oracle import overiva_oracle as orc

    X = orc.synth_iid(T, fs, M, seed=0)
    orc.overiva_faithful(X, n_src=K, n_iter=1, proj_back=False, model=MODEL)      # warm-up (page faults, BLAS threads)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        orc.overiva_faithful(X, n_src=K, n_iter=1, proj_back=False, model=MODEL)
        t1 = time.perf_counter()
        orc.overiva_faithful(X, n_src=K, n_iter=5, proj_back=False, model=MODEL)
        t2 = time.perf_counter()
        per = ((t2 - t1) - (t1 - t0)) / 4.0
        best = per if best is None else min(best, per)
    return max(best, 1e-6)


def cpu_baseline():
    """Reference-faithful NumPy restatement (oracle) on a bounded sample: 1024 of the 2048 bins, all 4000
    frames, 8 mics / 2 src (about 15 s of CPU work); work is linear in bins, so it/s at 2048 bins = it/s on the sample
    * 1024 / 2048.
    Timed with the default BLAS threading and, as the reference's own sweep pinned BLAS to one thread
    (overiva_sim.py:85-91), once more on a smaller sample with one thread."""
    fs = 1024
    per_iter = _time_oracle(fs)
    threads = os.cpu_count()
    one = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
        fs1 = fs        # same sample: a smaller one would sit in cache and flatter the CPU
        with threadpool_limits(limits=1):
            per1 = _time_oracle(fs1)
        one = {"value": (1.0 / per1) * fs1 / F, "cores": 1,
               "sample": f"{fs1} of {F} bins, one BLAS thread, {per1:.3f} s per iteration on the sample"}
    except Exception:
        pass
    # `cores` = the threads that actually did the work: the batched (M x T)(T x M) products of overiva.py:179 do not
    # thread in OpenBLAS (one thread gives the same rate, see single_thread), so however many BLAS threads the run is
    # allowed (blas_threads_allowed) it is one core
    effective = 1 if one and one["value"] > 0.8 * (1.0 / per_iter) * fs / F else threads
    return {"value": (1.0 / per_iter) * fs / F, "unit": "iterations/s", "cores": effective, "kind": "port",
            "blas_threads_allowed": threads,
            "cpu": _cpu_model(), "host_cpus": os.cpu_count(), "single_thread": one,
            "sample": f"oracle.overiva_faithful (NumPy, complex64 in / float64 r like overiva.py) on {fs} of {F} bins x "
                      f"{T} frames x {M} mics / {K} src, iterations 2-5, scaled by {fs}/{F}; "
                      f"{per_iter:.3f} s per iteration on the sample"}


def _time_plan(plan, args):
    """W untimed warm-up iterations, then exactly K timed ones bracketed by synchronisation; then the same K again
    with every kernel bracketed by HIP events on the plan's stream"""
    import torch

    plan.iterate(args.warmup)
    plan.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.iterate(args.steps)
    plan.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total_ms, stages = plan.iterate_timed(args.steps, per_kernel=True)
    return dt, total_ms, stages


def run_single(args):
    import torch

    import overiva_amd as oa

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    X = synth_x_device(torch, dev, 0, F)
    torch.cuda.synchronize()

    def make(mode):
        plan = oa.Plan(T, F, M, K, MODEL, device=0)
        plan.set_precision(mode)
        plan.set_x_device(X.data_ptr(), keepalive=X)
        plan.covariance()
        plan.set_w(None)
        if args.graph:
            plan.use_graph(True)
        return plan

    other_mode = None
    if not args.no_other_mode:
        # the other arithmetic mode on the same workload, same timing protocol (reported next to the metric; the
        # drop-in overiva() defaults to "precise", the metric's float32 tolerance is met by "fast" on this input).
        # Measured before the metric's own mode: a process's first tens of milliseconds of GPU work run 2-3 % slower
        # (clock ramp), and that should not land in `value`.
        other = "precise" if args.precision == "fast" else "fast"
        plan = make(other)
        dt2, total2, stages2 = _time_plan(plan, args)
        plan.close()
        other_mode = {"precision": other, "value": args.steps / dt2, "unit": "iterations/s", "ms_per_step": dt2 / args.steps * 1e3,
                      "stage_ms_per_step": {k: v / args.steps for k, v in stages2.items()}}
    plan = make(args.precision)
    dt, total_ms, stages = _time_plan(plan, args)
    W = plan.get_w()
    assert np.all(np.isfinite(W))
    splits = plan.cov_splits()
    plan.close()
    cov_ms = stages["weighted_cov"] / args.steps
    per_step = {k: v / args.steps for k, v in stages.items()}
    if M <= 8:
        bytes_cov = cov_algorithmic_bytes(T, F, M, K)
        achieved = bytes_cov / (cov_ms * 1e-3) / 1e9
        kname = f"cov_dma_kernel<{M}, {min(K, 2)}>" if args.precision == "fast" else f"cov_gram_kernel<{min(K, 2)}>"
        traffic, traffic_src = measured_traffic(kname)
        roofline = {"bound": "hbm", "kernel": f"{kname} (weighted spatial covariance of all sources in one pass, overiva.py:179)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_cov}
        # the whole iteration against the same roofline: X is read twice (demix/power pass, covariance pass)
        bytes_iter = bytes_cov + 8 * T * F * M + 4 * T * K * (1 + F // 64)
        roofline["iteration"] = {"algorithmic_bytes": bytes_iter, "achieved": bytes_iter / (total_ms / args.steps * 1e-3) / 1e9,
                                 "frac": bytes_iter / (total_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
    else:
        naive = 8.0 * K * M * M * T * F            # complex MACs counted as 8 real flops
        issued = 6.0 * K * M * M * T * F           # what the planar form issues: 3 MFMAs of 16x16x4 per 4 frames and source
        roofline = {"bound": "fp32-mfma", "kernel": "cov_mfma16_kernel<float, 16> (planar v_mfma_f32_16x16x4_f32, overiva.py:179)",
                    "achieved": issued / (cov_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": issued / (cov_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                    "issued_flops_per_launch": issued, "naive_complex_flops_per_launch": naive,
                    "naive_complex_tflops": naive / (cov_ms * 1e-3) / 1e12,
                    "note": "achieved/frac count ISSUED matrix-core flops; the naive-complex figure is algorithmic speed only"}
    roofline.update({"avg_launch_ms": cov_ms, "stage_ms_per_step": per_step, "event_timed_ms_per_step": total_ms / args.steps,
                     "cov_splits": splits})
    out = result_line(args, 1, dt)
    out["roofline"] = roofline
    if other_mode is not None:
        out["other_mode"] = other_mode
    out["cpu_baseline"] = cpu_baseline() if not args.no_cpu else None
    if out["cpu_baseline"]:
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def run_sharded(args):
    import torch
    import torch.distributed as dist

    from overiva_amd.sharded import HipEngine, shard_bounds

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    rank = int(os.environ.setdefault("RANK", "0"))
    world = int(os.environ.setdefault("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
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

        # transport of the per-iteration all-gather: RCCL through torch.distributed, or the library's push exchange
        # (validated against the collective on every rank before it is used, else it falls back; exchange.py)
        from overiva_amd.exchange import make_exchange

        xchg = make_exchange(eng, dist, None, rank, world, p_local, p_all, prefer=args.exchange)
        nparts = world * ppr

        def step():
            eng.power()
            eng.update_ptr(xchg.gather(), nparts)

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
        graph = None
        spg = 8            # iterations per captured graph (kernels + the RCCL all-gather)
        # Capturing RCCL collectives in a graph was verified with one rank only (this pool has 1-GPU boxes), so
        # the multi-rank default is eager launches (host cost per step ~45 us < device time); --graph 2 opts in.
        if args.graph >= 2 and args.steps >= spg and xchg.name == "collective":
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    for _ in range(spg):
                        step()
                graph.replay()
                stream.synchronize()
            except Exception as e:  # capture of the collective not supported: stay eager
                if rank == 0:
                    print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); eager", file=sys.stderr)
                graph = None
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
    exchange_name = xchg.name
    xchg.close()
    eng.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "bins": [f0, f1], **breakdown})
    out = result_line(args, world, float(tmax.item()))
    out["config"]["graph"] = graph is not None
    fl = f1 - f0
    if M <= 8:
        bytes_cov = cov_algorithmic_bytes(T, fl, M, K)
        out["roofline"] = {"bound": "hbm", "kernel": f"cov_dma_kernel<{M}, {min(K, 2)}> on rank 0's {fl} bins",
                           "achieved": bytes_cov / (cov_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_cov / (cov_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": bytes_cov, "avg_launch_ms": cov_ms, "per": "GPU"}
    else:
        issued = 6.0 * K * M * M * T * fl
        out["roofline"] = {"bound": "fp32-mfma", "kernel": f"cov_mfma16_kernel on rank 0's {fl} bins",
                           "achieved": issued / (cov_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": issued / (cov_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                           "avg_launch_ms": cov_ms, "per": "GPU"}
    out["cpu_baseline"] = None      # reported at N = 1 only
    out["ranks"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "exchange": exchange_name,
                    "per_rank_stage_ms": gathered,
                    "message_bytes_per_rank": int(p_local.numel() * 4)}
    dist.destroy_process_group()
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
        "dtype": "f32" if args.precision == "fast" else "f64",
        "data": "synthetic",
        "config": {"workload": WORKLOAD, "bins": F, "frames": T, "mics": M, "sources": K, "model": MODEL,
                   "precision": args.precision + (" (float32 everywhere; overiva() defaults to 'precise', see other_mode)"
                                                  if args.precision == "fast" else " (float64 covariance accumulation + per-bin algebra)"),
                   "parallelism": f"bins sharded over {n_gpus} GPU(s), one RCCL all-gather of (T,K) f32 per iteration"
                   if n_gpus > 1 else "single GPU", "graph": bool(args.graph)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", type=int, default=1,
                    help="0: eager; 1 (default): hipGraph replay on a single GPU, eager when sharded; 2: graph also when sharded")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline",
                    help="headline: BASELINE.json configs[2] (the metric's workload); cfg5: configs[4], 16 mics / 16 sources")
    ap.add_argument("--precision", choices=["fast", "precise"], default="fast",
                    help="arithmetic of the timed run (default fast = float32, the metric's dtype); the other mode is timed too")
    ap.add_argument("--no-other-mode", action="store_true", help="do not also time the other arithmetic mode")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (exercises RCCL + graph capture on 1 GPU)")
    ap.add_argument("--exchange", choices=["collective", "push"], default=os.environ.get("OIVA_EXCHANGE", "push"),
                    help="all-gather of the partial powers when sharded: the library's push exchange (default; validated against "
                         "the collective at start-up, falls back to it) or torch.distributed's collective (RCCL)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the sharded path (nccl = RCCL; tests use gloo)")
    ap.add_argument("--single-device", action="store_true",
                    help="tests on a 1-GPU box: every rank uses GPU 0 (needs --backend gloo: RCCL refuses two ranks on one device)")
    args = ap.parse_args()
    select_config(args.config)
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_sharded:
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
