"""Frequency-bin sharding over the GPUs of one node (one process per GPU, torch.distributed).

Every per-iteration quantity of the algorithm is per-bin except the source activation
``r[t,k] = f(sum_f |y_{t,f,k}|^2)`` (reference ``overiva.py:152-155``).  Each rank therefore owns a
contiguous range of bins end to end (its slice of X, Cx, V, W_hat, Y never leaves its GPU) and one
collective per iteration exchanges the per-rank partial powers: an all-gather of (T, K) float32
(one part per 64-bin batch, zero padded to the same count on every rank) followed by a sum in buffer order on every rank (bitwise identical ``r`` everywhere; an all-reduce
would leave the order to the ring).  With backend "nccl" this is RCCL over xGMI; the payload is
tiny (T*K*4 bytes per rank), so the step is latency bound.

torch is plumbing here (process group, the exchange buffers, the current stream); the arithmetic is
in the HIP kernels behind ``Plan``.  The engine is injectable so that the N > 1 control flow can be
exercised on CPU with the gloo backend in tests (``tests/test_sharded_gloo.py`` supplies an
oracle-backed engine; the product never does).
"""
import numpy as np

_active = None  # (group,) when overiva() should shard


def enable_bin_sharding(group=None, exchange=None):
    """Make ``overiva()`` shard bins over the ranks of ``group`` (default: the world group).
    Every rank must then call ``overiva()`` with the same arguments; every rank gets the full result.
    ``exchange``: "collective" (torch.distributed all-gather, the default), "push" (the library's own exchange,
    validated against the collective before use; see exchange.py), "resident" (the X-resident kernel with the exchange
    of the partial powers inside it, where every rank's shard fits on chip: ``BinShardedSolver``) or "fused" (the
    exchange inside the activation kernel of the four-launch iteration, any shard size, graphs replayed without the host);
    None reads $OIVA_EXCHANGE."""
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    global _active
    _active = (group, exchange)


def disable_bin_sharding():
    global _active
    _active = None


def active_group():
    return _active


def shard_bounds(n_freq, world):
    """bins [b[r], b[r+1]) belong to rank r; sizes differ by at most one"""
    return [n_freq * r // world for r in range(world + 1)]


def fused_blocks(bounds):
    """Block sums per rank for the exchange inside the activation kernel.  The library adds the 64-bin parts of the source
    powers in a canonical order: blocks of ceil(parts / 8) consecutive parts, each added sequentially, then the block sums
    sequentially (csrc/kernels_misc.hip).  Ranks with EQUAL shards made of whole blocks send their block sums and every rank
    forms exactly that sum -- the same bits as one GPU (2 / 4 / 8 ranks at 2048 bins: 4 / 2 / 1 blocks each); any other
    sharding sends one sum per rank (same bits on every rank, rounding-level difference to the single-GPU sum)."""
    F = bounds[-1]
    sizes = {bounds[r + 1] - bounds[r] for r in range(len(bounds) - 1)}
    bs = -(-(-(-F // 64)) // 8)                  # parts per block of the whole problem
    if len(sizes) == 1:
        fl = sizes.pop()
        if fl % (64 * bs) == 0 and 1 <= fl // (64 * bs) <= 8:
            return fl // (64 * bs)
    return 1


class HipEngine:
    """the product engine: one ``Plan`` on this rank's GPU, launched on torch's current stream"""

    def __init__(self, T, F_local, M, K, model, F_total, device, precision="fast"):
        import torch

        from .plan import Plan

        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # Kernels and the collective must share one stream so that they are ordered without host
        # syncs.  torch's default stream has the null handle, which the C ABI reads as "own stream",
        # so a dedicated torch stream is used unless the caller already switched to one.
        cur = torch.cuda.current_stream(self.device)
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device=self.device)
        self.plan = Plan(T, F_local, M, K, model, device=device, F_total=F_total, stream=self.stream.cuda_stream)
        self.plan.set_precision(precision)
        self.wdtype = np.complex64 if precision == "fast" else np.complex128
        self.T, self.K = T, K

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    @staticmethod
    def power_parts(n_bins):
        from .plan import Plan

        return Plan.power_parts(n_bins)

    def exchange_buffer(self, parts_per_rank):
        """(parts_per_rank * T, K) float32 device tensor aliasing the plan's partial-power buffer"""
        ptr, nbytes = self.plan.power_buffer(parts_per_rank)
        self.ppr = parts_per_rank

        class _Mem:
            pass

        m = _Mem()
        m.__cuda_array_interface__ = {"shape": (parts_per_rank * self.T, self.K), "typestr": "<f4", "data": (ptr, False),
                                      "version": 2, "strides": None}
        t = self.torch.as_tensor(m, device=self.device)
        assert t.data_ptr() == ptr, "torch copied the exchange buffer instead of aliasing it"
        return t

    def new_gather_buffer(self, world):
        # rank-major concatenation along dim 0 == (world, T, K) in memory (the layout every backend accepts)
        return self.torch.empty((world * self.ppr * self.T, self.K), dtype=self.torch.float32, device=self.device)

    def set_x(self, X, f0):
        self.plan.set_x(X, f0)

    def set_x_device(self, ptr, keepalive=None):
        self.plan.set_x_device(ptr, keepalive)

    def covariance(self):
        self.plan.covariance()

    def get_cx(self):
        return self.plan.get_cx()

    def set_w(self, W0):
        self.plan.set_w(W0)

    def set_w_eig(self):
        self.plan.set_w_eig()

    def setup_resident(self, dist, group, rank, world):
        """The X-resident kernel for this rank's shard, exchanging the partial source powers between the ranks itself
        (csrc/resident_kernel.inc).  Every rank's shard must qualify with the same frame-split geometry; the ranks agree
        on that (and on the mapping of each other's gather buffers) before anything is switched on.  Returns the reason it
        was NOT switched on, or None."""
        info = self.plan.resident_info()
        alls = [None] * world
        dist.all_gather_object(alls, (bool(info["qualifies"]), info["frame_splits"]), group=group)
        if not all(a[0] for a in alls):
            return "shard does not fit on chip on rank(s) " + ", ".join(str(r) for r, a in enumerate(alls) if not a[0])
        ns = min(a[1] for a in alls)           # uneven shards choose different split counts: the smallest fits everybody
        try:
            self.plan.set_resident_splits(ns)
            info = self.plan.resident_info()
            mine = (bool(info["qualifies"]), info["frame_splits"], info["frames_per_split"])
        except Exception:      # (any failure of a plan call on one rank must reach the others through the next exchange,
            mine = (False, 0, 0)   #  never escape between two collectives)
        dist.all_gather_object(alls, mine, group=group)
        if not all(a[0] for a in alls) or len({a[1:] for a in alls}) != 1:
            return f"the ranks found no common frame-split geometry: {alls}"
        if world == 1:
            try:
                self.plan.set_resident(True)
            except Exception as e:
                return f"{type(e).__name__}: {e}"
            return None
        from .exchange import PushExchange

        x, ok, why, mineh = None, True, "", b"\0" * 64
        try:
            x = PushExchange(self.device.index, rank, world, 0, info["frame_splits"] * info["frames_per_split"] * self.K * 4,
                             self.stream.cuda_stream)
            mineh = x.handle()
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        hs = [None] * world
        dist.all_gather_object(hs, (ok, mineh, why), group=group)
        if all(h[0] for h in hs):
            try:
                x.connect([h[1] for h in hs])
                self.plan.resident_connect(x.h)
            except Exception as e:
                ok, why = False, f"{type(e).__name__}: {e}"
        else:
            ok = False
            why = why or "; ".join(f"rank {r}: {h[2]}" for r, h in enumerate(hs) if not h[0])
        vs = [None] * world
        dist.all_gather_object(vs, (ok, why), group=group)
        if not all(v[0] for v in vs):
            try:
                self.plan.resident_connect(None)
            except Exception:
                pass
            if x is not None:
                x.close()
            return "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(vs) if not v[0])
        # switching it on allocates the kernel's exchange buffers: a failure on one rank is agreed on like the others
        try:
            self.plan.set_resident(True)
            ok, why = True, ""
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        dist.all_gather_object(vs, (ok, why), group=group)
        if not all(v[0] for v in vs):
            for undo in (lambda: self.plan.set_resident(False), lambda: self.plan.resident_connect(None), x.close):
                try:
                    undo()
                except Exception:
                    pass
            return "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(vs) if not v[0])
        self.resident_xchg = x
        return None

    def iterate_resident(self, n):
        self.plan.iterate(n)

    def setup_fused(self, dist, group, rank, world, nblk=1):
        """The exchange of the ranks' partial powers inside the activation kernel of the four-launch iteration (shards that do
        not fit on chip: 2 and 4 GPUs at the headline shape): no collective and no host in the loop, ``plan.iterate(n)``
        replays captured graphs of four kernels.  The ranks map each other's gather buffers (``PushExchange``, slot = T K 8
        bytes) and agree on the outcome.  Returns the reason it was NOT switched on, or None."""
        from .exchange import PushExchange

        x, ok, why, mineh = None, True, "", b"\0" * 64
        try:
            x = PushExchange(self.device.index, rank, world, 0, nblk * self.T * self.K * 8, self.stream.cuda_stream)
            mineh = x.handle()
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        hs = [None] * world
        dist.all_gather_object(hs, (ok, mineh, why), group=group)
        if all(h[0] for h in hs):
            try:
                if world > 1:
                    x.connect([h[1] for h in hs])
                self.plan.fused_connect(x.h)
                self.plan.use_graph(True)
            except Exception as e:
                ok, why = False, f"{type(e).__name__}: {e}"
        else:
            ok = False
            why = why or "; ".join(f"rank {r}: {h[2]}" for r, h in enumerate(hs) if not h[0])
        vs = [None] * world
        dist.all_gather_object(vs, (ok, why), group=group)
        if not all(v[0] for v in vs):
            for undo in (lambda: self.plan.fused_connect(None), lambda: x.close()):
                try:
                    undo()
                except Exception:
                    pass
            return "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(vs) if not v[0])
        self.fused_xchg = x
        return None

    def iterate_fused(self, n):
        self.plan.save_w()              # (on the stream, in front of the iterations: what a run falls back on, BinShardedSolver.iterate)
        self.plan.iterate(n)
        self.plan.sync()

    def drop_fused(self):
        """back to the state in front of the last ``iterate_fused`` call, on the plain activation kernel"""
        self.plan.fused_connect(None)
        self.plan.restore_w()
        if getattr(self, "fused_xchg", None) is not None:
            self.fused_xchg.close()
            self.fused_xchg = None

    def power(self):
        self.plan.power()

    def update(self, parts):
        self.plan.update(parts.data_ptr(), parts.shape[0] // self.T)

    def update_ptr(self, ptr, nparts):
        self.plan.update(ptr, nparts)

    def demix(self, proj_back):
        return self.plan.demix(proj_back)

    def get_w(self):
        return self.plan.get_w(self.wdtype)

    def to_comm(self, a):
        return self.torch.from_numpy(a).to(self.device)

    def sync(self):
        self.plan.sync()

    def close(self):
        self.plan.close()
        for name in ("resident_xchg", "fused_xchg"):
            if getattr(self, name, None) is not None:
                getattr(self, name).close()
                setattr(self, name, None)


class BinShardedSolver:
    """Same stage interface as the single-GPU solver in ``overiva.py``, over a process group."""

    def __init__(self, T, F, M, K, model, group=None, engine_factory=None, device=None, precision="fast", exchange=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if F < self.world:
            raise ValueError("fewer frequency bins than ranks")
        self.T, self.F, self.M, self.K = T, F, M, K
        self.bounds = shard_bounds(F, self.world)
        self.f0, self.f1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        if engine_factory is None:
            from .overiva import get_device

            dev = get_device() if device is None else device
            self.engine = HipEngine(T, self.f1 - self.f0, M, K, model, F, dev, precision)
        else:
            self.engine = engine_factory(T, self.f1 - self.f0, M, K, model, F)
        # every rank sends the same number of (T, K) parts: the largest batch count of any shard
        ppr = max(self.engine.power_parts(self.bounds[r + 1] - self.bounds[r]) for r in range(self.world))
        self.p_local = self.engine.exchange_buffer(ppr)
        self.p_all = self.engine.new_gather_buffer(self.world)
        self.nparts = self.world * ppr
        # transport of the per-iteration all-gather: torch.distributed's collective, or the library's push exchange
        # when asked for and validated (exchange.py)
        # "resident": one persistent launch per iterate() call with the shard on chip and the exchange inside the kernel;
        # agreed between the ranks, and off (with the reason kept) where a shard does not qualify
        import os

        want = exchange or os.environ.get("OIVA_EXCHANGE", "collective")
        choice = [want]
        dist.broadcast_object_list(choice, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        want = choice[0]
        self.resident, self.resident_refused = False, None
        if want == "resident" and hasattr(self.engine, "setup_resident"):
            if precision == "precise":      # (the kernel's float64 covariance exists for 4 channels on ONE rank only so far)
                self.resident_refused = "precise arithmetic (float64 covariance sums) runs the collective path when sharded"
            else:
                self.resident_refused = self.engine.setup_resident(dist, group, self.rank, self.world)
            self.resident = self.resident_refused is None
        # "fused": the exchange inside the activation kernel of the four-launch iteration (any shard size)
        self.fused, self.fused_refused = False, None
        if want == "fused" and hasattr(self.engine, "setup_fused"):
            self.fused_refused = self.engine.setup_fused(dist, group, self.rank, self.world, fused_blocks(self.bounds))
            self.fused = self.fused_refused is None
        if hasattr(self.engine, "plan"):
            from .exchange import make_exchange

            self.xchg = make_exchange(self.engine, dist, group, self.rank, self.world, self.p_local, self.p_all,
                                      prefer="collective" if want in ("resident", "fused") else want)
        else:
            self.xchg = None

    # ---- stages ---------------------------------------------------------------------------------
    def set_x(self, X):
        self.engine.set_x(X, self.f0)

    def covariance(self):
        self.engine.covariance()

    def get_cx(self):
        return self._gather_bins(self.engine.get_cx(), axis=0)

    def set_w(self, W0):
        if W0 is not None:
            W0 = np.broadcast_to(np.asarray(W0), (self.F, self.M, self.K))[self.f0:self.f1]
        self.engine.set_w(W0)

    def set_w_eig(self):
        """``init_eig`` (overiva.py:106-109): the eigenvectors are per bin, so every rank runs the device eigensolver on its
        own shard and nothing is exchanged; an engine without one (the CPU test engine) gets LAPACK's on the gathered Cx"""
        if hasattr(self.engine, "set_w_eig"):
            with self.engine.stream_ctx():
                self.engine.set_w_eig()
        else:
            from .overiva import eig_init

            self.set_w(eig_init(self.get_cx(), self.K))

    def iterate(self, n):
        if self.fused:
            # every rank replays the same n iterations from captured graphs; a wait that gave up is reported by the sync:
            # the ranks compare notes before anybody goes on
            try:
                with self.engine.stream_ctx():
                    self.engine.iterate_fused(n)
                mine = None
            except RuntimeError as e:
                mine = str(e)
            notes = [None] * self.world
            self.dist.all_gather_object(notes, mine, group=self.group)
            if not any(m is not None for m in notes):
                return
            # a rank did not deliver in time: EVERY rank goes back to the demixing matrices it saved in front of this call,
            # leaves the in-kernel exchange for good and repeats the n iterations through the collective (a warning and
            # last_solver_info()["exchange_degraded"] say so; nothing is silently different)
            import warnings

            why = "; ".join(f"rank {r}: {m}" for r, m in enumerate(notes) if m is not None)
            undo = None
            try:
                with self.engine.stream_ctx():
                    self.engine.drop_fused()
            except Exception as e:
                undo = f"{type(e).__name__}: {e}"
            self.dist.all_gather_object(notes, undo, group=self.group)
            if any(m is not None for m in notes):
                raise RuntimeError("sharded iteration (exchange inside the activation kernel) failed: " + why + "; and the fall-back "
                                   "to the collective failed too: " + "; ".join(f"rank {r}: {m}" for r, m in enumerate(notes) if m is not None))
            self.fused, self.fused_refused = False, "gave up during the run, continued on the collective: " + why
            warnings.warn("overiva_amd: the exchange inside the activation kernel gave up (" + why + "); the run continues on the collective path")
        if self.resident:
            # every rank launches the same n iterations; a launch that gave up (a rank's parts did not arrive) leaves W
            # unchanged on that rank only, so the ranks compare notes before anybody goes on
            try:
                with self.engine.stream_ctx():
                    self.engine.iterate_resident(n)
                mine = None
            except RuntimeError as e:
                mine = str(e)
            notes = [None] * self.world
            self.dist.all_gather_object(notes, mine, group=self.group)
            if any(m is not None for m in notes):
                raise RuntimeError("X-resident sharded iteration failed: " +
                                   "; ".join(f"rank {r}: {m}" for r, m in enumerate(notes) if m is not None))
            return
        with self.engine.stream_ctx():
            for _ in range(n):
                self.engine.power()                                # local sum_f |y|^2 -> p_local
                if self.xchg is not None:
                    self.engine.update_ptr(self.xchg.gather(), self.nparts)
                else:
                    self.dist.all_gather_into_tensor(self.p_all, self.p_local, group=self.group)
                    self.engine.update(self.p_all)                 # rank-order sum, r, V, IP1, J

    def demix(self, proj_back, dtype=np.complex64):
        return self._gather_bins(self.engine.demix(proj_back), axis=1).astype(dtype, copy=False)

    def get_w(self):
        # gather first, judge afterwards: a singular bin on ONE rank must raise on EVERY rank, not leave the
        # others waiting in the collective
        try:
            local, err = self.engine.get_w(), None
        except np.linalg.LinAlgError as e:
            local, err = np.full((self.f1 - self.f0, self.M, self.K), np.nan, getattr(self.engine, "wdtype", np.complex64)), e
        W = self._gather_bins(local, axis=0)
        if err is not None or not np.all(np.isfinite(W)):
            raise np.linalg.LinAlgError(str(err) if err is not None else
                                        "demixing matrix holds non-finite values on another rank (singular W_hat^H V)")
        return W

    def close(self):
        import sys

        _ov = sys.modules[__package__ + ".overiva"]      # (the package attribute `overiva` is the function, not the module)
        _ov._last_info = {"sharded": True, "world": self.world, "resident": getattr(self, "resident", False),
                          "resident_refused": getattr(self, "resident_refused", None),
                          "fused": getattr(self, "fused", False), "fused_refused": getattr(self, "fused_refused", None),
                          "exchange_degraded": getattr(self, "resident_refused", None) or getattr(self, "fused_refused", None),
                          "exchange": "resident" if getattr(self, "resident", False) else
                          ("fused" if getattr(self, "fused", False) else getattr(self.xchg, "name", None))}
        if getattr(self, "xchg", None) is not None:
            self.engine.sync()
            self.xchg.close()
        self.engine.close()

    # ---- helpers --------------------------------------------------------------------------------
    def _gather_bins(self, local, axis):
        """concatenate per-rank complex arrays along the bin axis on every rank"""
        import torch

        local = np.ascontiguousarray(np.moveaxis(local, axis, 0))          # (F_local, ...)
        cdt = local.dtype
        rdt = np.float64 if cdt == np.complex128 else np.float32
        rest = local.shape[1:]
        fmax = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        pad = np.zeros((fmax,) + rest, dtype=cdt)
        pad[: local.shape[0]] = local
        send = self.engine.to_comm(pad.view(rdt))
        recv = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype,
                           device=send.device)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        full = recv.cpu().numpy().view(cdt).reshape((self.world, fmax) + rest)
        parts = [full[r, : self.bounds[r + 1] - self.bounds[r]] for r in range(self.world)]
        return np.moveaxis(np.concatenate(parts, axis=0), 0, axis)
