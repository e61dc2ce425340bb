"""Transport of the per-iteration all-gather of the partial source powers between the ranks of a bin-sharded run.

Two interchangeable transports with the same result layout (rank-major concatenation of the parts):

* ``CollectiveExchange`` -- ``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI with backend "nccl");
* ``PushExchange``       -- the library's own exchange (``csrc/exchange.hip``): every rank stores its part straight into
  every rank's buffer and signals a counter; no collective launch, no protocol -- built for the 100 KB payload of this
  path, where a collective is pure latency.

``make_exchange`` validates the push transport against the collective before it is used -- three epochs of a known
pattern with host-side waits and a time-out, then one epoch in which the stream-side wait (``hipStreamWaitValue32``, the
primitive the iterations block on) is enqueued 50 ms BEFORE the parts are pushed: it has to hold the stream that long and
release it once every rank has pushed; a wait that does not return within 5 s is released by a host store -- and falls
back to the collective, on every rank alike (rank 0's choice of transport is broadcast, the verdicts are gathered), if
anything is off.  A platform on which peer mappings or peer atomics do not work thus costs a warning, not a wrong result
or a hang.  ``HSA_ENABLE_IPC_MODE_LEGACY=0`` must be in the environment before the process's first GPU call (this
platform's driver exports device memory through dmabuf only): launchers export it for their ranks
(``bench.py --gpus N``), and ``_lib.require_dmabuf_ipc`` sets it -- or warns that it is too late -- when a multi-process
exchange is created; importing the package no longer changes the environment.
"""
import ctypes as C
import os
import time
import warnings

import numpy as np

from . import _lib


class CollectiveExchange:
    name = "collective"
    fallback_reason = None      # why the push exchange was not used although it was asked for

    def __init__(self, engine, dist, group, world, p_local, p_all):
        self.dist, self.group, self.p_local, self.p_all = dist, group, p_local, p_all

    def gather(self):
        """all ranks' parts on this rank, ordered on the engine's stream; returns (device pointer, number of parts' rows)"""
        self.dist.all_gather_into_tensor(self.p_all, self.p_local, group=self.group)
        return self.p_all.data_ptr()

    def close(self):
        pass


class PushExchange:
    name = "push"

    def __init__(self, device, rank, world, part_ptr, part_bytes, stream_handle):
        if world > 1:
            _lib.require_dmabuf_ipc("push exchange between processes")
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        self.part_ptr, self.part_bytes, self.stream = int(part_ptr), int(part_bytes), int(stream_handle)
        h = C.c_void_p()
        _lib.check(self.lib.oiva_xchg_create(C.byref(h), int(device), int(rank), int(world), self.part_bytes))
        self.h = h
        self.epoch = 0

    def handle(self):
        buf = C.create_string_buffer(64)
        _lib.check(self.lib.oiva_xchg_export(self.h, buf))
        return buf.raw

    def connect(self, handles):
        blob = b"".join(handles)
        assert len(blob) == 64 * self.world
        _lib.check(self.lib.oiva_xchg_connect(self.h, blob))

    def push(self, part_ptr=None, stream=None, advance=True):
        if advance:
            self.epoch += 1
        _lib.check(self.lib.oiva_xchg_push(self.h, C.c_void_p(int(stream) if stream is not None else self.stream),
                                           C.c_void_p(int(part_ptr or self.part_ptr)), self.part_bytes, self.epoch))

    def force(self):
        """release a stream wait of the current epoch that would never be satisfied (failed validation)"""
        _lib.check(self.lib.oiva_xchg_force(self.h, self.epoch))

    def wait(self):
        _lib.check(self.lib.oiva_xchg_wait(self.h, C.c_void_p(self.stream), self.epoch))

    def gathered_ptr(self):
        p = C.c_void_p()
        _lib.check(self.lib.oiva_xchg_gathered(self.h, self.epoch, C.byref(p)))
        return p.value

    def poll(self, timeout_ms):
        ok = C.c_int()
        _lib.check(self.lib.oiva_xchg_poll(self.h, self.epoch, int(timeout_ms), C.byref(ok)))
        return bool(ok.value)

    def gather(self):
        self.push()
        self.wait()
        return self.gathered_ptr()

    def close(self):
        if getattr(self, "h", None):
            self.lib.oiva_xchg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_view(torch, ptr, nfloats, device):
    class _Mem:
        pass

    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(m, device=device)


def make_exchange(engine, dist, group, rank, world, p_local, p_all, prefer=None):
    """the transport for this run: the push exchange when it is asked for (``prefer="push"`` or ``OIVA_EXCHANGE=push``),
    available and validated on every rank, else the collective"""
    prefer = prefer or os.environ.get("OIVA_EXCHANGE", "collective")
    # every rank must take the same branch below (the branches hold collectives): rank 0's choice counts
    choice = [prefer]
    dist.broadcast_object_list(choice, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    prefer = choice[0]
    fallback = CollectiveExchange(engine, dist, group, world, p_local, p_all)
    if prefer != "push" or world > 16 or not hasattr(engine, "plan"):
        return fallback
    torch = engine.torch
    x, ok, why = None, True, ""
    try:
        part_bytes = p_local.numel() * 4
        x = PushExchange(engine.device.index, rank, world, p_local.data_ptr(), part_bytes, engine.stream.cuda_stream)
        mine = x.handle()
    except Exception as e:                      # allocation / export not supported here
        ok, why, mine = False, f"{type(e).__name__}: {e}", b"\0" * 64
    handles = [None] * world
    dist.all_gather_object(handles, (ok, mine), group=group)
    if all(h[0] for h in handles):
        try:
            x.connect([h[1] for h in handles])
            n = p_local.numel()
            flat = p_local.view(-1)
            saved = flat.clone()
            with torch.cuda.stream(engine.stream):
                for trial in range(3):          # both parities, and the reuse of the first
                    flat.copy_((torch.arange(n, device=p_local.device, dtype=torch.float32) % 251) + 1000.0 * rank + trial)
                    engine.stream.synchronize()
                    x.push()
                    if not x.poll(5000):
                        raise RuntimeError(f"parts of epoch {x.epoch} did not arrive within 5 s")
                    got = _device_view(torch, x.gathered_ptr(), n * world, p_local.device).clone().reshape(world, n)
                    want = torch.stack([(torch.arange(n, device=p_local.device, dtype=torch.float32) % 251) + 1000.0 * r + trial
                                        for r in range(world)])
                    if not torch.equal(got, want):
                        raise RuntimeError(f"epoch {x.epoch}: gathered parts differ from what the ranks sent")
                    x.wait()                    # the stream-side wait the iterations use (already satisfied here)
                    engine.stream.synchronize()
                # ... and once while it really BLOCKS: the wait is enqueued first, the push comes 50 ms later from a side
                # stream.  The wait must hold the stream for those 50 ms and let it go within 5 s after every rank has
                # pushed; a wait that never returns is released by a host store of the awaited value (force) -- the
                # iterations would have hung exactly there.
                flat.copy_((torch.arange(n, device=p_local.device, dtype=torch.float32) % 251) + 1000.0 * rank + 7)
                engine.stream.synchronize()
                side = torch.cuda.Stream(device=p_local.device)
                x.epoch += 1
                x.wait()
                done = torch.cuda.Event()
                done.record(engine.stream)
                time.sleep(0.05)
                held = not done.query()
                x.push(stream=side.cuda_stream, advance=False)
                t_end = time.monotonic() + 5.0
                while not done.query() and time.monotonic() < t_end:
                    time.sleep(0.001)
                if not done.query():
                    x.force()
                    engine.stream.synchronize()
                    raise RuntimeError("the stream-side wait did not return within 5 s of the pushes")
                if not held:
                    raise RuntimeError("the stream-side wait did not block while the parts were missing")
                got = _device_view(torch, x.gathered_ptr(), n * world, p_local.device).clone().reshape(world, n)
                want = torch.stack([(torch.arange(n, device=p_local.device, dtype=torch.float32) % 251) + 1000.0 * r + 7 for r in range(world)])
                if not torch.equal(got, want):
                    raise RuntimeError("blocking epoch: gathered parts differ from what the ranks sent")
                side.synchronize()
                flat.copy_(saved)
                engine.stream.synchronize()
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
    else:
        ok = False
        why = why or "another rank could not create the exchange"
    verdicts = [None] * world
    dist.all_gather_object(verdicts, (ok, why), group=group)
    if all(v[0] for v in verdicts):
        return x
    if rank == 0:
        warnings.warn("push exchange unavailable (" + "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(verdicts) if not v[0]) +
                      "); using the torch.distributed all-gather")
    if x is not None:
        x.close()
    fallback.fallback_reason = "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(verdicts) if not v[0])
    return fallback
