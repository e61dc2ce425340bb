#!/usr/bin/env python3
"""GPU box: is the eager bin-sharded step host-bound on one rank's share of the 8-GPU problem (256 of 2048 bins)?
Wall time per step of the product loop (power | exchange | activation + covariance + update) against the issue time
alone, for both transports (a 1-rank process group stands in for the node)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from overiva_amd.sharded import HipEngine
from overiva_amd.exchange import make_exchange
T, F, M, K, FT = 4000, 256, 8, 2, 2048
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device=dev))
for name in ("collective", "push"):
    eng = HipEngine(T, F, M, K, "laplace", FT, 0, precision="fast")
    with eng.stream_ctx():
        eng.set_x_device(X.data_ptr(), keepalive=X); eng.covariance(); eng.set_w(None)
        ppr = eng.power_parts(F)
        p_local = eng.exchange_buffer(ppr); p_all = eng.new_gather_buffer(1)
        x = make_exchange(eng, dist, None, 0, 1, p_local, p_all, prefer=name)
        def step():
            eng.power(); eng.update_ptr(x.gather(), ppr)
        for _ in range(50): step()
        eng.stream.synchronize()
        n = 1000
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter()
        eng.stream.synchronize()
        t2 = time.perf_counter()
    print(f"{x.name:10s}: issue {1e6 * (t1 - t0) / n:6.1f} us per step, wall {1e6 * (t2 - t0) / n:6.1f} us per step", flush=True)
    x.close(); eng.close()
dist.destroy_process_group()
