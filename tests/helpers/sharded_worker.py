"""Worker of tests/test_sharded_2proc_gpu.py: one of N processes that share GPU 0 and shard the bins of one
overiva() call between them -- the product's HipEngine and BinShardedSolver end to end, with the gloo backend
carrying the collectives (RCCL refuses two ranks on one device).  Rank 0 writes the result."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    out, T, F, M, K, model, precision, n_iter = sys.argv[1], *[int(a) for a in sys.argv[2:6]], sys.argv[6], sys.argv[7], int(sys.argv[8])
    exchange = sys.argv[9] if len(sys.argv) > 9 else "collective"
    init_eig = len(sys.argv) > 10 and sys.argv[10] == "eig"
    backend = sys.argv[11] if len(sys.argv) > 11 else "gloo"
    data = sys.argv[12] if len(sys.argv) > 12 else "mixture"
    import torch
    import torch.distributed as dist

    import overiva_amd as oa
    from oracle import overiva_oracle as orc          # test infrastructure: the input generator only

    if backend == "nccl":                              # RCCL: one rank per device (used with world = 1 on the 1-GPU box)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    oa.set_device(0)                                   # every rank on the one GPU of the box
    oa.set_precision(precision)
    X = orc.synth_mixture(T, F, M, K, seed=11) if data == "mixture" else orc.synth_iid(T, F, M, seed=11)
    oa.enable_bin_sharding(exchange=exchange)
    import warnings

    warnings.simplefilter("error")                    # a fall-back to the collective must fail the test, not pass silently
    seen = []
    if len(sys.argv) > 10 and sys.argv[10] == "pca":        # auxiva_pca under bin sharding (auxiva_pca.py:63-92): Y only
        Y = oa.auxiva_pca(X, n_src=K, n_iter=n_iter, proj_back=True, model=model)
        W = np.zeros((1,), X.dtype)
        seen.append(Y[:1])
    else:
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, model=model, return_filters=True, init_eig=init_eig,
                          callback=lambda y: seen.append(y.copy()))
    info = oa.last_solver_info()
    oa.disable_bin_sharding()
    if dist.get_rank() == 0:
        np.savez(out, Y=Y, W=W, cb=np.stack(seen), world=dist.get_world_size(), resident=bool(info.get("resident")),
                 refused=str(info.get("resident_refused") or info.get("fused_refused")), backend=dist.get_backend(),
                 exchange=str(info.get("exchange")))
    dist.barrier()
    dist.destroy_process_group()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
