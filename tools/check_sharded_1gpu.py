"""GPU-box check: the bin-sharded driver with a 1-rank RCCL group must reproduce the single-plan result bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import overiva_amd as oa
from oracle import overiva_oracle as orc
X = orc.synth_iid(300, 70, 4, seed=5)
Y0, W0 = oa.overiva(X, n_src=2, n_iter=7, proj_back=True, return_filters=True)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
oa.enable_bin_sharding()
Y1, W1 = oa.overiva(X, n_src=2, n_iter=7, proj_back=True, return_filters=True)
print("sharded(world=1) == single:", np.array_equal(Y0, Y1), np.array_equal(W0, W1), orc.rel_err(Y1, Y0))
got = []
Y2 = oa.overiva(X, n_src=2, n_iter=12, proj_back=True, callback=lambda y: got.append(y.copy()), init_eig=True)
print("callbacks", len(got), Y2.shape)
dist.destroy_process_group()
