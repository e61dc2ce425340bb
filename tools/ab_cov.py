import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = 4000, 2048, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
p = oa.Plan(T, F, M, K, "laplace")
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(2); p.sync()
import collections
d = collections.defaultdict(list)
for rnd in range(4):
    for ns in (4, 8, 12, 16):
        p.set_cov_splits(ns)
        d[ns].append(round(p.t_time_stage("weighted_cov", 10) * 1e3, 1))
print("COV_DMA", os.environ.get("OIVA_COV_DMA", "1"), dict(d))
