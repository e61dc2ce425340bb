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
res = []
for rnd in range(3):
    for ns in (8, 12, 16, 24, 32, 48, 64):
        p.set_pow_splits(ns)
        res.append((ns, p.t_time_stage("demix_power", 10) * 1e3))
import collections
d = collections.defaultdict(list)
for ns, t in res: d[ns].append(t)
print("LDS pad", os.environ.get("OIVA_POW_LDS_PAD", "0"), {ns: [round(x) for x in v] for ns, v in d.items()})
