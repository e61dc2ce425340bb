import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = 4000, 256, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
p = oa.Plan(T, F, M, K, "laplace")
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(2); p.sync()
out = {}
for ns in (4, 8, 16, 24, 32, 48, 64):
    p.set_cov_splits(ns)
    out[ns] = [round(p.t_time_stage("weighted_cov", 20) * 1e3, 1) for _ in range(2)]
print("cov  F=256 splits->us", out, "ablate" if os.environ.get("OIVA_UNUSED") else "")
out = {}
for ns in (8, 16, 32, 64, 96, 128, 192):
    p.set_pow_splits(ns)
    out[ns] = [round(p.t_time_stage("demix_power", 20) * 1e3, 1) for _ in range(2)]
print("pow  F=256 splits->us", out)
