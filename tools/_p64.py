import os, sys
sys.path.insert(0, os.getcwd())
import torch, overiva_amd as oa
T, F, M, K = (4000, 2048, 16, 16)
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
for mode in ("mixed", "precise"):
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    tc = min(p.t_time_stage('weighted_cov', 5) * 1e3 for _ in range(3)); tu = min(p.t_time_stage('ip_update', 5) * 1e3 for _ in range(3))
    print(f"{mode} splits {p.cov_splits():3d}: cov {tc:7.1f} us, update {tu:6.1f} us", flush=True)
    p.close()
