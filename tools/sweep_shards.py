import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, M, K = 4000, 8, 2
for F in (256, 512, 1024):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    torch.cuda.synchronize()
    p = oa.Plan(T, F, M, K, "laplace")
    p.set_x_device(X.data_ptr(), X)
    p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    d0 = (round(p.t_time_stage("demix_power", 20) * 1e3, 1), round(p.t_time_stage("weighted_cov", 20) * 1e3, 1), p.cov_splits())
    pw = {}
    for ns in (16, 24, 32, 48, 64, 96, 128, 192):
        p.set_pow_splits(ns)
        pw[ns] = round(min(p.t_time_stage("demix_power", 20) for _ in range(2)) * 1e3, 1)
    p.set_pow_splits(0)
    cv = {}
    for ns in (4, 8, 16, 24, 32):
        p.set_cov_splits(ns)
        cv[ns] = (round(min(p.t_time_stage("weighted_cov", 20) for _ in range(2)) * 1e3, 1), round(min(p.t_time_stage("ip_update", 20) for _ in range(2)) * 1e3, 1))
    print(f"F={F}: default pow/cov/splits {d0} | pow {pw} | cov(+update) {cv}")
    p.close()
