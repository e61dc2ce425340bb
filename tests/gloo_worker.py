"""Worker for tests/test_sharded_gloo.py: runs BinShardedSolver on CPU over a gloo process group
with a TEST-ONLY engine backed by the oracle's stage functions (the product engine is HipEngine)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class OracleEngine:
    """stand-in for HipEngine: same interface, NumPy arithmetic from oracle/ (tests only)"""

    def __init__(self, T, F_local, M, K, model, F_total):
        import torch

        self.torch = torch
        self.T, self.F, self.M, self.K, self.model, self.F_total = T, F_local, M, K, model, F_total
    @staticmethod
    def power_parts(n_bins):
        return (n_bins + 3) // 4          # test engine: one part per 4 bins, so shards differ in part count

    def exchange_buffer(self, parts_per_rank):
        self.ppr = parts_per_rank
        self._p = self.torch.zeros((parts_per_rank * self.T, self.K), dtype=self.torch.float32)
        return self._p

    def stream_ctx(self):
        import contextlib

        return contextlib.nullcontext()

    def new_gather_buffer(self, world):
        return self.torch.empty((world * self.ppr * self.T, self.K), dtype=self.torch.float32)

    def set_x(self, X, f0):
        self.X = np.asarray(X)[:, f0:f0 + self.F, :].astype(np.complex128)

    def covariance(self):
        from oracle import overiva_oracle as orc

        self.Cx = orc.input_covariance(self.X)

    def get_cx(self):
        return self.Cx.astype(np.complex64)

    def set_w(self, W0):
        from oracle import overiva_oracle as orc

        self.What = orc.init_demixing(self.Cx, self.K, W0=W0)

    def power(self):
        from oracle import overiva_oracle as orc

        buf = self._p.numpy().reshape(self.ppr, self.T, self.K)
        buf[:] = 0.0
        for part in range(self.power_parts(self.F)):
            sl = slice(4 * part, min(4 * part + 4, self.F))
            buf[part] = orc.demix_power(self.X[:, sl], self.What[sl, :, :self.K]).astype(np.float32)

    def update(self, parts):
        from oracle import overiva_oracle as orc

        p = np.zeros((self.T, self.K), dtype=np.float32)
        parts = parts.numpy().reshape(-1, self.T, self.K)
        for g in range(parts.shape[0]):          # fixed rank order, float32 like the kernel
            p = p + parts[g]
        rinv, wscale = orc.finalize_activation(p.astype(np.float64), self.F_total, self.model)
        self.What[:, :, :self.K] /= wscale[None, None, :]
        V = orc.weighted_cov_all(self.X, rinv)
        self.What = orc.ip_update_bin(self.What, V, self.Cx, self.K)

    # the in-kernel exchange of the product engine, as far as the driver sees it: set up by agreement, n iterations per call,
    # and -- $OIVA_TEST_FUSED_GIVES_UP = a rank -- a wait that gives up on that rank, which leaves garbage on EVERY rank
    def setup_fused(self, dist, group, rank, world, nblk=1):
        return None

    def iterate_fused(self, n):
        import torch.distributed as dist

        self._saved = self.What.copy()
        self.What = self.What * np.nan
        if os.environ.get("OIVA_TEST_FUSED_GIVES_UP") == str(dist.get_rank()):
            raise RuntimeError("the exchange inside the activation kernel gave up waiting (test)")

    def drop_fused(self):
        self.What = self._saved

    def demix(self, proj_back):
        from oracle import overiva_oracle as orc

        Y = np.einsum("tfm,fmk->tfk", self.X, np.conj(self.What[:, :, :self.K]))
        if proj_back:
            Y = Y * np.conj(orc.projection_back(Y, self.X[:, :, 0])[None])
        return Y.astype(np.complex64)

    # like HipEngine in the 'mixed' / 'precise' modes when $OIVA_TEST_WDTYPE=c128: W travels as complex128
    wdtype = np.complex128 if os.environ.get("OIVA_TEST_WDTYPE") == "c128" else np.complex64

    def get_w(self):
        import torch.distributed as dist

        if os.environ.get("OIVA_TEST_SINGULAR_RANK") == str(dist.get_rank()):
            raise np.linalg.LinAlgError("demixing matrix holds non-finite values (test)")
        return np.ascontiguousarray(self.What[:, :, :self.K]).astype(self.wdtype)

    def to_comm(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a))

    def close(self):
        pass


def main():
    import torch.distributed as dist

    from oracle import overiva_oracle as orc
    from overiva_amd.sharded import BinShardedSolver

    out_dir, model, n_iter = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T, F, M, K = 60, 11, 4, 2
    X = orc.synth_iid(T, F, M, seed=11)
    rng = np.random.default_rng(12)
    W0 = np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K)))
    s = BinShardedSolver(T, F, M, K, model, engine_factory=OracleEngine, exchange="fused" if os.environ.get("OIVA_TEST_FUSED_GIVES_UP") else None)
    assert s.fused == bool(os.environ.get("OIVA_TEST_FUSED_GIVES_UP"))
    assert s.f1 - s.f0 >= F // world
    s.set_x(X)
    s.covariance()
    Cx = s.get_cx()
    s.set_w(W0)
    s.iterate(n_iter)
    Y = s.demix(True)
    if os.environ.get("OIVA_TEST_SINGULAR_RANK") is not None:
        # one rank's bins are singular: EVERY rank must get the error (nobody may be left waiting in the gather)
        try:
            s.get_w()
            outcome = "no error"
        except np.linalg.LinAlgError as e:
            outcome = "LinAlgError: " + str(e)
        s.close()
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(outcome)
        dist.barrier()
        dist.destroy_process_group()
        return
    W = s.get_w()
    if os.environ.get("OIVA_TEST_FUSED_GIVES_UP"):
        assert not s.fused and "gave up" in s.fused_refused
    s.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), Y=Y, W=W, Cx=Cx, f0=s.f0, f1=s.f1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
