"""CPU-side checks: the C-ABI library loads and exports every symbol include/overiva_hip.h declares,
the ctypes table matches the header, host-side argument handling, and loud failure without a GPU or
without the built library.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, has_gpu

HEADER = os.path.join(REPO, "include", "overiva_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oiva_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from overiva_amd import build, _lib

    build.build_library()          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_declares_the_abi():
    syms = header_symbols()
    for must in ("oiva_plan_create", "oiva_plan_iterate", "oiva_plan_power", "oiva_plan_update",
                 "oiva_plan_demix", "oiva_plan_get_w", "oiva_last_error", "oiva_version"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in overiva_hip.h but not exported"
    assert lib.oiva_version() >= 100


def test_ctypes_table_matches_header(lib):
    from overiva_amd import _lib

    declared = set(header_symbols()) - {"oiva_version", "oiva_last_error"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_no_torch_types_in_abi():
    txt = open(HEADER).read()
    assert "torch" not in txt.lower().replace("pytorch", "") and "at::" not in txt and "std::" not in txt


def test_argument_validation_before_device(lib):
    import overiva_amd as oa

    X = (np.ones((8, 3, 2)) + 0j).astype(np.complex64)
    with pytest.raises(ValueError):
        oa.overiva(X, n_src=2, model="student")
    with pytest.raises(ValueError):
        oa.overiva(X, n_src=3)
    with pytest.raises(ValueError):
        oa.overiva(X, n_src=0)
    with pytest.raises(TypeError):
        oa.overiva(X.real)
    with pytest.raises(ValueError):
        oa.overiva(X[0])
    with pytest.raises(KeyError):      # auxiva_pca.py:86 pops 'proj_back' unconditionally
        oa.auxiva_pca(X, n_src=1, n_iter=1)


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_gpu(lib):
    """no CPU fallback: without a device the product path raises instead of computing"""
    import overiva_amd as oa

    X = (np.ones((8, 3, 2)) + 0j).astype(np.complex64)
    with pytest.raises(oa.HipError):
        oa.overiva(X, n_src=1, n_iter=1)
    n = ctypes.c_int()
    assert lib.oiva_device_count(ctypes.byref(n)) != 0
    assert b"hipGetDeviceCount" in lib.oiva_last_error()


def test_fails_loudly_without_library(monkeypatch):
    from overiva_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liboveriva_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "overiva_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_default_arithmetic_rule():
    """`auto` (overiva.py:89,126,131: the reference computes in the dtype of X): complex128 -> precise; complex64 -> precise on a
    short frame axis with up to 8 channels (overiva.py:179: the reference forms those covariances in complex128 too), else mixed,
    where the covariance pass hands float64 sums of short float32 chains to the float64 per-bin algebra"""
    from overiva_amd.overiva import SHORT_FRAME_AXIS, resolve_precision as rp

    assert rp(np.complex128, 4, "auto", 2) == "precise" and rp(np.complex128, 16, "auto", 2) == "precise"
    assert all(rp(np.complex64, m, "auto", k) == "mixed" for m in range(1, 9) for k in range(1, m + 1))
    assert all(rp(np.complex64, m, "auto", k) == "mixed" for m in range(9, 17) for k in (1, 2, 4, 5, m))
    assert rp(np.complex64, 16, "auto") == "mixed"
    assert rp(np.complex64, 16, "fast", 2) == "fast" and rp(np.complex128, 4, "mixed", 2) == "mixed"
    # the frame axis decides for complex64 input of up to 8 channels
    assert SHORT_FRAME_AXIS == 256
    assert all(rp(np.complex64, m, "auto", 2, n_frames=t) == "precise" for m in range(1, 9) for t in (2, 160, 235, 256))
    assert all(rp(np.complex64, m, "auto", 2, n_frames=t) == "mixed" for m in range(1, 9) for t in (257, 1000, 4000))
    assert all(rp(np.complex64, m, "auto", m, n_frames=160) == "mixed" for m in range(9, 17))
    assert rp(np.complex64, 8, "mixed", 2, n_frames=160) == "mixed" and rp(np.complex64, 8, "fast", 2, n_frames=160) == "fast"


def test_shard_bounds():
    from overiva_amd import shard_bounds

    for F in (8, 513, 2048, 2049):
        for G in (1, 2, 3, 4, 8):
            b = shard_bounds(F, G)
            assert b[0] == 0 and b[-1] == F and len(b) == G + 1
            sizes = np.diff(b)
            assert sizes.min() >= 1 and sizes.max() - sizes.min() <= 1


def test_eig_init_matches_reference_recipe():
    from overiva_amd.overiva import eig_init
    from oracle import overiva_oracle as orc

    X = orc.synth_iid(64, 5, 4, seed=3).astype(np.complex128)
    Cx = orc.input_covariance(X)
    W0 = eig_init(Cx, 2)
    ref = orc.init_demixing(Cx, 2, init_eig=True)[:, :, :2]
    assert np.allclose(W0, ref)


def test_plain_c_program_links_against_the_abi(lib, tmp_path):
    """the header is C (not C++) and the library is usable from a plain-C program"""
    import subprocess

    exe = tmp_path / "c_abi_demo"
    pkg = os.path.join(REPO, "overiva_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                        os.path.join(REPO, "examples", "c_abi_demo.c"), "-L", pkg, "-loveriva_hip",
                        f"-Wl,-rpath,{pkg}", "-lm", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not has_gpu():
        run = subprocess.run([str(exe)], capture_output=True, text=True)
        assert run.returncode == 2 and "no GPU" in run.stderr      # loud failure, no CPU path


def test_resident_kernels_use_no_scratch_memory(lib):
    """the X-resident kernels hold 128 covariance accumulators in the architectural registers and up to 128 floats of X in
    the accumulator file; a build whose register allocation falls over into scratch memory (per-lane stack in HBM) would
    still be correct and quietly several times slower, so the build keeps hipcc's resource remarks and this checks them"""
    import os

    from overiva_amd import build

    if not any(os.path.exists(os.path.join(build.OBJ, s + ".usage.txt")) for s in ("kernels_resident_m4", "kernels_resident_m8", "kernels_resident_m6", "kernels_resident_m2")):
        build.build_library(force=True)          # objects came from a build without remarks
    usage = build.resident_kernel_usage()
    assert len(usage) >= 26, usage.keys()      # 2, 4, 6, 8 channels x sources x frame residency x arithmetic
    for name, u in usage.items():
        assert u["scratch"] == 0, (name, u)
        assert u["vgprs"] + u["agprs"] <= 512
    # the variants that keep 8 frames per lane in registers need the accumulator file for them
    assert any(u["agprs"] >= 128 for name, u in usage.items() if "ILi8ELi2ELi8E" in name)


def test_no_kernel_of_the_iteration_uses_scratch_memory():
    """every kernel file leaves its compiler resource remarks next to its object (overiva_amd/build.py): no kernel on the path of
    an iteration, the prologue or the epilogue may spill registers or index a lane-private array dynamically (either ends in
    scratch memory = HBM traffic per lane).  Known exceptions: the one-time OGIVE initialisation kernels (per-thread matrices)."""
    from overiva_amd import build

    usage = build.kernel_usage()
    if len(usage) < 100:
        pytest.skip("resource remarks not available (library built without them)")
    allowed = ("ogive_init_kernel", "ogive_switch_kernel")
    bad = {n: u["scratch"] for n, u in usage.items() if u.get("scratch", 0) > 0 and not any(a in n for a in allowed)}
    assert not bad, bad


def test_occupancy_the_measured_kernels_were_built_for():
    """register budgets that decide how many waves a SIMD holds (512 registers per lane and SIMD): the matrix-core covariance kernel
    of configs[4] needs three workgroups per CU (<= 168 registers), the three-wave update kernel of configs[4] two workgroups
    per CU (<= 256 with its accumulator registers) -- a build that crosses either line still passes every parity test and
    loses 10-40 % of the stage"""
    from overiva_amd import build

    usage = build.kernel_usage()
    if len(usage) < 100:
        pytest.skip("resource remarks not available (library built without them)")
    hm = [u for n, u in usage.items() if "cov_hmfma_kernelILb1ELb1ELb1E" in n]
    rw = [u for n, u in usage.items() if "update_det16r_kernel" in n]
    assert hm and rw
    assert all(u["vgprs"] + u["agprs"] <= 168 for u in hm), hm
    assert all(u["vgprs"] + u["agprs"] <= 256 for u in rw), rw


def test_host_prefault_is_host_only_and_keeps_the_contents(lib):
    """oiva_host_prefault (csrc/host_io.hip: the copy-thread pool populating the pages of a destination) touches no device: it
    runs here.  Contents kept on an unaligned range, on a range shorter than a page, from four caller threads at once (the pool
    takes one call at a time); bad arguments are refused."""
    import ctypes as C
    import threading

    a = np.arange(2_000_001, dtype=np.float64)[1:]         # not page aligned
    keep = a.copy()
    assert lib.oiva_host_prefault(C.c_void_p(a.ctypes.data), a.nbytes) == 0
    assert np.array_equal(a, keep)
    small = np.arange(7, dtype=np.uint8)
    assert lib.oiva_host_prefault(C.c_void_p(small.ctypes.data), small.nbytes) == 0 and small.tolist() == list(range(7))
    assert lib.oiva_host_prefault(C.c_void_p(small.ctypes.data), 0) == 0
    assert lib.oiva_host_prefault(None, 16) != 0 and lib.oiva_host_prefault(C.c_void_p(small.ctypes.data), -1) != 0
    bufs = [np.full(1 << 20, i, np.int32) for i in range(4)]
    errs = []

    def work(b):
        for _ in range(20):
            if lib.oiva_host_prefault(C.c_void_p(b.ctypes.data), b.nbytes) != 0:
                errs.append(1)

    th = [threading.Thread(target=work, args=(b,)) for b in bufs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs and all(np.all(b == i) for i, b in enumerate(bufs))
