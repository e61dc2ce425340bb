"""Build recipe of liboveriva_hip.so: hipcc, gfx950 only, in-tree output (overiva_amd/liboveriva_hip.so).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
with the source tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "liboveriva_hip.so")
SOURCES = ["plan.hip", "host_io.hip", "kernels_cov.hip", "kernels_cov_mfma.hip", "kernels_cov_quad.hip", "kernels_cov_half16.hip", "kernels_cov_hmfma.hip", "kernels_cov_pair64.hip", "kernels_cov_pair32.hip", "kernels_demix.hip", "kernels_power_mfma.hip", "kernels_update.hip", "kernels_cov_update.hip", "kernels_update16.hip", "kernels_update16r.hip", "kernels_misc.hip", "kernels_ogive.hip", "kernels_evd.hip", "exchange.hip", "stft.hip", "resident.hip", "kernels_resident_m4.hip", "kernels_resident_m8.hip", "kernels_resident_m6.hip", "kernels_resident_m2.hip"]
HEADERS = [os.path.join(CSRC, "oiva_internal.h"), os.path.join(CSRC, "oiva_device.h"), os.path.join(CSRC, "update_chain.h"), os.path.join(CSRC, "cov_arith.h"), os.path.join(CSRC, "demix_arith.h"), os.path.join(CSRC, "resident.h"), os.path.join(CSRC, "host_io.h"), os.path.join(CSRC, "resident_kernel.inc"), os.path.join(REPO, "include", "overiva_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file flags (measured choices).  The X-resident kernels keep 128 accumulators in the architectural
# registers and 128 floats of X in the accumulator file: with the scheduler's default register-pressure model they end
# in scratch memory (40-450 bytes per lane), with the more exact trackers they do not.  The resource remarks of these
# files are kept next to the objects (tests/test_host.py checks that no resident kernel uses scratch).
_REMARKS = ("-Rpass-analysis=kernel-resource-usage",)
_RESIDENT = ("-mllvm", "-amdgpu-use-amdgpu-trackers=1") + _REMARKS
EXTRA_FLAGS = {"kernels_resident_m8.hip": _RESIDENT, "kernels_resident_m4.hip": _RESIDENT, "kernels_resident_m6.hip": _RESIDENT,
               "kernels_resident_m2.hip": _RESIDENT}
# every file that holds kernels leaves its resource remarks next to its object: tests/test_host.py checks that no kernel of the
# iteration spills to scratch memory (round 5: a select of an (re, im) pair put 80 bytes per lane of update_det16_kernel there)
for _src in SOURCES:
    if _src.startswith("kernels_") or _src == "stft.hip":
        EXTRA_FLAGS.setdefault(_src, _REMARKS)


def _compile(src, extra=()):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    extra = tuple(extra) + tuple(EXTRA_FLAGS.get(src, ())) + tuple(os.environ.get("OIVA_EXTRA_" + src.split(".")[0].upper(), "").split())
    cmd = [_hipcc(), *FLAGS, *extra, "-c", path, "-o", obj]
    # the command line is part of what an object depends on: a change of the per-file flags recompiles
    stamp = os.path.splitext(obj)[0] + ".cmd"
    same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
    if not same_cmd or _stale(obj, [path] + HEADERS):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(" ".join(cmd))
        if "-Rpass-analysis=kernel-resource-usage" in extra:
            # the resource remarks go next to the object; everything else the compiler said (warnings) is shown as usual
            remarks, rest, in_remark = [], [], False
            for line in r.stderr.splitlines(keepends=True):
                if "remark:" in line and "kernel-resource-usage" in line:
                    remarks.append(line)
                else:
                    rest.append(line)
            # (clang prints a source excerpt + caret under every remark: drop those lines too)
            # (... and the "In file included from" trail clang prints in front of a remark that sits in an included file)
            shown = [l for l in rest if l.strip() and not l.lstrip().startswith(("__global__", "^", "|", "In file included from"))
                     and "remarks generated" not in l and not l.lstrip()[:1].isdigit()]
            with open(os.path.splitext(obj)[0] + ".usage.txt", "w") as f:
                f.writelines(remarks)
            if shown:
                sys.stderr.writelines(shown)
        elif r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def resident_kernel_usage():
    """{kernel name: {"vgprs", "agprs", "scratch"}} of the X-resident kernels, from the remarks of the last build"""
    return kernel_usage(("kernels_resident_m4", "kernels_resident_m8", "kernels_resident_m6", "kernels_resident_m2"))


def kernel_usage(sources=None):
    """{kernel name: {"vgprs", "agprs", "scratch"}} of the kernels of the given sources (default: all that leave remarks)"""
    import re

    if sources is None:
        sources = [os.path.splitext(s)[0] for s in SOURCES if "-Rpass-analysis=kernel-resource-usage" in EXTRA_FLAGS.get(s, ())]
    out = {}
    for src in sources:
        path = os.path.join(OBJ, src + ".usage.txt")
        if not os.path.exists(path):
            continue
        name = None
        for line in open(path):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {}
                continue
            for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
                m = re.search(pat, line)
                if m and name:
                    out[name][key] = int(m.group(1))
    return out


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link liboveriva_hip.so.  Returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            path = os.path.join(OBJ, f)
            if os.path.isdir(path):                      # (variant_* directories of tools/build_variant.py)
                shutil.rmtree(path)
            else:
                os.remove(path)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lhipfft", "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
