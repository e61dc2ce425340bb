#!/usr/bin/env python3
"""Build container: a second copy of the library with extra compiler flags on some sources, for A/B measurements on the GPU
box (`OIVA_LIB=overiva_amd/liboveriva_hip_NAME.so python ...`).  usage: build_variant.py NAME "-DFLAG ..." [source.hip ...]
(default sources: the two X-resident translation units)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from overiva_amd import build as b

name, flags = sys.argv[1], sys.argv[2].split()
srcs = sys.argv[3:] or ["kernels_resident_m4.hip", "kernels_resident_m8.hip"]
b.build_library()
odir = os.path.join(b.OBJ, "variant_" + name)
os.makedirs(odir, exist_ok=True)
objs = []
for src in b.SOURCES:
    obj = os.path.join(b.OBJ, os.path.splitext(src)[0] + ".o")
    if src in srcs:
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        extra = [f for f in b.EXTRA_FLAGS.get(src, ()) if not f.startswith("-Rpass")]
        cmd = [b._hipcc(), *b.FLAGS, *extra, *flags, "-c", os.path.join(b.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
    objs.append(obj)
lib = os.path.join(b.PKG, f"liboveriva_hip_{name}.so")
r = subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lhipfft", "-o", lib], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print("built", lib)
