#!/usr/bin/env python3
"""Build A/B variants of libedison_hip.so without touching the product library.

    tools/lab/mkvariant.py name=FLAGS[@file.hip[,file2.hip]] ...

Each variant recompiles the named kernel files (default mfcc_kernels.hip) with the extra FLAGS and links them with
the product build's other objects (edison_amd/csrc/build/*.o) into edison_amd/csrc/abl/libedison_hip_<name>.so
(git-ignored, travels to the GPU box). Every variant is compiled with -DED_LAB: the kernel files accept their knobs only then
(and export an ed_lab_build_* marker, so that a lab library is never mistaken for the product one). A file given as path:alt.hip replaces the source text, e.g.
    new=-DX=1@mfcc_kernels.hip:tools/lab/mfcc_try.hip
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from edison_amd import build as B

def one(spec):
    name, rest = spec.split("=", 1)
    flags, _, files = rest.partition("@")
    files = [f for f in (files or "mfcc_kernels.hip").split(",") if f]
    odir = os.path.join(B.CSRC, "abl", name); os.makedirs(odir, exist_ok=True)
    replaced, objs = set(), []
    for f in files:
        base, _, alt = f.partition(":")
        src = os.path.join(ROOT, alt) if alt else os.path.join(B.CSRC, base)
        obj = os.path.join(odir, base + ".o")
        cmd = [B._hipcc(), "--offload-arch=" + B.ARCH, "-std=c++17", "-fno-slp-vectorize", "-O3", "-fPIC", "-I" + B.CSRC,
               "-Wno-unused-value", "-DED_LAB"] + B.PER_FILE_FLAGS.get(base, []) + flags.split() + ["-x", "hip", "-c", src, "-o", obj]
        subprocess.check_call(cmd)
        replaced.add(base + ".o"); objs.append(obj)
    for o in sorted(os.listdir(os.path.join(B.CSRC, "build"))):
        if o.endswith(".o") and o not in replaced:
            objs.append(os.path.join(B.CSRC, "build", o))
    out = os.path.join(B.CSRC, "abl", "libedison_hip_%s.so" % name)
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC"] + objs + ["-o", out, "-lm", "-ldl"])
    return out

if __name__ == "__main__":
    B.build()  # product objects must exist
    with ThreadPoolExecutor(max_workers=6) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print(o)
