"""Build libedison_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m edison_amd.build [--force]

Host-side C (tables.c, model.c, legacy.c) is compiled as C, the shim and the kernels as HIP; everything is
linked into edison_amd/csrc/libedison_hip.so, which is what ctypes (edison_amd/_lib.py), cgo or any other
FFI binds. The .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libedison_hip.so")
C_SOURCES = ["tables.c", "tables_q15.c", "tables_f32.c", "model.c", "model_net.c", "model_net_mm.c", "legacy.c"]
HIP_SOURCES = ["edison_hip.hip", "edison_q15.hip", "edison_f32.hip", "edison_net.hip", "edison_stream.hip", "edison_dist.hip", "mfcc_kernels.hip",
               "mfcc_q15_kernels.hip", "mfcc_f32_kernels.hip", "cnn_kernels.hip", "cnn_mfma_kernels.hip",
               "cnn_net_kernels.hip", "cnn_net_mfma_kernels.hip"]
HEADERS = ["edison_internal.h", "edison_ctx.h", "mfcc_fft.h", os.path.join("..", "..", "include", "edison_hip.h")]
ARCH = "gfx950"
# per-file code-generation flags. -amdgpu-sched-strategy=max-ilp: the machine scheduler orders the straight-line blocks of the two
# MFCC kernels for instruction-level parallelism instead of occupancy (which the launch bounds fix anyway: 160 / 155 VGPRs either
# way); measured interleaved in one process (tools/lab/ab_mfcc.py, round 3): float kernel +1.3 ... +2.4 %, Q15 kernel +0.4 %,
# CNN kernel -0.2 % (not applied there); outputs bit-identical.
PER_FILE_FLAGS = {
    "mfcc_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "mfcc_q15_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libedison_hip.so cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in C_SOURCES + HIP_SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    objs = []
    common = ["-O3", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter"] + os.environ.get("ED_CFLAGS", "").split()
    for src in C_SOURCES:
        obj = os.path.join(bdir, src + ".o")
        cmd = [hipcc, "-x", "c", "-std=gnu11"] + common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    for src in HIP_SOURCES:
        obj = os.path.join(bdir, src + ".o")
        # -fno-slp-vectorize: measured (tools/ubench/rates, profiles/r02_ubench_rates.txt) a v_pk_{add,mul,fma}_f32 costs 4.3-4.7
        # cycles of vector issue against 2.3-2.6 for a back-to-back VOP2 add / mul / fmac and 3.7-4.0 for a VOP3 v_fma_f32: packing
        # pays when whole register PAIRS stay pairs (the two-frame MFCC kernel is written that way by hand), not when the
        # compiler's SLP pass packs adjacent scalar ops and pays v_mov's to build the pairs (-15 % on the one-frame kernel)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-fno-slp-vectorize"] + common + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", OUT, "-lm", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


EXAMPLES_BIN = os.path.join(os.path.dirname(HERE), "examples", "bin")


def build_examples(force=False):
    """The C hosts under examples/ that bench.py runs (plain C against include/edison_hip.h, linked to the library with
    an rpath relative to the binary, so the pair travels with the tree): examples/bin/host_stream_latency."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "host_stream_latency.c")
    exe = os.path.join(EXAMPLES_BIN, "host_stream_latency")
    if not force and os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(OUT)):
        return exe
    cc = shutil.which("gcc") or shutil.which("cc") or _hipcc()
    os.makedirs(EXAMPLES_BIN, exist_ok=True)
    subprocess.check_call([cc, "-O2", "-Wall", src, "-I", os.path.join(root, "include"), "-L", CSRC, "-ledison_hip",
                           "-Wl,-rpath,$ORIGIN/../../edison_amd/csrc", "-o", exe])
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_examples(force="--force" in sys.argv))
