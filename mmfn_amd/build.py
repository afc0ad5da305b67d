"""Compile the gfx950 kernels into mmfn_amd/lib/libmmfn_hip.so (in-tree, travels with gpurun)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmmfn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("MMFN_EXTRA_FLAGS", "").split()


def _fresh(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mmfn_hip.h")]
    objs = []
    procs = []
    for src in srcs:
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and _fresh(obj, [src] + hdrs):
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or not _fresh(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_comm(force=force, verbose=verbose)
    return LIB


COMM_SRC = os.path.join(HERE, "csrc_comm", "comm.cpp")
COMM_LIB = os.path.join(LIBDIR, "libmmfn_comm.so")


def build_comm(force=False, verbose=True):
    """libmmfn_comm.so: the RCCL gradient all-reduce behind a C ABI (include/mmfn_comm.h); host code only, linked against
    librccl (the soname torch's own RCCL answers to, so both share one library instance in a process)."""
    hdr = os.path.join(HERE, "..", "include", "mmfn_comm.h")
    if not force and _fresh(COMM_LIB, [COMM_SRC, hdr]):
        return COMM_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", COMM_LIB, COMM_SRC, "-L%s/lib" % rocm, "-lrccl",
           "-Wl,-rpath,%s/lib" % rocm]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return COMM_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
