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
        # the compiler's per-kernel resource remarks (registers, scratch, LDS) are kept beside the object: check_scratch() reads them
        cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        log = open(obj[:-2] + ".resources.txt", "w")
        procs.append((src, subprocess.Popen(cmd, stderr=log), log))
    for src, p, log in procs:
        rc = p.wait()
        log.close()
        if rc != 0:
            sys.stderr.write(open(log.name).read()[-4000:])
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or not _fresh(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_comm(force=force, verbose=verbose)
    check_scratch(verbose=verbose)
    return LIB


SCRATCH_ALLOW = os.path.join(CSRC, "scratch_allowlist.txt")


def kernel_resources():
    """{kernel symbol: {"vgprs", "agprs", "scratch", "lds", "occupancy", "file"}} from the remarks hipcc wrote at compile time."""
    import re
    out = {}
    for path in sorted(glob.glob(os.path.join(LIBDIR, "*.resources.txt"))):
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {"file": os.path.basename(path)[:-len(".resources.txt")] + ".hip"}
                continue
            for key, tag in (("vgprs", "VGPRs"), ("agprs", "AGPRs"), ("scratch", "ScratchSize [bytes/lane]"),
                             ("lds", "LDS Size [bytes/block]"), ("occupancy", "Occupancy [waves/SIMD]")):
                m = re.search(r"remark:\s+%s: (\d+)" % re.escape(tag), line)
                if m and name:
                    out[name][key] = int(m.group(1))
    return out


def check_scratch(verbose=True):
    """Build-time gate: no kernel may use scratch memory (register spills, stack arrays) unless csrc/scratch_allowlist.txt names
    it (a substring of the mangled symbol per line, with the reason).  A spill in a hot kernel is a silent 5-10 % (round 5:
    attn_wg_dkv_kernel<128,3> carried 312 bytes per lane for two rounds before anybody looked)."""
    res = kernel_resources()
    if not res:
        return {}
    allow = []
    if os.path.exists(SCRATCH_ALLOW):
        allow = [l.split("#")[0].strip() for l in open(SCRATCH_ALLOW)]
        allow = [a for a in allow if a]
    bad = {k: v for k, v in res.items() if v.get("scratch", 0) > 0 and not any(a in k for a in allow)}
    if verbose:
        print("kernel resources: %d kernels, %d with scratch (%d allow-listed)" % (
            len(res), sum(1 for v in res.values() if v.get("scratch", 0) > 0),
            sum(1 for k, v in res.items() if v.get("scratch", 0) > 0) - len(bad)), flush=True)
    if bad:
        raise RuntimeError("kernels with scratch memory (spills) not in csrc/scratch_allowlist.txt:\n" + "\n".join(
            "  %s: %d bytes/lane (%s)" % (k, v["scratch"], v["file"]) for k, v in sorted(bad.items())))
    return res


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
