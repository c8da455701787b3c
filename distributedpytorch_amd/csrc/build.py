"""In-tree native build: hipcc cross-compiles every kernel for gfx950 (no GPU needed) and g++ builds
the host runtime.  Outputs land next to the package so they travel with the source tree:

  distributedpytorch_amd/lib/libdpamd_hip.so   HIP/CDNA4 kernels (C ABI, loaded with ctypes)
  distributedpytorch_amd/lib/libdpamd_host.so  C++ host runtime (sampler, IDX reader, bucket planner,
                                               TCP rendezvous store) - no HIP dependency
  distributedpytorch_amd/lib/libdpamd_comm.so  C++ RCCL communicator (own HIP comm stream + event fencing)

Usage: python -m distributedpytorch_amd.csrc.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(HERE, "build")
ROCM = os.environ.get("ROCM_HOME", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

HIP_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wno-unused-result", "-DNDEBUG"]
# -ffast-math only where it was measured to matter: the convolution translation units (MFMA builtins are unaffected;
# the epilogues' scalar math is).  The optimizer, BatchNorm, loss, pooling, elementwise and augmentation kernels divide,
# take square roots and are tested for inf / NaN propagation: they are built with IEEE semantics.
FAST_MATH_TUS = {"conv_igemm.hip", "conv_x6p.hip", "conv_x6s.hip", "conv_wgrad.hip"}
HOST_FLAGS = ["-O2", "-fPIC", "-std=c++17", "-Wall", "-pthread"]


def _sources(sub, exts):
    d = os.path.join(HERE, sub)
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def _deps_hash(src, extra):
    h = hashlib.sha1()
    for f in [src] + extra:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(cmd, src, obj, stamp, digest):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"compile failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(digest)
    return src, r.stderr


def _check_kernel_stubs(so):
    """A kernel whose body does not compile in the HOST pass (a device-only type or builtin outside a
    __HIP_DEVICE_COMPILE__ guard inside a template) silently loses its launch stub and the library then fails to load
    on the GPU box with 'undefined symbol: __device_stub__...'.  Catch that here, where the build runs."""
    r = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True)
    missing = [l.split()[-1] for l in r.stdout.splitlines() if "__device_stub__" in l]
    if missing:
        raise RuntimeError(f"{os.path.basename(so)}: kernel launch stubs missing (host pass did not instantiate them): "
                           + ", ".join(missing[:4]))


def build(force=False, jobs=None, verbose=True, libdir=None, defines=()):
    """libdir / defines: an A/B build of the kernels with extra -D flags into another directory (selected at run time with
    DPA_LIB_DIR); the default build is what ships."""
    global LIBDIR, OBJDIR
    if libdir:
        LIBDIR = os.path.abspath(libdir)
        OBJDIR = os.path.join(LIBDIR, "obj")
    extra = [f"-D{d}" for d in defines]
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = jobs or min(8, os.cpu_count() or 4)
    hip_src = _sources("hip", (".hip",))
    hip_hdr = _sources("hip", (".h",))
    host_src = _sources("host", (".cpp",))
    host_hdr = _sources("host", (".h",))
    comm_src = _sources("comm", (".cpp", ".hip"))
    comm_hdr = _sources("comm", (".h",))
    tasks = []
    objs = {"hip": [], "host": [], "comm": []}
    for kind, srcs, hdrs in (("hip", hip_src, hip_hdr), ("host", host_src, host_hdr), ("comm", comm_src, comm_hdr)):
        for src in srcs:
            base = os.path.splitext(os.path.basename(src))[0]
            obj = os.path.join(OBJDIR, f"{kind}_{base}.o")
            stamp = obj + ".sha1"
            objs[kind].append(obj)
            if kind == "hip":
                fm = ["-ffast-math"] if os.path.basename(src) in FAST_MATH_TUS else []
                cmd = [HIPCC] + HIP_FLAGS + fm + extra + ["-c", src, "-o", obj]
            elif kind == "comm" and src.endswith(".hip"):      # the mesh transport: host runtime + its reduce kernels
                cmd = [HIPCC] + HIP_FLAGS + [f"-I{ROCM}/include", "-pthread", "-c", src, "-o", obj]
            elif kind == "comm":
                cmd = ["g++"] + HOST_FLAGS + ["-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include", "-c", src, "-o", obj]
            else:
                cmd = ["g++"] + HOST_FLAGS + ["-c", src, "-o", obj]
            digest = _deps_hash(src, hdrs) + hashlib.sha1(" ".join(cmd).encode()).hexdigest()      # sources AND flags
            if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
                continue
            tasks.append((cmd, src, obj, stamp, digest))
    relink = bool(tasks) or force
    if tasks:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for src, err in ex.map(lambda t: _compile(*t), tasks):
                if verbose:
                    print(f"[build] compiled {os.path.relpath(src, PKG)}", flush=True)
                    if err.strip():
                        print(err, file=sys.stderr)
    hip_so = os.path.join(LIBDIR, "libdpamd_hip.so")
    host_so = os.path.join(LIBDIR, "libdpamd_host.so")
    if objs["hip"] and (relink or not os.path.exists(hip_so)):
        # No rpath on purpose: libamdhip64.so.7 resolves to the copy torch has already loaded, so the
        # kernels and torch share ONE HIP runtime (streams/events/pointers are interchangeable).
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", hip_so] + objs["hip"]
        subprocess.run(cmd, check=True)
        _check_kernel_stubs(hip_so)
        if verbose:
            print(f"[build] linked {os.path.relpath(hip_so, PKG)}", flush=True)
    if objs["host"] and (relink or not os.path.exists(host_so)):
        cmd = ["g++", "-shared", "-fPIC", "-pthread", "-o", host_so] + objs["host"]
        subprocess.run(cmd, check=True)
        if verbose:
            print(f"[build] linked {os.path.relpath(host_so, PKG)}", flush=True)
    comm_so = os.path.join(LIBDIR, "libdpamd_comm.so")
    if objs["comm"] and (relink or not os.path.exists(comm_so)):
        # like the kernel library: no rpath, so librccl.so.1 / libamdhip64.so.7 bind to the copies torch has loaded
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-pthread", "-o", comm_so] + objs["comm"] + \
              [f"-L{ROCM}/lib", "-lrccl", "-lrt"]
        subprocess.run(cmd, check=True)
        _check_kernel_stubs(comm_so)
        if verbose:
            print(f"[build] linked {os.path.relpath(comm_so, PKG)}", flush=True)
    return hip_so, host_so


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--libdir", default=None, help="A/B build into this directory (run with DPA_LIB_DIR=<dir>)")
    ap.add_argument("-D", dest="defines", action="append", default=[], help="extra preprocessor define for the kernel TUs")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, libdir=a.libdir, defines=a.defines)
