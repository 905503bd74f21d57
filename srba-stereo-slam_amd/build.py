"""Build libssk_hip.so (HIP kernels + C-ABI, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU.  Objects go to csrc/_obj (git-ignored); the shared library lands
next to this file so it travels with a gpurun snapshot.  No torch / pybind: the boundary is the plain
C-ABI of include/ssk_capi.h.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libssk_hip.so")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",          # float parity contract: no FMA contraction on host or device
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"),
] + os.environ.get("SSK_EXTRA_CFLAGS", "").split()     # e.g. -DSSK_LM_STAMPS for a diagnostic build


# per-source flags.  ssk_fast.hip: its lane-0 atomics are already one per wavefront or workgroup, and the optimizer's aggregated form waits for the
# returned value where the atomic is issued -- the tile draw of fast_nms_kernel is meant to return a whole tile later
SRC_FLAGS = {"ssk_fast.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS + SRC_FLAGS.get(src, [])).encode())
    inc = os.path.join(ROOT, "include")
    files = [os.path.join(CSRC, f) for f in [src] + sorted(x for x in os.listdir(CSRC) if x.endswith(".h"))]
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]     # every public header (ssk_brief_pattern.h too)
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + SRC_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(verbose=True, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if verbose:
        print("libssk_hip.so: %s (%d sources, %s)" % (LIB, len(srcs), "rebuilt" if rebuilt else "up to date"))
    return LIB


def build_host(verbose=True):
    """C++ facade (host/) + its demo driver, linked against libssk_hip.so (rpath: the package directory)."""
    host = os.path.join(HERE, "host")
    out = os.path.join(host, "facade_demo")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cpp") and not f.endswith("_tool.cpp")]
    # libssk_facade.so: the facade classes + the C-ABI of the RBA engine (include/ssk_rba.h), for callers without C++
    flib = os.path.join(HERE, "libssk_facade.so")
    lsrcs = [f for f in srcs if not f.endswith("facade_demo.cpp")]
    hdrs = [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h")]
    deps = srcs + hdrs + [LIB]
    # CPU-only drivers of host-side logic (no GPU library needed): keyframe record I/O (host/ssk_kf_io.h); keyframe policy
    # (host/ssk_kf_policy.h) + vocabulary file reader (host/ssk_bow_vocab_io.h, zlib)
    for name, libs in (("kf_io_tool", []), ("policy_tool", ["-lz"])):
        tool, tool_src = os.path.join(host, name), os.path.join(host, name + ".cpp")
        if not os.path.exists(tool) or os.path.getmtime(tool) < max(os.path.getmtime(d) for d in hdrs + [tool_src]):
            r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), tool_src, "-o", tool] + libs,
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("%s build failed:\n%s" % (name, r.stderr))
    if not os.path.exists(flib) or os.path.getmtime(flib) < max(os.path.getmtime(d) for d in lsrcs + hdrs + [LIB, os.path.join(ROOT, "include", "ssk_rba.h")]):
        cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include")] + lsrcs + \
              ["-L", HERE, "-lssk_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN", "-o", flib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("libssk_facade.so build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose:
            print("facade library: %s" % flib)
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-I", os.path.join(ROOT, "include")] + srcs + \
          ["-L", HERE, "-lssk_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/..", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host facade build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if verbose:
        print("host facade: %s" % out)
    return out


if __name__ == "__main__":
    build()
    build_host()
