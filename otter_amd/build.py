"""Builds libotter_gpu.so (hand-written HIP for gfx950) in-tree with hipcc.  No CPU fallback is built."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libotter_gpu.so")
SOURCES = ["otg_api.hip", "wfa_edit.hip", "myers_edit.hip", "wfa_affine.hip", "wfa_affine_reg.hip", "wfa_adaptive.hip", "cluster.hip", "poa.hip", "pipeline.hip", "emit.hip", "ingest.hip", "bedfa.hip", "dispatch.hip", "gather.hip"]
# -ffp-contract=off: the reference's clustering decisions are FP64 comparisons made without FMA
# contraction; fused operations are written explicitly where glibc uses them.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libotter_gpu.so (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "otter_gpu.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


TOOL_SRC = os.path.join(HERE, "..", "tools", "otter_assemble.cpp")
TOOL = os.path.join(HERE, "..", "tools", "otter_assemble")


def build_tool(force=False):
    """the command-line host of the dispatcher (plain C++ over the C-ABI); rebuilt when its source or the library is newer"""
    if not os.path.exists(TOOL_SRC) or not os.path.exists(LIB):
        return
    if not force and os.path.exists(TOOL) and os.path.getmtime(TOOL) > max(os.path.getmtime(TOOL_SRC), os.path.getmtime(LIB), os.path.getmtime(os.path.join(HERE, "..", "include", "otter_gpu.h"))):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", TOOL, TOOL_SRC, "-L" + HERE, "-lotter_gpu", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"])


def build(force=False, verbose=False, jobs=4):
    if not force and not needs_build():
        build_tool()
        return LIB
    cc = hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + [os.path.join(HERE, "..", "include", "otter_gpu.h")]
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) > max([os.path.getmtime(path)] + [os.path.getmtime(h) for h in headers]):
            continue
        cmd = [cc] + FLAGS + os.environ.get("OTG_EXTRA_HIPCC_FLAGS", "").split() + ["-c", path, "-o", obj]      # (measurement builds: -DOTG_REG_TIMING, ...)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        if len(procs) >= jobs:
            _drain(procs)
    _drain(procs)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-pthread", "-ldl"]      # zlib: BGZF blocks of the BAM ingest; threads: its region slices
    subprocess.check_call(cmd)
    build_tool(force=True)
    return LIB


def _drain(procs):
    while procs:
        src, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
        if out.strip():
            sys.stderr.write(out.decode(errors="replace"))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
