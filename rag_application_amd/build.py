"""Build libhx.so (HIP, gfx950) in-tree with hipcc.  `python -m rag_application_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libhx.so")
# (source, object, extra defines): sparse2.hip is built for both segment sizes (kernels.hpp)
SOURCES = [("scan.hip", "scan.o", ()), ("scan8.hip", "scan8.o", ()), ("select.hip", "select.o", ()),
           ("prep.hip", "prep.o", ()), ("shardx.hip", "shardx.o", ()),
           ("sparse2.hip", "sparse2_v32k.o", ("HX_SP_VARIANT=v32k", "HX_SEG_DOCS=32768", "HX_SP_THREADS=512")),
           ("sparse2.hip", "sparse2_v64k.o", ("HX_SP_VARIANT=v64k", "HX_SEG_DOCS=65536", "HX_SP_THREADS=1024")),
           ("sprescore.hip", "sprescore.o", ()),
           ("spbuild.hip", "spbuild.o", ()), ("engine.hip", "engine.o", ()), ("bm25.cpp", "bm25.o", ())]
HEADERS = ["hx_common.hpp", "kernels.hpp", "wsort.hpp", os.path.join("..", "..", "include", "hx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), lib: str = LIB, objdir: str = OBJ) -> str:
    """defines/lib/objdir: diagnostic variants (scripts/), never the shipped library."""
    OBJ_ = objdir
    os.makedirs(OBJ_, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s, o, extra in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_, o)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *[f"-D{d}" for d in (*extra, *defines)], "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_, o) for _, o, _ in SOURCES]
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
