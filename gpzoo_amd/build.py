"""Builds libgpzoo_hip.so (gfx950) in-tree with hipcc.

    python -m gpzoo_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpzoo_hip.so")
SOURCES = ["runtime.hip", "kfill.hip", "gemm.hip", "gemmw.hip", "gemmp.hip", "diag128.hip", "coop.hip", "factor.hip", "kgrad.hip", "poisson.hip", "svgp.hip", "vnngp.hip", "collective.hip"]
HEADERS = ["common.h", "gemm.h", "cov.h", "gemmw.h", "gemmp.h", "diag128.h", "factor.h", os.path.join("..", "..", "include", "gpzoo_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


HASH_MARK = b"GPZ_SRC_HASH:"


def source_hash():
    """sha256 (32 hex digits) over the names and contents of every file the library is built from, or None when the
    sources are not there (a binary-only deployment).  The library carries the value it was built from
    (gpz_source_hash): a binary older than its sources is recognised by content, whatever the file times say."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        path = os.path.join(CSRC, f)
        if not os.path.exists(path):
            return None
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(path, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:32]


def embedded_hash(lib_path: str = LIB):
    """The source hash a built library carries, read from the file (no dlopen); None if absent."""
    try:
        blob = open(lib_path, "rb").read()
    except OSError:
        return None
    i = blob.find(HASH_MARK)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(HASH_MARK):j].decode("ascii", "replace")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    return embedded_hash() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = [f'-DGPZ_SOURCE_HASH="{source_hash()}"'] if src == "runtime.hip" else []
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
