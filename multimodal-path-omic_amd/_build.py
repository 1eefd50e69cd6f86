"""Build libmpo_hip.so for gfx950 with hipcc (in-tree, next to this file).

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box
with the repo snapshot.  `python -m multimodal_path_omic_amd._build` or __graft_entry__.build().
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpo_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _headers_mtime()):
        return obj, None
    cmd = ["hipcc", *FLAGS, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    return obj, r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        results = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in results]
    rebuilt = any(msg is not None for _, msg in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[mpo build] {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, {len(objs)} objects, rebuilt={rebuilt})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
