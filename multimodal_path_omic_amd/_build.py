"""Build libmpo_hip.so for gfx950 with hipcc (in-tree, next to this file).

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box
with the repo snapshot.  `python -m multimodal_path_omic_amd._build` or __graft_entry__.build().
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpo_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]      # per-kernel registers / scratch / LDS -> <obj>.usage.json


# Per-file additions.  bag_selfattn.hip: its kernels are vector-ALU bound (softmax arithmetic per score element beside the
# MFMAs); left to its default the compiler parks MFMA results in AGPRs and spends 10-28 % of the vector instructions of
# every loop trip on v_accvgpr_read / _write copies.  None of these kernels needs the second register file.
FILE_FLAGS = {"bag_selfattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


_USAGE_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes", "VGPRs Spill": "vgpr_spill",
               "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "lds_bytes"}


def _parse_usage(stderr: str) -> dict:
    """The compiler's kernel-resource-usage remarks -> {mangled kernel name: {vgprs, scratch_bytes, ...}}.
    A kernel that quietly keeps an array in scratch memory loses its whole pipeline to it (r01: the key
    projection ran 310 us instead of 195), so tests/test_build_resources.py checks these for the hot kernels."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in _USAGE_KEYS:
                cur[_USAGE_KEYS[k.strip()]] = int(v)
    return out


def _compile(src):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _headers_mtime()):
        return obj, None
    cmd = ["hipcc", *FLAGS, *FILE_FLAGS.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    with open(obj[:-2] + ".usage.json", "w") as f:
        json.dump(_parse_usage(r.stderr), f, indent=1, sort_keys=True)
    return obj, r.stderr


def resource_usage() -> dict:
    """{mangled kernel name: usage} over every built translation unit (empty before the first build)."""
    out = {}
    if os.path.isdir(OBJ_DIR):
        for f in sorted(os.listdir(OBJ_DIR)):
            if f.endswith(".usage.json"):
                with open(os.path.join(OBJ_DIR, f)) as fh:
                    out.update(json.load(fh))
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        results = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in results if o.endswith(".o")]
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
