#!/usr/bin/env python3
"""Build a VARIANT of libmpo_hip.so next to the product library for same-box A/B timing (box-to-box spread of the bench is
+-2 %, more than most single changes).
    python tools/build_variant.py NAME -DFLAG [...]   ->  multimodal_path_omic_amd/libmpo_hip_NAME.so
On the GPU box swap it in for one bench run:  cp libmpo_hip.so keep.so; cp libmpo_hip_NAME.so libmpo_hip.so; bench; cp keep.so ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_path_omic_amd import _build as B      # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
objdir = os.path.join(B.CSRC, "build_" + name)
os.makedirs(objdir, exist_ok=True)
flags = [f for f in B.FLAGS if not f.startswith("-Rpass")] + extra


def comp(src):
    obj = os.path.join(objdir, src[:-4] + ".o")
    r = subprocess.run(["hipcc", *flags, *B.FILE_FLAGS.get(src, []), "-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return obj


with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(comp, B._sources()))
out = os.path.join(B.HERE, f"libmpo_hip_{name}.so")
subprocess.run(["hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out, *objs], check=True)
print(out, os.path.getsize(out) // 1024, "KiB")
