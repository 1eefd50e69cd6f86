"""Diagnostic: which part of the K2 key projection kernel bounds it?  Builds csrc/keyproj.hip with KP_VARIANT = 0..7
(bit 0: no stores, bit 1: no LDS reads / MFMAs, bit 2: no global loads) into /tmp and times each on 480 000 rows."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
CSRC = os.path.join(ROOT, "multimodal-path-omic_amd", "csrc")
STUB = "/tmp/kp_stub.hip"
open(STUB, "w").write('#include <cstdio>\n#include <cstdarg>\nvoid mpo_set_error(const char* f, ...) { va_list a; va_start(a, f); vfprintf(stderr, f, a); va_end(a); }\n')
dev = torch.device("cuda:0")
rows, E = 480000, 256
hs = [torch.relu(torch.randn(rows, E, device=dev)).to(torch.bfloat16) for _ in range(2)]
w = torch.randn(E, E, device=dev) / 16
b = torch.randn(E, device=dev)
outs = [torch.empty(rows, E, device=dev) for _ in range(2)]
s = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 6, 5]
for v in variants:
    so = f"/tmp/kp_v{v}.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared", f"-DKP_VARIANT={v}",
                    "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "keyproj.hip"), STUB, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    fn = lib._Z19mpo_launch_key_projPKvPKfS2_PfiiP12ihipStream_t
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    def run(i):
        rc = fn(hs[i & 1].data_ptr(), w.data_ptr(), b.data_ptr(), outs[i & 1].data_ptr(), rows, E, s)
        assert rc == 0
    for i in range(4): run(i)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20): run(i)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    print(f"variant {v} (stores {'off' if v & 1 else 'on'}, mfma {'off' if v & 2 else 'on'}, loads {'off' if v & 4 else 'on'}): {us:.1f} us", flush=True)
