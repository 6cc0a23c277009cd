#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of conv_gemm_f16_glds_kernel (needs build_native.py --stamp)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import ops, _native

def main():
    cin, cout, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    dev = torch.device("cuda", 0); T = 201; M = B * T
    x = torch.randn(M, cin, device=dev).half(); w = torch.randn(cout, cin, 1) / cin ** 0.5
    wp = ops.pack_weight(w, dev, torch.float16); bias = torch.randn(cout, device=dev)
    out = torch.empty(M, cout, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.conv1d_cl(x, wp, T, cin=cin, bias=bias, act="relu", out=out)
    torch.cuda.synchronize()
    lib = _native.load(); n = 8192 * 8
    buf = (C.c_ulonglong * n)()
    lib.sd_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.sd_debug_read_stamps(buf, n) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    nb = min(8192, ((M + 255) // 256) * ((cout + 127) // 128))
    st = st[:nb]
    d = np.diff(st[:, :7], axis=1) * 0.01   # us
    names = ["setup", "first data", "main loop", "ring->acc sync + LDS C tile", "store_tile issue", "store drain"]
    print(f"cin={cin} cout={cout} blocks={nb}")
    for i, nm in enumerate(names):
        print(f"  {nm:30s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
    print(f"  {'total per workgroup':30s} median {np.median((st[:, 6] - st[:, 0]) * 0.01):7.2f} us")
    t0 = st[:, 0].min(); print(f"  kernel span {(st[:, 6].max() - t0) * 0.01:.1f} us; starts p50 {np.median(st[:,0]-t0)*0.01:.1f} us")
main()
