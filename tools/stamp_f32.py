#!/usr/bin/env python3
"""Diagnostic (build_native.py --stamp): cycle shares of conv_gemm_f32_kernel's K loop, wave 0 of each workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import ops, _native
cin, cout = int(sys.argv[1]), int(sys.argv[2])
taps = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # e.g. `128 128 3 tee`: a Res2Net conv with its tee + tee_add epilogue
tee_mode = len(sys.argv) > 4 and sys.argv[4] == "tee"
dev = torch.device("cuda", 0); T = 201; M = int(os.environ.get("SEGS", "1024")) * T
x = torch.randn(M, cin, device=dev); w = torch.randn(cout, cin, taps) / (cin * taps) ** 0.5
wp = ops.pack_weight(w, dev); out = torch.empty(M, 1024 if tee_mode else cout, device=dev)
kw = dict(cin=cin, dil=2 if taps > 1 else 1, act="relu", bias=torch.randn(cout, device=dev), scale=torch.rand(cout, device=dev) + 0.5, shift=torch.randn(cout, device=dev))
if tee_mode:
    tee = torch.empty(M, cout, device=dev)
    kw.update(out=out, o_col0=128, tee=tee, tee_lo=0, tee_hi=cout, tee_add=out, ta_col0=256)
else:
    kw.update(out=out)
for _ in range(3):
    ops.conv1d_cl(x, wp, T, **kw)
torch.cuda.synchronize()
lib = _native.load(); n = 8192 * 10; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_c32_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_c32_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 10).astype(np.float64)
nk = taps * (cin // 32); tot = st[:, :4].sum(1)
print(f"cin={cin} cout={cout}: cycles per K step (64 MFMAs = 4096 cycles of one wave's matrix work; two waves share a SIMD)")
for i, nm in enumerate(["fragment reads + MFMA", "fetch issue", "vmcnt wait + stage write", "barrier"]):
    print(f"  {nm:26s} {np.median(st[:, i]) / nk:8.0f} cycles ({np.median(st[:, i] / np.maximum(tot, 1)) * 100:5.1f} %)")
print(f"  total                      {np.median(tot) / nk:8.0f} cycles per K step")
print(f"  prologue (launch -> K loop) {np.median(st[:, 4]):8.0f} cycles, epilogue (K loop -> stores retired) {np.median(st[:, 5]):8.0f} cycles, "
      f"K loop {np.median(tot):8.0f} cycles: loop share of the tile {np.median(tot / (tot + st[:, 4] + st[:, 5])) * 100:5.1f} %")
print(f"  epilogue parts: acc->LDS {np.median(st[:, 6]):6.0f}, barrier {np.median(st[:, 7]):6.0f}, params+LDS reads+math+store issue {np.median(st[:, 8]):6.0f}, "
      f"store drain {np.median(st[:, 5] - st[:, 6] - st[:, 7] - st[:, 8]):6.0f} cycles")
