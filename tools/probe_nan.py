import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from speech_diarization_amd import ops
dev = torch.device("cuda", 0)
v = torch.tensor([[float("nan"), float("inf"), -float("inf"), 1e6, -1e6, 1.5, 0.0, -2.25] * 4], device=dev)
p = ops.split16_pack(v).float()
print("pack hi", p[0, :8].tolist(), "lo", p[0, 32:40].tolist())
g = torch.Generator().manual_seed(5)
B, T, cin, cout = 2, 150, 128, 128
x = torch.randn(B * T, cin, generator=g).to(dev)
w = torch.randn(cout, cin, 3, generator=g) / np.sqrt(3 * cin)
ws, s = ops.pack_weight_split16(w, dev)
xn = x.clone(); xn[200, 5] = float("nan")
y = ops.conv1d_cl_split16(xn, ws, s, T, cin=cin, dil=2, act=None, narrow=True)
print("narrow split conv rows 198..202 col0:", y[198:203, 0].tolist())
wp = ops.pack_weight(w, dev)
y32 = ops.conv1d_cl(xn, wp, T, cin=cin, dil=2)
print("exact f32 conv rows 198..202 col0:", y32[198:203, 0].tolist())
y16 = ops.conv1d_cl(xn.half(), ops.pack_weight(w, dev, torch.float16), T, cin=cin, dil=2)
print("f16 conv rows 198..202 col0:", y16[198:203, 0].float().tolist())
