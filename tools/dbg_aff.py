import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from speech_diarization_amd import ops
n = 256
rng = np.random.default_rng(0)
x = rng.standard_normal((n, 192)).astype(np.float32)
xn = x / np.linalg.norm(x, axis=1, keepdims=True)
ref = (xn.astype(np.float64) @ xn.astype(np.float64).T)
got = ops.cosine_affinity(torch.from_numpy(x).cuda(), split16=True).cpu().numpy()
for name, blk, rb in (("direct", (slice(0, 128), slice(128, 256)), ref[0:128, 128:256]), ("mirror", (slice(128, 256), slice(0, 128)), ref[128:256, 0:128])):
    g = got[blk]
    bad = np.abs(g - rb) > 1e-5
    print(name, "bad", bad.sum(), "of", bad.size)
    if bad.any():
        idx = np.argwhere(bad)[:12]
        for (r, c) in idx:
            # where does the value come from?
            src = np.argwhere(np.abs(ref - g[r, c]) < 2e-6)
            print("  at", r, c, "value from", src[:3].tolist())
print("diag tiles bad", (np.abs(got[0:128, 0:128] - ref[0:128, 0:128]) > 1e-5).sum(), (np.abs(got[128:, 128:] - ref[128:, 128:]) > 1e-5).sum())
