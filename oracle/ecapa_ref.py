"""CPU restatement of speechbrain's ECAPA-TDNN forward.  TEST INFRASTRUCTURE.

PARITY UNPINNED: speechbrain and its checkpoints are absent (see oracle/__init__.py).
This restates `ECAPA_TDNN.forward` as reached from `encoder.encode_batch(x)`
[REF speech_encode.py:73-78] and `ECAPAEncoder.forward` [REF ecapa_annote.py:13-22]
(wav_lens = all ones, so zero-padded tails count as signal — the reference's
`embed_segments` pads and passes no lengths [REF anti_stick_diarize.py:163-168]);
layer definitions per SURVEY.md Appendix A.3.

Two independent formulations:
* `EcapaRef`       — torch (F.conv1d on [B, C, T]); float32 = the timed CPU baseline,
                     float64 = ground truth for parity.
* `ecapa_forward_numpy` — deliberately naive numpy on [B, T, C] with explicit reflect
                     gathers and einsum, for cross-checking `EcapaRef` at small width.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
ASP_EPS = 1e-12


def _geometry(sd: dict):
    n_blocks = 0
    while f"blocks.{n_blocks + 1}.tdnn1.conv.conv.weight" in sd:
        n_blocks += 1
    scale = sd["blocks.1.tdnn1.conv.conv.weight"].shape[0] // sd["blocks.1.res2net_block.blocks.0.conv.conv.weight"].shape[0]
    return n_blocks, scale


class EcapaRef:
    def __init__(self, state_dict: dict, dtype=torch.float32):
        self.dtype = dtype
        self.sd = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state_dict.items()}
        self.n_blocks, self.scale = _geometry(state_dict)

    # speechbrain Conv1d(padding="same", padding_mode="reflect")
    def _conv(self, x, name, dilation=1):
        w, b = self.sd[f"{name}.conv.weight"], self.sd[f"{name}.conv.bias"]
        pad = dilation * (w.shape[2] - 1) // 2
        if pad:
            x = F.pad(x, (pad, pad), mode="reflect")
        return F.conv1d(x, w, b, dilation=dilation)

    def _bn(self, x, name):
        g, b = self.sd[f"{name}.norm.weight"], self.sd[f"{name}.norm.bias"]
        rm, rv = self.sd[f"{name}.norm.running_mean"], self.sd[f"{name}.norm.running_var"]
        shp = (1, -1, 1) if x.dim() == 3 else (1, -1)
        return (x - rm.view(shp)) / torch.sqrt(rv.view(shp) + BN_EPS) * g.view(shp) + b.view(shp)

    # TDNNBlock: norm(activation(conv(x)))
    def _tdnn(self, x, name, dilation=1):
        return self._bn(torch.relu(self._conv(x, f"{name}.conv", dilation)), f"{name}.norm")

    def _se_res2net(self, x, i, dilation):
        p = f"blocks.{i}"
        residual = x
        x = self._tdnn(x, f"{p}.tdnn1")
        chunks = torch.chunk(x, self.scale, dim=1)
        ys = []
        for j, c in enumerate(chunks):
            if j == 0:
                y = c
            elif j == 1:
                y = self._tdnn(c, f"{p}.res2net_block.blocks.{j - 1}", dilation)
            else:
                y = self._tdnn(c + y, f"{p}.res2net_block.blocks.{j - 1}", dilation)
            ys.append(y)
        x = torch.cat(ys, dim=1)
        x = self._tdnn(x, f"{p}.tdnn2")
        s = x.mean(dim=2, keepdim=True)
        s = torch.relu(self._conv(s, f"{p}.se_block.conv1"))
        s = torch.sigmoid(self._conv(s, f"{p}.se_block.conv2"))
        return s * x + residual

    @torch.no_grad()
    def forward_features(self, feats: torch.Tensor, return_intermediates: bool = False):
        """feats [B, T, n_mels] (mean-normalised fbank) -> [B, emb]."""
        x = feats.to(self.dtype).transpose(1, 2)
        inter = {}
        x = self._tdnn(x, "blocks.0")
        inter["block0"] = x
        xl = []
        for i in range(1, self.n_blocks + 1):
            x = self._se_res2net(x, i, dilation=i + 1)
            xl.append(x)
            inter[f"block{i}"] = x
        x = torch.cat(xl, dim=1)
        x = self._tdnn(x, "mfa")
        inter["mfa"] = x
        # AttentiveStatisticsPooling(global_context=True), all-ones mask
        L = x.shape[2]
        mean = x.mean(dim=2)
        std = torch.sqrt(((x - mean.unsqueeze(2)) ** 2).mean(dim=2).clamp(min=ASP_EPS))
        attn = torch.cat([x, mean.unsqueeze(2).expand(-1, -1, L), std.unsqueeze(2).expand(-1, -1, L)], dim=1)
        attn = self._conv(torch.tanh(self._tdnn(attn, "asp.tdnn")), "asp.conv")
        attn = F.softmax(attn, dim=2)
        mu = (attn * x).sum(dim=2)
        sd_ = torch.sqrt((attn * (x - mu.unsqueeze(2)) ** 2).sum(dim=2).clamp(min=ASP_EPS))
        pooled = torch.cat([mu, sd_], dim=1).unsqueeze(2)
        inter["pooled"] = pooled
        pooled = self._bn(pooled, "asp_bn")
        emb = F.conv1d(pooled, self.sd["fc.conv.weight"], self.sd["fc.conv.bias"]).squeeze(2)
        if return_intermediates:
            return emb, inter
        return emb


# ------------------------------------------------------------------ naive numpy version

def _np_conv(x, w, b, dil):
    """x [B, T, Cin], w [Cout, Cin, k] -> [B, T, Cout], 'same' reflect padding."""
    B, T, _ = x.shape
    k = w.shape[2]
    out = np.zeros((B, T, w.shape[0]), dtype=x.dtype) + b[None, None, :]
    t = np.arange(T)
    for j in range(k):
        tt = t + (j - k // 2) * dil
        tt = np.where(tt < 0, -tt, tt)
        tt = np.where(tt >= T, 2 * (T - 1) - tt, tt)
        out += np.einsum("btc,oc->bto", x[:, tt, :], w[:, :, j])
    return out


def ecapa_forward_numpy(state_dict: dict, feats: np.ndarray, dtype=np.float64) -> np.ndarray:
    sd = {k: np.asarray(v, dtype=dtype) for k, v in state_dict.items()}
    n_blocks, scale = _geometry(state_dict)

    def bn(x, name):
        return (x - sd[f"{name}.norm.running_mean"]) / np.sqrt(sd[f"{name}.norm.running_var"] + BN_EPS) \
            * sd[f"{name}.norm.weight"] + sd[f"{name}.norm.bias"]

    def tdnn(x, name, dil=1):
        y = _np_conv(x, sd[f"{name}.conv.conv.weight"], sd[f"{name}.conv.conv.bias"], dil)
        return bn(np.maximum(y, 0.0), f"{name}.norm")

    x = tdnn(np.asarray(feats, dtype=dtype), "blocks.0")
    outs = []
    for i in range(1, n_blocks + 1):
        p = f"blocks.{i}"
        res = x
        t1 = tdnn(x, f"{p}.tdnn1")
        hid = t1.shape[2] // scale
        ys = [t1[:, :, :hid]]
        for j in range(1, scale):
            c = t1[:, :, j * hid:(j + 1) * hid]
            inp = c if j == 1 else c + ys[-1]
            ys.append(tdnn(inp, f"{p}.res2net_block.blocks.{j - 1}", i + 1))
        t2 = tdnn(np.concatenate(ys, axis=2), f"{p}.tdnn2")
        s = t2.mean(axis=1, keepdims=True)
        s = np.maximum(_np_conv(s, sd[f"{p}.se_block.conv1.conv.weight"], sd[f"{p}.se_block.conv1.conv.bias"], 1), 0.0)
        s = 1.0 / (1.0 + np.exp(-_np_conv(s, sd[f"{p}.se_block.conv2.conv.weight"], sd[f"{p}.se_block.conv2.conv.bias"], 1)))
        x = s * t2 + res
        outs.append(x)
    h = tdnn(np.concatenate(outs, axis=2), "mfa")
    T = h.shape[1]
    mean = h.mean(axis=1, keepdims=True)
    std = np.sqrt(np.maximum(((h - mean) ** 2).mean(axis=1, keepdims=True), ASP_EPS))
    a = np.concatenate([h, np.repeat(mean, T, 1), np.repeat(std, T, 1)], axis=2)
    a = np.tanh(tdnn(a, "asp.tdnn"))
    a = _np_conv(a, sd["asp.conv.conv.weight"], sd["asp.conv.conv.bias"], 1)
    a = np.exp(a - a.max(axis=1, keepdims=True))
    a = a / a.sum(axis=1, keepdims=True)
    mu = (a * h).sum(axis=1)
    sg = np.sqrt(np.maximum((a * (h - mu[:, None, :]) ** 2).sum(axis=1), ASP_EPS))
    pooled = np.concatenate([mu, sg], axis=1)
    pooled = (pooled - sd["asp_bn.norm.running_mean"]) / np.sqrt(sd["asp_bn.norm.running_var"] + BN_EPS) \
        * sd["asp_bn.norm.weight"] + sd["asp_bn.norm.bias"]
    return pooled @ sd["fc.conv.weight"][:, :, 0].T + sd["fc.conv.bias"]
