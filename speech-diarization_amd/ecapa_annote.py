"""Embedding-model adapters with the reference's class contract [REF ecapa_annote.py:6-33].

The reference subclasses `pyannote.audio.core.model.Model`; pyannote is not part of this
path, so these are plain `torch.nn.Module`s exposing the attributes pyannote's pipeline
reads (`.dimension`, `.forward(waveforms) -> [B, dimension]`).  `forward` also accepts the
`[B, 1, n]` layout pyannote 3.1 hands to its embedding callable, and `__call__(waveforms,
masks=None)` returns what `forward` returns.
"""
from __future__ import annotations

import torch

from .speech_encode import eres2netv2_encode_batch, using_ecapa_encoder, using_eres2netv2_encoder


class ECAPAEncoder(torch.nn.Module):
    def __init__(self, device: str | int = 0):
        super().__init__()
        self.model = using_ecapa_encoder(device)
        self.dimension = 192  # [REF ecapa_annote.py:11]
        self.sample_rate = 16000

    def forward(self, waveforms: torch.Tensor, masks: torch.Tensor | None = None) -> torch.Tensor:
        """waveforms: (batch, num_samples) or (batch, 1, num_samples) -> (batch, dimension),
        on the encoder's device [REF ecapa_annote.py:13-22]."""
        if waveforms.dim() == 3:
            if waveforms.shape[1] != 1:
                raise ValueError("expected mono waveforms [B, 1, n]")
            waveforms = waveforms[:, 0, :]
        if masks is not None and not bool(torch.all(masks != 0)):
            raise NotImplementedError("masked embedding is not part of the reference's adapter")
        return self.model.encode_batch(waveforms).squeeze(1)


class ERes2NetV2Encoder(torch.nn.Module):
    """Kept for interface parity [REF ecapa_annote.py:25-33]; constructing it fails like the reference
    does without its ONNX file (the ERes2NetV2 network is outside this hot path)."""

    def __init__(self, device: str | int = 0):
        super().__init__()
        self.model = using_eres2netv2_encoder()
        self.dimension = 192

    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        y = eres2netv2_encode_batch(waveforms.cpu().numpy())
        return torch.from_numpy(y).to(waveforms.device)
