"""CPU restatement of `ecapa_encode_batch` end to end, and the cosine call sites.  TEST INFRASTRUCTURE.

* `encode_batch_ref`  = [REF speech_encode.py:73-78]: `encoder.encode_batch(x).squeeze(1)`
  = Fbank -> sentence mean-norm -> ECAPA-TDNN with wav_lens = 1 (PARITY UNPINNED, see
  fbank_ref.py / ecapa_ref.py).
* `cosine_similarity_ref` = sklearn.metrics.pairwise.cosine_similarity, the function the
  reference calls [REF anti_stick_diarize.py:11,177] [REF diar_diag.py:14,215,219,278,355]
  (pinned: scikit-learn is installed).
* `adjacent_cosine_ref` = [REF anti_stick_diarize.py:102-104];
  `assign_windows_ref` = [REF anti_stick_diarize.py:429-434].
"""
from __future__ import annotations

import numpy as np
import torch
from sklearn.metrics.pairwise import cosine_similarity

from .ecapa_ref import EcapaRef
from .fbank_ref import speechbrain_fbank_ref, speechbrain_fbank_torch


def encode_batch_ref(state_dict: dict, wavs: np.ndarray, dtype=torch.float64, net: EcapaRef | None = None) -> np.ndarray:
    """wavs [B, n] -> [B, emb]. float64: ground truth for parity; float32: the CPU baseline."""
    net = net or EcapaRef(state_dict, dtype)
    if dtype == torch.float64:
        feats = torch.from_numpy(speechbrain_fbank_ref(np.asarray(wavs)))
    else:
        feats = speechbrain_fbank_torch(torch.from_numpy(np.asarray(wavs, dtype=np.float32)))
    return net.forward_features(feats).numpy()


def cosine_similarity_ref(x: np.ndarray) -> np.ndarray:
    return cosine_similarity(x)


def adjacent_cosine_ref(embs: np.ndarray) -> np.ndarray:
    return np.einsum("id,id->i", embs[:-1], embs[1:]) / (
        np.linalg.norm(embs[:-1], axis=1) * np.linalg.norm(embs[1:], axis=1) + 1e-8)


def assign_windows_ref(window_embs: np.ndarray, c_matrix: np.ndarray):
    w = window_embs / (np.linalg.norm(window_embs, axis=1, keepdims=True) + 1e-8)
    sim = np.dot(w, c_matrix.T)
    return np.argmax(sim, axis=1), sim
