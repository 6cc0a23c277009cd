"""Feature + embedding front door — drop-in for the reference's `speech_encode.py`.

Same callables, same array conventions (numpy in, fresh numpy out, `assert wavs.ndim == 2`):

* `fbank_batch(wavs, sr, n_mels, mean_nor)`          [REF speech_encode.py:10-38]
* `using_ecapa_encoder(device)` (lru_cache singleton)  [REF speech_encode.py:64-70]; its arithmetic ("f32" | "f16") is the
  process-wide switch `set_precision()` / env `SD_ECAPA_PRECISION`, cf. the reference's global precision knob
  [REF diarization_baseline.py:20-21]
* `ecapa_encode_batch(wavs)`                           [REF speech_encode.py:73-78]
* `using_eres2netv2_encoder()` / `eres2netv2_encode_batch(...)` [REF speech_encode.py:42-60]

Behind them: the HIP fbank kernel and the HIP ECAPA-TDNN forward (libsd_hip.so).  Tables
and weights are uploaded once per process, not per call, and a batch crosses PCIe once in
each direction (the reference rebuilds its transform per call and crosses three times).
There is no CPU fallback.
"""
from __future__ import annotations

import os
import threading
import warnings
from functools import lru_cache
from pathlib import Path

import numpy as np
import torch

from . import EMBEDDING_DIM
from .engine import EmbeddingEngine, fbank_device
from .features import FbankPlan

_THIS_DIR = Path(__file__).parent.resolve()
ECAPA_SOURCE = "LanceaKing/spkrec-ecapa-cnceleb"  # [REF speech_encode.py:67]
SYNTHETIC_SEED = 1234

# Arithmetic of the ECAPA-TDNN forward behind `using_ecapa_encoder()`: "f32" (exact f32 MFMA, the reference CPU path's
# arithmetic), "f32s" (f32-split16x3: f32 activations, the wide layers as three f16 MFMA products per value pair, f32-level
# accuracy at twice the rate), "f32ns" (the wide layers on exact f32 MFMA, only the narrow Res2Net / attention convs as split products)
# or "f16" (f16 operands, f32 accumulation: BASELINE.json configs[4]).  A process-wide switch, like the
# reference's own precision knob (`torch.backends.cuda.matmul.allow_tf32 = True`, [REF diarization_baseline.py:20-21]);
# initial value from the environment variable SD_ECAPA_PRECISION.
_PRECISIONS = ("f32", "f32s", "f32ns", "f16")
_precision = os.environ.get("SD_ECAPA_PRECISION", "f32").lower()
if _precision not in _PRECISIONS:
    raise ValueError(f"SD_ECAPA_PRECISION must be one of {_PRECISIONS}, got {_precision!r}")


def get_precision() -> str:
    return _precision


def set_precision(precision: str) -> None:
    """Select the arithmetic of the encoder singleton ("f32" | "f32s" | "f32ns" | "f16"); drops the cached encoder so that the next
    `using_ecapa_encoder()` / `ecapa_encode_batch()` call builds one with the new setting."""
    global _precision
    if precision not in _PRECISIONS:
        raise ValueError(f"precision must be one of {_PRECISIONS}, got {precision!r}")
    if precision != _precision:
        _precision = precision
        if hasattr(using_ecapa_encoder, "cache_clear"):
            using_ecapa_encoder.cache_clear()


def _cuda_device(device) -> torch.device:
    if isinstance(device, int):
        return torch.device("cuda", device)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"the HIP embedding path runs on the GPU; device={device!r} is not supported (no CPU fallback)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


@lru_cache(maxsize=4)
def _fbank_plan(kind: str, n_mels: int, sr: int, device_index: int) -> FbankPlan:
    with torch.cuda.device(device_index):
        return FbankPlan(kind, n_mels=n_mels, sr=sr)


def fbank_batch(wavs: np.ndarray,  # [B, n_samples]
                sr: int = 16000, n_mels: int = 80, mean_nor: bool = True) -> np.ndarray:
    """log-mel filterbank [B, T, n_mels]: 25 ms / 10 ms, 20 Hz .. sr/2-100 Hz, ln(x + 1e-6),
    per-utterance mean removal when `mean_nor`."""
    assert wavs.ndim == 2
    if not torch.cuda.is_available():
        raise RuntimeError("fbank_batch needs a GPU (the reference hard-codes 'cuda' too, [REF speech_encode.py:26,29])")
    dev = torch.device("cuda", torch.cuda.current_device())
    plan = _fbank_plan("torchaudio", n_mels, sr, dev.index)
    with torch.inference_mode():
        x = torch.from_numpy(np.ascontiguousarray(wavs, dtype=np.float32)).to(dev)
        feat = fbank_device(x, plan, mean_norm=mean_nor)
    return feat.cpu().numpy()  # [B, T, n_mels]


def _find_checkpoint() -> Path | None:
    env = os.environ.get("SD_ECAPA_CKPT")
    if env:
        p = Path(env)
        if not p.exists():
            raise FileNotFoundError(f"SD_ECAPA_CKPT={env} does not exist")
        return p
    for cand in (_THIS_DIR / "models" / "spkrec-ecapa-cnceleb" / "embedding_model.ckpt",
                 _THIS_DIR.parent / "models" / "spkrec-ecapa-cnceleb" / "embedding_model.ckpt"):
        if cand.exists():
            return cand
    return None


def load_ecapa_state_dict(path: str | os.PathLike | None = None) -> dict:
    """speechbrain `embedding_model.ckpt` state dict if one is available locally, else
    deterministic random-init weights of the same architecture (no network here)."""
    ckpt = Path(path) if path is not None else _find_checkpoint()
    if ckpt is not None:
        sd = torch.load(ckpt, map_location="cpu", weights_only=True)
        return {k: v.float().numpy() for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    from .synth import make_ecapa_state_dict
    warnings.warn(
        f"no local ECAPA checkpoint ({ECAPA_SOURCE} cannot be downloaded offline; set SD_ECAPA_CKPT): "
        f"using random-init weights (seed {SYNTHETIC_SEED}) — embeddings carry no trained speaker information",
        RuntimeWarning, stacklevel=2)
    return make_ecapa_state_dict(SYNTHETIC_SEED)


class HipEcapaEncoder:
    """Stands in for speechbrain's `EncoderClassifier`: `.encode_batch(wavs) -> Tensor[B, 1, 192]`."""

    def __init__(self, state_dict: dict, device, max_batch: int = 512, precision: str = "f32"):
        self.device = _cuda_device(device)
        self.precision = precision
        self.engine = EmbeddingEngine(state_dict, self.device, max_batch=max_batch, precision=precision)
        self.embedding_dim = self.engine.dim
        self._lanes = []
        self._lanes_lock = threading.Lock()

    @torch.inference_mode()
    def encode_batch(self, wavs: torch.Tensor, wav_lens: torch.Tensor | None = None, normalize: bool = False) -> torch.Tensor:
        if wavs.dim() == 1:
            wavs = wavs.unsqueeze(0)
        if wav_lens is not None and not bool(torch.all(wav_lens == 1)):
            raise NotImplementedError("relative lengths are not supported: the reference never passes wav_lens, "
                                      "zero-padded tails count as signal [REF anti_stick_diarize.py:163-168]")
        if normalize:
            raise NotImplementedError("normalize=True (speechbrain mean_var_norm_emb) is not used by the reference")
        x = wavs.to(self.device, dtype=torch.float32, non_blocking=True)
        return self.engine.embed(x).unsqueeze(1)

    __call__ = encode_batch

    # -- several independent batches in flight
    class _Lane:
        """One of the pipeline's slots: a stream, an engine (own workspace over the shared weights), pinned staging buffers."""

        def __init__(self, engine: EmbeddingEngine, device: torch.device):
            self.engine = engine
            self.stream = torch.cuda.Stream(device)
            self.done = torch.cuda.Event()
            self.host_in = None          # pinned f32, grows
            self.host_out = None         # pinned f32 [rows, dim], grows
            self.pending = None          # (index of the batch in flight, rows)

        def staging(self, rows: int, n: int, dim: int):
            if self.host_in is None or self.host_in.numel() < rows * n:
                self.host_in = torch.empty((rows * n,), dtype=torch.float32).pin_memory()
            if self.host_out is None or self.host_out.shape[0] < rows:
                self.host_out = torch.empty((rows, dim), dtype=torch.float32).pin_memory()
            return self.host_in[: rows * n].view(rows, n), self.host_out[:rows]

    @torch.inference_mode()
    def encode_batches(self, batches, lanes: int = 2) -> list:
        """`[ecapa_encode_batch(b) for b in batches]` with up to `lanes` batches in flight: the reference's callers embed a recording
        as a sequence of independent batches (32 segments zero-padded to the batch's longest [REF anti_stick_diarize.py:150-171], 128
        reassignment windows [REF :398-430]) and synchronise after every one, so the card idles while a batch crosses PCIe, and every
        kernel boundary of a small launch (the previous kernel's last workgroups draining, the next one's first ramping up) is
        exposed.  Here batch i + 1 is staged (pinned memory), copied and launched on a second stream while batch i computes: copies
        overlap compute and one launch's tail is filled by the other stream's kernels.  Every batch runs the same kernels on the
        same shapes as the one-at-a-time call, in a workspace of its own: results are BITWISE those of `ecapa_encode_batch(b)`.
        batches: iterable of float arrays [B_i, n_i] (any mix of shapes) -> list of fresh numpy arrays [B_i, 192]."""
        batches = [np.ascontiguousarray(b, dtype=np.float32) for b in batches]
        for b in batches:
            assert b.ndim == 2
        out: list = [None] * len(batches)
        if not batches:
            return out
        with self._lanes_lock:
            return self._encode_batches_locked(batches, out, max(1, min(int(lanes), len(batches))))

    def _encode_batches_locked(self, batches, out, n_lanes):
        # every lane runs on a SIBLING of self.engine (own workspace): a concurrent `encode_batch` on the main engine from another
        # thread (the reference's web UI calls the pipeline from a worker thread) never shares a workspace with a lane in flight
        while len(self._lanes) < n_lanes:
            self._lanes.append(self._Lane(self.engine.sibling(), self.device))
        use = self._lanes[:n_lanes]

        def harvest(lane):
            if lane.pending is not None:
                i, rows = lane.pending
                lane.done.synchronize()
                out[i] = lane.host_out[:rows].numpy().copy()
                lane.pending = None

        try:
            for i, b in enumerate(batches):
                lane = use[i % len(use)]
                harvest(lane)                                   # its previous batch: frees the lane's staging buffers
                rows, n = b.shape
                if rows == 0:
                    out[i] = np.empty((0, self.embedding_dim), dtype=np.float32)
                    continue
                h_in, h_out = lane.staging(rows, n, self.embedding_dim)
                h_in.numpy()[...] = b
                with torch.cuda.stream(lane.stream):
                    x = h_in.to(self.device, non_blocking=True)
                    emb = lane.engine.embed(x)
                    h_out.copy_(emb, non_blocking=True)
                    lane.done.record(lane.stream)
                lane.pending = (i, rows)
            for lane in use:
                harvest(lane)
        except BaseException:
            # a batch was refused (too short, wrong shape, ...) or the caller interrupted: let what is in flight finish and forget it, so
            # that the lanes' staging buffers are free and nothing of this call is harvested into the next one
            for lane in use:
                lane.stream.synchronize()
                lane.pending = None
            raise
        return out

    @torch.inference_mode()
    def encode_windows(self, signal, starts, n: int, rows_per_call: int = 8192, to_host: bool = True):
        """Embeddings [B, 192] of the windows `signal[starts[b] : starts[b] + n]` of one recording — what the reference's
        callers compute by slicing windows on the host and calling `ecapa_encode_batch` on batches of them
        [REF anti_stick_diarize.py:82-100, 396-430] — with ONE upload of the signal and no gathered copy
        (`EmbeddingEngine.embed_windows`).  signal: 1-d float array or tensor (host or device); starts: int array."""
        sig = signal if isinstance(signal, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32))
        sig = sig.to(self.device, dtype=torch.float32, non_blocking=True)
        st = starts if isinstance(starts, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int64))
        st = st.to(self.device, dtype=torch.int64)
        parts = [self.engine.embed_windows(sig, st[lo:lo + rows_per_call], n) for lo in range(0, int(st.numel()), rows_per_call)]
        out = torch.cat(parts) if parts else torch.empty((0, self.embedding_dim), dtype=torch.float32, device=self.device)
        return out.cpu().numpy() if to_host else out


@lru_cache(maxsize=1)
def using_ecapa_encoder(device: str | int = "cuda") -> HipEcapaEncoder:
    if not torch.cuda.is_available():
        raise RuntimeError("using_ecapa_encoder: no GPU visible; the HIP path has no CPU fallback")
    return HipEcapaEncoder(load_ecapa_state_dict(), device, precision=_precision)


def ecapa_encode_batch(wavs: np.ndarray) -> np.ndarray:
    encoder = using_ecapa_encoder()
    with torch.inference_mode():
        x = torch.from_numpy(np.ascontiguousarray(wavs)).float()
        y = encoder.encode_batch(x).squeeze(1).cpu().numpy()
    return y  # [B, 192]


def ecapa_encode_batches(batches, lanes: int = 2) -> list:
    """`[ecapa_encode_batch(b) for b in batches]`, bit for bit, with `lanes` batches in flight on the card
    (`HipEcapaEncoder.encode_batches`): what the batch loops of `embed_segments` / `frame_reassign` call."""
    return using_ecapa_encoder().encode_batches(batches, lanes=lanes)


@lru_cache(maxsize=1)
def using_eres2netv2_encoder():
    """The reference opens an ONNX file that is git-ignored and absent [REF speech_encode.py:42-50];
    the ERes2NetV2 network is outside this hot path (north_star names ECAPA)."""
    onnx_path = _THIS_DIR / "models/iic-speech_eres2netv2w24s4ep4_sv_zh-cn_16k-common.onnx"
    raise FileNotFoundError(
        f"{onnx_path}: the ERes2NetV2 ONNX model is not shipped and onnxruntime is not part of the MI355X path; "
        "use using_ecapa_encoder() / ecapa_encode_batch()")


def eres2netv2_encode_batch(wavs: np.ndarray, sr: int = 16000, num_mels: int = 80) -> np.ndarray:
    assert wavs.ndim == 2
    session = using_eres2netv2_encoder()
    features = fbank_batch(wavs, sr=sr, n_mels=num_mels, mean_nor=True)
    return session.run(None, dict(feature=features))[0]  # [B, 192]


assert EMBEDDING_DIM == 192
