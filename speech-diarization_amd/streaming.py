"""BASELINE.json configs[3]: streaming feed, fixed-shape hop captured in a hipGraph.

A real-time feed of `channels` microphones delivers `hop_s` of audio per hop; every hop the last
`window_s` of each channel is embedded.  The shape is fixed ([channels, window samples]), so the
whole hop — HIP fbank, the ~45 launches of the ECAPA-TDNN forward — is captured once into a
hipGraph and replayed (launch-bound inner loop: one graph launch instead of ~45 kernel launches).
Nothing in the reference streams; the embedding call being replaced is the same
`ecapa_encode_batch` [REF speech_encode.py:73-78].

`OnlineClusterer` is a host-side leader-follower agglomeration over the hop embeddings (cosine
threshold, running unit-norm centroids) — the "online AHC" of configs[3].
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import EmbeddingEngine


class StreamingEmbedder:
    """The last `window_s` of every channel live in ONE device buffer that is written in place: a doubled ring
    ([channels, 2 * win]; sample i of the ring is stored at i and at i + win, so the newest window is always the
    contiguous span starting at the write position q mod win).  The fbank kernel reads the windows where they lie
    (`sd_fbank_windows_f32` with a per-channel start offset held in a small device array), so a hop costs two
    hop-sized copies (four when the hop wraps around the end of the ring, i.e. when the window is not a whole number
    of hops) and no allocation; the captured graph replays against the same buffer and start array (its launches hold
    their addresses).  Any 0 < hop <= window is accepted."""

    def __init__(self, engine: EmbeddingEngine, channels: int = 16, window_s: float = 2.0, hop_s: float = 0.25, sr: int = 16000,
                 use_graph: bool = True):
        self.engine = engine
        self.channels, self.sr = channels, sr
        self.win = int(round(window_s * sr))
        self.hop = int(round(hop_s * sr))
        if not 0 < self.hop <= self.win:
            raise ValueError(f"hop ({self.hop} samples) must be positive and at most the window ({self.win})")
        dev = engine.device
        self._cap = 2 * self.win
        self._buf = torch.zeros((channels, self._cap), dtype=torch.float32, device=dev)
        # window start of channel c in the flat buffer = c * cap + (write position mod win)
        self._base = (torch.arange(channels, dtype=torch.int64) * self._cap).to(dev)
        self._starts = self._base.clone()               # static address (captured by the graph)
        self._q = 0                                     # write position in the ring, 0 <= q < win
        self._static_out = None
        self._graph = None
        self.use_graph = use_graph
        if use_graph:
            self._capture()

    @property
    def ring(self) -> torch.Tensor:
        """The current windows as a [channels, win] tensor (a gathered copy: diagnostics and tests only)."""
        idx = self._starts[:, None] + torch.arange(self.win, device=self._buf.device)[None, :]
        return self._buf.view(-1)[idx]

    def _embed(self) -> torch.Tensor:
        return self.engine.embed_windows(self._buf.view(-1), self._starts, self.win)

    def _capture(self) -> None:
        dev = self.engine.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                     # warm-up outside capture: allocates the engine workspace
                self._embed()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_out = self._embed()
        self.engine.freeze_workspace()

    def push(self, chunk: torch.Tensor) -> torch.Tensor:
        """chunk: f32 [channels, hop samples] (device or host) -> embeddings [channels, 192] of the updated windows."""
        if chunk.shape != (self.channels, self.hop):
            raise ValueError(f"expected a [{self.channels}, {self.hop}] hop, got {tuple(chunk.shape)}")
        chunk = chunk.to(self._buf.device, dtype=torch.float32, non_blocking=True)
        q, win = self._q, self.win
        first = min(self.hop, win - q)                                   # the piece up to the end of the ring ...
        for lo, n, src in ((q, first, chunk[:, :first]), (0, self.hop - first, chunk[:, first:])):   # ... and the wrapped rest
            if n > 0:
                self._buf[:, lo:lo + n].copy_(src)                       # in place, both images of the ring
                self._buf[:, lo + win:lo + win + n].copy_(src)
        self._q = (q + self.hop) % win
        torch.add(self._base, self._q, out=self._starts)                 # in place: the captured launches read this array
        if self._graph is None:
            return self._embed()
        self._graph.replay()
        return self._static_out


class OnlineClusterer:
    """Leader-follower online agglomeration: an embedding joins the closest centroid if its cosine is at
    least `threshold`, else it founds a new speaker; centroids are running sums re-normalised on use."""

    def __init__(self, threshold: float = 0.7, max_speakers: int = 16):
        self.threshold, self.max_speakers = threshold, max_speakers
        self.sums: list[np.ndarray] = []
        self.counts: list[int] = []

    def assign(self, embs: np.ndarray) -> np.ndarray:
        out = np.empty(len(embs), dtype=int)
        for i, e in enumerate(np.asarray(embs, dtype=np.float64)):
            e = e / (np.linalg.norm(e) + 1e-8)
            if self.sums:
                C = np.stack(self.sums)
                C = C / (np.linalg.norm(C, axis=1, keepdims=True) + 1e-8)
                sims = C @ e
                k = int(np.argmax(sims))
                if sims[k] >= self.threshold or len(self.sums) >= self.max_speakers:
                    self.sums[k] += e
                    self.counts[k] += 1
                    out[i] = k
                    continue
            self.sums.append(e.copy())
            self.counts.append(1)
            out[i] = len(self.sums) - 1
        return out
