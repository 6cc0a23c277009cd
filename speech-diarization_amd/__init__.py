"""MI355X-native embedding hot path for speaker diarization.

Modules carry the names of the reference's flat scripts they stand in for
(`speech_encode`, `ecapa_annote`, `vad`, `anti_stick_diarize`, `diarization_baseline`)
and keep their function signatures; the arithmetic runs in hand-written HIP
kernels for gfx950 behind the C ABI of `include/sd_hip.h` (`libsd_hip.so`).

Nothing here imports the CPU oracle (`oracle/`): that is test infrastructure.
"""

__version__ = "0.1.0"

EMBEDDING_DIM = 192  # [REF ecapa_annote.py:11] [REF speech_encode.py:78]
SAMPLE_RATE = 16000
