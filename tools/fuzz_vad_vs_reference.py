#!/usr/bin/env python3
"""Fuzz `vad.mask_to_segments` / `morph_open_close` / `hysteresis_binarize` against the REFERENCE's functions,
imported with tests/golden/make_golden.py's inert stubs.  Build container only (/root/reference must exist).

    python -B tools/fuzz_vad_vs_reference.py [n_masks]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402


def main(n_masks: int) -> int:
    make_golden.install_stubs()
    sys.path.insert(0, make_golden.REF)
    sys.dont_write_bytecode = True
    import vad as rvad
    from speech_diarization_amd import vad as mine

    rng = np.random.default_rng(7)
    bad = 0
    for it in range(n_masks):
        hop_ms = float(rng.choice([2.5, 5.0, 7.5, 10.0, 12.5, 16.0, 20.0, 32.0]))
        n = int(rng.integers(1, 1500))
        probs = np.clip(0.5 + 0.5 * np.sin(np.arange(n) / rng.uniform(3, 40)) + rng.normal(0, 0.2, n), 0, 1).astype(np.float32)
        on, off = [(0.6, 0.4), (0.5, 0.5), (0.7, 0.2)][it % 3]
        m_ref = rvad.hysteresis_binarize(probs, on, off)
        m = mine.hysteresis_binarize(probs, on, off)
        if not np.array_equal(m, m_ref):
            bad += 1
            print("hysteresis differs", it)
            continue
        open_ms, close_ms = float(rng.choice([0.0, 30.0, 80.0])), float(rng.choice([0.0, 40.0, 100.0]))
        a, b = rvad.morph_open_close(m, hop_ms, open_ms, close_ms), mine.morph_open_close(m, hop_ms, open_ms, close_ms)
        if not np.array_equal(a, b):
            bad += 1
            print("morph differs", it)
            continue
        if it % 2:
            a[n - int(rng.integers(0, 30)):] = True
        for pad in (0.0, 12.5, 40.0, 80.0, 200.0):
            ms, mg = float(rng.choice([25.0, 150.0, 250.0])), float(rng.choice([5.0, 100.0, 250.0]))
            want = [(float(x), float(y)) for x, y in rvad.mask_to_segments(a, hop_ms, ms, mg, pad)]
            got = mine.mask_to_segments(a, hop_ms, ms, mg, pad)
            if got != want:
                bad += 1
                print("segments differ", it, hop_ms, pad, [p for p in zip(got, want) if p[0] != p[1]][:2])
    print(f"{n_masks} masks x 5 pads: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1600))
