"""`flac.py`: the stems' FLAC writer [REF diarization_baseline.py:95-103] and the reader beside it.  No FLAC codec exists in the image to
decode the files with, so the format is pinned from the specification's side: the CRC check values of the two polynomials, every header
field of a written file read back BY HAND (bit offsets of the format, not the module's reader), STREAMINFO's MD5, lossless round trips
through the module's decoder over every subframe kind the writer emits, a stream BUILT BY HAND in this file with the features the writer
never emits (LPC subframes, mid / side and left / side channels, wasted bits, two Rice partitions with an escape, 5-bit Rice parameters) to
hold the decoder on its own, and corruption (a flipped bit, a truncated file, a wrong MD5) being refused."""
import hashlib
import struct

import numpy as np
import pytest

from speech_diarization_amd import audio_io, flac, synth


def test_crc_check_values():
    assert flac.crc8(b"123456789") == 0xF4            # CRC-8 (poly 0x07, init 0): the catalogue's check value
    assert flac.crc16(b"123456789") == 0xFEE8         # CRC-16/BUYPASS (poly 0x8005, init 0, no reflection)
    assert flac._utf8_number(0) == b"\x00" and flac._utf8_number(0x7F) == b"\x7f" and flac._utf8_number(0x80) == b"\xc2\x80"
    assert flac._utf8_number(0x7FF) == b"\xdf\xbf" and flac._utf8_number(0x800) == b"\xe0\xa0\x80" and flac._utf8_number(0xFFFF) == b"\xef\xbf\xbf"
    assert flac._utf8_number(0x10000) == b"\xf0\x90\x80\x80"


def _voice(seconds, sr=16000, seed=3, channels=1):
    conv = synth.synthetic_conversation(seconds, 2, seed=seed)
    y = conv.wav[: int(seconds * sr)].astype(np.float32)
    return y if channels == 1 else np.stack([y, 0.5 * np.roll(y, 7)] + [0.25 * y] * (channels - 2))[:channels]


def test_written_file_field_by_field(tmp_path):
    y = _voice(1.0)                                    # 16 000 samples: three blocks of 4096 + one of 3712
    p = tmp_path / "a.flac"
    flac.write_flac16(p, y, 16000)
    d = p.read_bytes()
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    assert d[:4] == b"fLaC" and d[4] == 0x80 and int.from_bytes(d[5:8], "big") == 34          # one, last, STREAMINFO block of 34 bytes
    info = d[8:42]
    assert struct.unpack(">HH", info[:4]) == (4096, 4096)
    packed = int.from_bytes(info[10:18], "big")
    assert packed >> 44 == 16000 and (packed >> 41) & 7 == 0 and (packed >> 36) & 31 == 15 and packed & ((1 << 36) - 1) == 16000
    assert info[18:34] == hashlib.md5(pcm.astype("<i2").tobytes()).digest()
    # frames: sync 0xFFF8, block size code, rate code 0101 (16 kHz), one channel, 16 bits, the frame number, CRC-8, ..., CRC-16
    pos, sizes = 42, []
    for number, n in enumerate((4096, 4096, 4096, 3712)):
        assert d[pos] == 0xFF and d[pos + 1] == 0xF8
        assert d[pos + 2] == ((12 if n == 4096 else 7) << 4 | 5) and d[pos + 3] == (0 << 4 | 4 << 1) and d[pos + 4] == number
        hdr = 5 + (2 if n != 4096 else 0)
        if n != 4096:
            assert struct.unpack(">H", d[pos + 5:pos + 7])[0] == n - 1
        assert flac.crc8(d[pos:pos + hdr]) == d[pos + hdr]
        nxt = d.find(b"\xff\xf8", pos + hdr)
        while nxt != -1 and not (flac.crc16(d[pos:nxt - 2]) == int.from_bytes(d[nxt - 2:nxt], "big")):
            nxt = d.find(b"\xff\xf8", nxt + 1)             # (a sync pattern inside the payload)
        end = len(d) if nxt == -1 else nxt
        assert flac.crc16(d[pos:end - 2]) == int.from_bytes(d[end - 2:end], "big")
        sizes.append(end - pos)
        pos = end
    assert pos == len(d)
    assert int.from_bytes(info[4:7], "big") == min(sizes) and int.from_bytes(info[7:10], "big") == max(sizes)
    assert len(d) < 0.75 * 2 * len(pcm)                    # fixed predictors + Rice: well under the PCM size on voiced audio


@pytest.mark.parametrize("case", ["voice", "stereo", "noise", "silence", "dc", "short", "one", "clip", "eight"])
def test_round_trip_is_lossless(tmp_path, case):
    rng = np.random.default_rng(5)
    y = {"voice": lambda: _voice(1.3), "stereo": lambda: _voice(0.7, channels=2), "noise": lambda: rng.uniform(-1, 1, 9000).astype(np.float32),
         "silence": lambda: np.zeros(5000, np.float32), "dc": lambda: np.full(4096 + 17, 0.25, np.float32), "short": lambda: _voice(0.01),
         "one": lambda: np.asarray([0.5], np.float32), "clip": lambda: np.clip(3.0 * _voice(0.6), -1, 1), "eight": lambda: _voice(0.3, channels=8)}[case]()
    p = tmp_path / "x.flac"
    audio_io.write_flac16(p, y, 16000)
    pcm, sr, bps = flac.read_flac(p)
    want = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int32)
    want = want[None, :] if want.ndim == 1 else want
    assert sr == 16000 and bps == 16 and np.array_equal(pcm, want)
    back, _ = audio_io.read_audio(p, 16000, mono=False)
    assert np.array_equal(back, (want / 32768.0).astype(np.float32))
    if case == "voice":                                    # the reader resamples and mixes down like the WAV path
        wav = tmp_path / "x.wav"
        audio_io.write_wav16(wav, y, 16000)
        for kw in (dict(sr=16000, mono=True), dict(sr=8000, mono=True)):
            assert np.array_equal(audio_io.read_audio(p, **kw)[0], audio_io.read_audio(wav, **kw)[0])


# ---- a stream built by hand, with what the writer never emits

class _W:
    def __init__(self):
        self.b = []

    def u(self, v, n):
        self.b += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def s(self, v, n):
        self.u(v & ((1 << n) - 1), n)

    def rice(self, r, k):
        u = (r << 1) if r >= 0 else ((-r) << 1) - 1
        self.b += [0] * (u >> k) + [1]
        if k:
            self.u(u & ((1 << k) - 1), k)

    def bytes(self):
        return np.packbits(np.asarray(self.b + [0] * (-len(self.b) % 8), dtype=np.uint8)).tobytes()


def _lpc_subframe(w, x, bps, coef, shift, prec, parts, k5=False, wasted=0):
    """An LPC subframe of the samples x (python ints, already shifted right by `wasted`), residual in 2^parts Rice partitions; the
    second partition (if any) escapes to raw 12-bit residuals."""
    order, n = len(coef), len(x)
    w.u(0, 1); w.u(31 + order, 6); w.u(1 if wasted else 0, 1)
    if wasted:
        w.b += [0] * (wasted - 1) + [1]
    for v in x[:order]:
        w.s(v, bps - wasted)
    w.u(prec - 1, 4); w.s(shift, 5)
    for c in coef:
        w.s(c, prec)
    res = [x[i] - (sum(c * x[i - 1 - j] for j, c in enumerate(coef)) >> shift) for i in range(order, n)]
    w.u(1 if k5 else 0, 2); w.u(parts, 4)
    per, at = n >> parts, 0
    for part in range(1 << parts):
        cnt = per - (order if part == 0 else 0)
        chunk = res[at:at + cnt]; at += cnt
        if part == 1:
            w.u(31 if k5 else 15, 5 if k5 else 4); w.u(12, 5)
            for r in chunk:
                assert -2048 <= r < 2048
                w.s(r, 12)
        else:
            w.u(3, 5 if k5 else 4)
            for r in chunk:
                w.rice(r, 3)


def _frame(number, n, chan_code, bps_code, body):
    hdr = bytes([0xFF, 0xF8, (6 << 4) | 0, (chan_code << 4) | (bps_code << 1)]) + flac._utf8_number(number) + bytes([n - 1])
    hdr += bytes([flac.crc8(hdr)])
    fr = hdr + body
    return fr + struct.pack(">H", flac.crc16(fr))


def _stream(frames, sr, ch, bps, total, pcm):
    raw = np.ascontiguousarray(np.asarray(pcm).T).astype("<i2").tobytes()
    info = struct.pack(">HH", 16, 64) + bytes(6) + ((sr << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | total).to_bytes(8, "big") + hashlib.md5(raw).digest()
    padding = bytes([0x81]) + (8).to_bytes(3, "big") + bytes(8)                  # a second (last) metadata block: PADDING, 8 bytes
    return b"fLaC" + bytes([0x00]) + (34).to_bytes(3, "big") + info + padding + b"".join(frames)


def test_decoder_on_a_hand_built_stream(tmp_path):
    rng = np.random.default_rng(11)
    n = 64
    t = np.arange(2 * n)
    left = (3000 * np.sin(t / 5.0) + rng.integers(-20, 20, 2 * n)).astype(int)
    right = (left * 0.9).astype(int) + rng.integers(-15, 15, 2 * n)
    frames = []
    # frame 0: mid / side, LPC order 2 (12-bit coefficients, shift 10) on mid (two partitions: Rice k = 3, then an escape), order 1 on side
    l, r = left[:n], right[:n]
    mid, side = (l + r) >> 1, l - r
    w = _W()
    _lpc_subframe(w, [int(v) for v in mid], 16, [1900, -900], 10, 12, 1)
    _lpc_subframe(w, [int(v) for v in side], 17, [1000], 10, 12, 0, k5=True)
    frames.append(_frame(0, n, 10, 4, w.bytes()))
    # frame 1: left / side; the left channel has two wasted bits (all samples multiples of 4), fixed-free LPC order 3
    l2, r2 = (left[n:] >> 2) << 2, right[n:]
    w = _W()
    _lpc_subframe(w, [int(v) >> 2 for v in l2], 16, [2500, -1400, 300], 10, 13, 0, wasted=2)
    _lpc_subframe(w, [int(v) for v in (l2 - r2)], 17, [900], 10, 11, 0)
    frames.append(_frame(1, n, 8, 4, w.bytes()))
    want = np.stack([np.concatenate([l, l2]), np.concatenate([r, r2])])
    p = tmp_path / "h.flac"
    p.write_bytes(_stream(frames, 22050, 2, 16, 2 * n, want))
    pcm, sr, bps = flac.read_flac(p)
    assert sr == 22050 and bps == 16 and np.array_equal(pcm, want)


def test_corruption_is_refused(tmp_path):
    p = tmp_path / "c.flac"
    flac.write_flac16(p, _voice(0.6), 16000)
    good = p.read_bytes()
    flac.read_flac(p)
    for where in (60, len(good) // 2, len(good) - 5):                            # inside a frame: some CRC (or the MD5) must notice
        bad = bytearray(good); bad[where] ^= 0x10
        p.write_bytes(bytes(bad))
        with pytest.raises(ValueError):
            flac.read_flac(p)
    p.write_bytes(good[: len(good) - 300])
    with pytest.raises(ValueError):
        flac.read_flac(p)
    bad = bytearray(good); bad[8 + 20] ^= 0xFF                                   # STREAMINFO's MD5
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="MD5"):
        flac.read_flac(p)
    p.write_bytes(b"RIFF" + good[4:])
    with pytest.raises(ValueError, match="not a FLAC"):
        flac.read_flac(p)
