import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import speech_encode, synth
enc = speech_encode.using_ecapa_encoder(); dev = enc.device
wavs = synth.synthetic_segments(5, 32, 32000)
def step_times(reps=50):
    acc = np.zeros(6)
    for _ in range(reps):
        t = [time.perf_counter()]
        with torch.inference_mode():
            x = torch.from_numpy(np.ascontiguousarray(wavs)).float(); t.append(time.perf_counter())
            xd = x.to(dev, dtype=torch.float32, non_blocking=True); t.append(time.perf_counter())
            e = enc.engine.embed(xd); t.append(time.perf_counter())
            e2 = e.unsqueeze(1).squeeze(1); t.append(time.perf_counter())
            c = e2.cpu(); t.append(time.perf_counter())
            y = c.numpy(); t.append(time.perf_counter())
        acc += np.diff(t)
    return acc / reps * 1e3
step_times(5)
print("from_numpy %.3f | to(dev) %.3f | embed enqueue %.3f | views %.3f | .cpu() (waits for the GPU) %.3f | numpy %.3f ms" % tuple(step_times()))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): speech_encode.ecapa_encode_batch(wavs)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
