#!/bin/bash
# A/B on the GPU box: the NaN-keeping ReLU (product) against the v_max form (variant built with -DSD_RELU_VMAX), interleaved processes.
set -e
v=speech-diarization_amd/variants/libsd_hip_vmax.so
for round in 1 2; do
  for p in f32 f16 f32s; do
    for b in 5000 32; do
      reps=12; [ $b = 32 ] && reps=300
      echo -n "keepnan "; python tools/loop_embed.py --batch $b --reps $reps --precision $p
      echo -n "vmax    "; SD_EXPERIMENT=1 SD_HIP_LIB=$v python tools/loop_embed.py --batch $b --reps $reps --precision $p
    done
  done
done
