#!/usr/bin/env python3
"""Parallel driver to the reference's teacher.py (train / reconstruct / checkpoint loop, teacher.py:27-112) on the
MI355X implementation: same model calls, no TensorFlow session, no plotting.

  python examples/teacher.py --teacher runs/teacher --steps 200                    # synthetic waves (simple_audio)
  python examples/teacher.py --teacher runs/teacher --tfrecord nsynth.tfrecord     # NSynth clips (nsynth.py)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-wavenet_amd", "dropin"))
import numpy as np                                   # noqa: E402
from model import WaveNetAutoEncoder                 # noqa: E402  (the dropin shim -> sr-wavenet_amd.model)
from nsynth import NsynthDataReader                  # noqa: E402
from simple_audio import generate_wave_batch         # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--teacher", type=str, default="teachers/%d" % int(time.time() * 1000))
    p.add_argument("--tfrecord", type=str, default=None, help="NSynth TFRecord file; default: synthetic waves")
    p.add_argument("--audio-max-length", type=int, default=16000)
    p.add_argument("--latent-channels", type=int, default=16)
    p.add_argument("--pool-stride", type=int, default=512)
    p.add_argument("--batch-size", type=int, default=4)
    p.add_argument("--num-samples", type=int, default=4096)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--print-steps", type=int, default=50)
    p.add_argument("--layers", type=int, default=30)
    a = p.parse_args(argv)
    dilations = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3)[:a.layers]
    data = NsynthDataReader(a.tfrecord, a.batch_size, a.num_samples, audio_max_length=a.audio_max_length) if a.tfrecord else None
    teacher = WaveNetAutoEncoder(input_size=a.num_samples, condition_size=0, num_mixtures=5, dilations=dilations,
                                 latent_channels=a.latent_channels, skip_channels=128, pool_stride=a.pool_stride,
                                 learning_rate=1e-4)                                     # teacher.py:61
    teacher.load(a.teacher)
    loss = None
    for step in range(a.steps):
        x = data.next()[0] if data else generate_wave_batch(a.batch_size, a.num_samples)[0].astype(np.float32)
        loss = teacher.train(x, None)                                                     # teacher.py:78
        if step % a.print_steps == 0:
            regen = teacher.reconstruct(x, None); enc = teacher.encode(x, None)           # teacher.py:83-84
            print(step, float(loss), "reconstruction rms %.3f" % float(np.sqrt(np.mean(regen ** 2))), "encoding", enc.shape, flush=True)
        teacher.save(a.teacher, step, force=False)                                        # once per minute, teacher.py:110
    teacher.save(a.teacher, a.steps - 1, force=True)
    return float(loss)


if __name__ == "__main__":
    main()
