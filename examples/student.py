#!/usr/bin/env python3
"""Parallel driver to the reference's student.py (distillation loop, student.py:57-160): encode with the frozen
teacher, draw logistic noise, ``train_fast``; same model calls, no TensorFlow session, no plotting.

  python examples/student.py --teacher runs/teacher --student runs/student --steps 200
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-wavenet_amd", "dropin"))
import numpy as np                                   # noqa: E402
from model import ParallelWaveNet                    # noqa: E402
from nsynth import NsynthDataReader                  # noqa: E402
from simple_audio import generate_wave_batch         # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--teacher", type=str, required=True, help="directory a WaveNetAutoEncoder was saved to")
    p.add_argument("--student", type=str, default="students/%d" % int(time.time() * 1000))
    p.add_argument("--tfrecord", type=str, default=None)
    p.add_argument("--audio-max-length", type=int, default=16000)
    p.add_argument("--latent-channels", type=int, default=16)
    p.add_argument("--pool-stride", type=int, default=512)
    p.add_argument("--batch-size", type=int, default=4)
    p.add_argument("--num-samples", type=int, default=4096)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--print-steps", type=int, default=50)
    p.add_argument("--layers", type=int, default=30)
    p.add_argument("--flows", type=int, default=4)
    a = p.parse_args(argv)
    dilations = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3)[:a.layers]
    data = NsynthDataReader(a.tfrecord, a.batch_size, a.num_samples, audio_max_length=a.audio_max_length) if a.tfrecord else None
    student = ParallelWaveNet(input_size=a.num_samples, condition_size=0, dilations=dilations, teacher=a.teacher,
                              dilation_channels=32, skip_channels=128, num_flows=a.flows,
                              latent_channels=a.latent_channels, pool_stride=a.pool_stride, alpha=1.0, beta=1.0,
                              gamma=1.0, learning_rate=1e-4)                              # student.py:82
    sess = None                                                                           # accepted and ignored
    student.load(sess, a.student)
    rng = np.random.default_rng(0)
    loss = None
    for step in range(a.steps):
        x = data.next()[0] if data else generate_wave_batch(a.batch_size, a.num_samples)[0].astype(np.float32)
        encoding = student.encode(sess, x, None)                                          # student.py:95
        noise = rng.logistic(0, 1, (a.batch_size, a.num_samples)).astype(np.float32)      # student.py:100
        loss, power_loss = student.train_fast(sess, noise, x, encoding, None)             # student.py:107
        if step % a.print_steps == 0:
            entropy = student.getEntropy_fast(sess, noise, encoding, None)
            print("Step: {:6d} | Entropy: {} | Power Loss: {:.4f} | Total Loss: {:.4f}".format(step, entropy, power_loss, loss), flush=True)
        student.save(sess, a.student, step, force=False)
    student.save(sess, a.student, a.steps - 1, force=True)
    return float(loss)


if __name__ == "__main__":
    main()
