"""BASELINE config 2 at FULL size (30 layers, 64/256 ch, 256-way softmax, batch 8 x 16000) through size-independent
properties.  (The direct oracle comparison at this depth and these dilations, on clips longer than the receptive
field, is tests/test_gpu_depth.py; a full 8 x 16000 batch would take the fp64 CPU oracle about a minute per pass, so
at the full batch the checks are structural.)

  causality          logits before a perturbed sample do not change (bit-exact)
  batch separability the batch-8 gradient is the mean of the two batch-4 half gradients; losses average
  directional FD     (loss(p + e v) - loss(p - e v)) / 2e == <grad, v>   (fp32 mode, whole-model direction)
  determinism        two steps from the same state are bit-identical (no atomics anywhere on the path)
  mu-law             decode(encode(x)) within half a quantisation step on 128 000 samples; encode(decode(c)) == c
"""
import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import DEV, dev

pytestmark = pytest.mark.gpu

DIL = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
B, T, R, S, C = 8, 16000, 64, 256, 256


def _engine(dt, batch=B, seed=0, share=None):
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=DIL, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=dt, learning_rate=1e-3)
    return EG.WaveNetEngine(cfg, batch, T, DEV, seed=seed, share_from=share)


def _inputs():
    K = sub("kernels")
    audio = dev(O.synthetic_audio(B, T, seed=0))
    return audio, K.mu_law_encode(audio, C)


def test_full_size_causality_and_loss_scale():
    eng = _engine(torch.bfloat16)
    audio, codes = _inputs()
    eng.set_inputs(audio, codes)
    lg0 = eng.forward(want_logits=True).clone()
    assert abs(float(eng.loss.item()) - np.log(C)) < 0.5          # near-uniform softmax at initialisation
    t0 = 9001
    a2 = audio.clone(); a2[:, t0] += 0.25
    eng.set_inputs(a2, codes)
    lg1 = eng.forward(want_logits=True)
    # RightShift (ops.py:78-80): logits[t] see audio[< t] only
    assert torch.equal(lg0[:, :t0 + 1], lg1[:, :t0 + 1])
    assert not torch.equal(lg0[:, t0 + 1], lg1[:, t0 + 1])
    # the receptive field is 1 + sum(d) = 3070 samples (+1 for the shift): nothing beyond it moves either
    rf = 1 + sum(DIL) + 1
    assert torch.equal(lg0[:, t0 + rf + 1:], lg1[:, t0 + rf + 1:])


def test_full_size_batch_separability_and_determinism():
    dt = torch.bfloat16
    full = _engine(dt)
    audio, codes = _inputs()
    full.set_inputs(audio, codes)
    full.forward(); full.backward()
    g8, l8 = full.grads.clone(), float(full.loss.item())
    full.forward(); full.backward()
    assert torch.equal(g8, full.grads) and l8 == float(full.loss.item())      # bit-reproducible
    half = _engine(dt, batch=B // 2, share=full)
    gs, ls = [], []
    for h in range(2):
        half.set_inputs(audio[h * 4:(h + 1) * 4], codes.view(B, T)[h * 4:(h + 1) * 4].contiguous())
        half.forward(); half.backward()
        gs.append(half.grads.clone()); ls.append(float(half.loss.item()))
    gm = 0.5 * (gs[0] + gs[1])
    # every op is per batch element: identical activations, only the fp32 summation order differs
    err = float((g8 - gm).norm() / g8.norm())
    assert err < 1e-4, err
    assert abs(l8 - 0.5 * (ls[0] + ls[1])) < 1e-5 * abs(l8)


def test_full_size_directional_finite_difference_fp32():
    eng = _engine(torch.float32)
    audio, codes = _inputs()
    eng.set_inputs(audio, codes)
    eng.forward(); eng.backward()
    g = eng.grads.clone()
    p0 = eng.params.clone()
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    v = torch.randn(p0.shape, device=DEV, generator=gen)
    v = v / v.norm() + g / g.norm()          # a random direction with a component along the gradient (so <g,v> is
    v = v / v.norm()                         # well above the fp32 resolution of the loss)
    ana = float((g.double() * v.double()).sum())
    eps = 1e-2
    ls = []
    for sgn in (1.0, -1.0):
        eng.params.copy_(p0 + sgn * eps * v); eng.repack()
        eng.forward()
        ls.append(float(eng.loss.double().item()))
    eng.params.copy_(p0); eng.repack()
    fd = (ls[0] - ls[1]) / (2 * eps)
    assert abs(fd - ana) < 2e-2 * abs(ana) + 1e-5, (fd, ana)


def test_full_size_mu_law_round_trip():
    K = sub("kernels")
    audio, codes = _inputs()
    dec = K.mu_law_decode(codes, C)
    # decode(encode(x)) lands in the same quantisation bin: re-encoding is the identity, all 256 bins included
    assert torch.equal(K.mu_law_encode(dec, C), codes)
    allc = torch.arange(C, dtype=torch.int32, device=DEV)
    assert torch.equal(K.mu_law_encode(K.mu_law_decode(allc, C), C), allc)
    # companded error bound: |x - decode(encode(x))| <= half a bin of the expanded grid (at most ~2.2 % of full scale)
    assert float((dec.view(-1) - audio.view(-1)).abs().max()) < 0.0222


def test_full_size_training_is_stable_and_bf16_tracks_fp32():
    """150 graph-replayed steps on one fixed batch: losses fall, stay finite, and the bf16 curve tracks the fp32 one."""
    K = sub("kernels")
    audio, codes = _inputs()
    final = {}
    for dt in (torch.bfloat16, torch.float32):
        eng = _engine(dt)
        eng.set_inputs(audio, codes)
        eng.train_step()
        l0 = float(eng.loss.item())
        eng.capture_graphs()
        for _ in range(150):
            eng.train_step_graphed()
        torch.cuda.synchronize()
        final[dt] = float(eng.loss.item())
        assert np.isfinite(final[dt]) and final[dt] < 0.6 * l0 and bool(torch.isfinite(eng.params).all())
        del eng
    assert abs(final[torch.bfloat16] - final[torch.float32]) < 0.02 * final[torch.float32], final


def test_full_size_bf16_forward_tracks_exact_fp32_mode():
    """Config 2: the bf16 engine's logits and loss against the exact-fp32 MFMA mode of the same engine (which the
    small-shape tests tie to the oracle at 1e-3)."""
    audio, codes = _inputs()
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        eng = _engine(dt, seed=3)
        eng.set_inputs(audio, codes)
        out[dt] = (eng.forward(want_logits=True).float().cpu().numpy(), float(eng.loss.item()))
        del eng
    ref, lref = out[torch.float32]
    got, lgot = out[torch.bfloat16]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 5e-2, err
    assert abs(lgot - lref) < 2e-3 * abs(lref), (lgot, lref)


# Bounds of test_full_size_bf16_gradients_track_exact_fp32_mode: TWICE the errors measured on MI355X in round 4
# (`SRWN_PRINT_ERR=1 pytest -s` prints them): relative L2 of each section of the flat gradient buffer of the bf16 default
# path against the exact-fp32 mode of the same engine on the same weights and batch -- (whole section, worst single layer).
# Measured: init_w 6.4e-3, init_b 6.6e-3, WF 7.2e-3 (worst layer 1.09e-2), BF 6.6e-3 (8.9e-3), WR 7.0e-3, BR 6.8e-3,
# WS 6.8e-3 (1.10e-2), BS 6.2e-3, head_w1 5.4e-3, head_b1 5.4e-3, head_w2 4.1e-3, head_b2 3.9e-3.
GRAD_BOUNDS = {"init_w": (1.3e-2, 1.3e-2), "init_b": (1.4e-2, 1.4e-2), "WF": (1.5e-2, 2.2e-2), "BF": (1.4e-2, 1.8e-2),
               "WR": (1.5e-2, 2.2e-2), "BR": (1.4e-2, 1.8e-2), "WS": (1.4e-2, 2.2e-2), "BS": (1.3e-2, 1.3e-2),
               "head_w1": (1.1e-2, 1.1e-2), "head_b1": (1.1e-2, 1.1e-2), "head_w2": (8.5e-3, 8.5e-3),
               "head_b2": (8e-3, 8e-3)}


def test_full_size_bf16_gradients_track_exact_fp32_mode():
    """Config 2 at the BENCHMARK'S OWN geometry -- 8 x 16000: 256 one-segment workgroups, the 17-tile three-tile body of the
    1..16 groups and the halo-free two-tile body of the 32..512 groups, the weight-gradient tiles, the skip weight
    gradients contracted from them, the one-launch head -- every gradient tensor of the bf16 default path against the
    exact-fp32 mode of the same engine (which tests/test_gpu_depth.py ties to the oracle at 1e-3 with every gradient, at
    this depth and these dilations, on a segment cut of T = 4300).  Per section of the flat gradient buffer (all layers of a
    kind together) and, for the per-layer kinds, the worst single layer."""
    import os
    audio, codes = _inputs()
    grads, secs = {}, None
    for dt in (torch.float32, torch.bfloat16):
        eng = _engine(dt, seed=3)
        if dt == torch.bfloat16:
            assert eng.fused_wt and eng.skip_wt and eng.head_chain      # the timed path
            assert eng.nslabs == 256 and sorted(set(eng.wt_seg_rows)) == [500]      # one 500-position segment per CU
        eng.set_inputs(audio, codes)
        eng.forward(); eng.backward()
        torch.cuda.synchronize()
        grads[dt] = eng.grads.double().cpu()
        secs = {n: (s.offset, s.numel, s.shape) for n, s in eng.sections.items()}
        del eng
    ref, got = grads[torch.float32], grads[torch.bfloat16]
    assert bool(torch.isfinite(got).all())
    report = {}
    for n, (off, num, shape) in secs.items():
        r, g = ref[off:off + num], got[off:off + num]
        err = float((g - r).norm() / r.norm())
        worst = err
        if len(shape) >= 2 and shape[0] == len(DIL):      # per-layer kinds: the worst layer on its own
            rl, gl = r.view(shape[0], -1), g.view(shape[0], -1)
            live = rl.norm(dim=1) > 0      # (the top layer's residual 1x1 gets no gradient: its dense output is unused,
            assert bool((gl[~live] == 0).all()), n      # model.py:45-50 -- exactly zero on both sides)
            worst = float(((gl - rl)[live].norm(dim=1) / rl[live].norm(dim=1)).max())
        report[n] = (err, worst)
    if os.environ.get("SRWN_PRINT_ERR"):
        print("MEASURED fullsize bf16-vs-fp32 gradient rel L2 (section, worst layer):",
              {k: ("%.3e" % v[0], "%.3e" % v[1]) for k, v in report.items()})
    for n, (err, worst) in report.items():
        assert err < GRAD_BOUNDS[n][0], (n, err)
        assert worst < GRAD_BOUNDS[n][1], (n, "worst layer", worst)
