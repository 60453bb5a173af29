"""Generates the committed golden fixtures under tests/golden/.

Run from the repo root: ``python tests/golden/make_golden.py``.

* ops_selfcheck.json -- the fixed inputs of the reference's only self-check
  (``ops.py:224-229,243-254``: x=[1..8], filters f1,f2,f3,f4, dilations 1,2,3,4,6)
  with the outputs that follow by hand from ``ops.py:6-10``
  (out[t] = sum_k x[t-(K-1-k)d] w[k]); the reference only prints them.
* mu_law.json -- closed-form mu-law values of ``ops.py:82-104`` (Q=256).
* layer_small.npz / stack_small.npz -- seeded outputs of oracle (i) in float64
  (``oracle/wavenet_np.py``); "parity unpinned" by the reference, see that header.

* student_small.npz / autoencoder_small.npz -- seeded outputs of oracle (i) for the Parallel-WaveNet student
  (model.py:290-535: flows, clipped output, entropy / power / cross-entropy / loss) and for the auto-encoder
  (model.py:136-216: encoding, logits, loss, the sampler ops.py:178-201 on fixed draws); "parity unpinned".

Nothing here reads /root/reference; fixtures are data only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import wavenet_np as O  # noqa: E402


def selfcheck():
    x = [1, 2, 3, 4, 5, 6, 7, 8]
    cases = [
        dict(ref="ops.py:243", filt=[1, 1], shape=[2, 1, 1], d=1, out=[[1, 3, 5, 7, 9, 11, 13, 15]]),
        dict(ref="ops.py:244", filt=[1, 0, 1], shape=[3, 1, 1], d=1, out=[[1, 2, 4, 6, 8, 10, 12, 14]]),
        dict(ref="ops.py:245", filt=[1, 0, 0, 0, 1], shape=[5, 1, 1], d=1, out=[[1, 2, 3, 4, 6, 8, 10, 12]]),
        dict(ref="ops.py:246", filt=[1, 1], shape=[2, 1, 1], d=2, out=[[1, 2, 4, 6, 8, 10, 12, 14]]),
        dict(ref="ops.py:247", filt=[1, 1], shape=[2, 1, 1], d=3, out=[[1, 2, 3, 5, 7, 9, 11, 13]]),
        dict(ref="ops.py:248", filt=[1, 1], shape=[2, 1, 1], d=4, out=[[1, 2, 3, 4, 6, 8, 10, 12]]),
        dict(ref="ops.py:249", filt=[1, 1], shape=[2, 1, 1], d=6, out=[[1, 2, 3, 4, 5, 6, 8, 10]]),
        dict(ref="ops.py:252", filt=[1, 2, 1, 2], shape=[2, 1, 2], d=1,
             out=[[1, 3, 5, 7, 9, 11, 13, 15], [2, 6, 10, 14, 18, 22, 26, 30]]),
    ]
    valid = dict(ref="ops.py:254", filt=[1, 2, 1, 2], shape=[2, 1, 2], d=1,
                 out=[[3, 5, 7, 9, 11, 13, 15], [6, 10, 14, 18, 22, 26, 30]])
    with open(os.path.join(HERE, "ops_selfcheck.json"), "w") as f:
        json.dump(dict(x=x, causal=cases, valid_nopad=valid), f, indent=1)


def mulaw():
    a = [-1.5, -1.0, -0.5, -0.01, 0.0, 0.01, 0.5, 1.0, 1.5]
    codes = [0, 0, 16, 98, 128, 157, 239, 255, 255]
    # closed form: code = trunc((sign(a)*ln(1+255|a|)/ln(256) + 1)/2*255 + 0.5)
    chk = []
    for v in a:
        s = (v > 0) - (v < 0)
        m = np.log1p(255 * min(abs(v), 1.0)) / np.log(256.0)
        chk.append(int((s * m + 1) / 2 * 255 + 0.5))
    assert chk == codes, chk
    allc = list(range(256))
    dec = O.mu_law_decode(np.array(allc), 256)
    rt = O.mu_law_encode(dec, 256).tolist()
    with open(os.path.join(HERE, "mu_law.json"), "w") as f:
        json.dump(dict(Q=256, audio=a, codes=codes, decode_all=[float(np.float32(v)) for v in dec],
                       decode_all_bits=[int(np.float32(v).view(np.uint32)) for v in dec],
                       roundtrip=rt), f, indent=1)


def layer_small():
    out = {}
    B, T, R, S = 2, 64, 8, 16
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, T, R))
    for d in (1, 4, 32):
        sp = O.init_stack_params(100 + d, [d], 2, R, S, 4, bias_scale=0.1)
        dense, skip, _ = O.residual_dilation_layer(x, sp.layers[0], d)
        l = sp.layers[0]
        for k in ("wf", "bf", "wr", "br", "ws", "bs"):
            out[f"d{d}_{k}"] = getattr(l, k)
        out[f"d{d}_dense"] = dense
        out[f"d{d}_skip"] = skip
    out["x"] = x
    np.savez_compressed(os.path.join(HERE, "layer_small.npz"), **out)


def stack_small():
    # config-1 shape: dilations [1,2,4,8,16]x2, R=32, T=256
    dil = [1, 2, 4, 8, 16] * 2
    B, T, R, S, C = 2, 256, 32, 32, 32
    sp = O.init_stack_params(11, dil, 2, R, S, C, bias_scale=0.05)
    audio = O.synthetic_audio(B, T, seed=3).astype(np.float64)
    codes = O.mu_law_encode(audio.astype(np.float32), C)
    logits, cache = O.stack_forward(sp, audio, shift_input=True)
    loss = O.softmax_ce_per_timestep(logits, codes)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(logits, codes))
    rng = np.random.default_rng(5)
    tg = rng.random((B, C)); tg /= tg.sum(-1, keepdims=True)
    logits_ns, cache_ns = O.stack_forward(sp, audio, shift_input=False)
    loss_p = O.wavenet_loss_pooled(logits_ns, tg)
    grads_p, _ = O.stack_backward(sp, cache_ns, O.dlogits_pooled(logits_ns, tg))
    out = dict(audio=audio, codes=codes, logits=logits, loss=np.float64(loss), targets=tg,
               logits_noshift=logits_ns, loss_pooled=np.float64(loss_p), dilations=np.array(dil))
    for n, a in O.flatten_named(sp, False):
        out["p." + n] = a
    for n, a in O.flatten_named(grads, False):
        out["g." + n] = a
    for n, a in O.flatten_named(grads_p, False):
        out["gp." + n] = a
    np.savez_compressed(os.path.join(HERE, "stack_small.npz"), **out)


def student_small():
    B, T, R, S, E, pool, F, M = 2, 768, 32, 128, 5, 64, 3, 5      # widths the GPU engine is built for
    dil = [1, 2, 4]
    rng = np.random.default_rng(21)
    noise = rng.logistic(0, 1, (B, T)) * 0.15
    cond = rng.standard_normal((B, T // pool, E))
    truth = O.synthetic_audio(B, T, seed=22).astype(np.float64)
    tl = rng.standard_normal((B, T, 4 * M)) * 0.5
    flows = [O.init_flow_params(30 + i, dil, 2, R, S, E, bias_scale=0.1) for i in range(F)]
    for p in flows:
        p.head_w2 = p.head_w2 * 0.3
    fw = O.student_forward(flows, noise, cond, pool)
    ls = O.student_loss(fw, tl, truth, 0.8, 1.2, 0.05)
    out = dict(noise=noise, cond=cond, truth=truth, teacher_logits=tl, dilations=np.array(dil), pool=np.int64(pool),
               seeds=np.array([30 + i for i in range(F)]), widths=np.array([R, S]), abg=np.array([0.8, 1.2, 0.05]), out=fw["out"],
               s_tot=fw["s_tot"], mu_tot=fw["mu_tot"], stft_power_truth=O.stft_power(truth),
               mol_dx=O.mol_dx(fw["out"], tl), **{k: np.float64(v) for k, v in ls.items()})
    np.savez_compressed(os.path.join(HERE, "student_small.npz"), **out)


def autoencoder_small():
    B, T, pool, EC, S, lat, cs, M, R = 2, 256, 32, 128, 128, 3, 2, 5, 32
    dil = [1, 2, 4]
    ep = O.init_encoder_params(41, len(dil), 2, EC, S, lat, bias_scale=0.1)
    dp_ = O.init_stack_params(42, dil, 2, R, S, 4 * M, cond_channels=lat + cs, bias_scale=0.1)
    x = O.synthetic_audio(B, T, seed=43).astype(np.float64)
    c = np.eye(cs)[[0, 1]]
    r = O.autoencoder_forward(ep, dp_, x, pool, c)
    rng = np.random.default_rng(44)
    u1 = rng.uniform(1e-5, 1 - 1e-5, (B, T, M)); u2 = rng.uniform(1e-5, 1 - 1e-5, (B, T))
    out = dict(x=x, conditions=c, dilations=np.array(dil), pool=np.int64(pool), seeds=np.array([41, 42]),
               widths=np.array([EC, S, R]),
               encoding=r["encoding"], logits=r["logits"], loss=np.float64(r["loss"]), u1=u1, u2=u2,
               sample=O.mol_sample(r["logits"], u1, u2))
    np.savez_compressed(os.path.join(HERE, "autoencoder_small.npz"), **out)


if __name__ == "__main__":
    selfcheck(); mulaw(); layer_small(); stack_small(); student_small(); autoencoder_small()
    print("golden fixtures written to", HERE)
