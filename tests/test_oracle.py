"""CPU tests: the oracle against the reference's fixed self-check inputs, the
closed-form mu-law values, the committed goldens, and oracle (i) vs oracle (ii)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from oracle import wavenet_torch as OT


def test_reference_selfcheck_vectors(golden_dir):
    """ops.py:243-254 fixed inputs; expected outputs by hand from ops.py:6-10."""
    g = json.load(open(os.path.join(golden_dir, "ops_selfcheck.json")))
    x = np.array(g["x"], dtype=np.float64).reshape(1, -1, 1)
    for c in g["causal"]:
        w = np.array(c["filt"], dtype=np.float64).reshape(c["shape"])
        y = O.dilated_causal_conv1d(x, w, c["d"])
        exp = np.array(c["out"], dtype=np.float64).T[None]
        assert np.array_equal(y, exp), c["ref"]
        yt = OT.causal_conv(torch.tensor(x).transpose(1, 2), torch.tensor(w), c["d"]).transpose(1, 2).numpy()
        assert np.array_equal(yt, exp), c["ref"]
    # ops.py:254: the un-padded VALID conv is the causal one minus its first K-1 steps
    c = g["valid_nopad"]
    w = np.array(c["filt"], dtype=np.float64).reshape(c["shape"])
    y = O.dilated_causal_conv1d(x, w, 1)[:, 1:, :]
    assert np.array_equal(y, np.array(c["out"], dtype=np.float64).T[None])


def test_mu_law_closed_form(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "mu_law.json")))
    codes = O.mu_law_encode(np.array(g["audio"], dtype=np.float32), g["Q"])
    assert codes.dtype == np.int32 and codes.tolist() == g["codes"]
    dec = O.mu_law_decode(np.arange(256), 256)
    assert dec.view(np.uint32).tolist() == g["decode_all_bits"]
    assert O.mu_law_encode(dec, 256).tolist() == g["roundtrip"] == list(range(256))
    assert dec[0] == -1.0 and dec[255] == 1.0 and abs(dec[128]) < 1e-4


def test_mu_law_edges():
    a = np.array([0.0, -0.0, 1e-9, -1e-9, 2.0, -2.0, np.float32(1) - np.float32(2 ** -24)], dtype=np.float32)
    c = O.mu_law_encode(a, 256)
    assert c.tolist() == [128, 128, 128, 127, 255, 0, 255]
    assert c.min() >= 0 and c.max() <= 255
    assert O.mu_law_encode(np.zeros((0,), np.float32), 256).shape == (0,)


def test_nn_upsample_and_shift():
    e = np.arange(2 * 3 * 2, dtype=np.float64).reshape(2, 3, 2)
    up = O.resize_embedding_nearest_neighbor(e, 12)
    assert np.array_equal(up, np.repeat(e, 4, axis=1))
    x = np.arange(8.0).reshape(1, 8, 1)
    assert O.right_shift(x)[0, :, 0].tolist() == [0, 0, 1, 2, 3, 4, 5, 6]


def test_goldens_reproduce(golden_dir):
    g = np.load(os.path.join(golden_dir, "layer_small.npz"))
    for d in (1, 4, 32):
        lp = O.LayerParams(g[f"d{d}_wf"], g[f"d{d}_bf"], None, None, g[f"d{d}_wr"], g[f"d{d}_br"],
                           g[f"d{d}_ws"], g[f"d{d}_bs"])
        dense, skip, _ = O.residual_dilation_layer(g["x"], lp, d)
        assert np.allclose(dense, g[f"d{d}_dense"], rtol=0, atol=1e-14)
        assert np.allclose(skip, g[f"d{d}_skip"], rtol=0, atol=1e-14)


def _mk(seed, dil, R, S, C, cond=0):
    return O.init_stack_params(seed, dil, 2, R, S, C, cond_channels=cond, bias_scale=0.1)


@pytest.mark.parametrize("gate_mode", ["reference", "wavenet"])
@pytest.mark.parametrize("with_cond", [False, True])
def test_oracle_i_vs_ii_forward_backward(gate_mode, with_cond):
    """Two independent restatements (hand backward vs autograd) agree to fp64 round-off."""
    dil = [1, 2, 4, 8, 1, 2]
    B, T, R, S, C, E, pool = 2, 64, 8, 16, 12, 5, 16
    sp = _mk(3, dil, R, S, C, cond=E if with_cond else 0)
    rng = np.random.default_rng(0)
    audio = rng.uniform(-1, 1, (B, T))
    codes = rng.integers(0, C, (B, T))
    cond = rng.standard_normal((B, T // pool, E)) if with_cond else None
    logits, cache = O.stack_forward(sp, audio, shift_input=True, cond=cond, pool_stride=pool, gate_mode=gate_mode)
    loss = O.softmax_ce_per_timestep(logits, codes)
    grads, extra = O.stack_backward(sp, cache, O.dlogits_per_timestep(logits, codes), cond=cond,
                                    pool_stride=pool, gate_mode=gate_mode)

    st = OT.TorchStack(sp)
    tc = None if cond is None else torch.tensor(cond, requires_grad=True)
    lt = st.forward(torch.tensor(audio), shift_input=True, cond=tc, pool_stride=pool, gate_mode=gate_mode)
    assert np.allclose(lt.detach().numpy(), logits, rtol=1e-10, atol=1e-12)
    ls = OT.loss_per_timestep(lt, torch.tensor(codes))
    assert abs(float(ls.detach()) - loss) < 1e-12
    ls.backward()
    gn = dict(O.flatten_named(grads, with_cond))
    for n, t in st.named(with_cond):
        ref = np.zeros_like(gn[n]) if t.grad is None else t.grad.numpy()
        assert np.allclose(gn[n], ref, rtol=1e-9, atol=1e-12), n
    if with_cond:
        assert np.allclose(extra["dcond"], tc.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_oracle_pooled_loss_grads():
    dil = [1, 2, 4]
    B, T, R, S, C = 3, 32, 8, 8, 10
    sp = _mk(9, dil, R, S, C)
    rng = np.random.default_rng(1)
    audio = rng.uniform(-1, 1, (B, T))
    tg = rng.random((B, C)); tg /= tg.sum(-1, keepdims=True)
    logits, cache = O.stack_forward(sp, audio)
    loss = O.wavenet_loss_pooled(logits, tg)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_pooled(logits, tg))
    st = OT.TorchStack(sp)
    lt = st.forward(torch.tensor(audio))
    ls = OT.loss_pooled(lt, torch.tensor(tg))
    assert abs(float(ls) - loss) < 1e-12
    ls.backward()
    gn = dict(O.flatten_named(grads, False))
    for n, t in st.named(False):
        ref = np.zeros_like(gn[n]) if t.grad is None else t.grad.numpy()
        assert np.allclose(gn[n], ref, rtol=1e-9, atol=1e-13), n
    # predict: softmax over pooled logits sums to one, shape [B,1,C]  (model.py:58-60)
    pr = O.wavenet_predict(sp, audio)
    assert pr.shape == (B, 1, C) and np.allclose(pr.sum(-1), 1)


def test_stack_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "stack_small.npz"))
    names = [k[2:] for k in g.files if k.startswith("p.")]
    dil = g["dilations"].tolist()
    L = len(dil)
    layers = [O.LayerParams(g[f"p.l{i}.wf"], g[f"p.l{i}.bf"], None, None, g[f"p.l{i}.wr"], g[f"p.l{i}.br"],
                            g[f"p.l{i}.ws"], g[f"p.l{i}.bs"]) for i in range(L)]
    sp = O.StackParams(g["p.init_w"], g["p.init_b"], layers, g["p.head_w1"], g["p.head_b1"],
                       g["p.head_w2"], g["p.head_b2"], tuple(dil))
    logits, _ = O.stack_forward(sp, g["audio"], shift_input=True)
    assert np.allclose(logits, g["logits"], rtol=0, atol=1e-12)
    assert abs(O.softmax_ce_per_timestep(logits, g["codes"]) - float(g["loss"])) < 1e-12
    assert len(names) == 2 + 6 * L + 4


def test_causality():
    """logits[:, :i+1] depend only on audio[:, :i+1] (pins incremental generation)."""
    sp = _mk(2, [1, 2, 4, 8], 8, 8, 8)
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, (1, 40))
    b = a.copy(); b[:, 25:] = rng.uniform(-1, 1, (1, 15))
    la, _ = O.stack_forward(sp, a); lb, _ = O.stack_forward(sp, b)
    assert np.array_equal(la[:, :25], lb[:, :25]) and not np.allclose(la[:, 25:], lb[:, 25:])
    # with the teacher-forcing shift, step i sees only audio[:i]
    la, _ = O.stack_forward(sp, a, shift_input=True); lb, _ = O.stack_forward(sp, b, shift_input=True)
    assert np.array_equal(la[:, :26], lb[:, :26])


def test_adam_tf_formula():
    rng = np.random.default_rng(0)
    th = rng.standard_normal(50); m = np.zeros(50); v = np.zeros(50)
    p = torch.tensor(th.copy(), requires_grad=True)
    opt = OT.TFAdam([p], lr=1e-2)
    for t in range(1, 6):
        g = rng.standard_normal(50)
        th, m, v = O.adam_step_tf(th, g, m, v, t, lr=1e-2)
        p.grad = torch.tensor(g)
        opt.step()
        assert np.allclose(p.detach().numpy(), th, rtol=1e-12, atol=1e-14)
    # first step moves every coordinate by ~lr*sign(g) (epsilon on the uncorrected sqrt(v))
    th1, _, _ = O.adam_step_tf(np.zeros(3), np.array([1.0, -2.0, 1e-3]), np.zeros(3), np.zeros(3), 1, lr=0.1)
    assert np.allclose(th1, [-0.1, 0.1, -0.1], rtol=1e-3)


def test_tf_variable_names():
    sp = _mk(0, [1, 2], 8, 8, 8)
    names = O.tf_variable_names(sp, "WaveNet")
    for k in ("WaveNet/causal_conv_Kernel", "WaveNet/dilated_conv_1_filter/dilated_conv_1_Kernel",
              "WaveNet/dilated_conv_0_gate/dilated_conv_0_Bias", "WaveNet/conv1d/kernel",
              "WaveNet/conv1d_3/kernel", "WaveNet/conv1d_5/bias"):
        assert k in names, k
    assert names["WaveNet/conv1d/kernel"].shape == (1, 8, 8)
    assert names["WaveNet/causal_conv_Bias"].shape == (1, 1, 8)


def test_mol_loss_np_vs_torch_autograd():
    """Mixture-of-logistics head (ops.py:124-175): hand-derived gradient == autograd, all four tf.where branches."""
    rng = np.random.default_rng(3)
    B, T, M = 3, 200, 5
    x = rng.uniform(-1, 1, (B, T))
    x[0, :5] = [-1.0, -0.9995, 0.9995, 1.0, 0.0]              # edge branches (ops.py:169)
    l = rng.standard_normal((B, T, 4 * M))
    l[..., 2 * M:3 * M] = rng.uniform(-9, 1, (B, T, M))       # log-scales across the -7 clamp
    l[1, :20, M:2 * M] += 30.0                                 # far means: the cdf_delta <= 1e-5 branch
    lp, aux = O.mol_log_probs(x, l)
    assert set(np.unique(aux["case"])) == {0, 1, 2, 3}
    lt = torch.tensor(l, requires_grad=True)
    loss_t = OT.mol_loss(torch.tensor(x), lt)
    assert abs(float(loss_t.detach()) - O.mol_loss(x, l)) < 1e-8 * abs(O.mol_loss(x, l))
    loss_t.backward()
    assert np.allclose(O.mol_dlogits(x, l), lt.grad.numpy(), rtol=1e-8, atol=1e-10)
    assert np.all(O.mol_dlogits(x, l)[..., 3 * M:] == 0)       # coeffs never reach the loss


# --------------------------------------------------------------------------
# Parallel-WaveNet student (model.py:290-537): oracle (i) == oracle (ii), closed forms, finite differences
# --------------------------------------------------------------------------
def _student_case(seed=0, B=2, T=1024, R=8, S=16, E=5, pool=64, F=3, M=4):
    rng = np.random.default_rng(seed)
    dil = [1, 2, 4]
    flows = [O.init_flow_params(10 + i, dil, 2, R, S, E, bias_scale=0.1) for i in range(F)]
    for p in flows:
        p.head_w2 = p.head_w2 * 0.3
    noise = rng.logistic(0, 1, (B, T)) * 0.15
    cond = rng.standard_normal((B, T // pool, E))
    truth = O.synthetic_audio(B, T, seed=seed + 1).astype(np.float64)
    tl = rng.standard_normal((B, T, 4 * M)) * 0.5
    return flows, noise, cond, truth, tl, pool


def test_stft_power_closed_form_and_both_oracles():
    """A pure tone at bin k has its power at bin k (Hann leaks into k-1, k+1 only); np.fft == DFT matrices."""
    T = 2048
    t = np.arange(T)
    x = np.cos(2 * np.pi * 32 * t / 512)[None]
    p = O.stft_power(x)
    assert p.shape == (1, 257)
    assert p[0].argmax() == 32 and p[0, 32] == pytest.approx((512 / 4) ** 2, rel=1e-9)
    assert p[0, 31] == pytest.approx((512 / 8) ** 2, rel=1e-9) and p[0, 34:].max() < 1e-12
    y = O.synthetic_audio(3, 1500, seed=2).astype(np.float64)
    assert np.allclose(O.stft_power(y), OT.stft_power(torch.tensor(y)).numpy(), rtol=1e-9, atol=1e-12)
    assert O.hann_periodic(512)[0] == 0 and O.hann_periodic(512)[256] == 1.0
    with pytest.raises(ValueError):
        O.stft_power(y[:, :500])


def test_student_forward_np_vs_torch_and_chain_identity():
    flows, noise, cond, truth, tl, pool = _student_case()
    fw = O.student_forward(flows, noise, cond, pool)
    # out = clip(z*s_tot + mu_tot) (model.py:535) is the clipped output of the flow chain
    assert np.allclose(fw["out"], np.clip(fw["x_last"], -1, 1), rtol=1e-12, atol=1e-12)
    ln = O.student_loss(fw, tl, truth, 0.7, 1.3, 0.01)
    ts = [OT.TorchStack(p) for p in flows]
    lt = OT.student_loss(ts, torch.tensor(noise), torch.tensor(cond), pool, torch.tensor(tl), torch.tensor(truth),
                         0.7, 1.3, 0.01)
    assert np.allclose(fw["out"], lt["out"].detach().numpy(), rtol=1e-10, atol=1e-12)
    for k in ("loss", "power_loss", "entropy", "cross_entropy"):
        assert float(lt[k].detach()) == pytest.approx(ln[k], rel=1e-7), k
    # entropy = sum over flows of the log-scales + 2 per sample (model.py:356)
    logs = sum(np.log(s).sum() for s in fw["scales"])
    assert ln["entropy"] == pytest.approx(logs + 2.0 * noise.size, rel=1e-10)


def test_student_autograd_vs_finite_differences():
    flows, noise, cond, truth, tl, pool = _student_case(seed=3, T=768, F=2)
    args = (0.9, 1.1, 0.02)

    def loss_of(fl):
        return O.student_loss(O.student_forward(fl, noise, cond, pool), tl, truth, *args)["loss"]

    ts = [OT.TorchStack(p) for p in flows]
    OT.student_loss(ts, torch.tensor(noise), torch.tensor(cond), pool, torch.tensor(tl), torch.tensor(truth),
                    *args)["loss"].backward()
    rng = np.random.default_rng(0)
    for fi, attr, li in ((0, "init_w", None), (0, "wf", 1), (1, "wr", 0), (1, "wc", 2), (0, "head_w2", None),
                         (1, "head_b2", None), (0, "bc", 0)):
        holder = flows[fi] if li is None else flows[fi].layers[li]
        arr = getattr(holder, attr)
        idx = tuple(rng.integers(0, s) for s in arr.shape)
        g = (getattr(ts[fi], attr) if li is None else ts[fi].layers[li][attr]).grad.numpy()[idx]
        old = arr[idx]
        h = 1e-5
        arr[idx] = old + h; lp = loss_of(flows)
        arr[idx] = old - h; lm = loss_of(flows)
        arr[idx] = old
        fd = (lp - lm) / (2 * h)
        assert fd == pytest.approx(g, rel=2e-4, abs=1e-6), (fi, attr, li, fd, g)
    # the skip 1x1s of a flow receive no gradient (model.py:440-449)
    assert all(l["ws"].grad is None for st in ts for l in st.layers)


def test_mol_dx_and_global_norm_clip():
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 100)); l = rng.standard_normal((2, 100, 12))
    xt = torch.tensor(x, requires_grad=True)
    OT.mol_loss(xt, torch.tensor(l)).backward()
    assert np.allclose(O.mol_dx(x, l), xt.grad.numpy(), rtol=1e-9, atol=1e-12)
    gs = [rng.standard_normal((3, 4)), rng.standard_normal(7)]
    c, gn = O.clip_by_global_norm(gs, 1.0)
    assert gn == pytest.approx(math.sqrt(sum((g ** 2).sum() for g in gs)))
    assert math.sqrt(sum((g ** 2).sum() for g in c)) == pytest.approx(1.0)
    small = [g * 1e-3 for g in gs]
    c2, _ = O.clip_by_global_norm(small, 1.0)
    assert all(np.array_equal(a, b) for a, b in zip(c2, small))


# --------------------------------------------------------------------------
# WaveNetAutoEncoder (model.py:75-285): encoder, joint loss, sampler
# --------------------------------------------------------------------------
def test_conv1d_same_padding_and_nc_layer():
    """tf.layers.conv1d SAME with K=2 pads one zero on the RIGHT: y[t] = x[t] w0 + x[t+1] w1 (ops.py:51)."""
    x = np.arange(1, 6, dtype=np.float64).reshape(1, 5, 1)
    w = np.array([10.0, 1.0]).reshape(2, 1, 1)
    assert O.conv1d_same(x, w)[0, :, 0].tolist() == [12.0, 23.0, 34.0, 45.0, 50.0]
    w3 = np.array([100.0, 10.0, 1.0]).reshape(3, 1, 1)          # K=3: one zero each side
    assert O.conv1d_same(x, w3)[0, :, 0].tolist() == [12.0, 123.0, 234.0, 345.0, 450.0]
    rng = np.random.default_rng(0)
    p = O.NCLayerParams(rng.standard_normal((2, 3, 4)), rng.standard_normal(4), rng.standard_normal((4, 4)),
                        rng.standard_normal(4), rng.standard_normal((4, 6)), rng.standard_normal(6))
    xx = rng.standard_normal((2, 7, 3))
    res, skip, a = O.residual_dilation_layer_nc(xx, p)
    assert res.shape == (2, 7, 4) and skip.shape == (2, 7, 6) and (a >= 0).all()
    assert np.allclose(res, a @ p.wr + p.br)                      # no "+ x": the NC layer returns the 1x1 alone


def test_autoencoder_np_vs_torch_and_finite_differences():
    B, T, pool, EC, S, lat, cs = 2, 128, 16, 8, 12, 3, 2
    dil = [1, 2, 4]
    ep = O.init_encoder_params(1, len(dil), 2, EC, S, lat, bias_scale=0.1)
    dp_ = O.init_stack_params(2, dil, 2, 8, S, 12, cond_channels=lat + cs, bias_scale=0.1)
    x = O.synthetic_audio(B, T, seed=1).astype(np.float64)
    c = np.eye(cs)[[0, 1]]
    r = O.autoencoder_forward(ep, dp_, x, pool, c)
    assert r["encoding"].shape == (B, T // pool, lat)
    te, td = OT.TorchEncoder(ep), OT.TorchStack(dp_)
    loss, e, lg = OT.autoencoder_loss(te, td, torch.tensor(x), pool, torch.tensor(c))
    assert np.allclose(r["encoding"], e.detach().numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(r["logits"], lg.detach().numpy(), rtol=1e-9, atol=1e-11)
    assert float(loss.detach()) == pytest.approx(r["loss"], rel=1e-10)
    loss.backward()
    g = dict(te.named())
    # variables outside the graph: the skip 1x1 of 'nc_conv' and the last layer's residual 1x1 (model.py:141-150)
    assert g["nc.ws"].grad is None and g["e%d.wr" % (len(dil) - 1)].grad is None and g["e0.wr"].grad is not None
    rng = np.random.default_rng(0)
    for holder, attr, name in ((ep.nc, "w", "nc.w"), (ep.layers[1], "w", "e1.w"), (ep.layers[0], "wr", "e0.wr"),
                               (ep.layers[2], "ws", "e2.ws"), (ep, "lat_w", "lat_w"), (ep.layers[1], "b", "e1.b")):
        arr = getattr(holder, attr)
        idx = tuple(rng.integers(0, s) for s in arr.shape)
        old = arr[idx]; h = 1e-6
        arr[idx] = old + h; lp = O.autoencoder_forward(ep, dp_, x, pool, c)["loss"]
        arr[idx] = old - h; lm = O.autoencoder_forward(ep, dp_, x, pool, c)["loss"]
        arr[idx] = old
        assert (lp - lm) / (2 * h) == pytest.approx(g[name].grad.numpy()[idx], rel=1e-4, abs=1e-6), name


def test_mol_sample_closed_forms():
    """ops.py:178-201 with given draws: u2 = 0.5 returns the selected mean; a dominant logit is always selected;
    the log-scale floor is -7; the output is clipped to [-1, 1]."""
    M = 3
    l = np.zeros((1, 4, 4 * M))
    l[..., :M] = [0.0, 50.0, 0.0]
    l[..., M:2 * M] = [0.1, -0.3, 0.7]
    l[..., 2 * M:3 * M] = [-1.0, -20.0, -1.0]
    u1 = np.full((1, 4, M), 0.5)
    u2 = np.array([[0.5, 0.9, 1e-5, 1 - 1e-5]])
    x = O.mol_sample(l, u1, u2)
    assert x[0, 0] == pytest.approx(-0.3)
    assert x[0, 1] == pytest.approx(-0.3 + math.exp(-7.0) * math.log(9.0))
    l[..., M:2 * M] = [0.1, 0.9999, 0.7]; l[..., 2 * M:3 * M] = 0.0
    assert O.mol_sample(l, u1, u2)[0].tolist()[2:] == [-1.0, 1.0]
    # Gumbel-max: with equal logits the largest -log(-log(u1)) wins
    l[..., :M] = 0.0
    u1[0, 0] = [0.2, 0.3, 0.9]
    assert O.mol_sample(l, u1, np.full((1, 4), 0.5))[0, 0] == pytest.approx(0.7)


def test_student_and_autoencoder_goldens(golden_dir):
    """The committed student / auto-encoder fixtures (tests/golden/make_golden.py) are reproduced by both oracles."""
    g = np.load(os.path.join(golden_dir, "student_small.npz"))
    dil = g["dilations"].tolist(); pool = int(g["pool"])
    E = g["cond"].shape[-1]
    R, S = (int(v) for v in g["widths"])
    flows = [O.init_flow_params(int(s), dil, 2, R, S, E, bias_scale=0.1) for s in g["seeds"]]
    for p in flows:
        p.head_w2 = p.head_w2 * 0.3
    fw = O.student_forward(flows, g["noise"], g["cond"], pool)
    assert np.allclose(fw["out"], g["out"], rtol=1e-12, atol=1e-12) and np.allclose(fw["s_tot"], g["s_tot"], rtol=1e-12)
    a, b, c = g["abg"]
    ls = O.student_loss(fw, g["teacher_logits"], g["truth"], a, b, c)
    for k in ("loss", "power_loss", "entropy", "cross_entropy"):
        assert ls[k] == pytest.approx(float(g[k]), rel=1e-12), k
    ts = [OT.TorchStack(p) for p in flows]
    lt = OT.student_loss(ts, torch.tensor(g["noise"]), torch.tensor(g["cond"]), pool, torch.tensor(g["teacher_logits"]),
                         torch.tensor(g["truth"]), a, b, c)
    assert float(lt["loss"].detach()) == pytest.approx(float(g["loss"]), rel=1e-7)
    assert np.allclose(O.stft_power(g["truth"]), g["stft_power_truth"], rtol=1e-12)
    assert np.allclose(O.mol_dx(g["out"], g["teacher_logits"]), g["mol_dx"], rtol=1e-10, atol=1e-12)

    g = np.load(os.path.join(golden_dir, "autoencoder_small.npz"))
    dil = g["dilations"].tolist(); pool = int(g["pool"])
    lat = g["encoding"].shape[-1]; cs = g["conditions"].shape[-1]; M = g["logits"].shape[-1] // 4
    EC, S, R = (int(v) for v in g["widths"])
    ep = O.init_encoder_params(int(g["seeds"][0]), len(dil), 2, EC, S, lat, bias_scale=0.1)
    dp_ = O.init_stack_params(int(g["seeds"][1]), dil, 2, R, S, 4 * M, cond_channels=lat + cs, bias_scale=0.1)
    r = O.autoencoder_forward(ep, dp_, g["x"], pool, g["conditions"])
    assert np.allclose(r["encoding"], g["encoding"], rtol=1e-12, atol=1e-14)
    assert np.allclose(r["logits"], g["logits"], rtol=1e-11, atol=1e-13) and r["loss"] == pytest.approx(float(g["loss"]), rel=1e-12)
    te, td = OT.TorchEncoder(ep), OT.TorchStack(dp_)
    lo, _, lg = OT.autoencoder_loss(te, td, torch.tensor(g["x"]), pool, torch.tensor(g["conditions"]))
    assert float(lo.detach()) == pytest.approx(float(g["loss"]), rel=1e-9)
    assert np.array_equal(O.mol_sample(g["logits"], g["u1"], g["u2"]), g["sample"])
