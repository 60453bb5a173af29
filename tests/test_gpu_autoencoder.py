"""GPU parity of the WaveNetAutoEncoder path (model.py:75-285; ops.py:48-58, 178-201) vs the CPU oracles.

Unpinned by the reference; anchored on oracle (i) == oracle (ii) (tests/test_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from oracle import wavenet_torch as OT
from tests._pkg import sub
from tests.test_gpu_kernels import DEV, dev, rel_err

pytestmark = pytest.mark.gpu


def _tap(x, ntaps, step, T, Cin, wp, bias, y, cout, aux, fadd, frames, pool, fscale, epi, dt):
    L = sub("_lib"); K = sub("kernels")
    rows = x.shape[0]
    L.call("srwn_tap_linear", x.data_ptr(), Cin, ntaps, step, T, Cin, wp, None if bias is None else bias.data_ptr(),
           y.data_ptr(), cout, cout, rows, None if aux is None else aux.data_ptr(), cout,
           None if fadd is None else fadd.data_ptr(), 0 if fadd is None else fadd.shape[-1], frames, pool, fscale, epi,
           K.abi_dtype(dt), torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("cout", [128, 256])
def test_tap_linear(dt, tol, cout):
    """Time-tap GEMM: the SAME-padded K=2 conv (ops.py:51), its data gradient, a 1x1 with the per-frame broadcast."""
    K = sub("kernels"); P = sub("packing")
    rng = np.random.default_rng(cout)
    B, T, Cin, pool = 3, 100, 128, 25
    frames = T // pool
    x = torch.tensor(rng.standard_normal((B * T, Cin)), dtype=dt, device=DEV)
    xq = x.double().cpu().numpy().reshape(B, T, Cin)
    w = rng.standard_normal((2, Cin, cout)) * 0.1
    b = rng.standard_normal(cout) * 0.1
    wq = torch.tensor(w, dtype=dt).double().numpy()
    flat = dev(w.reshape(-1))
    pk = K.Packer(DEV)
    off = pk.reserve(cout // 32, 2 * Cin // 16)
    P.fill_linear(pk, off, 0, 2 * Cin, cout, cout // 32, 2 * Cin // 16)      # [K*Cin, cout] natural: k = tap*Cin + i
    offT = pk.reserve(cout // 32, Cin // 16)                                 # one tap as a 1x1 (first Cin rows)
    P.fill_linear(pk, offT, 0, Cin, cout, cout // 32, Cin // 16)
    pk.finalize()
    packed = torch.zeros(pk.total, dtype=dt, device=DEV); pk.gather(flat, packed)
    wp = packed.data_ptr(); es = packed.element_size()
    # forward conv, relu
    y = torch.zeros((B * T, cout), dtype=dt, device=DEV)
    _tap(x, 2, 1, T, Cin, wp + off * es, dev(b), y, cout, None, None, 1, 1, 0.0, K.EPI_RELU, dt)
    ref = np.maximum(O.conv1d_same(xq, wq) + b, 0).reshape(B * T, cout)
    assert rel_err(y.double().cpu().numpy(), ref) < tol
    # taps going backwards in time (the data-gradient direction), masked by aux
    aux = torch.tensor(rng.standard_normal((B * T, cout)), dtype=dt, device=DEV)
    _tap(x, 2, -1, T, Cin, wp + off * es, None, y, cout, aux, None, 1, 1, 0.0, K.EPI_MASK, dt)
    back = np.zeros((B, T, cout))
    back += xq @ wq[0]
    back[:, 1:] += xq[:, :-1] @ wq[1]
    ref = (back.reshape(B * T, cout)) * (aux.double().cpu().numpy() > 0)
    assert rel_err(y.double().cpu().numpy(), ref) < tol
    # 1x1 with the per-frame broadcast term added before the mask
    fadd = dev(rng.standard_normal((B * frames, cout + 64)))
    _tap(x, 1, 0, T, Cin, wp + offT * es, dev(b), y, cout, aux, fadd, frames, pool, 0.5, K.EPI_MASK, dt)
    fa = np.repeat(fadd.cpu().numpy().reshape(B, frames, -1)[:, :, :cout], pool, axis=1).reshape(B * T, cout)
    ref = (xq.reshape(B * T, Cin) @ wq[0] + b + 0.5 * fa) * (aux.double().cpu().numpy() > 0)
    assert rel_err(y.double().cpu().numpy(), ref) < tol


def test_encoder_small_kernels():
    L = sub("_lib"); K = sub("kernels")
    rng = np.random.default_rng(2)
    st = torch.cuda.current_stream().cuda_stream
    # first encoder layer on the raw clip
    B, T, C = 2, 77, 128
    x = rng.standard_normal((B, T)); w = rng.standard_normal((2, 1, C)); b = rng.standard_normal(C)
    a = torch.zeros((B, T, C), device=DEV)
    xd, wd, bd = dev(x), dev(w), dev(b)     # (keep the device tensors alive across the raw-pointer call)
    L.call("srwn_nc_input_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), a.data_ptr(), B, T, C, 2, K.F32, st)
    ref = np.maximum(O.conv1d_same(np.maximum(x, 0)[:, :, None], w) + b, 0)
    assert rel_err(a.cpu().numpy(), ref) < 1e-5
    # its weight gradient: taps relu(x)[t+k]  (srwn_init_conv_wgrad with a negative shift)
    g = rng.standard_normal((B, T, C))
    gw = torch.zeros(2 * C, device=DEV); gb = torch.zeros(C, device=DEV)
    ws = torch.zeros(int(L.load().srwn_init_conv_wgrad_partials(B, T, C, 2)), device=DEV)
    K.init_conv_wgrad(dev(np.maximum(x, 0)), dev(g), gw, gb, 2, -1, ws)
    xr = np.maximum(x, 0)
    ref_w = np.stack([np.einsum("bt,btc->c", xr, g), np.einsum("bt,btc->c", xr[:, 1:], g[:, :-1])])
    assert rel_err(gw.cpu().numpy().reshape(2, C), ref_w) < 1e-4 and rel_err(gb.cpu().numpy(), g.sum((0, 1))) < 1e-4
    # small products with chunked operands
    M, N, R, Lc, E = 50, 6, 8, 3, 10
    A = rng.standard_normal((Lc, M, R)); W = rng.standard_normal((Lc, E, R)); bias = rng.standard_normal(N)
    Cq = torch.full((M, N), 1.0, device=DEV)
    Ad, Wd, biasd = dev(A), dev(W), dev(bias)
    L.call("srwn_small_gemm", Ad.data_ptr(), R, R, M * R, K.F32, Wd.data_ptr(), 1, R, R, E * R,
           biasd.data_ptr(), Cq.data_ptr(), N, K.F32, M, N, Lc * R, 1, st)
    ref = 1.0 + bias + np.einsum("lmr,lnr->mn", A, W[:, :N])
    assert rel_err(Cq.cpu().numpy(), ref) < 1e-5
    Ab = torch.tensor(A, dtype=torch.bfloat16, device=DEV)
    Cb = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    L.call("srwn_small_gemm", Ab.data_ptr(), R, R, M * R, K.BF16, Wd.data_ptr(), 1, R, R, E * R, None,
           Cb.data_ptr(), N, K.BF16, M, N, Lc * R, 0, st)
    assert rel_err(Cb.double().cpu().numpy(), np.einsum("lmr,lnr->mn", Ab.double().cpu().numpy(), W[:, :N])) < 1e-2
    Am = rng.standard_normal((M, 12)); D = rng.standard_normal((M, N))
    cw = torch.zeros((12, N), device=DEV); cb = torch.zeros(N, device=DEV)
    Amd, Dd = dev(Am), dev(D)
    L.call("srwn_small_wgrad", Amd.data_ptr(), 12, Dd.data_ptr(), N, cw.data_ptr(), cb.data_ptr(), M, 12, N, 0.5, st)
    assert rel_err(cw.cpu().numpy(), 0.5 * Am.T @ D) < 1e-5 and rel_err(cb.cpu().numpy(), 0.5 * D.sum(0)) < 1e-5
    # mixture sampler with given uniforms (ops.py:178-201)
    rows, Mx = 3000, 5
    lg = rng.standard_normal((rows, 4 * Mx)); lg[:, 2 * Mx:3 * Mx] = rng.uniform(-9, -1, (rows, Mx))
    u1 = rng.uniform(1e-5, 1 - 1e-5, (rows, Mx)).astype(np.float32); u2 = rng.uniform(1e-5, 1 - 1e-5, rows).astype(np.float32)
    out = torch.zeros(rows, device=DEV)
    lgd, u1d, u2d = dev(lg), dev(u1), dev(u2)
    L.call("srwn_mol_sample", lgd.data_ptr(), 4 * Mx, Mx, u1d.data_ptr(), u2d.data_ptr(), out.data_ptr(), rows, st)
    ref = O.mol_sample(lg.astype(np.float32).astype(np.float64)[None], u1.astype(np.float64)[None], u2.astype(np.float64)[None])[0]
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4


def _ae(dt, R, S, cs, B=2, T=256, pool=32, lat=8, M=5, dil=(1, 2, 4), lr=1e-3):
    EG = sub("engine"); EN = sub("encoder")
    dil = list(dil)
    ep = O.init_encoder_params(60, len(dil), 2, 128, S, lat, bias_scale=0.05)
    dp_ = O.init_stack_params(61, dil, 2, R, S, 4 * M, cond_channels=lat + cs, bias_scale=0.05)
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=4 * M,
                         cond_channels=lat + cs, pool_stride=pool, shift_input=True, head_mode="mol", dtype=dt,
                         learning_rate=lr)
    ae = EN.AutoEncoderEngine(cfg, B, T, 128, lat, cs, DEV)
    ae.enc.load_oracle_params(ep); ae.dec.load_oracle_params(dp_)
    x = O.synthetic_audio(B, T, seed=21).astype(np.float64)
    c = np.eye(max(cs, 1))[[(i * 7) % cs for i in range(B)]][:, :cs] if cs else None
    ae.set_inputs(dev(x), None if c is None else dev(c))
    return ae, ep, dp_, x, c, pool


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("R,S,cs,B,T,pool", [(64, 256, 3, 2, 256, 32), (32, 128, 0, 2, 256, 32), (64, 256, 0, 1, 32, 32),
                                             (32, 128, 2, 3, 96, 32), (64, 128, 0, 2, 50, 25)])
def test_autoencoder_forward_backward(dt, tol, R, S, cs, B, T, pool):
    ae, ep, dp_, x, c, pool = _ae(dt, R, S, cs, B=B, T=T, pool=pool)
    B, T = x.shape
    ref = O.autoencoder_forward(ep, dp_, x, pool, c)
    lg = ae.forward(want_logits=True)
    assert rel_err(ae.enc.enc.cpu().numpy().reshape(ref["encoding"].shape), ref["encoding"]) < tol
    assert rel_err(lg.cpu().numpy(), ref["logits"]) < tol
    assert abs(float(ae.loss.item()) - ref["loss"]) < tol * abs(ref["loss"])
    te, td = OT.TorchEncoder(ep), OT.TorchStack(dp_)
    loss, _, _ = OT.autoencoder_loss(te, td, torch.tensor(x), pool, None if c is None else torch.tensor(c))
    loss.backward()
    ae.backward()
    fp32 = dt == torch.float32
    def cmp(mine, want, name):
        a = mine.float().cpu().numpy()
        if want is None:                       # variables outside the graph get no gradient
            assert np.abs(a).max() == 0, name
            return
        r = want.numpy()
        e = (np.abs(a - r).max() / (np.abs(r).max() + 1e-30)) if fp32 else (np.linalg.norm(a - r) / (np.linalg.norm(r) + 1e-30))
        assert e < (tol if fp32 else 0.15), (name, e)
    mine = ae.enc.named_tensors(ae.enc.grads)
    for n, t in te.named():
        if n.startswith("nc.ws") or n.startswith("nc.bs"):
            continue
        cmp(mine[n], t.grad, "enc." + n)
    dm = ae.dec.named_tensors(ae.dec.grads)
    for n, t in td.named(include_cond=True):
        cmp(dm[n], t.grad, "dec." + n)


def test_autoencoder_trains():
    ae, *_ = _ae(torch.float32, 64, 256, 0, lr=1e-3)
    ae.train_step()
    l0 = float(ae.loss.item())
    for _ in range(10):
        ae.train_step()
    assert float(ae.loss.item()) < l0


def test_autoencoder_engine_vs_committed_golden(golden_dir):
    """The auto-encoder engine (fp32) and the sampler kernel against tests/golden/autoencoder_small.npz."""
    import os
    EG = sub("engine"); EN = sub("encoder"); L = sub("_lib")
    g = np.load(os.path.join(golden_dir, "autoencoder_small.npz"))
    dil = g["dilations"].tolist(); pool = int(g["pool"]); EC, S, R = (int(v) for v in g["widths"])
    B, T = g["x"].shape; lat = g["encoding"].shape[-1]; cs = g["conditions"].shape[-1]; M = g["logits"].shape[-1] // 4
    ep = O.init_encoder_params(int(g["seeds"][0]), len(dil), 2, EC, S, lat, bias_scale=0.1)
    dp_ = O.init_stack_params(int(g["seeds"][1]), dil, 2, R, S, 4 * M, cond_channels=lat + cs, bias_scale=0.1)
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=4 * M, cond_channels=lat + cs,
                         pool_stride=pool, shift_input=True, head_mode="mol", dtype=torch.float32)
    ae = EN.AutoEncoderEngine(cfg, B, T, EC, lat, cs, DEV)
    ae.enc.load_oracle_params(ep); ae.dec.load_oracle_params(dp_)
    ae.set_inputs(dev(g["x"]), dev(g["conditions"]))
    lg = ae.forward(want_logits=True)
    assert rel_err(ae.enc.enc.cpu().numpy().reshape(g["encoding"].shape), g["encoding"]) < 1e-3
    assert rel_err(lg.cpu().numpy(), g["logits"]) < 1e-3
    assert abs(float(ae.loss.item()) - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    # sampler on the fixture's logits and draws (ops.py:178-201)
    lgd = dev(g["logits"].reshape(B * T, 4 * M)); u1 = dev(g["u1"].reshape(B * T, M)); u2 = dev(g["u2"].reshape(B * T))
    out = torch.zeros(B * T, device=DEV)
    L.call("srwn_mol_sample", lgd.data_ptr(), 4 * M, M, u1.data_ptr(), u2.data_ptr(), out.data_ptr(), B * T,
           torch.cuda.current_stream().cuda_stream)
    assert (np.abs(out.cpu().numpy().reshape(B, T) - g["sample"]) < 1e-3).mean() > 0.995


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_wgrad_nc_layers_kernel(dt, tol):
    """srwn_wgrad_nc_layers: both conv taps (t, t+1 inside the clip), the 1x1 and the two bias gradients of every
    encoder layer in one pass."""
    L_, K = sub("_lib"), sub("kernels")
    rng = np.random.default_rng(12)
    L, B, T, C, ns = 3, 2, 173, 128, 5
    rows = B * T
    mk = lambda: torch.tensor(rng.standard_normal((L, rows, C)), dtype=dt, device=DEV)
    r, a, dp, dh = mk(), mk(), mk(), mk()
    f = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device=DEV)
    pw, pr, pb, pbr = f(L * ns * 2 * C * C), f(L * ns * C * C), f(L * ns * C), f(L * ns * C)
    L_.call("srwn_wgrad_nc_layers", r.data_ptr(), a.data_ptr(), dp.data_ptr(), dh.data_ptr(), rows * C, L, pw.data_ptr(),
            pr.data_ptr(), pb.data_ptr(), pbr.data_ptr(), rows, T, ns, C, 2, K.abi_dtype(dt),
            torch.cuda.current_stream().cuda_stream)
    ow, orr = torch.empty((L, 2, C, C), device=DEV), torch.empty((L, C, C), device=DEV)
    ob, obr = torch.empty((L, C), device=DEV), torch.empty((L, C), device=DEV)
    K.reduce_partials(pw, ns, 2 * C * C, L, True, 1.0, ow.data_ptr(), 2 * C * C)
    K.reduce_partials(pr, ns, C * C, L, True, 1.0, orr.data_ptr(), C * C)
    K.reduce_partials(pb, ns, C, L, True, 1.0, ob.data_ptr(), C); K.reduce_partials(pbr, ns, C, L, True, 1.0, obr.data_ptr(), C)
    q = lambda t: t.double().cpu().numpy().reshape(L, B, T, C)
    rq, aq, pq, hq = q(r), q(a), q(dp), q(dh)
    for l in range(L):
        w0 = np.einsum("bti,bto->io", rq[l], pq[l])
        w1 = np.einsum("bti,bto->io", rq[l][:, 1:], pq[l][:, :-1])          # r[t+1]^T dpre[t], nothing across clips
        assert rel_err(ow[l, 0].cpu().numpy(), w0) < tol and rel_err(ow[l, 1].cpu().numpy(), w1) < tol
        assert rel_err(orr[l].cpu().numpy(), np.einsum("btn,btm->nm", aq[l], hq[l])) < tol
        assert rel_err(ob[l].cpu().numpy(), pq[l].sum((0, 1))) < tol and rel_err(obr[l].cpu().numpy(), hq[l].sum((0, 1))) < tol


@pytest.mark.parametrize("B,T,pool", [(2, 256, 32), (3, 96, 32), (1, 33, 33), (2, 50, 25), (1, 1, 1)])
def test_fused_nc_layers_match_the_two_launch_path(B, T, pool, monkeypatch):
    """srwn_nc_layer_fwd/_bwd (one launch per encoder layer, accumulator-chained) against srwn_tap_linear x 2 in bf16:
    same products on the same bf16-rounded operands, so activations agree to an ulp of bf16 and gradients closely;
    canary rows beyond each saved tensor stay untouched."""
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SRWN_NC_FUSED", fused)
        ae, ep, dp_, x, c, _ = _ae(torch.bfloat16, 32, 128, 0, B=B, T=T, pool=pool)
        assert ae.enc.fused == (fused == "1")
        ae.forward(); ae.backward(); torch.cuda.synchronize()
        e = ae.enc
        out[fused] = dict(a=e.a.float().cpu().numpy(), r=e.r.float().cpu().numpy(), dh=e.dh.float().cpu().numpy(),
                          dpre=e.dpre.float().cpu().numpy(), enc=e.enc.cpu().numpy(), g=e.grads.cpu().numpy(),
                          loss=float(ae.loss.item()))
    f, u = out["1"], out["0"]
    L = f["a"].shape[0] - 1
    assert rel_err(f["a"], u["a"]) < 1e-2 and rel_err(f["r"][:L], u["r"][:L]) < 1e-2
    assert rel_err(f["enc"], u["enc"]) < 1e-2 and abs(f["loss"] - u["loss"]) < 1e-2 * abs(u["loss"])
    assert rel_err(f["dpre"], u["dpre"]) < 3e-2 and rel_err(f["dh"][:L], u["dh"][:L]) < 3e-2
    assert np.linalg.norm(f["g"] - u["g"]) < 3e-2 * np.linalg.norm(u["g"])


def test_nc_layer_kernels_reject_other_shapes():
    L_ = sub("_lib")
    z = torch.zeros(64 * 128, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(128, device=DEV)
    zp, bp = z.data_ptr(), b.data_ptr()
    with pytest.raises(RuntimeError, match="128 channels"):
        L_.call("srwn_nc_layer_fwd", zp, zp, zp, bp, bp, zp, zp, None, None, 1, 64, 64, 2, 1, None)     # 64 channels
    with pytest.raises(RuntimeError, match="bf16"):
        L_.call("srwn_nc_layer_fwd", zp, zp, zp, bp, bp, zp, zp, None, None, 1, 64, 128, 2, 0, None)    # fp32
    with pytest.raises(RuntimeError, match="wresT"):
        L_.call("srwn_nc_layer_bwd", zp, zp, zp, zp, None, None, 0, 0, 1, 1.0, None, zp, 1, 64, 128, 2, 1, None)
    assert L_.call("srwn_nc_layer_fwd", zp, zp, zp, bp, bp, zp, zp, None, None, 0, 64, 128, 2, 1, None) == 0   # empty batch
    assert int(L_.load().srwn_nc_mask_words(3, 33)) == 3 * 2 * 64


def test_nc_mask_bits_match_the_forward_kernels_words():
    """The relu-mask words the fused forward writes equal srwn_nc_mask_bits of the tensors it stored."""
    L_ = sub("_lib")
    ae, *_ = _ae(torch.bfloat16, 32, 128, 0, B=3, T=96, pool=32)
    ae.forward(); torch.cuda.synchronize()
    e = ae.enc
    assert e.fused
    for l in range(1, e.L + 1):
        for t_, bits in ((e.a[l], e.abits[l]),) + (((e.r[l], e.rbits[l]),) if l < e.L else ()):
            want = torch.zeros_like(bits)
            L_.call("srwn_nc_mask_bits", t_.data_ptr(), want.data_ptr(), e.B, e.T, 128, 1, None)
            assert torch.equal(bits, want)
            assert int((bits != 0).sum()) > 0
