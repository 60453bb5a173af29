"""GPU parity of the Parallel-WaveNet student path (model.py:290-537) vs the CPU oracles.

Nothing in the reference pins these results: parity is "unpinned by the reference" and anchored on
oracle (i) (NumPy fp64, np.fft STFT) == oracle (ii) (torch autograd, DFT-matrix STFT) -- tests/test_oracle.py.
fp32 mode meets 1e-3 relative; bf16 rounds every stored activation and is judged loosely."""
import math

import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from oracle import wavenet_torch as OT
from tests._pkg import sub
from tests.test_gpu_kernels import DEV, dev, rel_err

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("R", [32, 64])
@pytest.mark.parametrize("rows", [1, 255, 1000])
def test_flow_affine_fwd_bwd(dt, tol, R, rows):
    K = sub("kernels")
    rng = np.random.default_rng(rows + R)
    h = torch.tensor(rng.standard_normal((rows, R)), dtype=dt, device=DEV)
    hq = h.double().cpu().numpy()
    w2 = rng.standard_normal((R, 2)) * 0.2; b2 = rng.standard_normal(2) * 0.1
    x = rng.standard_normal(rows); dxo = rng.standard_normal(rows)
    nb = K.flow_partials(rows)
    prm = torch.zeros((rows, 2), device=DEV); xo = torch.zeros(rows, device=DEV); ent = torch.zeros(nb, device=DEV)
    K.flow_affine_fwd(h, dev(w2), dev(b2), dev(x), prm, xo, ent)
    a = np.maximum(hq, 0)
    p_ref = a @ w2 + b2
    assert rel_err(prm.cpu().numpy(), p_ref) < 1e-5
    assert rel_err(xo.cpu().numpy(), x * np.exp(p_ref[:, 0]) + p_ref[:, 1]) < 1e-5
    assert abs(float(ent.sum().item()) - p_ref[:, 0].sum()) < 1e-4 * max(1.0, abs(p_ref[:, 0]).sum())
    g = torch.zeros((rows, R), dtype=dt, device=DEV); dxi = torch.zeros(rows, device=DEV)
    parts = torch.zeros((nb, 2 * R + 2), device=DEV)
    eg = -0.37
    K.flow_affine_bwd(h, dev(w2), prm, dev(x), dev(dxo), eg, g, dxi, parts)
    pq = prm.double().cpu().numpy()
    sc = np.exp(pq[:, 0])
    d = np.stack([dxo * x * sc + eg, dxo], 1)
    assert rel_err(dxi.cpu().numpy(), dxo * sc) < 1e-5
    assert rel_err(g.double().cpu().numpy(), (hq > 0) * (d @ w2.T)) < tol
    got = parts.double().sum(0).cpu().numpy()
    assert rel_err(got[:2 * R].reshape(R, 2), a.T @ d) < 1e-4
    assert rel_err(got[2 * R:], d.sum(0)) < 1e-4


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-5)])
def test_causal_conv1d_dgrad(dt, tol):
    K = sub("kernels")
    rng = np.random.default_rng(4)
    B, T, Cout, Kw = 2, 300, 64, 2
    for Cin, d, shift in ((1, 1, 1), (1, 1, 0), (3, 4, 0)):
        dy = torch.tensor(rng.standard_normal((B, T, Cout)), dtype=dt, device=DEV)
        w = rng.standard_normal((Kw, Cin, Cout))
        dyq = dy.double().cpu().numpy()
        dx_ref, _ = O._conv_backward(np.zeros((B, T, Cin)), w, d, dyq)          # adjoint of ops.py:6-10
        if shift:                                                                # adjoint of RightShift (ops.py:78-80)
            dx_ref = np.concatenate([dx_ref[:, shift:], np.zeros((B, shift, Cin))], 1)
        dx = torch.full((B, T, Cin), 2.0, device=DEV)
        K.causal_conv1d_dgrad(dy, dev(w), dx, d, shift=shift, accumulate=True, scale=0.5)
        assert rel_err(dx.cpu().numpy(), 2.0 + 0.5 * dx_ref) < 1e-5
        K.causal_conv1d_dgrad(dy, dev(w), dx, d, shift=shift)
        assert rel_err(dx.cpu().numpy(), dx_ref) < 1e-5


@pytest.mark.parametrize("T", [512, 1000, 4096])
def test_stft_power_fwd_bwd(T):
    """model.py:360-371 vs oracle (i) (np.fft) and the gradient of oracle (ii) (autograd through DFT matrices)."""
    K = sub("kernels")
    B = 3
    x = O.synthetic_audio(B, T, seed=5).astype(np.float64) + 0.01
    nf = K.stft_frames(T)
    assert nf == 1 + (T - 512) // 256
    spec = torch.zeros((B, nf, 257, 2), device=DEV); fp = torch.zeros((B, nf, 257), device=DEV)
    pw = torch.zeros((B, 257), device=DEV)
    K.stft_power(dev(x), spec, fp, pw)
    ref = O.stft_power(x)
    assert rel_err(pw.cpu().numpy(), ref) < 1e-4
    truth = O.synthetic_audio(B, T, seed=6).astype(np.float64)
    pt = torch.zeros((B, 257), device=DEV)
    K.stft_power(dev(truth), None, fp, pt)
    dpow = torch.zeros((B, 257), device=DEV); loss = torch.zeros(1, device=DEV)
    gamma, gs = 0.7, 0.25
    K.power_loss(pt, pw, gamma, gs, dpow, loss)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lref = ((OT.stft_power(torch.tensor(truth)) - OT.stft_power(xt)) ** 2).sum() * gamma
    lref.backward()
    assert abs(float(loss.item()) - float(lref)) < 1e-3 * abs(float(lref))
    dx = torch.full((B, T), 1.0, device=DEV)
    K.stft_power_bwd(spec, dpow, dx, accumulate=True)
    assert rel_err(dx.cpu().numpy() - 1.0, gs * xt.grad.numpy()) < 1e-3
    K.stft_power_bwd(spec, dpow, dx)
    assert rel_err(dx.cpu().numpy(), gs * xt.grad.numpy()) < 1e-3
    with pytest.raises(RuntimeError):   # a clip shorter than one frame (the reference would average zero frames)
        xd = dev(x)
        sub("_lib").call("srwn_stft_power", xd.data_ptr(), None, fp.data_ptr(), pw.data_ptr(), B, 300,
                         torch.cuda.current_stream().cuda_stream)


def test_mol_loss_dx_and_clamp():
    K = sub("kernels")
    L = sub("_lib")
    rng = np.random.default_rng(3)
    N, M = 3000, 10
    x = rng.uniform(-1, 1, N); x[:5] = [-1.0, -0.9995, 0.9995, 1.0, 0.0]
    l = rng.standard_normal((N, 4 * M)); l[:, 2 * M:3 * M] = rng.uniform(-9, 1, (N, M)); l[100:160, M:2 * M] += 30.0
    lg = torch.zeros((N, 64), dtype=torch.float32, device=DEV); lg[:, :4 * M] = dev(l)
    parts = torch.zeros((N + 255) // 256, device=DEV); dx = torch.zeros(N, device=DEV)
    K.mol_loss_dx(lg, dev(x), M, parts, dx, 0.5)
    ref = O.mol_loss(x[None], l[None])
    assert abs(float(parts.sum().item()) - ref) < 1e-3 * abs(ref)
    assert rel_err(dx.cpu().numpy(), 0.5 * O.mol_dx(x[None], l[None])[0]) < 1e-3
    # autograd agrees with the hand-derived d/dx (oracle ii vs oracle i)
    xt = torch.tensor(x, requires_grad=True)
    OT.mol_loss(xt[None], torch.tensor(l)[None]).backward()
    assert rel_err(O.mol_dx(x[None], l[None])[0], xt.grad.numpy()) < 1e-9
    v = np.array([-2.0, -1.0, -0.5, 1.0, 1.5], np.float32)
    y = torch.zeros(5, device=DEV); g = torch.zeros(5, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    dvv = dev(v); ones = torch.ones(5, device=DEV)
    L.call("srwn_clamp", dvv.data_ptr(), y.data_ptr(), 5, -1.0, 1.0, st)
    L.call("srwn_clamp_bwd", dvv.data_ptr(), ones.data_ptr(), g.data_ptr(), 5, -1.0, 1.0, st)
    assert y.cpu().tolist() == [-1.0, -1.0, -0.5, 1.0, 1.0]
    assert g.cpu().tolist() == [0.0, 1.0, 1.0, 1.0, 0.0]


def test_clip_by_global_norm_and_scaled_adam():
    K = sub("kernels")
    rng = np.random.default_rng(9)
    n = 10000
    for gscale, world in ((3.0, 1), (1e-3, 1), (5.0, 2)):
        g = rng.standard_normal(n) * gscale
        p0 = rng.standard_normal(n)
        parts = torch.zeros(K.sumsq_partials(n), device=DEV); out = torch.zeros(2, device=DEV)
        K.sumsq(dev(g), parts)
        K.clip_scale(parts, 1.0, 1.0 / world, out)
        clipped, gn = O.clip_by_global_norm([g / world], 1.0)
        assert abs(float(out[1]) - gn) < 1e-4 * gn
        p = dev(p0); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
        step = torch.zeros(1, dtype=torch.int64, device=DEV)
        K.adam_step_scaled(p, dev(g), m, v, step, 1e-2, out, True)
        ref, _, _ = O.adam_step_tf(p0, clipped[0], np.zeros(n), np.zeros(n), 1, lr=1e-2)
        assert int(step.item()) == 1
        assert rel_err(p.cpu().numpy(), ref) < 1e-5


# ---------------------------------------------------------------------------------------------------
# student engine
# ---------------------------------------------------------------------------------------------------
def _setup(dt, R, S, F, B=2, T=1024, E=5, pool=64, M=5, dil=(1, 2, 4, 8), alpha=0.8, beta=1.2, gamma=0.05):
    EG = sub("engine"); ST = sub("student")
    rng = np.random.default_rng(11)
    dil = list(dil)
    tsp = O.init_stack_params(40, dil, 2, 64, 256, 4 * M, cond_channels=E, bias_scale=0.05)
    flows = [O.init_flow_params(50 + i, dil, 2, R, S, E, bias_scale=0.05) for i in range(F)]
    for p in flows:                      # keep scale = exp(.) moderate so the output is not saturated by the clip
        p.head_w2 = p.head_w2 * 0.3
    noise = rng.logistic(0, 1, (B, T)) * 0.15
    cond = rng.standard_normal((B, T // pool, E))
    truth = O.synthetic_audio(B, T, seed=12).astype(np.float64)
    tcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * M,
                          cond_channels=E, pool_stride=pool, shift_input=True, dtype=dt, head_mode="mol")
    teacher = EG.WaveNetEngine(tcfg, B, T, DEV); teacher.load_oracle_params(tsp)
    fcfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, cond_channels=E, pool_stride=pool,
                          dtype=dt)
    stu = ST.StudentEngine(teacher, fcfg, F, alpha=alpha, beta=beta, gamma=gamma, learning_rate=1e-3)
    for f, p in zip(stu.flows, flows):
        f.load_oracle_params(p)
    stu.set_inputs(dev(noise), dev(truth), dev(cond))
    tl, _ = O.stack_forward(tsp, truth, shift_input=True, cond=cond, pool_stride=pool)
    return stu, flows, noise, cond, truth, tl, pool, (alpha, beta, gamma)


def _oracle_grads(flows, noise, cond, pool, tl, truth, abg):
    ts = [OT.TorchStack(p) for p in flows]
    res = OT.student_loss(ts, torch.tensor(noise), torch.tensor(cond), pool, torch.tensor(tl), torch.tensor(truth), *abg)
    res["loss"].backward()
    grads = [{n: t.grad.numpy() for n, t in OT.flow_named(st)} for st in ts]
    return res, grads


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("R,S,F,B,T,pool", [(64, 256, 2, 2, 1024, 64), (32, 128, 3, 2, 1024, 64), (64, 256, 1, 1, 512, 64),
                                            (32, 128, 2, 3, 640, 128)])
def test_student_forward_backward_step(dt, tol, R, S, F, B, T, pool):
    stu, flows, noise, cond, truth, tl, pool, abg = _setup(dt, R, S, F, B=B, T=T, pool=pool)
    B, T = noise.shape
    fw = O.student_forward(flows, noise, cond, pool)
    assert (np.abs(fw["out"]) < 1).mean() > 0.9
    ref = O.student_loss(fw, tl, truth, *abg)
    stu.forward()
    assert rel_err(stu.teacher.logits32[:, :tl.shape[-1]].cpu().numpy().reshape(tl.shape), tl) < tol
    assert rel_err(stu.out.cpu().numpy().reshape(B, T), fw["out"]) < tol
    got = stu.losses()
    for k in ("entropy", "power_loss", "cross_entropy", "loss"):
        assert abs(got[k] - ref[k]) < tol * max(abs(ref[k]), 1.0), (k, got[k], ref[k])
    res, grads = _oracle_grads(flows, noise, cond, pool, tl, truth, abg)
    assert abs(float(res["loss"]) - ref["loss"]) < 1e-6 * abs(ref["loss"])
    stu.backward()
    worst = 0.0
    for f, g in zip(stu.flows, grads):
        mine = f.named_tensors(f.grads)
        for n, r in g.items():
            a = mine[n].float().cpu().numpy()
            if dt == torch.float32:
                e = np.abs(a - r).max() / (np.abs(r).max() + 1e-30)
            else:
                e = np.linalg.norm(a - r) / (np.linalg.norm(r) + 1e-30)
            worst = max(worst, e)
            assert e < (tol if dt == torch.float32 else 0.15), (n, e)
    # clip_by_global_norm + Adam (model.py:382-385, 401), fp32 only (bf16 gradients differ by rounding)
    if dt == torch.float32:
        flat = [g[n] for g in grads for n in g]
        clipped, gn = O.clip_by_global_norm(flat, 1.0)
        before = [{n: t.float().cpu().numpy().copy() for n, t in f.named_tensors().items()} for f in stu.flows]
        stu.optimizer_step()
        assert abs(float(stu.clip[1].item()) - gn) < 1e-3 * gn
        it = iter(clipped)
        for f, b, g in zip(stu.flows, before, grads):
            now = f.named_tensors()
            for n in g:
                c = next(it)
                want, _, _ = O.adam_step_tf(b[n], c, np.zeros_like(c), np.zeros_like(c), 1, lr=1e-3)
                assert np.abs(now[n].float().cpu().numpy() - want).max() < 2e-5, n


def test_student_trains_and_graph_replay():
    """A few steps lower the loss; the captured-graph step equals the eager step."""
    stu, *_ = _setup(torch.float32, 64, 256, 2)
    stu.train_step()
    l0 = stu.losses()["loss"]
    for _ in range(8):
        stu.train_step()
    l1 = stu.losses()["loss"]
    assert l1 < l0, (l0, l1)
    a, *_ = _setup(torch.bfloat16, 64, 256, 2)
    b, *_ = _setup(torch.bfloat16, 64, 256, 2)
    a.train_step(); b.train_step()
    b.capture_graphs()
    for _ in range(3):
        a.train_step(); b.train_step_graphed()
    torch.cuda.synchronize()
    assert torch.equal(a.storage.params, b.storage.params)
    assert a.losses()["loss"] == b.losses()["loss"]


def test_student_slow_train_path():
    """ParallelWaveNet.train (model.py:599-632): per noise row, the 1-row noise broadcasts over the whole batch of
    encodings/truths, the loss divides by 1, the gradient is clipped on its own; the mean of the clipped gradients is
    applied.  Checked against the oracle run the same way."""
    stu, flows, noise, cond, truth, tl, pool, abg = _setup(torch.float32, 64, 256, 2)
    B = noise.shape[0]
    acc = None
    ref_losses = []
    for i in range(B):
        ts = [OT.TorchStack(p) for p in flows]
        ni = np.repeat(noise[i:i + 1], B, axis=0)
        res = OT.student_loss(ts, torch.tensor(ni), torch.tensor(cond), pool, torch.tensor(tl), torch.tensor(truth), *abg)
        loss = res["loss"] * B                       # the oracle divides by its batch; the reference by 1 row
        loss.backward()
        g = [v.grad.numpy() for st in ts for _, v in OT.flow_named(st)]
        c, _ = O.clip_by_global_norm(g, 1.0)
        acc = [a / B for a in c] if acc is None else [x + a / B for x, a in zip(acc, c)]
        ref_losses.append(float(loss))
    before = [{n: t.float().cpu().numpy().copy() for n, t in f.named_tensors().items()} for f in stu.flows]
    l, p = stu.train_per_sample()
    assert abs(l - np.mean(ref_losses)) < 1e-3 * abs(np.mean(ref_losses))
    it = iter(acc)
    for f, b in zip(stu.flows, before):
        now = f.named_tensors()
        for n, _ in OT.flow_named(OT.TorchStack(flows[0])):
            gref = next(it)
            want, _, _ = O.adam_step_tf(b[n], gref, np.zeros_like(gref), np.zeros_like(gref), 1, lr=1e-3)
            assert np.abs(now[n].float().cpu().numpy() - want).max() < 3e-5, n
    assert torch.equal(stu.noise, dev(noise))        # the staged batch is restored


def test_student_engine_vs_committed_golden(golden_dir):
    """The student engine (fp32) against the committed fixture tests/golden/student_small.npz."""
    import os
    EG = sub("engine"); ST = sub("student")
    g = np.load(os.path.join(golden_dir, "student_small.npz"))
    dil = g["dilations"].tolist(); pool = int(g["pool"]); R, S = (int(v) for v in g["widths"])
    B, T = g["noise"].shape; E = g["cond"].shape[-1]; M = g["teacher_logits"].shape[-1] // 4
    flows = [O.init_flow_params(int(s), dil, 2, R, S, E, bias_scale=0.1) for s in g["seeds"]]
    for p in flows:
        p.head_w2 = p.head_w2 * 0.3
    tcfg = EG.StackConfig(dilations=dil, dilation_channels=64, skip_channels=256, output_channels=4 * M, cond_channels=E,
                          pool_stride=pool, shift_input=True, dtype=torch.float32, head_mode="mol")
    teacher = EG.WaveNetEngine(tcfg, B, T, DEV)
    fcfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, cond_channels=E, pool_stride=pool,
                          dtype=torch.float32)
    a, b, c = (float(v) for v in g["abg"])
    stu = ST.StudentEngine(teacher, fcfg, len(flows), alpha=a, beta=b, gamma=c)
    for f, p in zip(stu.flows, flows):
        f.load_oracle_params(p)
    stu.set_inputs(dev(g["noise"]), dev(g["truth"]), dev(g["cond"]))
    stu.forward()
    # the fixture's teacher logits are given data: overwrite what the (random) teacher engine produced and redo the loss
    K = sub("kernels")
    teacher.logits32.zero_(); teacher.logits32[:, :4 * M].copy_(dev(g["teacher_logits"].reshape(B * T, 4 * M)))
    K.mol_loss_dx(teacher.logits32, stu.out, M, stu.ce_parts, stu.dx, stu.beta / B)
    K.reduce_loss(stu.ce_parts, stu.ce_parts.numel(), 1.0, stu.ce)
    assert rel_err(stu.out.cpu().numpy().reshape(B, T), g["out"]) < 1e-3
    got = stu.losses()
    for k in ("entropy", "power_loss", "cross_entropy", "loss"):
        assert abs(got[k] - float(g[k])) < 1e-3 * max(abs(float(g[k])), 1.0), k
    assert rel_err(stu.dx.cpu().numpy().reshape(B, T) * B / b, g["mol_dx"]) < 1e-3
