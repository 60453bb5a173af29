"""GPU parity of the whole stack (forward logits, loss, every gradient, one Adam step) vs the oracle.

fp32 mode must meet the north-star tolerance (1e-3 relative); bf16 mode is checked loosely (it
rounds every stored activation to 8 significant bits)."""
import os

import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import DEV, dev, rel_err

pytestmark = pytest.mark.gpu


def _engine(sp, B, T, R, S, C, dt, shift=True, E=0, pool=1, lr=1e-3):
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=list(sp.dilations), dilation_channels=R, skip_channels=S, output_channels=C,
                         cond_channels=E, pool_stride=pool, shift_input=shift, dtype=dt, learning_rate=lr)
    eng = EG.WaveNetEngine(cfg, B, T, DEV)
    eng.load_oracle_params(sp)
    return eng


@pytest.fixture(autouse=True)
def _shipped_default(monkeypatch):
    """Every test of this file runs the SHIPPED default (SRWN_WT_STORE_X=0: the inputs of the layers inside a group are
    not stored, only their transposed tiles) unless it asks otherwise: the two bf16 tests that rebuild the oracle's
    backward from the engine's saved activations build a second engine with the rows kept (_with_inner_inputs) and
    hold the default engine's gradients to that one's bit for bit."""
    monkeypatch.setenv("SRWN_WT_STORE_X", "0")


def _with_inner_inputs(monkeypatch, build):
    """`build()` once more with SRWN_WT_STORE_X=1 (the rows of every layer input kept for inspection)."""
    monkeypatch.setenv("SRWN_WT_STORE_X", "1")
    try:
        eng = build()
        assert eng.wt_store_x
        return eng
    finally:
        monkeypatch.setenv("SRWN_WT_STORE_X", "0")


def _bwd_oracle_on_engine_forward(eng, sp, cond=None, pool=1):
    """bf16 mode: relu masks flip where an activation is within rounding of zero, which changes single
    gradient entries by O(1) and says nothing about the backward kernels.  So the backward is judged
    given the forward: the oracle's backward runs on the engine's own saved activations."""
    f = lambda t: t.double().cpu().numpy()
    B, T, L = eng.B, eng.T, eng.L
    layers = []
    for l in range(L):
        x = f(eng.xs[l])          # the layer's complete input: the conditioning bias is already in it
        z = f(eng.zs[l]); sg = 1 / (1 + np.exp(-z))
        layers.append(dict(x=x, z=z, s=sg, c=z * sg, g=None))
    x0 = f(eng.audio)[:, :, None]
    if eng.cfg.shift_input:
        x0 = O.right_shift(x0)
    r0 = f(eng.r0).reshape(B, T, -1); r1 = f(eng.r1).reshape(B, T, -1)
    cache = dict(x0=x0, layers=layers, total=r0, r0=r0, a1=r1, r1=r1)
    dlog = f(eng.dlogits).reshape(B, T, -1)[:, :, :eng.C]
    grads, _ = O.stack_backward(sp, cache, dlog, cond=cond, pool_stride=pool)
    return grads


def _check_grads(eng, grads, tol, with_cond=False):
    gn = dict(O.flatten_named(grads, with_cond))
    got = eng.named_tensors(eng.grads)
    worst = 0.0
    for n, ref in gn.items():
        g = got[n].float().cpu().numpy()
        scale = np.abs(ref).max()
        if scale < 1e-12:
            assert np.abs(g).max() < 1e-6, n
            continue
        if tol > 1e-2:   # bf16: relu-mask flips make single entries noisy; judge the tensor in L2
            e = np.linalg.norm(g - ref) / np.linalg.norm(ref)
        else:
            e = np.abs(g - ref).max() / scale
        worst = max(worst, e)
        assert e < tol, (n, e)
    return worst


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_stack_small_golden(golden_dir, dt, tol):
    g = np.load(os.path.join(golden_dir, "stack_small.npz"))
    dil = g["dilations"].tolist(); L = len(dil)
    layers = [O.LayerParams(g[f"p.l{i}.wf"], g[f"p.l{i}.bf"], None, None, g[f"p.l{i}.wr"], g[f"p.l{i}.br"],
                            g[f"p.l{i}.ws"], g[f"p.l{i}.bs"]) for i in range(L)]
    sp = O.StackParams(g["p.init_w"], g["p.init_b"], layers, g["p.head_w1"], g["p.head_b1"], g["p.head_w2"],
                       g["p.head_b2"], tuple(dil))
    B, T = g["audio"].shape
    eng = _engine(sp, B, T, 32, 32, 32, dt)
    eng.set_inputs(dev(g["audio"]), dev(g["codes"], torch.int32))
    logits = eng.forward(want_logits=True)
    assert rel_err(logits.cpu().numpy(), g["logits"]) < tol
    assert abs(float(eng.loss.item()) - float(g["loss"])) < tol * float(g["loss"])
    eng.backward()
    got = eng.named_tensors(eng.grads)
    for k in g.files:
        if k.startswith("g."):
            ref = g[k]; scale = np.abs(ref).max()
            d = got[k[2:]].cpu().numpy() - ref
            e = (np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30)) if tol > 1e-2 else np.abs(d).max() / max(scale, 1e-12)
            assert e < tol or scale < 1e-12, (k, e)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("R,S,C,B,T", [(64, 256, 256, 2, 300), (32, 128, 30, 1, 515), (64, 64, 100, 3, 64)])
def test_stack_forward_backward_adam(monkeypatch, dt, tol, R, S, C, B, T):
    dil = [1, 2, 4, 8, 16, 32, 64, 1, 2]
    sp = O.init_stack_params(21, dil, 2, R, S, C, bias_scale=0.05)
    rng = np.random.default_rng(T)
    audio = O.synthetic_audio(B, T, seed=2).astype(np.float64)
    codes = rng.integers(0, C, (B, T))
    logits, cache = O.stack_forward(sp, audio, shift_input=True)
    loss = O.softmax_ce_per_timestep(logits, codes)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(logits, codes))
    eng = _engine(sp, B, T, R, S, C, dt, lr=1e-2)
    eng.set_inputs(dev(audio), dev(codes, torch.int32))
    lg = eng.forward(want_logits=True)
    assert rel_err(lg.cpu().numpy(), logits) < tol
    assert abs(float(eng.loss.item()) - loss) < tol * loss
    eng.backward()
    assert not eng.wt_store_x
    if dt == torch.bfloat16:
        # the oracle's backward on the saved activations of a twin that keeps every layer input; the default engine
        # (same kernels, fewer stores) must have formed exactly the same gradients
        twin = _with_inner_inputs(monkeypatch, lambda: _engine(sp, B, T, R, S, C, dt, lr=1e-2))
        twin.set_inputs(dev(audio), dev(codes, torch.int32))
        twin.forward(want_logits=True); twin.backward()      # (the same head launches as `eng` ran)
        assert torch.equal(twin.grads, eng.grads) and float(twin.loss.item()) == float(eng.loss.item())
        grads = _bwd_oracle_on_engine_forward(twin, sp)
    _check_grads(eng, grads, tol)
    # one TF-Adam step on the oracle's gradients vs the engine's update
    before = {k: v.clone() for k, v in eng.named_tensors().items()}
    eng.optimizer_step()
    gn = dict(O.flatten_named(grads, False)); pn = dict(O.flatten_named(sp, False))
    after = eng.named_tensors()
    for n in gn:
        th, _, _ = O.adam_step_tf(pn[n], gn[n], np.zeros_like(gn[n]), np.zeros_like(gn[n]), 1, lr=1e-2)
        if dt == torch.float32:
            # first Adam step moves by ~lr*sign(g): compare only where the gradient is not ~0
            mask = np.abs(gn[n]) > 1e-6 * max(np.abs(gn[n]).max(), 1e-30)
            d = np.abs(after[n].cpu().numpy() - th)[mask]
            assert d.size == 0 or d.max() < 2e-4, (n, d.max())
    # training loop sanity: the loss goes down
    l0 = float(eng.loss.item())
    for _ in range(5):
        eng.train_step()
    assert float(eng.loss.item()) < l0


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("E", [16, 20])
def test_stack_conditioned_decoder(monkeypatch, dt, tol, E):
    """createDecoder variant (model.py:158-196): RightShift + per-layer conditioning add."""
    dil = [1, 2, 4, 8, 16, 1, 2, 4]
    R, S, C, B, T, pool = 64, 128, 64, 2, 256, 32
    sp = O.init_stack_params(5, dil, 2, R, S, C, cond_channels=E, bias_scale=0.05)
    rng = np.random.default_rng(0)
    audio = O.synthetic_audio(B, T, seed=4).astype(np.float64)
    codes = rng.integers(0, C, (B, T))
    cond = rng.standard_normal((B, T // pool, E))
    logits, cache = O.stack_forward(sp, audio, shift_input=True, cond=cond, pool_stride=pool)
    loss = O.softmax_ce_per_timestep(logits, codes)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(logits, codes), cond=cond, pool_stride=pool)
    eng = _engine(sp, B, T, R, S, C, dt, E=E, pool=pool)
    eng.set_inputs(dev(audio), dev(codes, torch.int32), dev(cond))
    lg = eng.forward(want_logits=True)
    assert rel_err(lg.cpu().numpy(), logits) < tol
    assert abs(float(eng.loss.item()) - loss) < tol * loss
    eng.backward()
    assert not eng.wt_store_x
    if dt == torch.bfloat16:
        twin = _with_inner_inputs(monkeypatch, lambda: _engine(sp, B, T, R, S, C, dt, E=E, pool=pool))
        twin.set_inputs(dev(audio), dev(codes, torch.int32), dev(cond))
        twin.forward(want_logits=True); twin.backward()      # (the same head launches as `eng` ran)
        assert torch.equal(twin.grads, eng.grads) and float(twin.loss.item()) == float(eng.loss.item())
        grads = _bwd_oracle_on_engine_forward(twin, sp, cond=cond, pool=pool)
    _check_grads(eng, grads, tol, with_cond=True)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("C", [30, 256])
def test_pooled_classifier_head(dt, tol, C):
    """class WaveNet (model.py:8-72): clip-level softmax, soft labels, mean CE."""
    EG = sub("engine")
    dil = [1, 2, 4, 8, 16, 32]
    R, S, B, T = 32, 128, 3, 200
    sp = O.init_stack_params(8, dil, 2, R, S, C, bias_scale=0.05)
    rng = np.random.default_rng(C)
    audio = O.synthetic_audio(B, T, seed=6).astype(np.float64)
    tg = rng.random((B, C)); tg /= tg.sum(-1, keepdims=True)
    logits, cache = O.stack_forward(sp, audio)
    loss = O.wavenet_loss_pooled(logits, tg)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_pooled(logits, tg))
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, dtype=dt,
                         head_mode="pooled")
    eng = EG.WaveNetEngine(cfg, B, T, DEV); eng.load_oracle_params(sp)
    eng.set_inputs(dev(audio), dev(tg))
    eng.forward()
    assert rel_err(eng.probs.cpu().numpy(), O.wavenet_predict(sp, audio)[:, 0, :]) < tol
    assert abs(float(eng.loss.item()) - loss) < tol * loss
    eng.backward()
    if dt == torch.float32:
        _check_grads(eng, grads, tol)
    l0 = float(eng.loss.item())
    for _ in range(5):
        eng.train_step()
    assert float(eng.loss.item()) < l0


def test_deterministic_and_tf_names():
    dil = [1, 2, 4]
    sp = O.init_stack_params(1, dil, 2, 64, 64, 32)
    eng = _engine(sp, 2, 128, 64, 64, 32, torch.bfloat16)
    a = dev(O.synthetic_audio(2, 128)); c = dev(np.zeros((2, 128)), torch.int32)
    eng.set_inputs(a, c); eng.forward(); eng.backward(); g1 = eng.grads.clone(); l1 = eng.loss.clone()
    eng.set_inputs(a, c); eng.forward(); eng.backward()
    assert torch.equal(g1, eng.grads) and torch.equal(l1, eng.loss)   # slab reductions are order-fixed
    names = eng.tf_variables("WaveNet", decoder=False)
    assert names["WaveNet/conv1d_3/kernel"].shape == (1, 64, 64)
    assert names["WaveNet/dilated_conv_2_gate/dilated_conv_2_Kernel"].shape == (2, 64, 64)
    assert names["WaveNet/conv1d_7/kernel"].shape == (1, 64, 32)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("M", [5, 10])
def test_mixture_of_logistics_teacher(dt, tol, M):
    """The reference's live teacher loss (model.py:114; ops.py:124-175) on the decoder stack: loss (a SUM over
    batch and time), the gradient wrt the head parameters, and every stack gradient vs the oracle."""
    EG = sub("engine")
    K = sub("kernels")
    dil = [1, 2, 4, 8, 16, 1, 2]
    R, S, B, T = 64, 256, 2, 200
    C = 4 * M
    sp = O.init_stack_params(31, dil, 2, R, S, C, bias_scale=0.05)
    audio = O.synthetic_audio(B, T, seed=8).astype(np.float64)
    audio[0, :3] = [-1.0, 1.0, 0.9995]                         # the edge branches of ops.py:169
    logits, cache = O.stack_forward(sp, audio, shift_input=True)
    loss = O.mol_loss(audio, logits)
    grads, _ = O.stack_backward(sp, cache, O.mol_dlogits(audio, logits))
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=dt, head_mode="mol", learning_rate=1e-4)
    eng = EG.WaveNetEngine(cfg, B, T, DEV); eng.load_oracle_params(sp)
    eng.set_inputs(dev(audio))
    lg = eng.forward(want_logits=True)
    assert rel_err(lg.cpu().numpy(), logits) < tol
    assert abs(float(eng.loss.item()) - loss) < tol * abs(loss)
    eng.backward()
    if dt == torch.float32:
        _check_grads(eng, grads, tol)
    else:
        # bf16: judge the backward on the engine's own forward (see _bwd_oracle_on_engine_forward)
        f = lambda t: t.double().cpu().numpy()
        g = O.mol_dlogits(audio, f(eng.logits32)[:, :C].reshape(B, T, C))
        assert rel_err(f(eng.dlogits)[:, :C].reshape(B, T, C), g) < tol
    l0 = float(eng.loss.item())
    for _ in range(5):
        eng.train_step()
    assert float(eng.loss.item()) < l0


def test_mol_kernel_branches():
    """All four tf.where branches + the -7 clamp through srwn_mol_loss, vs the oracle."""
    K = sub("kernels")
    rng = np.random.default_rng(3)
    N, M = 3000, 5
    x = rng.uniform(-1, 1, N); x[:5] = [-1.0, -0.9995, 0.9995, 1.0, 0.0]
    l = rng.standard_normal((N, 4 * M)); l[:, 2 * M:3 * M] = rng.uniform(-9, 1, (N, M)); l[100:160, M:2 * M] += 30.0
    _, aux = O.mol_log_probs(x[None], l[None])
    assert set(np.unique(aux["case"])) == {0, 1, 2, 3}
    lg = torch.zeros((N, 32), dtype=torch.float32, device=DEV); lg[:, :4 * M] = dev(l)
    parts = torch.zeros((N + 255) // 256, dtype=torch.float32, device=DEV)
    dl = torch.full((N, 32), float("nan"), dtype=torch.float32, device=DEV)
    K.mol_loss(lg, dev(x), M, parts, dl, 1.0)
    assert abs(float(parts.sum().item()) - O.mol_loss(x[None], l[None])) < 1e-3 * abs(O.mol_loss(x[None], l[None]))
    ref = O.mol_dlogits(x[None], l[None])[0]
    assert rel_err(dl[:, :4 * M].cpu().numpy(), ref) < 1e-3
    assert torch.all(dl[:, 3 * M:] == 0)


def test_bucketed_allreduce_schedule_matches_plain_step(monkeypatch):
    """Data-parallel schedule: {forward, upper backward} | all-reduce of the skip+head bucket in flight |
    {lower backward} | all-reduce of the layer bucket | {Adam}.  With one rank the collectives are identities, so the
    split schedule (eager and as three hipGraphs) must reproduce the plain step bit for bit."""
    import socket
    import torch.distributed as dist
    EG = sub("engine")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    try:
        dil = [1, 2, 4, 8, 16, 32] * 3
        B, T, R, S, C = 2, 400, 64, 256, 256
        audio = O.synthetic_audio(B, T, seed=3)
        codes = O.mu_law_encode(audio, C)

        def run(buckets, graph):
            monkeypatch.setenv("SRWN_FORCE_DIST", "1")
            monkeypatch.setenv("SRWN_BUCKETS", buckets)
            cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                                 dtype=torch.bfloat16, learning_rate=1e-3)
            eng = EG.WaveNetEngine(cfg, B, T, DEV, seed=5)
            assert eng.bucketed == (buckets == "1")
            eng.set_inputs(dev(audio), dev(codes, torch.int32))
            eng.train_step()
            if graph:
                eng.capture_graphs()
                assert (eng._g_b2 is not None) == (buckets == "1")
            for _ in range(3):
                (eng.train_step_graphed if graph else eng.train_step)()
            torch.cuda.synchronize()
            return eng.params.clone(), float(eng.loss.item()), eng

        p_plain, l_plain, e0 = run("0", False)
        assert 0 < e0.split_layer < len(dil) and e0.bucket_off == e0.sections["WS"].offset
        for graph in (False, True):
            p, l, _ = run("1", graph)
            assert torch.equal(p, p_plain) and l == l_plain, graph
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,T", [(1, 1), (3, 7), (1, 31), (2, 33), (5, 65)])
@pytest.mark.parametrize("R,S", [(64, 256), (32, 128)])
def test_stack_degenerate_clip_lengths(B, T, R, S):
    """Clips shorter than one 32-step tile, shorter than the dilations, one sample long: forward, loss and every
    gradient against the oracle (fp32), with canaries around the activation buffers to catch out-of-bounds stores."""
    dil = [1, 2, 4, 8, 16, 32, 1, 2]
    C = 256
    sp = O.init_stack_params(33, dil, 2, R, S, C, bias_scale=0.05)
    rng = np.random.default_rng(B * 100 + T)
    audio = rng.uniform(-1, 1, (B, T))
    codes = rng.integers(0, C, (B, T))
    logits, cache = O.stack_forward(sp, audio, shift_input=True)
    loss = O.softmax_ce_per_timestep(logits, codes)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(logits, codes))
    eng = _engine(sp, B, T, R, S, C, torch.float32)
    # re-home the big activation stacks inside canary-padded storage
    pad = 4096
    guards = []
    assert eng.fused_wt and not eng.wt_store_x      # the shipped default path, its tiles and partial slabs included
    for name in ("xs", "zs", "dfs", "gs", "dcs", "r0", "r1", "da1", "dtotal", "dlogits", "xTs", "cTs", "pl_f", "pl_r",
                 "pl_bf", "pl_br"):
        if not hasattr(eng, name):
            continue
        t = getattr(eng, name)
        big = torch.full((t.numel() + 2 * pad,), 12345.0, dtype=t.dtype, device=DEV)
        big[pad:pad + t.numel()] = 0
        setattr(eng, name, big[pad:pad + t.numel()].view(t.shape))
        guards.append((name, big, t.numel()))
    eng.set_inputs(dev(audio), dev(codes, torch.int32))
    lg = eng.forward(want_logits=True)
    assert rel_err(lg.cpu().numpy(), logits) < 1e-3
    assert abs(float(eng.loss.item()) - loss) < 1e-3 * loss
    eng.backward()
    _check_grads(eng, grads, 1e-3)
    for name, big, n in guards:
        assert bool((big[:pad] == 12345.0).all()) and bool((big[pad + n:] == 12345.0).all()), name


@pytest.mark.gpu
def test_ctypes_binding_runs_the_same_smoke_step():
    """The suite runs on the pybind11 binding (the default); the ctypes binding of the same library takes the same
    arguments: __graft_entry__.smoke() -- a forward against the oracle and three training steps -- in a child process
    with SRWN_BINDING=ctypes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRWN_BINDING="ctypes")
    code = ("import importlib, __graft_entry__ as g; g.smoke(); "
            "L = importlib.import_module('sr-wavenet_amd._lib'); assert L.BINDING == 'ctypes', L.BINDING; print('binding', L.BINDING)")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout and "binding ctypes" in r.stdout


def test_training_step_loss_equals_forward_loss():
    """The training step leaves the final sum of its loss partials to the skip / head reduction launch of the backward pass
    (one more job there instead of a launch of its own); the value must be the one forward() alone reports, to the bit."""
    EG = sub("engine")
    dil = [1, 2, 4, 8, 16, 32] * 2
    B, T, R, S, C = 2, 700, 64, 256, 256
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=torch.bfloat16, learning_rate=1e-3)
    eng = EG.WaveNetEngine(cfg, B, T, DEV, seed=5)
    audio = O.synthetic_audio(B, T, seed=3)
    eng.set_inputs(dev(audio), dev(O.mu_law_encode(audio, C), torch.int32))
    assert eng.head_chain and eng.batch_reduce
    eng.forward()
    l_fwd = float(eng.loss.item())
    eng.loss.fill_(-1.0)
    eng.train_step()                       # (the loss is of the parameters BEFORE the update)
    assert float(eng.loss.item()) == l_fwd
    eng.forward()
    l2 = float(eng.loss.item())
    eng.capture_graphs()
    eng.loss.fill_(-1.0)
    eng.train_step_graphed()
    torch.cuda.synchronize()
    assert float(eng.loss.item()) == l2 and l2 != l_fwd
