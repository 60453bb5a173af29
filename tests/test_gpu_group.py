"""Multi-layer (group) kernels vs their per-layer twins: the fused launches must reproduce what one launch per
layer stores (forward: bit for bit -- same MFMA order; backward: to accumulation-order round-off), for every
stride / halo / segment shape the grouping produces, in both dtypes.  The per-layer path itself is held to the
oracle in test_gpu_kernels / test_gpu_bwd / test_gpu_engine (and, fused path on by default, so is this one)."""
import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import DEV

pytestmark = pytest.mark.gpu

PART16_TOL = 5e-3      # bf16 partial blocks vs fp32 partials, kernel gradients: 2 x the worst measured (1.5e-3 .. 2.4e-3 over these
#                        shapes, few segments each; config 2 at full size: 1.3e-4 .. 5.7e-4, tools/partial16_probe.py)


def _pair(monkeypatch, dil, B, T, R, S, C, dt, E=0, pool=1, seg_rows=0, seed=3, ref_fuse="0", fuse_wt="0"):
    """The same model twice: one launch per layer (SRWN_FUSE=0) and the multi-layer kernels.  fuse_wt = "1": the second
    model also sums the layer weight gradients inside its 8-wave backward group kernel from the forward kernel's
    weight-gradient tiles (the default path of the engine); ref_fuse = "1" makes the first model the group kernels + the
    separate weight-gradient pass (those kernels' direct twin)."""
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=list(dil), dilation_channels=R, skip_channels=S, output_channels=C,
                         cond_channels=E, pool_stride=pool, shift_input=True, dtype=dt)
    monkeypatch.setenv("SRWN_WG_SLAB_ROWS", "3072")   # the same partial-sum slabs on both sides (the fused path sizes its own)
    monkeypatch.setenv("SRWN_FUSE", ref_fuse)
    monkeypatch.setenv("SRWN_FUSE_WT", "0")
    monkeypatch.setenv("SRWN_SEG_ROWS", str(seg_rows))
    ref = EG.WaveNetEngine(cfg, B, T, DEV, seed=seed)
    monkeypatch.setenv("SRWN_FUSE", "1")
    monkeypatch.setenv("SRWN_FUSE_WT", fuse_wt)
    monkeypatch.setenv("SRWN_WT_STORE_X", "0")     # (the production setting: inner layers' input rows are not stored)
    fus = EG.WaveNetEngine(cfg, B, T, DEV, seed=seed)
    assert ref.fuse_fwd == (ref_fuse == "1") and fus.fuse_fwd
    # biases are zero at init (tf.layers.conv1d defaults): make every one of them count
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name in ("BF", "BR", "BS", "init_b", "head_b1"):
        ref.view(name).copy_(0.1 * torch.randn(ref.view(name).shape, generator=g))
    fus.params.copy_(ref.params)
    ref.repack(); fus.repack()
    rng = np.random.default_rng(seed)
    audio = torch.tensor(np.clip(0.5 * np.sin(np.arange(B * T).reshape(B, T) * 0.05) + 0.1 * rng.normal(size=(B, T)), -1, 1),
                         dtype=torch.float32, device=DEV)
    tg = torch.tensor(rng.integers(0, C, size=(B, T)), dtype=torch.int32, device=DEV)
    cond = None
    if E:
        cond = torch.tensor(rng.normal(size=(B, T // pool, E)), dtype=torch.float32, device=DEV)
    for e in (ref, fus):
        e.set_inputs(audio, tg, cond)
    return ref, fus


SHAPES = [
    # dilations,                      B, T,    R,  S,   seg_rows
    ([1, 2, 4, 8, 16],                2, 700,  64, 256, 0),      # one stride-1 group, several segments per clip
    ([1, 2, 4, 8, 16] * 2,            1, 333,  32, 128, 0),      # the reference scripts' width, ragged length
    ([32, 64, 128, 256, 512],         2, 2100, 64, 256, 0),      # stride 32: residue classes, lengths differ by one
    ([4, 8, 16, 32],                  2, 515,  64, 256, 96),     # stride 4, forced short segments (several per class)
    ([1, 2, 4, 8, 16, 32, 64, 128],   1, 1100, 64, 256, 64),     # cut into {1..16} and {32..128}
    ([3, 6, 1, 2],                    2, 257,  32, 128, 0),      # gcd 3 then gcd 1
    ([1, 2, 4],                       3, 1,    64, 256, 0),      # T = 1
    ([2, 2, 2],                       1, 31,   64, 256, 0),      # fewer steps than one tile, halo 3 at stride 2
    ([512, 1],                        1, 600,  64, 256, 0),      # singletons fall back to the per-layer kernel
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil,B,T,R,S,seg", SHAPES)
def test_group_forward_equals_per_layer(monkeypatch, dt, dil, B, T, R, S, seg):
    ref, fus = _pair(monkeypatch, dil, B, T, R, S, 64, dt, seg_rows=seg)
    assert any(l1 - l0 >= 2 for l0, l1 in fus.groups) or dil == [512, 1]
    ref.forward(); fus.forward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.zs[l], fus.zs[l]), "z of layer %d" % l
        assert torch.equal(ref.xs[l + 1], fus.xs[l + 1]), "x of layer %d" % (l + 1)
    assert torch.equal(ref.loss, fus.loss)


def test_group_forward_eight_wave_body(monkeypatch):
    """The plain forward group kernel has two bodies for bf16 / 64 channels: twelve waves of two tiles (default) and eight
    of three (SRWN_GF_WAVES=8; also what the weight-gradient-tile mode runs).  Same arithmetic, same bits; the choice is
    read once per process, so the eight-wave run is a fresh child process."""
    import subprocess, sys, os, textwrap
    from tests._pkg import ROOT
    code = textwrap.dedent("""
        import importlib, sys, torch, numpy as np
        sys.path.insert(0, %r)
        EG = importlib.import_module("sr-wavenet_amd.engine")
        cfg = EG.StackConfig(dilations=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512], dilation_channels=64, skip_channels=256,
                             output_channels=64, shift_input=True, dtype=torch.bfloat16)
        eng = EG.WaveNetEngine(cfg, 2, 2100, "cuda", seed=3)
        rng = np.random.default_rng(3)
        eng.set_inputs(torch.tensor(np.clip(0.3 * rng.normal(size=(2, 2100)), -1, 1), dtype=torch.float32, device="cuda"),
                       torch.tensor(rng.integers(0, 64, size=(2, 2100)), dtype=torch.int32, device="cuda"))
        eng.forward(); torch.cuda.synchronize()
        torch.save({"zs": eng.zs.cpu(), "xtop": eng.xs[[5, 10]].cpu(), "loss": eng.loss.cpu()}, sys.argv[1])
    """ % ROOT)
    outs = []
    for waves in ("12", "8"):
        out = os.path.join(str(os.environ.get("TMPDIR", "/tmp")), "srwn_gf_waves_%s_%d.pt" % (waves, os.getpid()))
        env = dict(os.environ, SRWN_GF_WAVES=waves, SRWN_FUSE_WT="0", SRWN_FUSE="1")
        subprocess.run([sys.executable, "-c", code, out], env=env, check=True, cwd=ROOT, timeout=300)
        outs.append(torch.load(out, weights_only=True))
        os.remove(out)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_hand_counted_waits_against_full_drains():
    """The kernels that retire their LDS-DMA with a hand-counted s_waitcnt vmcnt(N) -- group_fwd, rowgemm, colgemm, the
    one-launch head, the skip weight gradient -- leave N younger loads / stores in flight; N is right only while the
    compiler emits as many of them as the source counts.  SRWN_SAFE_WAIT=1 replaces every such wait by vmcnt(0): one
    training step of the default bf16 path (hipGraph-free) must give the same bits either way (a count grown too large
    would let MFMAs read a buffer the DMA has not finished: different bits, or the same by luck -- run under both)."""
    import subprocess, sys, os, textwrap
    from tests._pkg import ROOT
    code = textwrap.dedent("""
        import importlib, sys, torch, numpy as np
        sys.path.insert(0, %r)
        EG = importlib.import_module("sr-wavenet_amd.engine")
        cfg = EG.StackConfig(dilations=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512], dilation_channels=64, skip_channels=256,
                             output_channels=256, shift_input=True, dtype=torch.bfloat16)
        eng = EG.WaveNetEngine(cfg, 2, 2100, "cuda", seed=3)
        assert eng.fused_wt and eng.skip_wt and eng.head_chain
        rng = np.random.default_rng(3)
        eng.set_inputs(torch.tensor(np.clip(0.3 * rng.normal(size=(2, 2100)), -1, 1), dtype=torch.float32, device="cuda"),
                       torch.tensor(rng.integers(0, 256, size=(2, 2100)), dtype=torch.int32, device="cuda"))
        for _ in range(2):
            eng.forward(); eng.backward(); eng.optimizer_step()
        torch.cuda.synchronize()
        torch.save({"zs": eng.zs.cpu(), "loss": eng.loss.cpu(), "grads": eng.grads.cpu(), "dtotal": eng.dtotal.cpu(),
                    "dcs": eng.dcs.cpu(), "params": eng.params.cpu()}, sys.argv[1])
    """ % ROOT)
    outs = []
    for safe in ("0", "1"):
        out = os.path.join(str(os.environ.get("TMPDIR", "/tmp")), "srwn_safe_wait_%s_%d.pt" % (safe, os.getpid()))
        env = dict(os.environ, SRWN_SAFE_WAIT=safe)
        subprocess.run([sys.executable, "-c", code, out], env=env, check=True, cwd=ROOT, timeout=300)
        outs.append(torch.load(out, weights_only=True))
        os.remove(out)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("E,pool", [(16, 8), (40, 50)])
def test_group_forward_conditioned(monkeypatch, dt, E, pool):
    dil = [1, 2, 4, 8, 16, 32, 64]
    ref, fus = _pair(monkeypatch, dil, 2, 400, 64, 256, 64, dt, E=E, pool=pool, seg_rows=128)
    ref.forward(); fus.forward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.zs[l], fus.zs[l]), "z of layer %d" % l
        assert torch.equal(ref.xs[l + 1], fus.xs[l + 1]), "x of layer %d" % (l + 1)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil,B,T,R,S,seg", SHAPES)
def test_group_backward_equals_per_layer(monkeypatch, dt, dil, B, T, R, S, seg):
    """fp32: the fused chain is the same arithmetic in the same order -> identical bits.  bf16: the gradient handed
    from layer to layer is the stored (rounded) one, where one launch per layer chains the unrounded tile from its UP
    half into its DOWN half -> equal to bf16 rounding."""
    ref, fus = _pair(monkeypatch, dil, B, T, R, S, 64, dt, seg_rows=seg)
    assert fus.fused_bwd and not ref.fused_bwd
    for e in (ref, fus):
        e.forward(); e.backward()
    torch.cuda.synchronize()
    L = len(dil)
    for l in range(L):
        if dt == torch.float32:
            assert torch.equal(ref.dfs[l], fus.dfs[l]), "df of layer %d" % l
            assert torch.equal(ref.gs[l], fus.gs[l]), "G of layer %d" % l
        else:
            assert _rel(fus.dfs[l], ref.dfs[l]) < 2e-2, "df of layer %d: %g" % (l, _rel(fus.dfs[l], ref.dfs[l]))
            assert _rel(fus.gs[l], ref.gs[l]) < 2e-2, "G of layer %d: %g" % (l, _rel(fus.gs[l], ref.gs[l]))
    gr, gf = ref.named_tensors(ref.grads), fus.named_tensors(fus.grads)
    for n in gr:
        if dt == torch.float32:
            assert torch.equal(gr[n], gf[n]), n
        else:
            assert _rel(gf[n], gr[n]) < 3e-2, "%s: %g" % (n, _rel(gf[n], gr[n]))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil,B,T,R,S,seg", SHAPES)
def test_group_wt_equals_twin(monkeypatch, dt, dil, B, T, R, S, seg):
    """The engine's default path -- srwn_residual_group_fwd_wt / srwn_residual_group_bwd_wt: the forward kernel also writes
    the transposed layer inputs and gate outputs, the backward kernel contracts them with df / G from its LDS image and
    stores neither -- against the plain group kernels followed by srwn_wgrad_layers: same activations bit for bit, same
    chain (the groups' bottom gradients agree bit for bit), the weight gradients are the same products summed per
    segment instead of per slab of rows."""
    ref, fus = _pair(monkeypatch, dil, B, T, R, S, 64, dt, seg_rows=seg, fuse_wt="1", ref_fuse="1")
    assert fus.fused_wt and ref.fused_bwd and not ref.fused_wt
    for e in (ref, fus):
        e.forward(); e.backward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.zs[l], fus.zs[l]), "z of layer %d" % l
    for _, l1 in fus.groups:      # (in this mode only the groups' top layers store their output rows)
        assert torch.equal(ref.xs[l1], fus.xs[l1]), "x of layer %d" % l1
    assert torch.equal(ref.loss, fus.loss)
    for l0, _ in fus.groups:
        assert torch.equal(ref.gs[l0], fus.gs[l0]), "bottom gradient of the group at layer %d" % l0
    gr, gf = ref.named_tensors(ref.grads), fus.named_tensors(fus.grads)
    tol = 2e-5 if dt == torch.float32 else 4e-3      # same operands; only the order of the fp32 sums differs
    for n in gr:
        assert torch.isfinite(gf[n]).all(), n
        assert _rel(gf[n], gr[n]) < tol, "%s: %g" % (n, _rel(gf[n], gr[n]))


@pytest.mark.parametrize("dil,B,T,seg", [([1, 2, 4, 8, 16], 2, 700, 0), ([32, 64, 128, 256, 512], 2, 2100, 0),
                                         ([4, 8, 16, 32], 2, 515, 96), ([1, 2, 4, 8, 16, 32, 64, 128], 1, 1100, 64),
                                         ([1, 2, 4], 3, 1, 0), ([2, 2, 2], 1, 31, 0), ([512, 1], 1, 600, 0),
                                         ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3, 1, 2000, 0)])
@pytest.mark.parametrize("part16", ["0", "1"])
def test_skip_wgrad_from_tiles_equals_wgrad256(monkeypatch, dil, B, T, seg, part16):
    """srwn_wgrad_skip_wt (the skip 1x1s' weight gradients contracted from the forward kernel's transposed gate outputs,
    dskip through LDS-DMA) against srwn_wgrad256 on z in the same engine.  The operands differ by one bf16 rounding: the
    forward kernel gates the unrounded tanh (the c its own residual 1x1 multiplies), srwn_wgrad256 gates the stored,
    rounded z (the c the skip sum multiplied) -- measured 3e-4 to 7e-4 relative on the gradient; a dropped tile or a
    wrong row would show as 1e-2 or more.  The bias gradient (column sums of dskip: no c) agrees to the order of the
    fp32 sums, by a block's idle waves or, when every block has four layers, by the column-sum kernel."""
    # part16 = "0": fp32 partial slabs -- the kernel's contraction is held to the exact one below at 2e-5; "1" (the
    # default): its partial slabs are bf16 blocks, one more rounding per partial sum (bound: 2 x the 1.0e-3 .. 1.4e-3 measured)
    monkeypatch.setenv("SRWN_PART16", part16)
    _, eng = _pair(monkeypatch, dil, B, T, 64, 256, 256 if len(dil) != 4 else 64, torch.bfloat16, seg_rows=seg,
                   fuse_wt="1", ref_fuse="1")     # (64 classes: the engine's other reduction schedule)
    assert eng.fused_wt and eng.skip_wt and (eng.skip_parts16 is not None) == (part16 == "1")
    K = sub("kernels")
    calls = []
    real = K.wgrad_skip_wt
    monkeypatch.setattr(K, "wgrad_skip_wt", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    out = {}
    for mode in (True, False):
        eng.skip_wt = mode
        eng.grads.zero_()
        eng.forward(); eng.backward()
        torch.cuda.synchronize()
        assert len(calls) == 1          # (the kernel under test really ran, once, in the first mode only)
        out[mode] = {n: eng.view(n, eng.grads).clone() for n in ("WS", "BS")}
    for n, tol in (("WS", 5e-3), ("BS", 2e-5)):
        assert torch.isfinite(out[True][n]).all()
        assert _rel(out[True][n], out[False][n]) < tol, "%s: %g" % (n, _rel(out[True][n], out[False][n]))
    # ... and exactly: the same contraction in torch from the tiles themselves.  Element (c, kg, j) of tile k of segment
    # (clip b, residue r, first position j0) is time r + st (j0 + 32 k + kordW(kg, j)) (csrc/srwn_group.h)
    L, R, Tn = len(dil), 64, T
    d = eng.dtotal.double()
    kg, j = np.meshgrid(np.arange(4), np.arange(8), indexing="ij")
    kord = 16 * (kg >> 1) + 2 * (kg & 1) + (j >> 2) + 4 * (j & 3)                      # [4, 8]
    for l in range(L):
        st, W = eng.wt_layer_st[l], eng.wt_layer_seg[l]
        J = -(-Tn // st); nsub = -(-J // W); KT = -(-W // 32)
        nseg = B * st * nsub
        # (a tile = four fragments of 16 channels, each [kg][channel][j]: csrc/srwn_group.h wt_load)
        tiles = eng.cTs[l][:nseg * KT * R * 32].view(nseg, KT, R // 16, 4, 16, 8).permute(0, 1, 2, 4, 3, 5)
        tiles = tiles.reshape(nseg, KT, R, 4, 8).double()
        seg = np.arange(nseg)
        b, rem = seg // (st * nsub), seg % (st * nsub)
        r, j0 = (rem, np.zeros_like(rem)) if nsub == 1 else (rem // nsub, (rem % nsub) * W)
        Jr = (Tn - r + st - 1) // st
        wseg = np.minimum(Jr - j0, W)
        pos = 32 * np.arange(KT)[None, :, None, None] + kord[None, None]                 # [1, KT, 4, 8]
        ok = pos < wseg[:, None, None, None]
        row = b[:, None, None, None] * Tn + r[:, None, None, None] + st * (j0[:, None, None, None] + pos)
        row = torch.tensor(np.where(ok, row, 0), device=d.device)                        # [nseg, KT, 4, 8]
        okt = torch.tensor(ok, device=d.device)
        c = torch.where(okt[:, :, None], tiles, torch.zeros((), dtype=torch.float64, device=d.device))
        dg = d[row.reshape(-1)].view(nseg, KT, 4, 8, -1)                                 # dskip rows of every tile element
        want = torch.einsum("sknqj,skqjc->nc", c, dg)
        got = out[True]["WS"][l].double()
        assert _rel(got, want) < (2e-5 if part16 == "0" else 3e-3), "layer %d: %g" % (l, _rel(got, want))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_group_wt_conditioned(monkeypatch, dt):
    """Conditioned decoder (model.py:176-189) in the weight-gradient-tile mode: the stored layer inputs include the
    conditioning bias, and the backward kernel also writes every layer's input gradient (summed per frame for the
    conditioning 1x1's gradient, model.py:180)."""
    dil = [1, 2, 4, 8, 16, 32, 64]
    ref, fus = _pair(monkeypatch, dil, 2, 400, 64, 256, 64, dt, E=16, pool=8, seg_rows=128, fuse_wt="1", ref_fuse="1")
    assert fus.fused_wt
    for e in (ref, fus):
        e.forward(); e.backward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.gs[l], fus.gs[l]), "G of layer %d" % l
    gr, gf = ref.named_tensors(ref.grads), fus.named_tensors(fus.grads)
    tol = 2e-5 if dt == torch.float32 else 4e-3
    for n in gr:
        assert _rel(gf[n], gr[n]) < tol, "%s: %g" % (n, _rel(gf[n], gr[n]))


def test_group_backward_vs_oracle_fp32(monkeypatch):
    """The fused path against the CPU oracle directly (two groups of a 1..128 cycle, ragged length, short segments)."""
    EG = sub("engine")
    from tests.test_gpu_kernels import dev, rel_err
    dil = [1, 2, 4, 8, 16, 32, 64, 128]
    B, T, R, S, C = 2, 300, 64, 256, 64
    monkeypatch.setenv("SRWN_FUSE", "1")
    monkeypatch.setenv("SRWN_SEG_ROWS", "100")
    sp = O.init_stack_params(5, dil, 2, R, S, C, bias_scale=0.05)
    rng = np.random.default_rng(5)
    audio = O.synthetic_audio(B, T, seed=1).astype(np.float64)
    codes = rng.integers(0, C, (B, T))
    ref_logits, cache = O.stack_forward(sp, audio, shift_input=True)
    loss = O.softmax_ce_per_timestep(ref_logits, codes)
    grads, _ = O.stack_backward(sp, cache, O.dlogits_per_timestep(ref_logits, codes))
    cfg = EG.StackConfig(dilations=dil, dilation_channels=R, skip_channels=S, output_channels=C, shift_input=True,
                         dtype=torch.float32)
    eng = EG.WaveNetEngine(cfg, B, T, DEV)
    eng.load_oracle_params(sp)
    assert eng.fused_bwd and eng.fuse_fwd and len(eng.groups) >= 2
    eng.set_inputs(dev(audio), dev(codes, torch.int32))
    logits = eng.forward(want_logits=True)
    eng.backward()
    torch.cuda.synchronize()
    assert rel_err(logits.cpu().numpy(), ref_logits) < 1e-3
    assert abs(float(eng.loss.item()) - loss) < 1e-3 * loss
    named = eng.named_tensors(eng.grads)
    for n, ref in O.flatten_named(grads, False):
        g = named[n].cpu().numpy()
        assert np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-3, n


def test_group_plan():
    Kn = sub("kernels")
    d = [2 ** i for i in range(10)] * 3
    assert Kn.group_plan(d, 31, 8) == [(0, 5), (5, 10), (10, 15), (15, 20), (20, 25), (25, 30)]
    assert Kn.group_plan(d, 63, 8) == [(0, 6), (6, 10), (10, 16), (16, 20), (20, 26), (26, 30)]
    assert Kn.group_plan([1, 2, 4, 8, 16] * 2, 31, 8) == [(0, 5), (5, 10)]
    assert Kn.group_plan([1, 2, 4, 8, 16] * 2, 31, 3) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert Kn.group_plan([512, 1], 31, 8) == [(0, 1), (1, 2)]
    assert Kn.group_plan([], 31, 8) == []


@pytest.mark.parametrize("B,T,C", [(2, 700, 256), (1, 333, 250), (3, 1, 256), (1, 8191, 256)])
def test_head_chain_equals_separate_launches(monkeypatch, B, T, C):
    """One-launch head (csrc/srwn_head.hip) against head 1x1 + softmax head + two head data gradients: the same
    products, every intermediate rounded to bf16 before it feeds the next product in both paths; the bias joins the sum
    last instead of first, so values agree to a bf16 ulp, not bit for bit.  Ragged row counts, classes padded to 256,
    T = 1."""
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=[1, 2, 4], dilation_channels=64, skip_channels=256, output_channels=C,
                         shift_input=True, dtype=torch.bfloat16)
    monkeypatch.setenv("SRWN_HEAD_CHAIN", "0")
    ref = EG.WaveNetEngine(cfg, B, T, DEV, seed=9)
    monkeypatch.setenv("SRWN_HEAD_CHAIN", "1")
    fus = EG.WaveNetEngine(cfg, B, T, DEV, seed=9)
    assert fus.head_chain and not ref.head_chain
    g = torch.Generator(device="cpu").manual_seed(9)
    for name in ("BS", "head_b1", "head_b2"):
        ref.view(name).copy_(0.2 * torch.randn(ref.view(name).shape, generator=g))
    ref.view("head_b2")[C:] = 0
    fus.params.copy_(ref.params)
    ref.repack(); fus.repack()
    rng = np.random.default_rng(9)
    audio = torch.tensor(np.clip(0.3 * rng.normal(size=(B, T)), -1, 1), dtype=torch.float32, device=DEV)
    tg = torch.tensor(rng.integers(0, C, size=(B, T)), dtype=torch.int32, device=DEV)
    for e in (ref, fus):
        e.set_inputs(audio, tg)
        e.forward(); e.backward()
    torch.cuda.synchronize()
    assert fus._head_bwd_done and not ref._head_bwd_done
    assert torch.equal(ref.r0, fus.r0)
    for name in ("r1", "dlogits", "da1", "dtotal"):
        a, b = getattr(ref, name).float(), getattr(fus, name).float()
        assert torch.isfinite(b).all()
        assert _rel(b, a) < 6e-3, "%s: %g" % (name, _rel(b, a))
        # element-wise: one bf16 ulp of the larger values, except where a relu mask flips on a ~0 activation
        bad = ((a - b).abs() > 2.0 ** -7 * a.abs().clamp_min(float(a.abs().max()) * 2.0 ** -6)).float().mean()
        assert float(bad) < 1e-3, "%s: %g of the elements off by more than an ulp" % (name, float(bad))
    assert abs(float(ref.loss) - float(fus.loss)) < 1e-4 * abs(float(ref.loss))
    gr, gf = ref.named_tensors(ref.grads), fus.named_tensors(fus.grads)
    for n in gr:
        assert _rel(gf[n], gr[n]) < 2e-2, "%s: %g" % (n, _rel(gf[n], gr[n]))


def _blk16(mat):
    """[rows, cols] fp32 -> the SRWN_PARTIALS_BLK16 order: 16 x 16 blocks (row block major), each as 64 lanes x 4 values --
    lane l holds rows 4 (l >> 4) + 0..3 of column l & 15 (the accumulator layout of v_mfma_f32_16x16x32_bf16)."""
    rows, cols = mat.shape
    b = mat.reshape(rows // 16, 4, 4, cols // 16, 16)          # [rb][l>>4][rr][cb][l&15]
    return b.permute(0, 3, 1, 4, 2).reshape(-1)               # [rb][cb][l>>4][l&15][rr]


@pytest.mark.parametrize("rows,cols,nslabs,nbatch", [(128, 64, 256, 3), (64, 64, 37, 2), (64, 32, 8, 1), (32, 32, 300, 5)])
def test_reduce_partials_bf16_blocks(rows, cols, nslabs, nbatch):
    """srwn_reduce_partials_multi on the bf16 16 x 16-block layout the backward group kernel writes with part16: equal (to
    the bit) to the fp32-slab reduction of the same (bf16-representable) values."""
    K = sub("kernels")
    g = torch.Generator(device="cpu").manual_seed(rows + nslabs)
    vals = torch.randn(nbatch, nslabs, rows, cols, generator=g).bfloat16()
    blocks = torch.stack([torch.stack([_blk16(vals[l, s].float()) for s in range(nslabs)]) for l in range(nbatch)]).bfloat16()
    n = rows * cols
    out16 = torch.full((nbatch, n + 8), float("nan"), dtype=torch.float32, device=DEV)
    out32 = torch.full((nbatch, n + 8), float("nan"), dtype=torch.float32, device=DEV)
    p16 = blocks.to(DEV).contiguous().view(-1)
    p32 = vals.float().to(DEV).contiguous().view(-1)
    K.reduce_partials_multi([(p16, nslabs, n, nbatch, True, 0.5, out16.data_ptr(), n + 8, cols)])
    K.reduce_partials_multi([(p32, nslabs, n, nbatch, True, 0.5, out32.data_ptr(), n + 8)])
    assert torch.equal(out16[:, :n], out32[:, :n])
    assert bool(torch.isnan(out16[:, n:]).all())              # nothing written beyond a batch's block
    ref = 0.5 * vals.double().sum(1).reshape(nbatch, n)
    assert float((out16[:, :n].cpu().double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("dil,B,T,R,S,seg", SHAPES)
def test_group_wt_bf16_partial_blocks_vs_fp32_partials(monkeypatch, dil, B, T, R, S, seg):
    """SRWN_PART16 (default): the backward group kernels store their per-workgroup weight-gradient partials as bf16 blocks
    instead of fp32.  Same chain, same products: activations, bottom gradients and every gradient that does not pass
    through those partials are bit-equal to the fp32-partial build of the same path; the conv-tap, residual 1x1 and (64 / 256
    channels: srwn_wgrad_skip_wt) skip 1x1 kernel gradients differ by one bf16 rounding per partial sum (bound: 2 x the worst measured over these shapes)."""
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=list(dil), dilation_channels=R, skip_channels=S, output_channels=64, shift_input=True,
                         dtype=torch.bfloat16)
    monkeypatch.setenv("SRWN_SEG_ROWS", str(seg))
    engs = []
    for p16 in ("0", "1"):
        monkeypatch.setenv("SRWN_PART16", p16)
        e = EG.WaveNetEngine(cfg, B, T, DEV, seed=3)
        assert e.fused_wt and e.part16 == (p16 == "1")
        engs.append(e)
    ref, fus = engs
    rng = np.random.default_rng(3)
    audio = torch.tensor(np.clip(0.5 * np.sin(np.arange(B * T).reshape(B, T) * 0.05) + 0.1 * rng.normal(size=(B, T)), -1, 1),
                         dtype=torch.float32, device=DEV)
    tg = torch.tensor(rng.integers(0, 64, size=(B, T)), dtype=torch.int32, device=DEV)
    for e in engs:
        e.set_inputs(audio, tg)
        e.forward(); e.backward()
    torch.cuda.synchronize()
    assert float(ref.loss.item()) == float(fus.loss.item())
    for l0, _ in fus.groups:
        assert torch.equal(ref.gs[l0], fus.gs[l0])
    import os
    worst = 0.0
    for name, sec in fus.sections.items():
        a, b = ref.view(name, ref.grads), fus.view(name, fus.grads)
        if name in ("WF", "WR") or (name == "WS" and fus.skip_parts16 is not None):      # (64 / 256 channels: the skip kernels'
            assert bool(torch.isfinite(b).all())                                         #  partial slabs are bf16 blocks as well)
            e = _rel(b, a)
            worst = max(worst, e)
            assert e < PART16_TOL, (name, e)
        elif name in ("init_w", "init_b") and fus.fuse_icg != ref.fuse_icg:
            # (bf16 mode: the first group's backward launch forms the input conv's gradient itself only together with the
            # bf16 partial blocks; with fp32 slabs srwn_init_conv_wgrad does -- the same products in another order)
            assert _rel(b, a) < 2e-5, (name, _rel(b, a))
        else:
            assert torch.equal(a, b), name
    if os.environ.get("SRWN_PRINT_ERR"):
        print("MEASURED part16 vs fp32 partials, WF/WR rel (%s B=%d T=%d R=%d): %.3e" % (dil, B, T, R, worst))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil,B,T,R,S,seg", [SHAPES[0], SHAPES[1], SHAPES[3], ([1, 2, 4], 3, 1, 64, 256, 0), ([512, 1], 1, 600, 64, 256, 0)])
def test_input_conv_fused_into_the_first_group(monkeypatch, dt, dil, B, T, R, S, seg):
    """SRWN_FUSE_IC (default): the stack's input conv (model.py:40 / 172-173, RightShift folded in) is computed inside the
    first layer group's forward kernel instead of being a launch that writes the group's input rows and a read that
    fetches them back.  Same arithmetic: every stored activation, the loss and every gradient are bit-equal to the
    separate launch; xs[0] is not written in this mode."""
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=list(dil), dilation_channels=R, skip_channels=S, output_channels=64, shift_input=True, dtype=dt)
    monkeypatch.setenv("SRWN_SEG_ROWS", str(seg))
    engs = []
    for ic in ("0", "1"):
        monkeypatch.setenv("SRWN_FUSE_IC", ic)
        e = EG.WaveNetEngine(cfg, B, T, DEV, seed=3)
        g = torch.Generator(device="cpu").manual_seed(5)
        for name in ("init_b", "BF", "BR"):
            e.view(name).copy_(0.1 * torch.randn(e.view(name).shape, generator=g))
        e.repack()
        engs.append(e)
    ref, fus = engs
    rng = np.random.default_rng(3)
    audio = torch.tensor(np.clip(0.5 * np.sin(np.arange(B * T).reshape(B, T) * 0.05) + 0.1 * rng.normal(size=(B, T)), -1, 1),
                         dtype=torch.float32, device=DEV)
    tg = torch.tensor(rng.integers(0, 64, size=(B, T)), dtype=torch.int32, device=DEV)
    for e in engs:
        e.xs.fill_(7.0)
        e.set_inputs(audio, tg)
        e.forward(); e.backward()
    torch.cuda.synchronize()
    assert fus.fused_wt and fus._ic_fused and not ref._ic_fused
    assert bool((fus.xs[0] == 7.0).all()) and not bool((ref.xs[0] == 7.0).any())
    assert torch.equal(ref.zs, fus.zs) and torch.equal(ref.xTs, fus.xTs) and torch.equal(ref.cTs, fus.cTs)
    for l0, l1 in fus.groups:
        assert torch.equal(ref.xs[l1], fus.xs[l1])
    assert float(ref.loss.item()) == float(fus.loss.item())
    # ... and the first group's BACKWARD launch forms the input conv's kernel + bias gradient from its bottom gradient
    # while it is on the chip (the audio as a high + a low bf16 part on the matrix pipe; per segment) instead of
    # srwn_init_conv_wgrad's pass over gs[0] (FMAs; slabs of 256 rows): the same products in another order
    assert fus.fuse_icg and not ref.fuse_icg
    for name in fus.sections:
        a, b = ref.view(name, ref.grads), fus.view(name, fus.grads)
        if name in ("init_w", "init_b"):
            assert bool(torch.isfinite(b).all()) and _rel(b, a) < 2e-5, (name, _rel(b, a))
        else:
            assert torch.equal(a, b), name
