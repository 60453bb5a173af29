"""GPU parity tests of the backward kernels (layer data gradient, weight gradients, Adam) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import DEV, TOL, dev, rel_err

pytestmark = pytest.mark.gpu


def _gate_grad(z):
    s = 1 / (1 + np.exp(-z))
    return (s + z * s * (1 - s)) * (1 - z * z)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,S", [(64, 256), (32, 64), (32, 128)])
@pytest.mark.parametrize("B,T,d_up", [(2, 64, 1), (1, 100, 8), (2, 300, 64), (1, 50, 128), (3, 1000, 16),
                                      (1, 1100, 256), (1, 2100, 512), (2, 700, 512)])   # the benchmark's upper dilations
def test_residual_layer_bwd(dt, R, S, B, T, d_up):
    K = sub("kernels"); P = sub("packing")
    rng = np.random.default_rng(R + T + d_up)
    wf_up = rng.standard_normal((2, R, R)) / np.sqrt(2 * R)
    wr = rng.standard_normal((R, R)) / np.sqrt(R)
    ws = rng.standard_normal((R, S)) / np.sqrt(S)
    flat = torch.cat([dev(wf_up).flatten(), dev(wr).flatten(), dev(ws).flatten()])
    pk = K.Packer(DEV)
    oc = P.pack_conv_T(pk, 0, 2, R)
    orr = P.pack_linear_T(pk, 2 * R * R, R, R, R, perm=True)
    osk = P.pack_linear_T(pk, 3 * R * R, R, S, R)
    pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV); pk.gather(flat, buf)
    es = buf.element_size(); base = buf.data_ptr()
    g_in = dev(rng.standard_normal((B, T, R)), dt)
    df_up = dev(rng.standard_normal((B, T, R)), dt)
    dtotal = dev(rng.standard_normal((B, T, S)), dt)
    z = dev(np.tanh(rng.standard_normal((B, T, R))), dt)
    g_out = torch.full((B, T, R), float("nan"), dtype=dt, device=DEV); df_out = torch.full_like(g_out, float("nan"))
    K.residual_layer_bwd(g_in, df_up, base + oc * es, g_out, base + orr * es, base + osk * es, dtotal, z, df_out,
                         B, T, R, S, 2, d_up, True, True, dt)
    # oracle: data gradient of the conv + residual path, then dc/df
    q = lambda t: t.double().cpu().numpy()
    wq = lambda w: dev(w, dt).double().cpu().numpy()
    dxc, _ = O._conv_backward(np.zeros((B, T, R)), wq(wf_up), d_up, q(df_up))
    G = q(g_in) * O.SQRT_HALF + dxc
    assert rel_err(q(g_out), G) < TOL[dt]
    Gq = q(g_out) if dt == torch.bfloat16 else G  # the kernel feeds the rounded tile forward only via registers
    dc = (G * O.SQRT_HALF) @ wq(wr).T + q(dtotal) @ wq(ws).T
    df = dc * _gate_grad(q(z))
    assert rel_err(q(df_out), df) < TOL[dt]
    # same through the precomputed skip term (srwn_skip_dgrad_all), two "layers" sharing dtotal
    if (R, S) in ((64, 256), (32, 128)):
        ws2 = rng.standard_normal((R, S)) / np.sqrt(S)
        flat2 = torch.cat([dev(ws).flatten(), dev(ws2).flatten()])
        pk2 = K.Packer(DEV)
        o_all = pk2.reserve(2 * (R // 32), S // 16)
        per = (R // 32) * (S // 16) * 512
        P.fill_linear_T(pk2, o_all, 0, R, S, R // 32, S // 16)
        P.fill_linear_T(pk2, o_all + per, R * S, R, S, R // 32, S // 16)
        pk2.finalize()
        buf2 = torch.empty(pk2.total, dtype=dt, device=DEV); pk2.gather(flat2, buf2)
        dcs = torch.full((2, B * T, R), float("nan"), dtype=dt, device=DEV)
        K.skip_dgrad_all(dtotal, buf2.data_ptr() + o_all * es, dcs, R, S)
        assert rel_err(q(dcs[0]).reshape(B, T, R), q(dtotal) @ wq(ws).T) < TOL[dt]
        assert rel_err(q(dcs[1]).reshape(B, T, R), q(dtotal) @ wq(ws2).T) < TOL[dt]
        g3 = torch.full_like(g_out, float("nan")); df3 = torch.full_like(g_out, float("nan"))
        K.residual_layer_bwd(g_in, df_up, base + oc * es, g3, base + orr * es, None, None, z, df3, B, T, R, S, 2, d_up,
                             True, True, dt, dcs=dcs[0].view(B, T, R))
        assert rel_err(q(g3), G) < TOL[dt]
        dc3 = (G * O.SQRT_HALF) @ wq(wr).T + q(dcs[0]).reshape(B, T, R)
        assert rel_err(q(df3), dc3 * _gate_grad(q(z))) < TOL[dt]
    # DOWN only (top layer) and UP only (below layer 0)
    df2 = torch.full_like(df_out, float("nan"))
    K.residual_layer_bwd(None, None, None, None, None, base + osk * es, dtotal, z, df2, B, T, R, S, 2, 1, False, True, dt)
    assert rel_err(q(df2), (q(dtotal) @ wq(ws).T) * _gate_grad(q(z))) < TOL[dt]
    g2 = torch.full_like(g_out, float("nan"))
    K.residual_layer_bwd(None, df_up, base + oc * es, g2, None, None, None, None, None, B, T, R, S, 2, d_up, True, False, dt)
    assert rel_err(q(g2), dxc) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,B,T,shift,gate", [(64, 64, 2, 300, 4, False), (64, 256, 2, 150, 0, True),
                                                       (256, 256, 1, 200, 0, False), (32, 32, 3, 70, 1, False),
                                                       (64, 32, 1, 5000, 512, True), (40, 64, 2, 64, 0, False)])
def test_wgrad(dt, cin, cout, B, T, shift, gate):
    K = sub("kernels")
    rows = B * T
    rng = np.random.default_rng(cin + cout + T)
    L = 3
    x = dev(np.tanh(rng.standard_normal((L, B, T, cin))), dt)
    dy = dev(rng.standard_normal((2 if cout != 256 else L, B, T, cout)), dt)
    shared = dy.shape[0] != L
    shifts = [shift, 0, 2 * shift]
    ns = K.wgrad_slabs(rows)
    parts = torch.full((L * ns * cin * cout,), float("nan"), dtype=torch.float32, device=DEV)
    bparts = torch.full((L * ns * cout,), float("nan"), dtype=torch.float32, device=DEV)
    es = x.element_size()
    K.wgrad(x.data_ptr(), rows * cin, cin, dy.data_ptr(), 0 if shared else rows * cout, cout, shifts, L, parts, bparts,
            rows, T, ns, dt, pro=K.PRO_GATE if gate else K.PRO_NONE)
    out = torch.empty((L, cin, cout), dtype=torch.float32, device=DEV)
    K.reduce_partials(parts, ns, cin * cout, L, True, 0.5, out.data_ptr(), cin * cout)
    bout = torch.empty((L, cout), dtype=torch.float32, device=DEV)
    K.reduce_partials(bparts, ns, cout, L, True, 1.0, bout.data_ptr(), cout)
    xs = x.double().cpu().numpy(); ds = dy.double().cpu().numpy()
    if gate:
        xs = xs / (1 + np.exp(-xs)) if False else xs * (1 / (1 + np.exp(-xs)))
        if dt == torch.bfloat16:
            xs = dev(xs, dt).double().cpu().numpy()
    for l in range(L):
        d = ds[0 if shared else l]
        s = shifts[l]
        xsh = np.zeros_like(xs[l]); 
        if s < T:
            xsh[:, s:, :] = xs[l][:, :T - s, :]
        ref = 0.5 * np.einsum("bti,bto->io", xsh, d)
        assert rel_err(out[l].cpu().numpy(), ref) < TOL[dt], l
        assert rel_err(bout[l].cpu().numpy(), d.sum((0, 1))) < TOL[dt], l


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_wgrad_cond(dt):
    K = sub("kernels")
    B, T, cin, cout, pool, shift = 2, 128, 64, 64, 16, 8
    rows = B * T
    rng = np.random.default_rng(0)
    x = dev(rng.standard_normal((1, B, T, cin)), dt); dy = dev(rng.standard_normal((1, B, T, cout)), dt)
    cb = dev(rng.standard_normal((1, B, T // pool, cin)), dt)
    ns = K.wgrad_slabs(rows)
    parts = torch.empty(ns * cin * cout, dtype=torch.float32, device=DEV)
    K.wgrad(x.data_ptr(), 0, cin, dy.data_ptr(), 0, cout, [shift], 1, parts, None, rows, T, ns, dt,
            cond_ptr=cb.data_ptr(), cond_frames=T // pool, pool_stride=pool)
    out = torch.empty((cin, cout), dtype=torch.float32, device=DEV)
    K.reduce_partials(parts, ns, cin * cout, 1, True, 1.0, out.data_ptr(), 0)
    xin = x[0].double().cpu().numpy() + np.repeat(cb[0].double().cpu().numpy(), pool, axis=1)
    if dt == torch.bfloat16:
        xin = dev(xin, dt).double().cpu().numpy()
    xsh = np.zeros_like(xin); xsh[:, shift:, :] = xin[:, :T - shift, :]
    ref = np.einsum("bti,bto->io", xsh, dy[0].double().cpu().numpy())
    assert rel_err(out.cpu().numpy(), ref) < TOL[dt]


def test_frame_sum_and_init_conv_wgrad():
    K = sub("kernels")
    rng = np.random.default_rng(1)
    B, T, R, pool = 2, 96, 64, 16
    g = dev(rng.standard_normal((B, T, R)))
    fs = K.frame_sum(g, T // pool, pool).cpu().numpy()
    assert rel_err(fs, g.cpu().numpy().reshape(B, T // pool, pool, R).sum(2)) < 1e-5
    audio = dev(rng.uniform(-1, 1, (B, T)))
    for shift in (0, 1):
        gw = torch.empty(2 * R, dtype=torch.float32, device=DEV); gb = torch.empty(R, dtype=torch.float32, device=DEV)
        ws = torch.empty(int(sub("_lib").load().srwn_init_conv_wgrad_partials(B, T, R, 2)), dtype=torch.float32, device=DEV)
        K.init_conv_wgrad(audio, g, gw, gb, 2, shift, ws)
        x0 = audio.double().cpu().numpy()[:, :, None]
        if shift:
            x0 = O.right_shift(x0)
        _, dw = O._conv_backward(x0, np.zeros((2, 1, R)), 1, g.double().cpu().numpy())
        assert rel_err(gw.cpu().numpy().reshape(2, 1, R), dw) < 1e-5
        assert rel_err(gb.cpu().numpy(), g.double().cpu().numpy().sum((0, 1))) < 1e-5


def test_adam_matches_tf_formula():
    K = sub("kernels")
    rng = np.random.default_rng(0)
    n = 1000
    th = rng.standard_normal(n); m = np.zeros(n); v = np.zeros(n)
    p = dev(th); pm = torch.zeros(n, device=DEV); pv = torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for t in range(1, 6):
        g = rng.standard_normal(n) * 0.1
        th, m, v = O.adam_step_tf(th, 0.5 * g, m, v, t, lr=1e-2)
        K.adam_step(p, dev(g), pm, pv, step, 1e-2, grad_scale=0.5)
    assert int(step.item()) == 5
    assert rel_err(p.cpu().numpy(), th) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode,rows", [("stack", 5000), ("stack", 331), ("flat", 2100)])
def test_wgrad256(dt, mode, rows):
    """All skip kernels at once (stack of z, gate prologue) and the 256x256 head kernels."""
    K = sub("kernels")
    rng = np.random.default_rng(rows)
    d = dev(rng.standard_normal((rows, 256)), dt)
    if mode == "stack":
        L = 6
        a = dev(np.tanh(rng.standard_normal((L, rows, 64))), dt)
        ns = K.wgrad256_slabs(rows, L)
        parts = torch.full((ns * L * 64 * 256,), float("nan"), dtype=torch.float32, device=DEV)
        bparts = torch.full((ns * 256,), float("nan"), dtype=torch.float32, device=DEV)
        K.wgrad256(a.data_ptr(), rows * 64, 64, L, d, parts, bparts, rows, ns, pro=K.PRO_GATE)
        out = torch.empty((L * 64, 256), dtype=torch.float32, device=DEV)
        K.reduce_partials(parts, ns, L * 64 * 256, 1, True, 1.0, out.data_ptr(), 0)
        az = a.double().cpu().numpy()
        c = az * (1 / (1 + np.exp(-az)))
        if dt == torch.bfloat16:
            c = dev(c, dt).double().cpu().numpy()
        ref = np.einsum("lrm,rn->lmn", c, d.double().cpu().numpy()).reshape(L * 64, 256)
    else:
        a = dev(rng.standard_normal((rows, 256)), dt)
        ns = K.wgrad256_slabs(rows, 4)
        parts = torch.full((ns * 256 * 256,), float("nan"), dtype=torch.float32, device=DEV)
        bparts = torch.full((ns * 256,), float("nan"), dtype=torch.float32, device=DEV)
        K.wgrad256(a.data_ptr(), 64, 256, 4, d, parts, bparts, rows, ns)
        out = torch.empty((256, 256), dtype=torch.float32, device=DEV)
        K.reduce_partials(parts, ns, 256 * 256, 1, True, 1.0, out.data_ptr(), 0)
        ref = a.double().cpu().numpy().T @ d.double().cpu().numpy()
    bout = torch.empty(256, dtype=torch.float32, device=DEV)
    K.reduce_partials(bparts, ns, 256, 1, True, 1.0, bout.data_ptr(), 0)
    assert rel_err(out.cpu().numpy(), ref) < TOL[dt]
    assert rel_err(bout.cpu().numpy(), d.double().cpu().numpy().sum(0)) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_cond", [False, True])
@pytest.mark.parametrize("R", [64, 32])
@pytest.mark.parametrize("T,dil", [(352, [1, 16, 300]), (1120, [256, 512, 1]), (640, [512, 2, 700])])
def test_wgrad_layers_fused(dt, with_cond, R, T, dil):
    """conv taps + 1x1 residual gradients of several layers in one pass (srwn_wgrad_layers); dilations up to the
    benchmark's 512, and one beyond the clip (every delayed tap zero)."""
    K = sub("kernels")
    L, B, pool = 3, 2, 32
    rows = B * T
    rng = np.random.default_rng(7)
    x = dev(rng.standard_normal((L, rows, R)), dt); z = dev(np.tanh(rng.standard_normal((L, rows, R))), dt)
    df = dev(rng.standard_normal((L, rows, R)), dt); g = dev(rng.standard_normal((L, rows, R)), dt)
    cond = dev(rng.standard_normal((B * (T // pool), L * R)), dt) if with_cond else None
    ns = K.wgrad_slabs(rows)
    pf = torch.full((L * ns * 2 * R * R,), float("nan"), dtype=torch.float32, device=DEV)
    pr = torch.full((L * ns * R * R,), float("nan"), dtype=torch.float32, device=DEV)
    pbf = torch.full((L * ns * R,), float("nan"), dtype=torch.float32, device=DEV); pbr = torch.full_like(pbf, float("nan"))
    ckw = dict(cond_ptr=cond.data_ptr(), cond_layer_stride=R, cond_frames=T // pool, pool_stride=pool,
               cond_row_stride=L * R) if with_cond else {}
    K.wgrad_layers(x, z, df, g.data_ptr(), dil, pf, pr, pbf, pbr, T, ns, **ckw)
    of = torch.empty((L, 2, R, R), dtype=torch.float32, device=DEV); orr = torch.empty((L, R, R), dtype=torch.float32, device=DEV)
    obf = torch.empty((L, R), dtype=torch.float32, device=DEV); obr = torch.empty((L, R), dtype=torch.float32, device=DEV)
    K.reduce_partials(pf, ns, 2 * R * R, L, True, 1.0, of.data_ptr(), 2 * R * R)
    K.reduce_partials(pr, ns, R * R, L, True, 1.0, orr.data_ptr(), R * R)
    K.reduce_partials(pbf, ns, R, L, True, 1.0, obf.data_ptr(), R); K.reduce_partials(pbr, ns, R, L, True, 1.0, obr.data_ptr(), R)
    q = lambda t: t.double().cpu().numpy()
    for l in range(L):
        xin = q(x[l]).reshape(B, T, R)
        if with_cond:
            cb = q(cond).reshape(B, T // pool, L, R)[:, :, l, :]
            xin = xin + np.repeat(cb, pool, axis=1)
            if dt == torch.bfloat16:
                xin = dev(xin, dt).double().cpu().numpy()
        dfl = q(df[l]).reshape(B, T, R); gl = q(g[l]).reshape(B, T, R)
        zl = q(z[l]).reshape(B, T, R); c = zl / (1 + np.exp(-zl))
        if dt == torch.bfloat16:
            c = dev(c, dt).double().cpu().numpy()
        _, dw = O._conv_backward(xin, np.zeros((2, R, R)), dil[l], dfl)
        assert rel_err(of[l].cpu().numpy(), dw) < TOL[dt], l
        assert rel_err(orr[l].cpu().numpy(), np.einsum("btn,btm->nm", c, gl)) < TOL[dt], l
        assert rel_err(obf[l].cpu().numpy(), dfl.sum((0, 1))) < TOL[dt] and rel_err(obr[l].cpu().numpy(), gl.sum((0, 1))) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,S,L,rows", [(32, 128, 30, 3000), (32, 128, 6, 331), (64, 128, 5, 700), (32, 256, 10, 1500)])
def test_wgrad_wide_reference_widths(dt, R, S, L, rows):
    """The wide weight-gradient GEMM at the widths the reference's scripts use (dilation_channels=32,
    skip_channels=128): A in chunks of R channels, D of S columns."""
    K = sub("kernels")
    rng = np.random.default_rng(rows + R + S)
    d = dev(rng.standard_normal((rows, S)), dt)
    a = dev(np.tanh(rng.standard_normal((L, rows, R))), dt)
    ns = K.wgrad256_slabs(rows, L, R)
    parts = torch.full((ns * L * R * S,), float("nan"), dtype=torch.float32, device=DEV)
    bparts = torch.full((ns * S,), float("nan"), dtype=torch.float32, device=DEV)
    K.wgrad256(a.data_ptr(), rows * R, R, L, d, parts, bparts, rows, ns, pro=K.PRO_GATE, chunk_width=R)
    out = torch.empty((L * R, S), dtype=torch.float32, device=DEV)
    K.reduce_partials(parts, ns, L * R * S, 1, True, 1.0, out.data_ptr(), 0)
    az = a.double().cpu().numpy()
    c = az * (1 / (1 + np.exp(-az)))
    if dt == torch.bfloat16:
        c = dev(c, dt).double().cpu().numpy()
    ref = np.einsum("lrm,rn->lmn", c, d.double().cpu().numpy()).reshape(L * R, S)
    bout = torch.empty(S, dtype=torch.float32, device=DEV)
    K.reduce_partials(bparts, ns, S, 1, True, 1.0, bout.data_ptr(), 0)
    assert rel_err(out.cpu().numpy(), ref) < TOL[dt]
    assert rel_err(bout.cpu().numpy(), d.double().cpu().numpy().sum(0)) < TOL[dt]
    # a [rows, S] tensor against a [rows, 128] one (the head 1x1 of a 128-channel skip path)
    if S == 128:
        x = dev(rng.standard_normal((rows, 128)), dt)
        ns2 = K.wgrad256_slabs(rows, 2)
        p2 = torch.full((ns2 * 128 * 128,), float("nan"), dtype=torch.float32, device=DEV)
        K.wgrad256(x.data_ptr(), 64, 128, 2, d, p2, None, rows, ns2)
        o2 = torch.empty((128, 128), dtype=torch.float32, device=DEV)
        K.reduce_partials(p2, ns2, 128 * 128, 1, True, 1.0, o2.data_ptr(), 0)
        assert rel_err(o2.cpu().numpy(), x.double().cpu().numpy().T @ d.double().cpu().numpy()) < TOL[dt]


def test_reduce_partials_multi_matches_single_launches():
    """Several partial buffers finished by one launch: every job bit-identical to its own srwn_reduce_partials call
    (both kernel shapes: few outputs x many slabs, many outputs x few slabs; batched and shared partials)."""
    Kn = sub("kernels")
    g = torch.Generator(device="cpu").manual_seed(3)
    specs = [  # nslabs, n, nbatch, partials_batched, scale, out_batch_stride
        (42, 2 * 64 * 64, 5, True, 1.0, 2 * 64 * 64),
        (42, 64, 5, True, 0.70710678, 64),
        (32, 30 * 64 * 256, 1, True, 1.0, 0),
        (32, 256, 30, False, 1.0, 256),
        (64, 256 * 256, 1, True, 1.0, 0),
        (7, 100, 3, True, 2.0, 128),
    ]
    jobs, singles = [], []
    for ns, n, nb, batched, scale, stride in specs:
        parts = torch.randn((nb if batched else 1) * ns * n, generator=g).to(DEV)
        out_a = torch.zeros(max(stride, n) * nb, device=DEV)
        out_b = torch.zeros_like(out_a)
        Kn.reduce_partials(parts, ns, n, nb, batched, scale, out_a.data_ptr(), stride)
        jobs.append((parts, ns, n, nb, batched, scale, out_b.data_ptr(), stride))
        singles.append((out_a, out_b))
    Kn.reduce_partials_multi(jobs)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(singles):
        assert torch.equal(a, b), "job %d" % k
        assert float(a.abs().max()) > 0
    with pytest.raises(RuntimeError):
        Kn.reduce_partials_multi(jobs * 3)       # 18 jobs: more than one launch takes


def test_adam_step_counter_ticks_inside_the_update_launch():
    """The device step counter is advanced by the update launch itself (the block whose arrival comes last stores the new
    count; the upper half of the 64-bit word is the arrival counter and is zero again afterwards): a grid of ~1000
    blocks, several steps, eager and replayed from a hipGraph -- every block must have used the same t, the word must
    read as a plain int64 count between launches."""
    K = sub("kernels")
    rng = np.random.default_rng(1)
    n = 1_003_457          # ~980 blocks, a ragged tail
    th = rng.standard_normal(n); m = np.zeros(n); v = np.zeros(n)
    p = dev(th); pm = torch.zeros(n, device=DEV); pv = torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    g = dev(rng.standard_normal(n) * 0.1)
    gh = g.double().cpu().numpy()
    for t in range(1, 4):
        th, m, v = O.adam_step_tf(th, gh, m, v, t, lr=1e-2)
        K.adam_step(p, g, pm, pv, step, 1e-2)
        assert int(step.item()) == t
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        K.adam_step(p, g, pm, pv, step, 1e-2)
    for t in range(4, 8):          # (the capture itself does not run the kernel)
        th, m, v = O.adam_step_tf(th, gh, m, v, t, lr=1e-2)
        gr.replay()
    torch.cuda.synchronize()
    assert int(step.item()) == 7
    assert rel_err(p.cpu().numpy(), th) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pack_gather_with_column_sums(dt):
    """srwn_pack_gather_rowsum: the image gather and, by extra blocks of the same launch, the column sums of a matrix (the
    sum of the layers' skip biases, model.py:50) -- bit-equal to the plain gather and to srwn_reduce_partials."""
    K = sub("kernels"); P = sub("packing")
    rng = np.random.default_rng(7)
    L, R, S = 30, 64, 256
    params = dev(rng.standard_normal(L * R * S + L * S))
    pk = K.Packer(DEV)
    o = pk.reserve(S // 32, L * R // 16)
    for l in range(L):
        P.fill_linear(pk, o, l * R * S, R, S, S // 32, L * R // 16, ks_offset=l * R // 16, ks_count=R // 16)
    pk.finalize()
    plain = torch.zeros(pk.total, dtype=dt, device=DEV)
    pk.gather(params, plain)
    both = torch.zeros(pk.total, dtype=dt, device=DEV)
    bias = params[L * R * S:].view(L, S)
    got = torch.full((S,), float("nan"), dtype=torch.float32, device=DEV)
    pk.gather(params, both, rowsum=(bias, got))
    want = torch.empty(S, dtype=torch.float32, device=DEV)
    K.reduce_partials(bias.reshape(-1), L, S, 1, True, 1.0, want.data_ptr(), 0)
    assert torch.equal(both, plain) and torch.equal(got, want)
    # a prefix of the image (what the training step re-gathers) + the tail (what generate() re-gathers) == the whole
    parts = torch.zeros(pk.total, dtype=dt, device=DEV)
    cut = (pk.total // 3) // 8 * 8
    pk.gather(params, parts, 0, cut)
    assert bool((parts[cut:] == 0).all())
    pk.gather(params, parts, cut, None)
    assert torch.equal(parts, plain)
