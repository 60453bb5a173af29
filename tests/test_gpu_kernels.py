"""GPU parity tests of the individual HIP kernels against the CPU oracle, through the C-ABI.

Tolerances: fp32 mode 1e-3 relative (north_star; measured error is ~1e-6), bf16 mode 3e-2 of the
tensor scale (bf16 has 8 significant bits; operands are rounded once, accumulation is fp32).
mu-law integer codes must be bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 1e-3, torch.bfloat16: 3e-2}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def dev(a, dt=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()


def test_mu_law_bit_exact(golden_dir):
    K = sub("kernels")
    g = json.load(open(os.path.join(golden_dir, "mu_law.json")))
    codes = K.mu_law_encode(dev(np.array(g["audio"], np.float32)), 256).cpu().numpy()
    assert codes.tolist() == g["codes"]
    dec = K.mu_law_decode(torch.arange(256, dtype=torch.int32, device=DEV), 256).cpu().numpy()
    assert dec.view(np.uint32).tolist() == g["decode_all_bits"]
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-1.2, 1.2, 200000), rng.normal(0, 0.01, 100000), [0.0, -0.0, 1e-9, -1e-9]])
    a = a.astype(np.float32)
    for Q in (256, 32, 2):
        got = K.mu_law_encode(dev(a), Q).cpu().numpy()
        assert np.array_equal(got, O.mu_law_encode(a, Q)), Q
    c = rng.integers(0, 256, 50000).astype(np.int32)
    got = K.mu_law_decode(dev(c, torch.int32), 256).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.mu_law_decode(c, 256).view(np.uint32))
    assert K.mu_law_encode(torch.empty(0, device=DEV), 256).numel() == 0
    # round trip on every code
    rt = K.mu_law_encode(K.mu_law_decode(torch.arange(256, dtype=torch.int32, device=DEV), 256), 256)
    assert rt.cpu().tolist() == list(range(256))


def test_causal_conv_selfcheck_vectors(golden_dir):
    """The reference's own fixed inputs (ops.py:243-252) through the HIP conv."""
    K = sub("kernels")
    g = json.load(open(os.path.join(golden_dir, "ops_selfcheck.json")))
    x = dev(np.array(g["x"], np.float32).reshape(1, -1, 1))
    for c in g["causal"]:
        w = dev(np.array(c["filt"], np.float32).reshape(c["shape"]))
        y = K.causal_conv1d_fwd(x, w, None, dilation=c["d"]).cpu().numpy()
        assert np.array_equal(y[0].T, np.array(c["out"], np.float32)), c["ref"]


@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_causal_conv_random(odt):
    K = sub("kernels")
    rng = np.random.default_rng(1)
    for (B, T, Cin, Cout, Kw, d, shift) in [(2, 50, 1, 64, 2, 1, 1), (1, 33, 3, 10, 3, 4, 0), (2, 7, 1, 32, 2, 1, 0)]:
        x = rng.standard_normal((B, T, Cin)).astype(np.float32)
        w = rng.standard_normal((Kw, Cin, Cout)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        xs = O.right_shift(x, shift) if shift else x
        ref = O.dilated_causal_conv1d_bias(xs.astype(np.float64), w.astype(np.float64), b.astype(np.float64), d)
        y = K.causal_conv1d_fwd(dev(x), dev(w), dev(b), d, shift, odt).float().cpu().numpy()
        assert rel_err(y, ref) < (1e-5 if odt == torch.float32 else 1e-2)


def _layer_setup(seed, R, S, d, E=0):
    sp = O.init_stack_params(seed, [d], 2, R, S, 8, cond_channels=E, bias_scale=0.2)
    return sp, sp.layers[0]


def _pack_layer(K, l, R, dt):
    P = sub("packing")
    flat = torch.cat([dev(l.wf).flatten(), dev(l.wr).flatten()])
    pk = K.Packer(DEV)
    oc = P.pack_conv(pk, 0, 2, R)
    orr = P.pack_res(pk, 2 * R * R, R)
    pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV)
    pk.gather(flat, buf)
    es = buf.element_size()
    return buf, buf.data_ptr() + oc * es, buf.data_ptr() + orr * es


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R", [32, 64])
@pytest.mark.parametrize("B,T,d", [(2, 64, 1), (1, 100, 4), (2, 300, 32), (1, 40, 64), (1, 513, 512)])
def test_residual_layer_fwd(dt, R, B, T, d):
    K = sub("kernels")
    sp, l = _layer_setup(10 + d, R, 16, d)
    rng = np.random.default_rng(d)
    x = rng.standard_normal((B, T, R))
    xq = dev(x, dt)
    dense, _, cache = O.residual_dilation_layer(xq.double().cpu().numpy(), l, d)
    buf, pc, pr = _pack_layer(K, l, R, dt)
    h = torch.full((B, T, R), float("nan"), dtype=dt, device=DEV); z = torch.full_like(h, float("nan"))
    K.residual_layer_fwd(xq, None, pc, pr, dev(l.bf), dev(l.br), h, z, 2, d)
    torch.cuda.synchronize()
    assert rel_err(z.float().cpu().numpy(), cache["z"]) < TOL[dt]
    assert rel_err(h.float().cpu().numpy(), dense) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_residual_layer_fwd_cond(dt):
    K = sub("kernels")
    R, B, T, d, pool = 64, 2, 96, 8, 16
    sp, l = _layer_setup(5, R, 16, d)
    rng = np.random.default_rng(2)
    xq = dev(rng.standard_normal((B, T, R)), dt)
    cbq = dev(rng.standard_normal((B, T // pool, R)), dt)
    # the kernel takes the layer's complete input and adds the NEXT layer's conditioning bias (model.py:181-183) to
    # the row it stores; srwn_add_frame_bias puts the first layer's bias onto the input conv's output
    dense, _, cache = O.residual_dilation_layer(xq.double().cpu().numpy(), l, d)
    buf, pc, pr = _pack_layer(K, l, R, dt)
    h = torch.empty((B, T, R), dtype=dt, device=DEV); z = torch.empty_like(h)
    K.residual_layer_fwd(xq, cbq, pc, pr, dev(l.bf), dev(l.br), h, z, 2, d, pool)
    up = np.repeat(cbq.double().cpu().numpy(), pool, axis=1)
    assert rel_err(z.float().cpu().numpy(), cache["z"]) < TOL[dt]
    assert rel_err(h.float().cpu().numpy(), dense + up) < TOL[dt]
    x2 = xq.clone()
    sub("_lib").call("srwn_add_frame_bias", x2.data_ptr(), cbq.data_ptr(), R, B, T, R, T // pool, pool, K.abi_dtype(dt),
                     torch.cuda.current_stream().cuda_stream)
    want = torch.tensor(xq.double().cpu().numpy() + up).to(dt).double().numpy()
    assert np.array_equal(x2.double().cpu().numpy(), want)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,rows", [(64, 256, 200), (256, 256, 70), (256, 32, 129), (32, 64, 64), (64, 96, 31),
                                           (128, 128, 300), (256, 128, 129), (64, 128, 1)])
def test_pw_linear_relu(dt, Cin, Cout, rows):
    K = sub("kernels"); P = sub("packing")
    rng = np.random.default_rng(Cin + Cout)
    x = dev(rng.standard_normal((rows, Cin)), dt)
    w = rng.standard_normal((Cin, Cout)) / np.sqrt(Cin); b = rng.standard_normal(Cout)
    flat = dev(w).flatten()
    pk = K.Packer(DEV); off = P.pack_linear(pk, 0, Cin, Cout, Cout); pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV); pk.gather(flat, buf)
    wq = buf  # operands rounded to dt
    y = torch.full((rows, Cout), float("nan"), dtype=dt, device=DEV)
    K.pw_linear(x.data_ptr(), Cin, 0, Cin, Cin, buf.data_ptr() + off * buf.element_size(), dev(b), y, Cout, Cout,
                rows, epi=K.EPI_RELU)
    wr = dev(w, dt).double().cpu().numpy()
    ref = np.maximum(x.double().cpu().numpy() @ wr + b, 0)
    assert rel_err(y.float().cpu().numpy(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pw_linear_skip_sum_gate_chunks(dt):
    """The sum of all layers' skip 1x1s as one contraction over a [L, rows, R] stack of z."""
    K = sub("kernels"); P = sub("packing")
    L, rows, R, S = 5, 150, 64, 256
    rng = np.random.default_rng(3)
    z = dev(np.tanh(rng.standard_normal((L, rows, R))), dt)
    ws = rng.standard_normal((L, R, S)) / 8; bs = rng.standard_normal((L, S))
    flat = dev(ws).flatten()
    pk = K.Packer(DEV)
    off = pk.reserve(S // 32, L * R // 16)
    for l in range(L):
        P.fill_linear(pk, off, l * R * S, R, S, S // 32, L * R // 16, ks_offset=l * R // 16, ks_count=R // 16)
    pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV); pk.gather(flat, buf)
    y = torch.empty((rows, S), dtype=dt, device=DEV)
    K.pw_linear(z.data_ptr(), R, rows * R, R, L * R, buf.data_ptr() + off * buf.element_size(), dev(bs.sum(0)), y, S,
                S, rows, pro=K.PRO_GATE, epi=K.EPI_RELU)
    zz = z.double().cpu().numpy()
    c = zz * (1 / (1 + np.exp(-zz)))
    if dt == torch.bfloat16:
        c = dev(c, dt).double().cpu().numpy()
    wq = dev(ws, dt).double().cpu().numpy()
    ref = np.maximum(np.einsum("lrn,lns->rs", c, wq) + bs.sum(0), 0)
    assert rel_err(y.float().cpu().numpy(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pw_linear_mask(dt):
    K = sub("kernels"); P = sub("packing")
    rows, Cin, Cout = 100, 256, 64
    rng = np.random.default_rng(4)
    x = dev(rng.standard_normal((rows, Cin)), dt); aux = dev(rng.standard_normal((rows, Cout)), dt)
    w = rng.standard_normal((Cout, Cin)) / 16  # forward weight [Cout(in of dgrad... )]; dgrad uses W^T
    flat = dev(w).flatten()
    pk = K.Packer(DEV); off = P.pack_linear_T(pk, 0, Cout, Cin, Cout); pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV); pk.gather(flat, buf)
    y = torch.empty((rows, Cout), dtype=dt, device=DEV)
    K.pw_linear(x.data_ptr(), Cin, 0, Cin, Cin, buf.data_ptr() + off * buf.element_size(), None, y, Cout, Cout, rows,
                aux=aux, epi=K.EPI_MASK)
    ref = (x.double().cpu().numpy() @ dev(w, dt).double().cpu().numpy().T) * (aux.float().cpu().numpy() > 0)
    assert rel_err(y.float().cpu().numpy(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,rows", [(256, 100), (32, 64), (30, 45), (100, 33)])
def test_head_softmax_ce(dt, C, rows):
    K = sub("kernels"); P = sub("packing")
    S = 64
    Cp = (C + 31) // 32 * 32
    rng = np.random.default_rng(C)
    x = dev(rng.standard_normal((rows, S)), dt)
    w = rng.standard_normal((S, C)) / 4; b = rng.standard_normal(C)
    tg = rng.integers(0, C, rows).astype(np.int32)
    pk = K.Packer(DEV); off = P.pack_linear(pk, 0, S, C, Cp); pk.finalize()
    buf = torch.empty(pk.total, dtype=dt, device=DEV); pk.gather(dev(w).flatten(), buf)
    parts = torch.zeros((rows + 31) // 32, dtype=torch.float32, device=DEV)
    dl = torch.full((rows, Cp), float("nan"), dtype=dt, device=DEV)
    lo = torch.empty((rows, C), dtype=torch.float32, device=DEV)
    scale = 1.0 / rows
    K.head_softmax_ce(x, buf.data_ptr() + off * buf.element_size(), dev(b), dev(tg, torch.int32), parts, dl, lo, Cp, C,
                      scale)
    loss = torch.empty(1, dtype=torch.float32, device=DEV)
    K.reduce_loss(parts, parts.numel(), scale, loss)
    logits = x.double().cpu().numpy() @ dev(w, dt).double().cpu().numpy() + b
    assert rel_err(lo.cpu().numpy(), logits) < TOL[dt]
    ref_loss = O.softmax_ce_per_timestep(logits[None], tg[None])
    assert abs(float(loss.item()) - ref_loss) < TOL[dt] * max(1.0, abs(ref_loss))
    ref_d = O.dlogits_per_timestep(logits[None], tg[None])[0]
    got = dl.float().cpu().numpy()
    assert np.all(got[:, C:] == 0)
    assert rel_err(got[:, :C], ref_d) < TOL[dt]
