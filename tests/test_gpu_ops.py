"""Every free function of the reference's ops.py through the `ops` seam (sr-wavenet_amd/ops.py, dropin/ops.py) against
the CPU oracle: the names `from ops import *` gives model.py (model.py:6) all resolve to HIP-backed code."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import rel_err

pytestmark = pytest.mark.gpu

REFERENCE_NAMES = ["_DilatedCausalConv1d", "DilatedCausalConv1d", "ResidualDilationLayer", "ResidualDilationLayerNC",
                   "ResizeEmbeddingNearestNeighbor", "RightShift", "mu_law_encode", "mu_law_decode",
                   "categorical_sample", "log_prob_from_logits", "log_sum_exp", "discretized_mix_logistic_loss",
                   "sample_from_discretized_mix_logistic", "probs_logistic"]     # def statements of ops.py:6-214


def test_every_ops_name_is_exported():
    ops = sub("ops")
    d = os.path.join(os.path.dirname(sub("model").__file__), "dropin")
    sys.path.insert(0, d)
    try:
        import importlib
        shim = importlib.import_module("ops")
        for n in REFERENCE_NAMES:
            assert callable(getattr(ops, n)), n
            assert getattr(shim, n) is getattr(ops, n), n
    finally:
        sys.path.remove(d)
        sys.modules.pop("ops", None)


def _lp(V, name, K):
    f = lambda k: V[k].cpu().numpy().astype(np.float64)
    return O.LayerParams(f(name + "_filter/" + name + "_Kernel"), f(name + "_filter/" + name + "_Bias").reshape(-1),
                         None, None, f(name + "/residual/kernel")[0], f(name + "/residual/bias"),
                         f(name + "/skip/kernel")[0], f(name + "/skip/bias"))


def test_reference_self_check_calls():
    """The two layer constructions of the reference's own self-check (ops.py:231-234): a K=3 conv with 4 channels and
    an 8-channel ResidualDilationLayer, both on a 1-channel input at dilation 4 (`inputs + residual` broadcasts)."""
    ops = sub("ops")
    rng = np.random.default_rng(0)
    x = rng.random((1, 8, 1)).astype(np.float32)
    conv1 = ops.DilatedCausalConv1d(x, kernel_size=3, channels=4, dilation_rate=4, name="causal_conv1")
    V = ops.VARIABLES
    V["causal_conv1_Bias"].copy_(torch.tensor(rng.standard_normal((1, 1, 4)), dtype=torch.float32))
    conv1 = ops.DilatedCausalConv1d(x, kernel_size=3, channels=4, dilation_rate=4, name="causal_conv1")
    ref = O.dilated_causal_conv1d_bias(x.astype(np.float64), V["causal_conv1_Kernel"].cpu().numpy().astype(np.float64),
                                       V["causal_conv1_Bias"].cpu().numpy().astype(np.float64), 4)
    assert rel_err(conv1.cpu().numpy(), ref) < 1e-5
    ones = np.ones((1, 8, 1), np.float32)
    ops.ResidualDilationLayer(ones, kernel_size=2, dilation_channels=8, skip_channels=4, dilation_rate=4,
                              name="dilation_layer1")
    for k in ("dilation_layer1_filter/dilation_layer1_Bias", "dilation_layer1/residual/bias", "dilation_layer1/skip/bias"):
        V[k].copy_(torch.tensor(0.3 * rng.standard_normal(tuple(V[k].shape)), dtype=torch.float32))
    dense, skip = ops.ResidualDilationLayer(ones, kernel_size=2, dilation_channels=8, skip_channels=4, dilation_rate=4,
                                            name="dilation_layer1")
    assert tuple(dense.shape) == (1, 8, 8) and tuple(skip.shape) == (1, 8, 4)
    d_ref, s_ref, _ = O.residual_dilation_layer(ones.astype(np.float64), _lp(V, "dilation_layer1", 2), 4)
    assert rel_err(dense.cpu().numpy(), d_ref) < 1e-5 and rel_err(skip.cpu().numpy(), s_ref) < 1e-5
    assert "dilation_layer1_gate/dilation_layer1_Kernel" in V     # created, never used (ops.py:31-33)


@pytest.mark.parametrize("K,cin,R,S,d,T", [(3, 64, 64, 256, 2, 70), (2, 48, 48, 40, 5, 33), (5, 1, 16, 8, 3, 50),
                                           (2, 64, 64, 100, 1, 40)])
def test_residual_dilation_layer_generic_shapes(K, cin, R, S, d, T):
    """filter_width != 2, channel counts outside {32, 64}, skip widths that are not multiples of 32."""
    ops = sub("ops")
    rng = np.random.default_rng(K + R)
    x = rng.standard_normal((2, T, cin)).astype(np.float32)
    name = "gen_%d_%d_%d_%d" % (K, cin, R, S)
    ops.ResidualDilationLayer(x, K, R, S, dilation_rate=d, name=name)
    V = ops.VARIABLES
    for k in (name + "_filter/" + name + "_Bias", name + "/residual/bias", name + "/skip/bias"):
        V[k].copy_(torch.tensor(0.2 * rng.standard_normal(tuple(V[k].shape)), dtype=torch.float32))
    dense, skip = ops.ResidualDilationLayer(x, K, R, S, dilation_rate=d, name=name)
    d_ref, s_ref, _ = O.residual_dilation_layer(x.astype(np.float64), _lp(V, name, K), d)
    assert rel_err(dense.cpu().numpy(), d_ref) < 1e-3 and rel_err(skip.cpu().numpy(), s_ref) < 1e-3
    with pytest.raises(ValueError):
        ops.ResidualDilationLayer(rng.standard_normal((1, 8, 3)).astype(np.float32), 2, 8, 4, name="bad_bcast")


@pytest.mark.parametrize("K,cin,R,S", [(2, 1, 128, 128), (2, 128, 128, 128), (3, 16, 24, 12), (4, 8, 8, 8)])
def test_residual_dilation_layer_nc(K, cin, R, S):
    ops = sub("ops")
    rng = np.random.default_rng(K * 7 + cin)
    x = rng.standard_normal((2, 45, cin)).astype(np.float32)
    name = "nc_%d_%d_%d" % (K, cin, R)
    ops.ResidualDilationLayerNC(x, K, R, S, dilation_rate=9, name=name)
    V = ops.VARIABLES
    for k in (name + "_NC/conv1d/bias", name + "/residual_nc/bias", name + "/skip_nc/bias"):
        V[k].copy_(torch.tensor(0.2 * rng.standard_normal(tuple(V[k].shape)), dtype=torch.float32))
    res, skip = ops.ResidualDilationLayerNC(x, K, R, S, dilation_rate=9, name=name)   # the dilation is ignored (ops.py:51)
    f = lambda k: V[k].cpu().numpy().astype(np.float64)
    p = O.NCLayerParams(f(name + "_NC/conv1d/kernel"), f(name + "_NC/conv1d/bias"), f(name + "/residual_nc/kernel")[0],
                        f(name + "/residual_nc/bias"), f(name + "/skip_nc/kernel")[0], f(name + "/skip_nc/bias"))
    r_ref, s_ref, _ = O.residual_dilation_layer_nc(x.astype(np.float64), p)
    assert rel_err(res.cpu().numpy(), r_ref) < 1e-4 and rel_err(skip.cpu().numpy(), s_ref) < 1e-4


@pytest.mark.parametrize("shape", [(3, 7, 256), (5, 40), (2, 3, 4, 1000), (1, 1)])
def test_log_prob_from_logits_and_log_sum_exp(shape):
    ops = sub("ops")
    rng = np.random.default_rng(len(shape))
    x = (rng.standard_normal(shape) * 20).astype(np.float32)      # large logits: the max subtraction matters
    lp = ops.log_prob_from_logits(x).cpu().numpy()
    lse = ops.log_sum_exp(x).cpu().numpy()
    assert lp.shape == shape and lse.shape == shape[:-1]
    assert np.abs(lp - O.log_prob_from_logits(x.astype(np.float64))).max() < 1e-4
    assert np.abs(lse - O.log_sum_exp(x.astype(np.float64))).max() < 1e-4


def test_categorical_sample_follows_softmax():
    ops = sub("ops")
    rng = np.random.default_rng(3)
    logits = np.tile((rng.standard_normal((1, 12)) * 2).astype(np.float32), (40000, 1))
    s = ops.categorical_sample(logits, 12, seed=5)
    assert s.dtype == torch.int64 and tuple(s.shape) == (40000,)
    assert int(s.min()) >= 0 and int(s.max()) < 12
    p = np.exp(O.log_prob_from_logits(logits[0].astype(np.float64)))
    freq = np.bincount(s.cpu().numpy(), minlength=12) / 40000.0
    assert np.abs(freq - p).max() < 0.01
    assert torch.equal(s, ops.categorical_sample(logits, 12, seed=5))           # counter-based: reproducible per seed
    assert not torch.equal(s, ops.categorical_sample(logits, 12, seed=6))
    big = np.full((3, 5), -1e4, np.float32); big[:, 2] = 1e4                      # overflow-safe, degenerate
    assert ops.categorical_sample(big, 5).cpu().tolist() == [2, 2, 2]


@pytest.mark.parametrize("M", [1, 5, 10])
def test_discretized_mix_logistic_loss_and_sampler(M):
    ops = sub("ops")
    rng = np.random.default_rng(M)
    B, T = 3, 200
    l = rng.standard_normal((B, T, 4 * M)).astype(np.float32)
    l[..., 2 * M:3 * M] = l[..., 2 * M:3 * M] * 2 - 3           # log-scales on both sides of the -7 floor
    x = np.clip(rng.standard_normal((B, T, 1)) * 0.6, -1, 1).astype(np.float32)
    x[0, :10] = -1.0; x[0, 10:20] = 1.0                         # the two edge branches of ops.py:169
    l[1, :30, 2 * M:3 * M] = -9.0; l[1, :30, M:2 * M] = 5.0     # the cdf_delta <= 1e-5 branch
    ref_rows = -O.log_sum_exp(O.mol_log_probs(x[..., 0].astype(np.float64), l.astype(np.float64))[0])
    got_rows = ops.discretized_mix_logistic_loss(x, l, sum_all=False).cpu().numpy()
    assert got_rows.shape == (B, T, 1)
    assert np.abs(got_rows[..., 0] - ref_rows).max() < 1e-3 * max(1.0, np.abs(ref_rows).max())
    tot = float(ops.discretized_mix_logistic_loss(x, l))
    assert abs(tot - O.mol_loss(x[..., 0].astype(np.float64), l.astype(np.float64))) < 1e-4 * abs(tot)
    s = ops.sample_from_discretized_mix_logistic(l, M, seed=1)
    assert tuple(s.shape) == (B, T, 1) and float(s.min()) >= -1 and float(s.max()) <= 1
    assert torch.equal(s, ops.sample_from_discretized_mix_logistic(l, M, seed=1))
    # one sharp component: samples concentrate on its mean
    l2 = np.zeros((1, 500, 4 * M), np.float32); l2[..., 0] = 50.0; l2[..., M] = 0.25; l2[..., 2 * M:3 * M] = -7.0
    s2 = ops.sample_from_discretized_mix_logistic(l2, M, seed=2).cpu().numpy()
    assert abs(np.median(s2) - 0.25) < 2e-3


def test_probs_logistic():
    ops = sub("ops")
    rng = np.random.default_rng(9)
    scale = np.abs(rng.standard_normal((4, 100))).astype(np.float32) * 0.1
    scale[0, :5] = 0.0                                          # clipped at exp(log_scale_min)
    mu = rng.standard_normal((4, 100)).astype(np.float32) * 0.3
    y = np.clip(rng.standard_normal((4, 100)), -1, 1).astype(np.float32)
    for nc, lsm in ((256, -14), (16, -7)):
        got = ops.probs_logistic(scale, mu, y, nc, lsm).cpu().numpy()
        ref = O.probs_logistic(scale, mu, y, nc, lsm)
        assert np.abs(got - ref).max() < 2e-6
    assert tuple(ops.probs_logistic(scale[:1], mu, y[:, :1]).shape) == (4, 100)     # broadcasting like the TF ops


@pytest.mark.parametrize("K,cin,R,S,d,T", [(2, 64, 64, 256, 4, 90), (3, 32, 32, 40, 2, 50), (2, 1, 8, 4, 4, 16)])
def test_residual_dilation_layer_gate_modes(K, cin, R, S, d, T):
    """gate_mode: "reference" = the graph the reference runs (ops.py:33: the gate conv's result is overwritten,
    c = z sigmoid(z)); "wavenet" = the canonical unit ops.py:31-32 builds and discards, tanh(filter conv) * sigmoid(gate
    conv) with the `_gate` variables -- both against the oracle's restatement of the same two graphs (fp32, 1e-3).  On the
    stack's own shape the "reference" mode runs the fused MFMA kernel, "wavenet" the generic ones."""
    ops = sub("ops")
    rng = np.random.default_rng(K + R + d)
    x = rng.standard_normal((2, T, cin)).astype(np.float32)
    name = "gm_%d_%d_%d_%d" % (K, cin, R, S)
    ops.ResidualDilationLayer(x, K, R, S, dilation_rate=d, name=name)
    V = ops.VARIABLES
    for k in (name + "_filter/" + name + "_Bias", name + "_gate/" + name + "_Bias", name + "/residual/bias", name + "/skip/bias"):
        V[k].copy_(torch.tensor(0.2 * rng.standard_normal(tuple(V[k].shape)), dtype=torch.float32))
    f = lambda k: V[k].cpu().numpy().astype(np.float64)
    lp = _lp(V, name, K)
    lp.wg, lp.bg = f(name + "_gate/" + name + "_Kernel"), f(name + "_gate/" + name + "_Bias").reshape(-1)
    out = {}
    for mode in ("reference", "wavenet"):
        dense, skip = ops.ResidualDilationLayer(x, K, R, S, dilation_rate=d, name=name, gate_mode=mode)
        d_ref, s_ref, _ = O.residual_dilation_layer(x.astype(np.float64), lp, d, gate_mode=mode)
        assert rel_err(dense.cpu().numpy(), d_ref) < 1e-3 and rel_err(skip.cpu().numpy().reshape(s_ref.shape), s_ref) < 1e-3, mode
        out[mode] = dense.cpu().numpy()
    assert rel_err(out["wavenet"], out["reference"]) > 1e-2          # two different graphs
    with pytest.raises(ValueError):
        ops.ResidualDilationLayer(x, K, R, S, dilation_rate=d, name=name, gate_mode="glu")
    L = sub("_lib")
    assert L.load().srwn_gated_activation(1, None, 1, 1, 8, 1, None) == -3        # wavenet mode needs g
    assert L.load().srwn_gated_activation(1, 1, 1, 1, 8, 2, None) == -4           # unknown mode
