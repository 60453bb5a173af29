"""Host-side mirror of the reference's ``ops.py`` free functions on the HIP kernels.

Same names and argument meaning as ``/root/reference/ops.py``; tensors are torch CUDA/HIP tensors
(NumPy arrays are accepted and moved to the device), channels-last ``[B,T,C]`` with filters
``[K,Cin,Cout]`` (ops.py:4-5).  ``tf.get_variable`` is replaced by a module-level ``VARIABLES`` store
keyed by the same variable names so layers can be re-applied with shared weights.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import kernels as K
from . import packing as P

VARIABLES: Dict[str, torch.Tensor] = {}
_rng = np.random.default_rng(0)


def _dev(x, dtype=torch.float32):
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    return t.to(device="cuda", dtype=dtype).contiguous()


def get_variable(name, shape, init="xavier"):
    """tf.get_variable stand-in (ops.py:14-18): Xavier-uniform kernels, constant-0 biases."""
    if name not in VARIABLES:
        if init == "xavier":
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            v = _rng.uniform(-lim, lim, size=shape)
        else:
            v = np.zeros(shape)
        VARIABLES[name] = _dev(v)
    v = VARIABLES[name]
    if tuple(v.shape) != tuple(shape):
        raise ValueError("variable %s exists with shape %s, requested %s" % (name, tuple(v.shape), tuple(shape)))
    return v


def _DilatedCausalConv1d(inputs, filters, dilation_rate=1):
    """ops.py:6-10."""
    return K.causal_conv1d_fwd(_dev(inputs), _dev(filters), None, dilation_rate)


def DilatedCausalConv1d(inputs, kernel_size, channels, dilation_rate=1, name="", dtype=None, use_bias=True):
    """ops.py:13-20."""
    x = _dev(inputs)
    filters = get_variable(name + "_Kernel", (kernel_size, x.shape[-1], channels))
    bias = get_variable(name + "_Bias", (1, 1, channels), init="zeros").reshape(-1) if use_bias else None
    return K.causal_conv1d_fwd(x, filters, bias, dilation_rate)


def _layer_variables(name, kernel_size, cin, R, skip_channels, with_gate=False):
    wf = get_variable(name + "_filter/" + name + "_Kernel", (kernel_size, cin, R))
    bf = get_variable(name + "_filter/" + name + "_Bias", (1, 1, R), init="zeros").reshape(-1)
    wg = get_variable(name + "_gate/" + name + "_Kernel", (kernel_size, cin, R))       # dead in the reference's graph (ops.py:32-33)
    bg = get_variable(name + "_gate/" + name + "_Bias", (1, 1, R), init="zeros").reshape(-1)
    wr = get_variable(name + "/residual/kernel", (1, R, R)); br = get_variable(name + "/residual/bias", (R,), "zeros")
    ws = get_variable(name + "/skip/kernel", (1, R, skip_channels))
    bs = get_variable(name + "/skip/bias", (skip_channels,), "zeros")
    if with_gate:
        return wf, bf, wr, br, ws, bs, wg, bg
    return wf, bf, wr, br, ws, bs


GATE_MODES = {"reference": 0, "wavenet": 1}      # include/srwn.h: SRWN_GATE_REFERENCE / SRWN_GATE_WAVENET


def ResidualDilationLayer(inputs, kernel_size, dilation_channels, skip_channels, dilation_rate=1, name="",
                          dtype=None, use_bias=True, gate_mode="reference"):
    """ops.py:23-46 -> (dense, skip).

    gate_mode (no reference counterpart): "reference" (default) is the graph the reference RUNS -- ops.py:33
    overwrites the gate conv's result, so combined = z * sigmoid(z) with z = tanh(filter conv); "wavenet" is the canonical
    unit ops.py:31-32 builds and discards, tanh(filter conv) * sigmoid(gate conv) with the `_gate` variables (forward,
    fp32, on the generic kernels; the fused training kernels implement "reference" only).

    Like ops.py:33 the gate conv is created but its result is discarded (kept for checkpoint parity).
    The stack's shape (input channels == dilation_channels in {32, 64}, kernel_size 2, skip_channels a multiple of 32)
    runs on the fused MFMA layer kernel; every other shape the reference accepts -- any filter width, any channel
    counts, a 1-channel input that broadcasts in ``inputs + residual`` (the self-check of ops.py:232-236) -- runs the
    same arithmetic on the generic kernels.
    """
    x = _dev(inputs)
    B, T, cin = x.shape
    R = dilation_channels
    if gate_mode not in GATE_MODES:
        raise ValueError("gate_mode %r: 'reference' or 'wavenet'" % (gate_mode,))
    wf, bf, wr, br, ws, bs, wg, bg = _layer_variables(name, kernel_size, cin, R, skip_channels, with_gate=True)
    if not use_bias:
        bf = torch.zeros_like(bf); bg = torch.zeros_like(bg)
    if gate_mode == "wavenet" or cin != R or R not in (32, 64) or kernel_size != 2 or skip_channels % 32:
        if cin not in (1, R):
            raise ValueError("ResidualDilationLayer: inputs with %d channels cannot be added to a %d-channel residual "
                             "(ops.py:40)" % (cin, R))
        from ._lib import call
        st = K._stream()
        f = K.causal_conv1d_fwd(x, wf, bf, dilation_rate)                                   # ops.py:27
        z = torch.empty_like(f); c = torch.empty_like(f)
        g = K.causal_conv1d_fwd(x, wg, bg, dilation_rate) if gate_mode == "wavenet" else None   # ops.py:32
        call("srwn_gated_activation", f.data_ptr(), None if g is None else g.data_ptr(), z.data_ptr(), c.data_ptr(),
             f.numel(), GATE_MODES[gate_mode], st)                                          # ops.py:28,33,36
        res = K.causal_conv1d_fwd(c, wr, br.reshape(-1), 1)                                 # ops.py:39
        dense = torch.empty_like(res)
        call("srwn_residual_combine", x.data_ptr(), cin, res.data_ptr(), R, dense.data_ptr(), B * T, st)   # ops.py:40
        skip = K.causal_conv1d_fwd(c, ws, bs.reshape(-1), 1)                                # ops.py:44
        return dense, skip
    flat = torch.cat([wf.reshape(-1), wr.reshape(-1), ws.reshape(-1)])
    pk = K.Packer(x.device)
    oc = P.pack_conv(pk, 0, 2, R)
    orr = P.pack_res(pk, 2 * R * R, R)
    osk = P.pack_linear(pk, 3 * R * R, R, skip_channels, skip_channels)
    pk.finalize()
    buf = torch.empty(pk.total, dtype=torch.float32, device=x.device)
    pk.gather(flat, buf)
    dense = torch.empty_like(x); z = torch.empty_like(x)
    K.residual_layer_fwd(x, None, buf.data_ptr() + 4 * oc, buf.data_ptr() + 4 * orr, bf, br.reshape(-1), dense, z,
                         2, dilation_rate)
    skip = torch.empty((B * T, skip_channels), dtype=torch.float32, device=x.device)
    K.pw_linear(z.data_ptr(), R, 0, R, R, buf.data_ptr() + 4 * osk, bs.reshape(-1), skip, skip_channels,
                skip_channels, B * T, pro=K.PRO_GATE)
    return dense, skip.view(B, T, skip_channels)


def ResidualDilationLayerNC(inputs, kernel_size, dilation_channels, skip_channels, dilation_rate=1, name="",
                            dtype=None, use_bias=True):
    """ops.py:48-57 -> (residual, skip): relu -> tf.layers.conv1d(kernel_size, 'SAME') -> relu, then a 1x1 residual and
    a 1x1 skip.  ``dilation_rate`` is accepted and, as in the reference, never used.  (The encoder of
    WaveNetAutoEncoder runs these layers on the MFMA kernels of encoder.py; this is the ops-level function.)"""
    from ._lib import call
    x = _dev(inputs)
    B, T, cin = x.shape
    R = dilation_channels
    w = get_variable(name + "_NC/conv1d/kernel", (kernel_size, cin, R))
    b = get_variable(name + "_NC/conv1d/bias", (R,), "zeros")
    wr = get_variable(name + "/residual_nc/kernel", (1, R, R)); br = get_variable(name + "/residual_nc/bias", (R,), "zeros")
    ws = get_variable(name + "/skip_nc/kernel", (1, R, skip_channels))
    bs = get_variable(name + "/skip_nc/bias", (skip_channels,), "zeros")
    st = K._stream()
    r = torch.empty_like(x)
    call("srwn_relu", x.data_ptr(), r.data_ptr(), x.numel(), st)                            # ops.py:49
    # SAME padding of a stride-1 conv: pad_left = (K-1)//2 -> the causal kernel with its taps moved ceil((K-1)/2) ahead
    a = K.causal_conv1d_fwd(r, w, b.reshape(-1), 1, shift=-((kernel_size - 1) - (kernel_size - 1) // 2))   # ops.py:51
    call("srwn_relu", a.data_ptr(), a.data_ptr(), a.numel(), st)                            # ops.py:52
    return K.causal_conv1d_fwd(a, wr, br.reshape(-1), 1), K.causal_conv1d_fwd(a, ws, bs.reshape(-1), 1)    # ops.py:54-55


def ResizeEmbeddingNearestNeighbor(inputs, output_size):
    """ops.py:64-74: out[b,t,c] = in[b, floor(t*E/output_size), c].  Index plumbing only -- on the hot
    path this upsample never materialises (the layer kernel reads cond[b, t // pool_stride])."""
    x = _dev(inputs)
    E = x.shape[1]
    idx = torch.clamp((torch.arange(output_size, device=x.device) * (E / float(output_size))).long(), max=E - 1)
    return x.index_select(1, idx)


def RightShift(inputs, shift_size=1):
    """ops.py:78-80, as a 1-tap causal conv with an identity kernel and the shift folded into the tap."""
    x = _dev(inputs)
    C = x.shape[-1]
    eye = torch.eye(C, device=x.device, dtype=torch.float32).reshape(1, C, C)
    return K.causal_conv1d_fwd(x, eye, None, 1, shift_size)


def mu_law_encode(audio, quantization_channels):
    """ops.py:82-93 -> int32 codes."""
    return K.mu_law_encode(_dev(audio), quantization_channels)


def mu_law_decode(output, quantization_channels):
    """ops.py:96-104."""
    return K.mu_law_decode(_dev(output, torch.int32), quantization_channels)


def _rows(x):
    x = _dev(x)
    C = x.shape[-1]
    return x.reshape(-1, C), x.shape


def categorical_sample(logits, d=None, seed=0):
    """ops.py:106-109: one class index per row of ``logits`` [rows, C], drawn from softmax(logits) (``d`` is unused there
    too).  tf.multinomial's random stream cannot be reproduced; draws here are counter-based per (seed, row, class)."""
    from ._lib import call
    x, shp = _rows(logits)
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    call("srwn_categorical_sample", x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], int(seed), K._stream())
    return out.reshape(shp[:-1]).long()


def log_prob_from_logits(x):
    """ops.py:111-115: numerically stable log-softmax over the last axis."""
    from ._lib import call
    r, shp = _rows(x)
    y = torch.empty_like(r)
    call("srwn_log_softmax", r.data_ptr(), y.data_ptr(), None, r.shape[0], r.shape[1], K._stream())
    return y.reshape(shp)


def log_sum_exp(x):
    """ops.py:117-122: log-sum-exp over the last axis."""
    from ._lib import call
    r, shp = _rows(x)
    lse = torch.empty(r.shape[0], dtype=torch.float32, device=r.device)
    call("srwn_log_softmax", r.data_ptr(), None, lse.data_ptr(), r.shape[0], r.shape[1], K._stream())
    return lse.reshape(shp[:-1])


def discretized_mix_logistic_loss(x, l, sum_all=True):
    """ops.py:124-175: x [B,T,1] in [-1,1], l [B,T,4*nr_mix] -> -sum of the log-likelihood (sum_all) or the
    per-position negative log-likelihood [B,T,1]."""
    from ._lib import call
    lg = _dev(l)
    B, T, C4 = lg.shape
    M = C4 // 4
    xv = _dev(x).reshape(B * T)
    out = torch.empty(B * T, dtype=torch.float32, device=lg.device)
    call("srwn_mol_nll_rows", lg.data_ptr(), C4, xv.data_ptr(), M, out.data_ptr(), B * T, K._stream())
    if not sum_all:
        return out.reshape(B, T, 1)
    # (the training path fuses this sum and the gradient into srwn_mol_loss; here the rows are summed in a fixed order)
    tot = torch.empty(1, dtype=torch.float32, device=lg.device)
    K.reduce_loss(out, B * T, 1.0, tot)
    return tot[0]


def sample_from_discretized_mix_logistic(l, nr_mix, seed=None):
    """ops.py:178-201 -> x [B,T,1] in [-1,1]; the two uniform draws come from the device generator
    (tf.random_uniform(minval=1e-5, maxval=1-1e-5), ops.py:187,196)."""
    from ._lib import call
    lg = _dev(l)
    B, T, C4 = lg.shape
    gen = None
    if seed is not None:
        gen = torch.Generator(device=lg.device); gen.manual_seed(int(seed))
    lo, hi = 1e-5, 1.0 - 1e-5
    u1 = torch.rand((B * T, nr_mix), device=lg.device, generator=gen) * (hi - lo) + lo
    u2 = torch.rand((B * T,), device=lg.device, generator=gen) * (hi - lo) + lo
    out = torch.empty(B * T, dtype=torch.float32, device=lg.device)
    call("srwn_mol_sample", lg.data_ptr(), C4, nr_mix, u1.data_ptr(), u2.data_ptr(), out.data_ptr(), B * T, K._stream())
    return out.reshape(B, T, 1)


def probs_logistic(scale, mu, y, num_classes=256, log_scale_min=-14):
    """ops.py:203-214: probability mass of the bin around y under logistic(mu, scale)."""
    from ._lib import call
    s, m, yy = _dev(scale), _dev(mu), _dev(y)
    s, m, yy = torch.broadcast_tensors(s, m, yy)
    s, m, yy = s.contiguous(), m.contiguous(), yy.contiguous()
    out = torch.empty_like(s)
    call("srwn_probs_logistic", s.data_ptr(), m.data_ptr(), yy.data_ptr(), out.data_ptr(), s.numel(), int(num_classes),
         float(log_scale_min), K._stream())
    return out
