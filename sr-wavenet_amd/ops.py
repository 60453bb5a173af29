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


def ResidualDilationLayer(inputs, kernel_size, dilation_channels, skip_channels, dilation_rate=1, name="",
                          dtype=None, use_bias=True):
    """ops.py:23-46 -> (dense, skip), through the fused HIP layer kernel (fp32 MFMA).

    Like ops.py:33 the gate conv is created but its result is discarded (kept for checkpoint parity).
    """
    x = _dev(inputs)
    B, T, R = x.shape
    if R != dilation_channels or R not in (32, 64) or kernel_size != 2 or skip_channels % 32:
        raise NotImplementedError("fused layer: input channels == dilation_channels in {32,64}, kernel_size 2, "
                                  "skip_channels multiple of 32 (got %d -> %d, K=%d, S=%d)"
                                  % (R, dilation_channels, kernel_size, skip_channels))
    wf = get_variable(name + "_filter/" + name + "_Kernel", (kernel_size, R, R))
    bf = get_variable(name + "_filter/" + name + "_Bias", (1, 1, R), init="zeros").reshape(-1)
    get_variable(name + "_gate/" + name + "_Kernel", (kernel_size, R, R))       # dead (ops.py:32-33)
    get_variable(name + "_gate/" + name + "_Bias", (1, 1, R), init="zeros")
    wr = get_variable(name + "/residual/kernel", (1, R, R)); br = get_variable(name + "/residual/bias", (R,), "zeros")
    ws = get_variable(name + "/skip/kernel", (1, R, skip_channels))
    bs = get_variable(name + "/skip/bias", (skip_channels,), "zeros")
    if not use_bias:
        bf = torch.zeros_like(bf)
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
