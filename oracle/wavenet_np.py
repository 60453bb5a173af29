"""CPU oracle (i): explicit NumPy restatement of the reference's WaveNet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may use it, and there only as the checker.

Every function cites the reference lines (``/root/reference``) it restates.

PARITY PINNING.  The reference cannot be imported in the authoring container
(``import tensorflow`` -> ModuleNotFoundError; no network) and it holds no
expected outputs: its only self-check (``ops.py:221-264``) *prints* results of
``_DilatedCausalConv1d`` on fixed inputs.  The oracle is therefore pinned

* for the dilated causal conv (a1) by the fixed inputs of ``ops.py:224-254``
  whose outputs follow by hand from ``ops.py:6-10`` (tests/golden/ops_selfcheck.json);
* for mu-law by closed-form values of ``ops.py:82-104``;
* for everything else (layer, stack, head, losses, Adam): **parity unpinned** by
  the reference; pinned instead by two independent restatements agreeing with
  each other (this file, explicit loops/einsum + hand-written backward, versus
  ``oracle/wavenet_torch.py``, torch-CPU conv1d + autograd).

Arithmetic is float64 unless a dtype is passed; layout is the reference's
channels-last ``[B, T, C]`` with filters ``[K, Cin, Cout]`` (``ops.py:4-5``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SQRT_HALF = 0.7071067811865476  # literal at ops.py:40


# --------------------------------------------------------------------------
# ops.py restatements
# --------------------------------------------------------------------------
def dilated_causal_conv1d(x: np.ndarray, w: np.ndarray, dilation: int = 1) -> np.ndarray:
    """``_DilatedCausalConv1d`` (ops.py:6-10).

    Left-pad ``d*(K-1)`` zeros then VALID cross-correlation with dilation ``d``:
    ``y[b,t,o] = sum_k sum_i x[b, t-(K-1-k)*d, i] * w[k,i,o]`` (zeros for t<0).
    """
    B, T, Cin = x.shape
    K, Cin2, Cout = w.shape
    assert Cin == Cin2
    y = np.zeros((B, T, Cout), dtype=np.result_type(x, w))
    for k in range(K):
        shift = (K - 1 - k) * dilation
        if shift >= T:
            continue
        # y[:, shift:, :] += x[:, :T-shift, :] @ w[k]
        y[:, shift:, :] += np.einsum("bti,io->bto", x[:, : T - shift, :], w[k])
    return y


def dilated_causal_conv1d_bias(x, w, b, dilation=1):
    """``DilatedCausalConv1d`` (ops.py:13-20): conv + bias of shape [1,1,C]."""
    y = dilated_causal_conv1d(x, w, dilation)
    if b is not None:
        y = y + np.reshape(b, (1, 1, -1))
    return y


def sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def residual_dilation_layer(x, lp: "LayerParams", dilation: int, gate_mode: str = "reference"):
    """``ResidualDilationLayer`` (ops.py:23-46) -> (dense, skip, cache).

    gate_mode="reference" reproduces ops.py:33 exactly: the gate conv result is
    discarded and ``gated_conv = sigmoid(filter_conv)`` where ``filter_conv`` is
    already ``tanh(conv)``; i.e. ``combined = z * sigmoid(z)``, ``z = tanh(f)``.
    gate_mode="wavenet" is the canonical ``tanh(f) * sigmoid(g)`` (opt-in).
    """
    f = dilated_causal_conv1d_bias(x, lp.wf, lp.bf, dilation)  # ops.py:27
    z = np.tanh(f)  # ops.py:28
    if gate_mode == "reference":
        s = sigmoid(z)  # ops.py:33 (the bug, kept for parity)
        g = None
    elif gate_mode == "wavenet":
        g = dilated_causal_conv1d_bias(x, lp.wg, lp.bg, dilation)  # ops.py:32
        s = sigmoid(g)
    else:
        raise ValueError(gate_mode)
    c = z * s  # ops.py:36
    res = c @ lp.wr + lp.br  # tf.layers.conv1d k=1, bias on (ops.py:39)
    dense = (x + res) * SQRT_HALF  # ops.py:40
    skip = c @ lp.ws + lp.bs  # ops.py:44
    return dense, skip, dict(x=x, z=z, s=s, c=c, g=g)


def resize_embedding_nearest_neighbor(e: np.ndarray, output_size: int) -> np.ndarray:
    """``ResizeEmbeddingNearestNeighbor`` (ops.py:64-74).

    tf.image.resize_nearest_neighbor, align_corners=False, on [B,E,C,1] with the
    channel axis unchanged: ``out[b,t,c] = in[b, floor(t*E/output_size), c]``.
    """
    B, E, C = e.shape
    idx = np.minimum((np.arange(output_size) * (E / float(output_size))).astype(np.int64), E - 1)
    # TF computes floor(dst * scale) with scale = in/out in float32; for the integer
    # ratios used by the reference (output = pool_stride * E) this equals t // pool_stride.
    return e[:, idx, :]


def right_shift(x: np.ndarray, shift: int = 1) -> np.ndarray:
    """``RightShift`` (ops.py:78-80): out[:,t] = in[:,t-shift], zero fill."""
    out = np.zeros_like(x)
    if shift < x.shape[1]:
        out[:, shift:, :] = x[:, : x.shape[1] - shift, :]
    return out


def _log1p_f32(v: np.ndarray) -> np.ndarray:
    """Correctly-rounded float32 log1p: evaluate in float64, round once.

    TF's fp32 kernel approximates this value to ~1 ulp; fixing the definition
    lets the HIP kernel (which does the same in f64) be bit-exact against it.
    """
    return np.log1p(v.astype(np.float64)).astype(np.float32)


def mu_law_encode(audio: np.ndarray, quantization_channels: int) -> np.ndarray:
    """``mu_law_encode`` (ops.py:82-93) in float32, op by op -> int32 codes.

    tf.to_int32 truncates toward zero (values are >= 0 here).
    """
    a = np.asarray(audio, dtype=np.float32)
    mu = np.float32(quantization_channels - 1)
    safe = np.minimum(np.abs(a), np.float32(1.0))
    magnitude = _log1p_f32(mu * safe) / _log1p_f32(np.asarray(mu))
    signal = np.sign(a) * magnitude
    q = (signal + np.float32(1.0)) / np.float32(2.0) * mu + np.float32(0.5)
    return q.astype(np.int32)  # truncation


def mu_law_decode(codes: np.ndarray, quantization_channels: int) -> np.ndarray:
    """``mu_law_decode`` (ops.py:96-104) in float32 -> waveform in [-1,1].

    ``(1 + mu) ** abs(signal)`` is evaluated in float64 and rounded once (the
    correctly rounded value of TF's fp32 pow); ``1/mu`` is the Python double
    1/255 rounded to float32 when it meets the float32 tensor.
    """
    mu = quantization_channels - 1
    o = np.asarray(codes).astype(np.float32)
    signal = np.float32(2.0) * (o / np.float32(mu)) - np.float32(1.0)
    p = np.power(np.float64(1 + mu), np.abs(signal).astype(np.float64)).astype(np.float32)
    magnitude = np.float32(1.0 / mu) * (p - np.float32(1.0))
    return (np.sign(signal) * magnitude).astype(np.float32)


def log_prob_from_logits(x: np.ndarray) -> np.ndarray:
    """``log_prob_from_logits`` (ops.py:111-115): max-subtracted log-softmax."""
    m = np.max(x, axis=-1, keepdims=True)
    return x - m - np.log(np.sum(np.exp(x - m), axis=-1, keepdims=True))


def log_sum_exp(x: np.ndarray) -> np.ndarray:
    """``log_sum_exp`` (ops.py:117-122)."""
    m = np.max(x, axis=-1)
    m2 = np.max(x, axis=-1, keepdims=True)
    return m + np.log(np.sum(np.exp(x - m2), axis=-1))


def probs_logistic(scale, mu, y, num_classes: int = 256, log_scale_min: float = -14.0):
    """``probs_logistic`` (ops.py:203-214): mass of the bin of half-width 1/(num_classes-1) around y under a logistic
    with mean mu and scale clipped below at exp(log_scale_min)."""
    scale = np.clip(np.asarray(scale, dtype=np.float64), np.exp(log_scale_min), np.inf)
    c = np.asarray(y, dtype=np.float64) - np.asarray(mu, dtype=np.float64)
    h = 1.0 / (num_classes - 1)
    return sigmoid((c + h) / scale) - sigmoid((c - h) / scale)


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
@dataclass
class LayerParams:
    wf: np.ndarray  # [K,R,R]  "<name>_filter/<name>_Kernel"
    bf: np.ndarray  # [R]      "<name>_filter/<name>_Bias" ([1,1,R] in TF)
    wg: np.ndarray  # [K,R,R]  dead in gate_mode="reference" (ops.py:32-33)
    bg: np.ndarray  # [R]
    wr: np.ndarray  # [R,R]    tf.layers.conv1d kernel [1,R,R]
    br: np.ndarray  # [R]
    ws: np.ndarray  # [R,S]
    bs: np.ndarray  # [S]
    wc: Optional[np.ndarray] = None  # [E,R] conditioning 1x1 (model.py:180), decoder only
    bc: Optional[np.ndarray] = None  # [R]


@dataclass
class StackParams:
    init_w: np.ndarray  # [K,1,R] causal_conv_Kernel (model.py:40)
    init_b: np.ndarray  # [R]
    layers: List[LayerParams]
    head_w1: np.ndarray  # [S,S] (model.py:53)
    head_b1: np.ndarray
    head_w2: np.ndarray  # [S,C] (model.py:56)
    head_b2: np.ndarray
    dilations: Tuple[int, ...] = ()

    def astype(self, dt):
        def c(a):
            return None if a is None else a.astype(dt)

        return StackParams(
            c(self.init_w), c(self.init_b),
            [LayerParams(*[c(getattr(l, f)) for f in ("wf", "bf", "wg", "bg", "wr", "br", "ws", "bs", "wc", "bc")])
             for l in self.layers],
            c(self.head_w1), c(self.head_b1), c(self.head_w2), c(self.head_b2), tuple(self.dilations))


def xavier_uniform(rng: np.random.Generator, shape: Sequence[int]) -> np.ndarray:
    """tf.contrib.layers.xavier_initializer() (uniform) / tf.layers glorot_uniform.

    limit = sqrt(6/(fan_in+fan_out)), fan_in = K*Cin, fan_out = K*Cout for a
    [K,Cin,Cout] kernel (ops.py:15; tf.layers.conv1d default kernel_initializer).
    """
    shape = tuple(shape)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def init_stack_params(seed: int, dilations: Sequence[int], K: int, R: int, S: int, C: int,
                      cond_channels: int = 0, bias_scale: float = 0.0) -> StackParams:
    """Seeded init following the reference's initialisers (biases 0 unless
    ``bias_scale`` is given, which tests use to make biases matter)."""
    rng = np.random.default_rng(seed)

    def bias(n):
        return rng.normal(0, bias_scale, size=(n,)) if bias_scale else np.zeros((n,))

    init_w = xavier_uniform(rng, (K, 1, R))
    init_b = bias(R)
    layers = []
    for _ in dilations:
        wf = xavier_uniform(rng, (K, R, R)); bf = bias(R)
        wg = xavier_uniform(rng, (K, R, R)); bg = bias(R)
        wc = bc = None
        if cond_channels:
            wc = xavier_uniform(rng, (1, cond_channels, R))[0]; bc = bias(R)
        wr = xavier_uniform(rng, (1, R, R))[0]; br = bias(R)
        ws = xavier_uniform(rng, (1, R, S))[0]; bs = bias(S)
        layers.append(LayerParams(wf, bf, wg, bg, wr, br, ws, bs, wc, bc))
    w1 = xavier_uniform(rng, (1, S, S))[0]; b1 = bias(S)
    w2 = xavier_uniform(rng, (1, S, C))[0]; b2 = bias(C)
    return StackParams(init_w, init_b, layers, w1, b1, w2, b2, tuple(int(d) for d in dilations))


def tf_variable_names(p: StackParams, scope: str = "WaveNet", decoder: bool = False) -> Dict[str, np.ndarray]:
    """Reference variable names -> arrays in TF shapes (SURVEY §8a naming rules).

    WaveNet class: layer i residual = conv1d_{2i}, skip = conv1d_{2i+1}, head =
    conv1d_{2L}, conv1d_{2L+1} (first unnamed layer is plain "conv1d").  Decoder
    (model.py:180; ops.py:39,44): cond = conv1d_{3i}, residual = {3i+1}, skip = {3i+2}.
    """
    out: Dict[str, np.ndarray] = {}

    def cname(j):
        return "conv1d" if j == 0 else "conv1d_%d" % j

    out[f"{scope}/causal_conv_Kernel"] = p.init_w
    out[f"{scope}/causal_conv_Bias"] = p.init_b.reshape(1, 1, -1)
    per = 3 if decoder else 2
    for i, l in enumerate(p.layers):
        n = f"dilated_conv_{i}"
        out[f"{scope}/{n}_filter/{n}_Kernel"] = l.wf
        out[f"{scope}/{n}_filter/{n}_Bias"] = l.bf.reshape(1, 1, -1)
        out[f"{scope}/{n}_gate/{n}_Kernel"] = l.wg
        out[f"{scope}/{n}_gate/{n}_Bias"] = l.bg.reshape(1, 1, -1)
        j = per * i
        if decoder:
            out[f"{scope}/{cname(j)}/kernel"] = l.wc[None]
            out[f"{scope}/{cname(j)}/bias"] = l.bc
            j += 1
        out[f"{scope}/{cname(j)}/kernel"] = l.wr[None]
        out[f"{scope}/{cname(j)}/bias"] = l.br
        out[f"{scope}/{cname(j + 1)}/kernel"] = l.ws[None]
        out[f"{scope}/{cname(j + 1)}/bias"] = l.bs
    L = len(p.layers)
    out[f"{scope}/{cname(per * L)}/kernel"] = p.head_w1[None]
    out[f"{scope}/{cname(per * L)}/bias"] = p.head_b1
    out[f"{scope}/{cname(per * L + 1)}/kernel"] = p.head_w2[None]
    out[f"{scope}/{cname(per * L + 1)}/bias"] = p.head_b2
    return out


# --------------------------------------------------------------------------
# model.py restatements: forward
# --------------------------------------------------------------------------
def stack_forward(p: StackParams, audio: np.ndarray, *, shift_input: bool = False,
                  cond: Optional[np.ndarray] = None, pool_stride: int = 1,
                  gate_mode: str = "reference"):
    """The residual stack + head of ``WaveNet.createNetwork`` (model.py:33-56) and,
    with ``shift_input``/``cond``, of ``createDecoder`` (model.py:158-196).

    audio [B,T]; cond = encoding_w_condition [B,E,Cc] with T == pool_stride*E.
    Returns per-timestep logits [B,T,C] and a cache for the backward pass.
    """
    x0 = audio[:, :, None]  # tf.expand_dims(inputs, 2)  model.py:35
    if shift_input:
        x0 = right_shift(x0)  # model.py:172
    h = dilated_causal_conv1d_bias(x0, p.init_w, p.init_b, 1)  # model.py:40 / 173
    caches = []
    skips = None
    for l, d in zip(p.layers, p.dilations):
        up = None
        if cond is not None:
            cb = cond @ l.wc + l.bc  # model.py:180
            up = resize_embedding_nearest_neighbor(cb, pool_stride * cb.shape[1])  # model.py:181
            h = h + up  # model.py:183
        h, skip, cache = residual_dilation_layer(h, l, d, gate_mode)  # model.py:45 / 185
        cache["up"] = up
        caches.append(cache)
        skips = skip if skips is None else skips + skip  # tf.reduce_sum(skip_layers, 0)  model.py:50
    total = skips
    r0 = np.maximum(total, 0)  # model.py:51
    a1 = r0 @ p.head_w1 + p.head_b1  # model.py:53
    r1 = np.maximum(a1, 0)  # model.py:54
    logits = r1 @ p.head_w2 + p.head_b2  # model.py:56
    return logits, dict(x0=x0, layers=caches, total=total, r0=r0, a1=a1, r1=r1, h_last=h)


def wavenet_pooled_logits(logits_t: np.ndarray) -> np.ndarray:
    """tf.nn.pool AVG, window = input_size = T, VALID (model.py:58): [B,T,C] -> [B,1,C]."""
    return logits_t.mean(axis=1, keepdims=True)


def softmax(x):
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def wavenet_predict(p: StackParams, audio: np.ndarray, gate_mode="reference") -> np.ndarray:
    """``WaveNet.predict`` (model.py:58-60,70-72): softmax of time-pooled logits [B,1,C]."""
    logits_t, _ = stack_forward(p, audio, gate_mode=gate_mode)
    return softmax(wavenet_pooled_logits(logits_t))


def wavenet_loss_pooled(logits_t: np.ndarray, targets: np.ndarray) -> float:
    """``WaveNet`` loss (model.py:24-29): mean over [B,1] of softmax-CE-v2 against
    ``labels = targets[:,None,:]`` (soft labels allowed): -sum(labels*log_softmax)."""
    lp = log_prob_from_logits(wavenet_pooled_logits(logits_t))
    ce = -(targets[:, None, :] * lp).sum(-1)
    return float(ce.mean())


def softmax_ce_per_timestep(logits_t: np.ndarray, codes: np.ndarray) -> float:
    """mu-law softmax teacher loss: the head the reference carries commented out at
    model.py:100-112 (``targets = one_hot(mu_law_encode(inputs))``;
    ``loss = reduce_mean(softmax_CE(logits, targets))``): mean over [B,T]."""
    lp = log_prob_from_logits(logits_t)
    B, T, _ = logits_t.shape
    picked = np.take_along_axis(lp, codes[:, :, None].astype(np.int64), axis=2)[..., 0]
    return float(-picked.mean())


# --------------------------------------------------------------------------
# hand-written backward (independent of autograd; cross-checked in tests)
# --------------------------------------------------------------------------
def _conv_backward(x, w, dilation, dy):
    """Gradients of dilated_causal_conv1d wrt x, w (bias grad = dy.sum((0,1)))."""
    B, T, _ = x.shape
    K = w.shape[0]
    dx = np.zeros_like(x, dtype=dy.dtype)
    dw = np.zeros_like(w, dtype=dy.dtype)
    for k in range(K):
        s = (K - 1 - k) * dilation
        if s >= T:
            continue
        dw[k] = np.einsum("bti,bto->io", x[:, : T - s, :], dy[:, s:, :])
        dx[:, : T - s, :] += np.einsum("bto,io->bti", dy[:, s:, :], w[k])
    return dx, dw


def stack_backward(p: StackParams, cache, dlogits: np.ndarray, *, cond=None, pool_stride=1,
                   gate_mode: str = "reference"):
    """Backward of :func:`stack_forward` for d(loss)/d(logits_t) = dlogits [B,T,C].

    Returns a StackParams-shaped gradient object (dead gate params get zeros in
    gate_mode="reference", where TF reports None gradients for them).
    """
    L = len(p.layers)
    g_w2 = np.einsum("bts,btc->sc", cache["r1"], dlogits)
    g_b2 = dlogits.sum((0, 1))
    dr1 = dlogits @ p.head_w2.T
    da1 = dr1 * (cache["a1"] > 0)
    g_w1 = np.einsum("bts,btu->su", cache["r0"], da1)
    g_b1 = da1.sum((0, 1))
    dr0 = da1 @ p.head_w1.T
    dtotal = dr0 * (cache["total"] > 0)

    glayers: List[Optional[LayerParams]] = [None] * L
    dh = np.zeros_like(cache["layers"][-1]["x"], dtype=dlogits.dtype)  # grad wrt last dense: unused output
    dcond = None if cond is None else np.zeros_like(cond, dtype=dlogits.dtype)
    for i in range(L - 1, -1, -1):
        l, d, c = p.layers[i], p.dilations[i], cache["layers"][i]
        x, z, s, cc = c["x"], c["z"], c["s"], c["c"]
        dres = dh * SQRT_HALF
        g_wr = np.einsum("btn,btm->nm", cc, dres); g_br = dres.sum((0, 1))
        g_ws = np.einsum("btn,bts->ns", cc, dtotal); g_bs = dtotal.sum((0, 1))
        dc = dres @ l.wr.T + dtotal @ l.ws.T
        if gate_mode == "reference":
            dz = dc * (s + z * s * (1 - s))
            df = dz * (1 - z * z)
            dxc, g_wf = _conv_backward(x, l.wf, d, df)
            g_bf = df.sum((0, 1))
            g_wg = np.zeros_like(l.wg); g_bg = np.zeros_like(l.bg)
        else:
            dz = dc * s
            df = dz * (1 - z * z)
            dg = dc * z * s * (1 - s)
            dxc, g_wf = _conv_backward(x, l.wf, d, df)
            dxg, g_wg = _conv_backward(x, l.wg, d, dg)
            dxc = dxc + dxg
            g_bf = df.sum((0, 1)); g_bg = dg.sum((0, 1))
        dx = dres + dxc  # x is the (conditioned) layer input; dense = (x+res)*sqrt(.5)
        g_wc = g_bc = None
        if cond is not None:
            B, T, R = dx.shape
            E = cond.shape[1]
            dcb = dx.reshape(B, E, pool_stride, R).sum(2)  # adjoint of NN upsample
            g_wc = np.einsum("bec,ber->cr", cond, dcb)
            g_bc = dcb.sum((0, 1))
            dcond += dcb @ l.wc.T
        glayers[i] = LayerParams(g_wf, g_bf, g_wg, g_bg, g_wr, g_br, g_ws, g_bs, g_wc, g_bc)
        dh = dx
    dx0, g_iw = _conv_backward(cache["x0"], p.init_w, 1, dh)
    g_ib = dh.sum((0, 1))
    grads = StackParams(g_iw, g_ib, glayers, g_w1, g_b1, g_w2, g_b2, tuple(p.dilations))
    return grads, dict(dx0=dx0, dcond=dcond, dtotal=dtotal)


def dlogits_pooled(logits_t: np.ndarray, targets: np.ndarray) -> np.ndarray:
    """d/dlogits_t of :func:`wavenet_loss_pooled`."""
    B, T, C = logits_t.shape
    sm = softmax(wavenet_pooled_logits(logits_t))[:, 0, :]
    # d CE / d pooled = softmax*sum(labels) - labels ; mean over B ; pooled = mean over T
    dp = (sm * targets.sum(-1, keepdims=True) - targets) / B
    return np.broadcast_to(dp[:, None, :] / T, (B, T, C)).copy()


def dlogits_per_timestep(logits_t: np.ndarray, codes: np.ndarray) -> np.ndarray:
    """d/dlogits_t of :func:`softmax_ce_per_timestep`."""
    B, T, C = logits_t.shape
    sm = softmax(logits_t)
    oh = np.zeros_like(sm)
    np.put_along_axis(oh, codes[:, :, None].astype(np.int64), 1.0, axis=2)
    return (sm - oh) / (B * T)


# --------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------
def adam_step_tf(theta, g, m, v, t: int, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (model.py:31): ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t)``,
    ``theta -= lr_t * m / (sqrt(v) + eps)`` (epsilon added to the *uncorrected*
    sqrt(v) -- differs from torch.optim.Adam).  ``t`` counts from 1."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------
# flat parameter vector (same order the product uses; tests compare by name)
# --------------------------------------------------------------------------
def flatten_named(p: StackParams, include_cond: bool) -> List[Tuple[str, np.ndarray]]:
    out = [("init_w", p.init_w), ("init_b", p.init_b)]
    for i, l in enumerate(p.layers):
        out += [(f"l{i}.wf", l.wf), (f"l{i}.bf", l.bf), (f"l{i}.wr", l.wr), (f"l{i}.br", l.br),
                (f"l{i}.ws", l.ws), (f"l{i}.bs", l.bs)]
        if include_cond:
            out += [(f"l{i}.wc", l.wc), (f"l{i}.bc", l.bc)]
    out += [("head_w1", p.head_w1), ("head_b1", p.head_b1), ("head_w2", p.head_w2), ("head_b2", p.head_b2)]
    return out


# --------------------------------------------------------------------------
# synthetic input (mirrors simple_audio.py:40-61 in spirit)
# --------------------------------------------------------------------------
def synthetic_audio(B: int, T: int, seed: int = 0, sample_rate: int = 16000) -> np.ndarray:
    """x[b,t] = 0.5*sin(2*pi*f_b*t/sr) + 0.05*N(0,1), f_b = 110*(b+1) Hz, clipped to [-1,1], float32."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)[None, :]
    f = 110.0 * (np.arange(B, dtype=np.float64)[:, None] % 16 + 1)
    x = 0.5 * np.sin(2 * np.pi * f * t / sample_rate) + 0.05 * rng.standard_normal((B, T))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


# --------------------------------------------------------------------------
# discretised mixture-of-logistics head of the reference's live teacher (ops.py:124-175)
# --------------------------------------------------------------------------
def _softplus(v):
    return np.logaddexp(0.0, v)


def mol_log_probs(x: np.ndarray, l: np.ndarray):
    """Per-mixture log-probabilities of ``discretized_mix_logistic_loss`` (ops.py:131-169).

    x [B,T] in [-1,1]; l [B,T,4M] = (logit_probs, means, log_scales, coeffs); the coeffs (ops.py:138) are
    computed by the reference but never used for single-channel audio.  Returns (log_probs [B,T,M], aux).
    """
    M = l.shape[-1] // 4
    logit_probs = l[..., :M]
    means = l[..., M:2 * M]
    raw_ls = l[..., 2 * M:3 * M]
    log_scales = np.maximum(raw_ls, -7.0)                      # ops.py:137
    xx = x[..., None]
    centered = xx - means                                       # ops.py:147
    inv = np.exp(-log_scales)
    plus_in = inv * (centered + 1.0 / 255.0)
    min_in = inv * (centered - 1.0 / 255.0)
    cdf_plus = sigmoid(plus_in); cdf_min = sigmoid(min_in)
    log_cdf_plus = plus_in - _softplus(plus_in)                 # ops.py:153
    log_one_minus_cdf_min = -_softplus(min_in)                  # ops.py:154
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * _softplus(mid_in)  # ops.py:157
    case = np.where(xx < -0.999, 0, np.where(xx > 0.999, 1, np.where(cdf_delta > 1e-5, 2, 3)))   # ops.py:169
    comp = np.where(case == 0, log_cdf_plus,
                    np.where(case == 1, log_one_minus_cdf_min,
                             np.where(case == 2, np.log(np.maximum(cdf_delta, 1e-12)), log_pdf_mid - np.log(127.5))))
    lp = comp + log_prob_from_logits(logit_probs)               # ops.py:171
    return lp, dict(case=case, inv=inv, plus_in=plus_in, min_in=min_in, mid_in=mid_in, cdf_plus=cdf_plus,
                    cdf_min=cdf_min, cdf_delta=cdf_delta, clamp=(raw_ls > -7.0), logit_probs=logit_probs)


def mol_loss(x: np.ndarray, l: np.ndarray) -> float:
    """``discretized_mix_logistic_loss(x, l, sum_all=True)`` (ops.py:173-174): -sum over B,T of logsumexp."""
    lp, _ = mol_log_probs(x, l)
    return float(-log_sum_exp(lp).sum())


def mol_dlogits(x: np.ndarray, l: np.ndarray) -> np.ndarray:
    """Hand-derived d(mol_loss)/dl [B,T,4M] (tf.where routes the gradient to the selected branch only)."""
    M = l.shape[-1] // 4
    lp, a = mol_log_probs(x, l)
    w = softmax(lp)                                             # d(-LSE)/d lp_m = -w_m
    sm = softmax(a["logit_probs"])
    g = np.zeros_like(l)
    g[..., :M] = -(w - sm)
    sp, smn, smd = sigmoid(a["plus_in"]), sigmoid(a["min_in"]), sigmoid(a["mid_in"])
    inv, case = a["inv"], a["case"]
    # d comp / d mean and d comp / d log_scale per branch
    dm0 = (1 - sp) * (-inv);            ds0 = (1 - sp) * (-a["plus_in"])
    dm1 = (-smn) * (-inv);              ds1 = (-smn) * (-a["min_in"])
    num_m = sp * (1 - sp) * (-inv) - smn * (1 - smn) * (-inv)
    num_s = sp * (1 - sp) * (-a["plus_in"]) - smn * (1 - smn) * (-a["min_in"])
    den = np.maximum(a["cdf_delta"], 1e-12)
    dm2 = num_m / den;                  ds2 = num_s / den
    dm3 = (1 - 2 * smd) * (-inv);       ds3 = (1 - 2 * smd) * (-a["mid_in"]) - 1.0
    dm = np.select([case == 0, case == 1, case == 2], [dm0, dm1, dm2], dm3)
    ds = np.select([case == 0, case == 1, case == 2], [ds0, ds1, ds2], ds3)
    g[..., M:2 * M] = -w * dm
    g[..., 2 * M:3 * M] = -w * ds * a["clamp"]
    return g


def mol_dx(x: np.ndarray, l: np.ndarray) -> np.ndarray:
    """d(mol_loss)/dx [B,T]: x enters only through ``centered = x - means`` (ops.py:147), so it is minus the
    sum over mixtures of the gradient wrt the means (the tf.where conditions on x carry no gradient)."""
    M = l.shape[-1] // 4
    return -mol_dlogits(x, l)[..., M:2 * M].sum(-1)


# --------------------------------------------------------------------------
# Parallel-WaveNet student (model.py:290-537): inverse-autoregressive flows distilled against a frozen teacher
# --------------------------------------------------------------------------
def init_flow_params(seed: int, dilations: Sequence[int], K: int, R: int, S: int, cond_channels: int,
                     bias_scale: float = 0.0) -> StackParams:
    """One flow of ``createPartialFlow`` (model.py:415-453): a conditioned decoder-style stack whose skip 1x1s
    exist as variables but are unused (model.py:440-449 commented out), with head relu -> 1x1 R->2
    (model.py:451-452) stored as ``head_w2``/``head_b2`` (``head_w1`` is unused)."""
    p = init_stack_params(seed, dilations, K, R, S, 2, cond_channels=cond_channels, bias_scale=bias_scale)
    rng = np.random.default_rng(seed + 7919)
    p.head_w1 = None; p.head_b1 = None
    p.head_w2 = xavier_uniform(rng, (1, R, 2))[0]
    p.head_b2 = rng.normal(0, bias_scale, size=(2,)) if bias_scale else np.zeros((2,))
    return p


def flow_forward(p: StackParams, x_in: np.ndarray, cond: np.ndarray, pool_stride: int):
    """``createFlow`` (model.py:456-487): params = partial flow(x_in); scale = exp(params[...,0]),
    mean = params[...,1]; out = x_in*scale + mean.  x_in [B,T]; cond [B,E,Cc]."""
    h = dilated_causal_conv1d_bias(right_shift(x_in[:, :, None]), p.init_w, p.init_b, 1)   # model.py:423-424
    for l, d in zip(p.layers, p.dilations):
        cb = cond @ l.wc + l.bc                                                            # model.py:431
        h = h + resize_embedding_nearest_neighbor(cb, pool_stride * cb.shape[1])           # model.py:432-435
        h, _skip, _ = residual_dilation_layer(h, l, d, "reference")                        # model.py:438-440
    prm = np.maximum(h, 0) @ p.head_w2 + p.head_b2                                         # model.py:451-452
    scale = np.exp(prm[..., 0]); mean = prm[..., 1]                                        # model.py:479-480
    return scale, mean, x_in * scale + mean, prm


def student_forward(flows: Sequence[StackParams], noise: np.ndarray, cond: np.ndarray, pool_stride: int):
    """``ParallelWaveNet.createNetwork`` (model.py:490-535).  Returns out [B,T] (clipped), s_tot, mu_tot and the
    per-flow scales/means."""
    x = noise
    scales, means = [], []
    for p in flows:
        s, m, x, _ = flow_forward(p, x, cond, pool_stride)
        scales.append(s); means.append(m)
    s_tot = np.ones_like(noise); mu_tot = np.zeros_like(noise)
    for i in range(len(flows)):                                                            # model.py:517-533
        s_tot = s_tot * scales[i]
        mu = means[i]
        for j in range(i + 1, len(flows)):
            mu = mu * scales[j]
        mu_tot = mu_tot + mu
    out = np.minimum(np.maximum(noise * s_tot + mu_tot, -1.0), 1.0)                        # model.py:535
    return dict(out=out, s_tot=s_tot, mu_tot=mu_tot, scales=scales, means=means, x_last=x)


def hann_periodic(n: int) -> np.ndarray:
    """tf.contrib.signal.hann_window(n, periodic=True), the default window of tf.contrib.signal.stft."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_power(x: np.ndarray, frame_length: int = 512, frame_step: int = 256) -> np.ndarray:
    """``reduce_mean(abs(stft(x, 512, 256))**2, 1)`` (model.py:360-368): frames without end padding
    (1 + (T-512)//256 of them), periodic Hann window, fft_length = 512 -> [B, 257]."""
    B, T = x.shape
    nf = 1 + (T - frame_length) // frame_step
    if T < frame_length or nf < 1:
        raise ValueError("stft_power: clip shorter than one frame")
    frames = np.stack([x[:, i * frame_step:i * frame_step + frame_length] for i in range(nf)], 1)
    X = np.fft.rfft(frames * hann_periodic(frame_length), n=frame_length, axis=-1)
    return (np.abs(X) ** 2).mean(1)


def student_loss(fw, teacher_logits: np.ndarray, truth: np.ndarray, alpha=1.0, beta=1.0, gamma=1.0):
    """model.py:356-379: entropy = sum(log s_tot + 2); power = gamma*||phi(truth) - phi(out)||_F^2;
    loss = (beta*MoL_NLL(out | teacher) - alpha*entropy + power) / B."""
    B = truth.shape[0]
    entropy = float((np.log(fw["s_tot"]) + 2.0).sum())
    diff = stft_power(truth) - stft_power(fw["out"])
    power = float((diff ** 2).sum()) * gamma
    ce = mol_loss(np.clip(fw["out"], -1, 1), teacher_logits) * beta
    return dict(loss=(ce - alpha * entropy + power) / B, power_loss=power, entropy=entropy, cross_entropy=ce)


def clip_by_global_norm(grads: Sequence[np.ndarray], clip_norm: float = 1.0):
    """tf.clip_by_global_norm (model.py:385): g * clip_norm / max(global_norm, clip_norm)."""
    gn = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    s = clip_norm / max(gn, clip_norm)
    return [g * s for g in grads], gn


# --------------------------------------------------------------------------
# WaveNetAutoEncoder (model.py:75-285): non-causal encoder + the conditioned decoder above + mixture sampler
# --------------------------------------------------------------------------
@dataclass
class NCLayerParams:
    w: np.ndarray    # [K,Cin,EC]  "<name>_NC/conv1d/kernel" (ops.py:51)
    b: np.ndarray    # [EC]
    wr: np.ndarray   # [EC,EC]     residual 1x1 (ops.py:54)
    br: np.ndarray
    ws: np.ndarray   # [EC,S]      skip 1x1 (ops.py:55)
    bs: np.ndarray


@dataclass
class EncoderParams:
    nc: NCLayerParams            # 'nc_conv' on the raw clip, Cin = 1; its skip output is discarded (model.py:141)
    layers: List[NCLayerParams]
    lat_w: np.ndarray            # [S, latent] (model.py:152)
    lat_b: np.ndarray


def init_encoder_params(seed: int, nlayers: int, K: int, EC: int, S: int, latent: int,
                        bias_scale: float = 0.0) -> EncoderParams:
    rng = np.random.default_rng(seed)
    bias = lambda n: rng.normal(0, bias_scale, size=(n,)) if bias_scale else np.zeros((n,))

    def layer(cin):
        return NCLayerParams(xavier_uniform(rng, (K, cin, EC)), bias(EC), xavier_uniform(rng, (1, EC, EC))[0], bias(EC),
                             xavier_uniform(rng, (1, EC, S))[0], bias(S))

    return EncoderParams(layer(1), [layer(EC) for _ in range(nlayers)], xavier_uniform(rng, (1, S, latent))[0],
                         bias(latent))


def conv1d_same(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """tf.layers.conv1d(strides=1, padding='SAME') (ops.py:51): cross-correlation with pad_left = (K-1)//2,
    pad_right = K-1-pad_left -- for K = 2: y[t] = x[t] w[0] + x[t+1] w[1], zero beyond the clip."""
    B, T, _ = x.shape
    K = w.shape[0]
    pl = (K - 1) // 2
    y = np.zeros((B, T, w.shape[2]), dtype=np.result_type(x, w))
    for k in range(K):
        o = k - pl                      # y[t] += x[t+o] w[k]
        lo, hi = max(0, -o), min(T, T - o)
        if hi > lo:
            y[:, lo:hi] += x[:, lo + o:hi + o] @ w[k]
    return y


def residual_dilation_layer_nc(x: np.ndarray, p: NCLayerParams):
    """ops.py:48-58.  ``dilation_rate`` is accepted by the reference and never used.  Returns (residual, skip, a)."""
    a = np.maximum(conv1d_same(np.maximum(x, 0), p.w) + p.b, 0)
    return a @ p.wr + p.br, a @ p.ws + p.bs, a


def encoder_forward(ep: EncoderParams, inputs: np.ndarray, pool_stride: int) -> np.ndarray:
    """createEncoder (model.py:136-156): inputs [B,T] -> encoding [B, T//pool_stride, latent]."""
    h, _, _ = residual_dilation_layer_nc(inputs[:, :, None], ep.nc)
    total = None
    for p in ep.layers:
        h, skip, _ = residual_dilation_layer_nc(h, p)
        total = skip if total is None else total + skip
    reduced = total @ ep.lat_w + ep.lat_b                                        # model.py:152
    B, T, C = reduced.shape
    E = T // pool_stride
    return reduced[:, :E * pool_stride].reshape(B, E, pool_stride, C).mean(2)    # tf.nn.pool AVG VALID, model.py:154


def with_conditions(encoding: np.ndarray, conditions: Optional[np.ndarray]) -> np.ndarray:
    """model.py:161-167: tile the clip-level condition over frames and concatenate."""
    if conditions is None:
        return encoding
    c = np.repeat(conditions[:, None, :], encoding.shape[1], axis=1)
    return np.concatenate([encoding, c], axis=2)


def autoencoder_forward(ep: EncoderParams, dp_: StackParams, inputs: np.ndarray, pool_stride: int,
                        conditions: Optional[np.ndarray] = None):
    """createNetwork (model.py:203-216) + the loss (model.py:103,114): labels = inputs, decoder fed RightShift(inputs)."""
    enc = encoder_forward(ep, inputs, pool_stride)
    logits, _ = stack_forward(dp_, inputs, shift_input=True, cond=with_conditions(enc, conditions),
                              pool_stride=pool_stride)
    return dict(encoding=enc, logits=logits, loss=mol_loss(inputs, logits))


def mol_sample(l: np.ndarray, u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    """sample_from_discretized_mix_logistic (ops.py:178-201) with the two uniform draws given (the reference draws
    them from U(1e-5, 1-1e-5)): l [B,T,4M], u1 [B,T,M], u2 [B,T] -> x [B,T] in [-1,1]."""
    M = l.shape[-1] // 4
    sel = np.argmax(l[..., :M] - np.log(-np.log(u1)), axis=-1)                   # ops.py:187
    mean = np.take_along_axis(l[..., M:2 * M], sel[..., None], -1)[..., 0]
    ls = np.maximum(np.take_along_axis(l[..., 2 * M:3 * M], sel[..., None], -1)[..., 0], -7.0)
    x = mean + np.exp(ls) * (np.log(u2) - np.log(1.0 - u2))                      # ops.py:197
    return np.minimum(np.maximum(x, -1.0), 1.0)
