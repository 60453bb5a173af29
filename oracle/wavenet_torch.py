"""CPU oracle (ii): torch-CPU restatement (``F.conv1d`` + autograd) of the same graph.

TEST INFRASTRUCTURE ONLY (see oracle/wavenet_np.py header).  Independent of
oracle (i): convolutions go through ``torch.nn.functional.conv1d`` (weights
``[Cout,Cin,K] = tf_filters.permute(2,1,0)``, explicit left pad ``d*(K-1)``) and
gradients come from autograd instead of the hand-written backward.  Also the
``cpu_baseline`` of bench.py ("port": CPU restatement of the reference graph --
TensorFlow itself is not installable here).

Parity pinning: same statement as oracle/wavenet_np.py -- a1 pinned by the
reference's fixed self-check inputs, everything else **parity unpinned** by the
reference and pinned by oracle (i) == oracle (ii).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SQRT_HALF = 0.7071067811865476  # ops.py:40


def causal_conv(x_bct: torch.Tensor, w_kio: torch.Tensor, d: int) -> torch.Tensor:
    """ops.py:6-10 on channels-first tensors: pad left d*(K-1), VALID, dilation d."""
    K = w_kio.shape[0]
    w = w_kio.permute(2, 1, 0)  # [Cout, Cin, K]
    return F.conv1d(F.pad(x_bct, (d * (K - 1), 0)), w, dilation=d)


class TorchStack:
    """Holds the parameters of oracle (i)'s StackParams as leaf tensors."""

    def __init__(self, sp, dtype=torch.float64, requires_grad=True):
        def t(a):
            return None if a is None else torch.tensor(np.asarray(a), dtype=dtype, requires_grad=requires_grad)

        self.dilations = tuple(sp.dilations)
        self.init_w, self.init_b = t(sp.init_w), t(sp.init_b)
        self.layers = [dict(wf=t(l.wf), bf=t(l.bf), wg=t(l.wg), bg=t(l.bg), wr=t(l.wr), br=t(l.br),
                            ws=t(l.ws), bs=t(l.bs), wc=t(l.wc), bc=t(l.bc)) for l in sp.layers]
        self.head_w1, self.head_b1 = t(sp.head_w1), t(sp.head_b1)
        self.head_w2, self.head_b2 = t(sp.head_w2), t(sp.head_b2)
        self.dtype = dtype

    def named(self, include_cond: bool):
        out = [("init_w", self.init_w), ("init_b", self.init_b)]
        for i, l in enumerate(self.layers):
            out += [(f"l{i}.wf", l["wf"]), (f"l{i}.bf", l["bf"]), (f"l{i}.wr", l["wr"]), (f"l{i}.br", l["br"]),
                    (f"l{i}.ws", l["ws"]), (f"l{i}.bs", l["bs"])]
            if include_cond:
                out += [(f"l{i}.wc", l["wc"]), (f"l{i}.bc", l["bc"])]
        out += [("head_w1", self.head_w1), ("head_b1", self.head_b1),
                ("head_w2", self.head_w2), ("head_b2", self.head_b2)]
        return out

    def forward(self, audio: torch.Tensor, *, shift_input=False, cond: Optional[torch.Tensor] = None,
                pool_stride: int = 1, gate_mode: str = "reference", return_hidden=False):
        """model.py:33-56 / 158-196 -> per-timestep logits [B,T,C]."""
        x0 = audio[:, None, :]  # [B,1,T]
        if shift_input:  # ops.py:78-80
            x0 = F.pad(x0, (1, 0))[:, :, :-1]
        h = causal_conv(x0, self.init_w, 1) + self.init_b[None, :, None]
        total = None
        hidden = []
        for l, d in zip(self.layers, self.dilations):
            if cond is not None:
                cb = cond @ l["wc"] + l["bc"]  # [B,E,R]   model.py:180
                up = cb.repeat_interleave(pool_stride, dim=1)  # NN upsample by integer factor (ops.py:64-74)
                h = h + up.transpose(1, 2)
            f = causal_conv(h, l["wf"], d) + l["bf"][None, :, None]
            z = torch.tanh(f)
            if gate_mode == "reference":
                s = torch.sigmoid(z)  # ops.py:33
            else:
                s = torch.sigmoid(causal_conv(h, l["wg"], d) + l["bg"][None, :, None])
            c = z * s
            res = torch.einsum("bnt,nm->bmt", c, l["wr"]) + l["br"][None, :, None]
            skip = torch.einsum("bnt,ns->bst", c, l["ws"]) + l["bs"][None, :, None]
            if return_hidden:
                hidden.append(dict(x=h.transpose(1, 2), z=z.transpose(1, 2)))
            h = (h + res) * SQRT_HALF
            total = skip if total is None else total + skip
        r0 = torch.relu(total)
        a1 = torch.einsum("bst,su->but", r0, self.head_w1) + self.head_b1[None, :, None]
        r1 = torch.relu(a1)
        logits = torch.einsum("bst,sc->bct", r1, self.head_w2) + self.head_b2[None, :, None]
        logits = logits.transpose(1, 2)  # [B,T,C]
        if return_hidden:
            return logits, hidden
        return logits


def loss_pooled(logits_t: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """model.py:24-29,58: mean_B( -sum(labels * log_softmax(mean_T logits)) )."""
    pooled = logits_t.mean(dim=1)
    return -(targets * torch.log_softmax(pooled, dim=-1)).sum(-1).mean()


def loss_per_timestep(logits_t: torch.Tensor, codes: torch.Tensor) -> torch.Tensor:
    """mu-law softmax teacher loss (model.py:100-112, commented-out head): mean over [B,T]."""
    B, T, C = logits_t.shape
    return F.cross_entropy(logits_t.reshape(B * T, C), codes.reshape(B * T).long(), reduction="mean")


class TFAdam:
    """tf.train.AdamOptimizer semantics (see oracle/wavenet_np.adam_step_tf)."""

    def __init__(self, params: List[torch.Tensor], lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.params = params
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            m.mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)
            p.grad = None


def cpu_train_steps(sp, audio_np: np.ndarray, codes_np: np.ndarray, steps: int, threads: int,
                    lr: float = 1e-3) -> Dict[str, float]:
    """fp32 fwd+bwd+Adam on CPU for bench.py's cpu_baseline; returns samples/s."""
    import time

    torch.set_num_threads(threads)
    st = TorchStack(sp, dtype=torch.float32)
    params = [t for _, t in st.named(include_cond=False)]
    opt = TFAdam(params, lr=lr)
    audio = torch.tensor(audio_np, dtype=torch.float32)
    codes = torch.tensor(codes_np, dtype=torch.int64)
    B, T = audio.shape

    def one():
        loss = loss_per_timestep(st.forward(audio, shift_input=True), codes)
        loss.backward()
        opt.step()
        return float(loss.detach())

    one()  # warm-up
    t0 = time.perf_counter()
    last = 0.0
    for _ in range(steps):
        last = one()
    dt = time.perf_counter() - t0
    return dict(samples_per_s=steps * B * T / dt, seconds=dt, loss=last, steps=steps, B=B, T=T)


def mol_loss(x: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    """ops.py:124-175 in torch (autograd gives the reference gradient, incl. tf.where routing)."""
    M = l.shape[-1] // 4
    logit_probs = l[..., :M]
    means = l[..., M:2 * M]
    log_scales = torch.clamp(l[..., 2 * M:3 * M], min=-7.0)
    xx = x[..., None].expand_as(means)
    centered = xx - means
    inv = torch.exp(-log_scales)
    plus_in = inv * (centered + 1.0 / 255.0)
    min_in = inv * (centered - 1.0 / 255.0)
    cdf_plus = torch.sigmoid(plus_in); cdf_min = torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    comp = torch.where(xx < -0.999, log_cdf_plus,
                       torch.where(xx > 0.999, log_one_minus_cdf_min,
                                   torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)),
                                               log_pdf_mid - math.log(127.5))))
    lp = comp + torch.log_softmax(logit_probs, dim=-1)
    return -torch.logsumexp(lp, dim=-1).sum()


def mol_loss_sum(x: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    return mol_loss(x, l)


# --------------------------------------------------------------------------
# Parallel-WaveNet student (model.py:290-537) with autograd
# --------------------------------------------------------------------------
def flow_forward(st: TorchStack, x_in: torch.Tensor, cond: torch.Tensor, pool_stride: int):
    """model.py:415-487 on a TorchStack whose head_w2/head_b2 are the R->2 1x1 (oracle/wavenet_np.init_flow_params)."""
    x0 = F.pad(x_in[:, None, :], (1, 0))[:, :, :-1]
    h = causal_conv(x0, st.init_w, 1) + st.init_b[None, :, None]
    for l, d in zip(st.layers, st.dilations):
        cb = cond @ l["wc"] + l["bc"]
        h = h + cb.repeat_interleave(pool_stride, dim=1).transpose(1, 2)
        z = torch.tanh(causal_conv(h, l["wf"], d) + l["bf"][None, :, None])
        c = z * torch.sigmoid(z)
        h = (h + torch.einsum("bnt,nm->bmt", c, l["wr"]) + l["br"][None, :, None]) * SQRT_HALF
    prm = torch.einsum("brt,rc->btc", torch.relu(h), st.head_w2) + st.head_b2
    scale = torch.exp(prm[..., 0]); mean = prm[..., 1]
    return scale, mean, x_in * scale + mean, prm


def stft_power(x: torch.Tensor, frame_length: int = 512, frame_step: int = 256) -> torch.Tensor:
    """DFT-matrix restatement of model.py:360-368 (independent of np.fft in oracle (i))."""
    B, T = x.shape
    nf = 1 + (T - frame_length) // frame_step
    n = torch.arange(frame_length, dtype=x.dtype)
    win = 0.5 - 0.5 * torch.cos(2 * math.pi * n / frame_length)
    k = torch.arange(frame_length // 2 + 1, dtype=x.dtype)
    ang = 2 * math.pi * torch.outer(n, k) / frame_length
    frames = x.unfold(1, frame_length, frame_step)[:, :nf] * win
    re = frames @ torch.cos(ang); im = -(frames @ torch.sin(ang))
    return (re * re + im * im).mean(1)


def student_loss(flows: List[TorchStack], noise: torch.Tensor, cond: torch.Tensor, pool_stride: int,
                 teacher_logits: torch.Tensor, truth: torch.Tensor, alpha=1.0, beta=1.0, gamma=1.0):
    """model.py:490-535 + 356-379; teacher_logits are constants (stop_gradient, model.py:334)."""
    x = noise
    scales, means = [], []
    for st in flows:
        s, m, x, _ = flow_forward(st, x, cond, pool_stride)
        scales.append(s); means.append(m)
    s_tot = torch.ones_like(noise); mu_tot = torch.zeros_like(noise)
    for i in range(len(flows)):
        s_tot = s_tot * scales[i]
        mu = means[i]
        for j in range(i + 1, len(flows)):
            mu = mu * scales[j]
        mu_tot = mu_tot + mu
    out = torch.minimum(torch.maximum(noise * s_tot + mu_tot, torch.tensor(-1.0, dtype=noise.dtype)),
                        torch.tensor(1.0, dtype=noise.dtype))
    entropy = (torch.log(s_tot) + 2.0).sum()
    diff = stft_power(truth) - stft_power(out)
    power = (diff ** 2).sum() * gamma
    ce = mol_loss(torch.clamp(out, -1, 1), teacher_logits) * beta
    loss = (ce - alpha * entropy + power) / noise.shape[0]
    return dict(loss=loss, power_loss=power, entropy=entropy, cross_entropy=ce, out=out, s_tot=s_tot, mu_tot=mu_tot)


def flow_named(st: TorchStack):
    """Trained variables of one flow (the skip 1x1s get no gradient: model.py:440-449)."""
    out = [("init_w", st.init_w), ("init_b", st.init_b)]
    for i, l in enumerate(st.layers):
        out += [(f"l{i}.wf", l["wf"]), (f"l{i}.bf", l["bf"]), (f"l{i}.wr", l["wr"]), (f"l{i}.br", l["br"]),
                (f"l{i}.wc", l["wc"]), (f"l{i}.bc", l["bc"])]
    out += [("head_w2", st.head_w2), ("head_b2", st.head_b2)]
    return out


# --------------------------------------------------------------------------
# WaveNetAutoEncoder (model.py:75-285) with autograd
# --------------------------------------------------------------------------
class TorchEncoder:
    def __init__(self, ep, dtype=torch.float64):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)
        lay = lambda p: dict(w=t(p.w), b=t(p.b), wr=t(p.wr), br=t(p.br), ws=t(p.ws), bs=t(p.bs))
        self.nc = lay(ep.nc)
        self.layers = [lay(p) for p in ep.layers]
        self.lat_w, self.lat_b = t(ep.lat_w), t(ep.lat_b)

    @staticmethod
    def _nc(x_bct, p):
        """ops.py:48-58 channels-first; SAME padding of a K-tap kernel = (K-1)//2 left, rest right."""
        K = p["w"].shape[0]
        pl = (K - 1) // 2
        a = torch.relu(F.conv1d(F.pad(torch.relu(x_bct), (pl, K - 1 - pl)), p["w"].permute(2, 1, 0)) +
                       p["b"][None, :, None])
        res = torch.einsum("bct,cd->bdt", a, p["wr"]) + p["br"][None, :, None]
        skip = torch.einsum("bct,cs->bst", a, p["ws"]) + p["bs"][None, :, None]
        return res, skip

    def forward(self, inputs: torch.Tensor, pool_stride: int) -> torch.Tensor:
        h, _ = self._nc(inputs[:, None, :], self.nc)
        total = None
        for p in self.layers:
            h, s = self._nc(h, p)
            total = s if total is None else total + s
        red = torch.einsum("bst,sl->btl", total, self.lat_w) + self.lat_b
        B, T, C = red.shape
        E = T // pool_stride
        return red[:, :E * pool_stride].reshape(B, E, pool_stride, C).mean(2)

    def named(self):
        out = [("nc." + k, v) for k, v in self.nc.items()]
        for i, p in enumerate(self.layers):
            out += [(f"e{i}.{k}", v) for k, v in p.items()]
        return out + [("lat_w", self.lat_w), ("lat_b", self.lat_b)]


def autoencoder_loss(enc: TorchEncoder, dec: TorchStack, inputs: torch.Tensor, pool_stride: int,
                     conditions: Optional[torch.Tensor] = None):
    e = enc.forward(inputs, pool_stride)
    cond = e if conditions is None else torch.cat([e, conditions[:, None, :].expand(-1, e.shape[1], -1)], dim=2)
    logits = dec.forward(inputs, shift_input=True, cond=cond, pool_stride=pool_stride)
    return mol_loss(inputs, logits), e, logits
