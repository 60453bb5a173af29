"""Parallel-WaveNet student on one MI355X: inverse-autoregressive flows distilled against a frozen teacher.

Mirrors ``ParallelWaveNet`` (model.py:290-537):

  flow i     x_{i+1} = x_i * exp(p_i[...,0]) + p_i[...,1],  p_i = head(stack_i(RightShift(x_i), encoding))
             (createPartialFlow / createFlow, model.py:415-487: the conditioned residual stack WITHOUT its skip
             path -- model.py:440-449 is commented out -- and a relu -> 1x1 R->2 head)
  out        clip(z * s_tot + mu_tot, -1, 1) (model.py:517-535) -- algebraically x_F, which is what is computed
  loss       (beta * MoL_NLL(out | teacher logits on the TRUTH clip) - alpha * sum(log s_tot + 2)
              + gamma * ||mean_frames|STFT(truth)|^2 - mean_frames|STFT(out)|^2||_F^2) / B      (model.py:356-379)
             The teacher runs on ``inputs_truth`` (input_map, model.py:318-324) and is frozen (stop_gradient,
             model.py:334): the student's gradient enters the cross entropy through x only.
  update     tf.clip_by_global_norm(grads, 1.0) then Adam (model.py:382-401, the train_fast path student.py:107 uses)

All flows share one flat fp32 parameter / gradient / Adam buffer (one all-reduce, one norm, one update).
"""
from __future__ import annotations

import math
from dataclasses import replace
from typing import Dict, List, Optional

import numpy as np
import torch

from . import dp
from . import kernels as K
from ._lib import call
from .engine import SQRT_HALF, Section, StackConfig, WaveNetEngine, _Span


class FlowStorage:
    """Flat fp32 buffers shared by all flows of one student."""

    def __init__(self, n: int, device):
        z = lambda: torch.zeros(n, dtype=torch.float32, device=device)
        self.params, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z()
        self.adam_step = torch.zeros(1, dtype=torch.int64, device=device)


class FlowStack(WaveNetEngine):
    """One flow: ``createPartialFlow`` + the affine transform of ``createFlow`` (model.py:415-487)."""

    def __init__(self, cfg: StackConfig, batch: int, length: int, device="cuda", seed: int = 0,
                 storage: Optional[FlowStorage] = None, slot: int = 0):
        if not cfg.cond_channels:
            raise ValueError("a flow is conditioned on the teacher's encoding (model.py:431): cond_channels > 0")
        cfg = replace(cfg, head_mode="flow", shift_input=True, output_channels=2)
        self._storage, self._slot = storage, slot
        super().__init__(cfg, batch, length, device=device, seed=seed)

    # -- parameters ----------------------------------------------------------------------------------
    @staticmethod
    def layout(cfg: StackConfig) -> Dict[str, Section]:
        L, R, Kw, E = len(cfg.dilations), cfg.dilation_channels, cfg.filter_width, cfg.cond_channels
        secs: Dict[str, Section] = {}
        off = 0
        for name, shape in (("init_w", (Kw, 1, R)), ("init_b", (R,)), ("WF", (L, Kw, R, R)), ("BF", (L, R)),
                            ("WR", (L, R, R)), ("BR", (L, R)), ("WC", (L, E, R)), ("BC", (L, R)),
                            ("flow_w", (R, 2)), ("flow_b", (2,))):   # flow_w | flow_b stay adjacent (one reduce)
            secs[name] = Section(name, off, shape)
            off += secs[name].numel
        return secs

    @staticmethod
    def param_count(cfg: StackConfig) -> int:
        secs = FlowStack.layout(cfg)
        last = secs["flow_b"]
        return last.offset + last.numel

    def _build_params(self, seed):
        self.sections = self.layout(self.cfg)
        self.nparams = self.param_count(self.cfg)
        st = self._storage or FlowStorage(self.nparams, self.dev)
        lo = self._slot * self.nparams
        self.params, self.grads = st.params[lo:lo + self.nparams], st.grads[lo:lo + self.nparams]
        self.adam_m, self.adam_v = st.adam_m[lo:lo + self.nparams], st.adam_v[lo:lo + self.nparams]
        self.adam_step = st.adam_step
        L, R, S, Kw = self.L, self.R, self.S, self.Kw
        # variables the reference creates but never trains in a flow: the dead gate conv (ops.py:31-33) and the
        # unused skip 1x1s (ops.py:44, model.py:440-449); kept for checkpoint interchange
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self.dead_gate = {"WG": f(L, Kw, R, R), "BG": f(L, R)}
        self.dead_skip = {"WS": f(L, R, S), "BS": f(L, S)}
        self.init_parameters(seed)

    def init_parameters(self, seed: int):
        rng = np.random.default_rng(seed)

        def xav(shape, fan_in, fan_out):
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return torch.tensor(rng.uniform(-lim, lim, size=shape), dtype=torch.float32)

        L, R, S, Kw, E = self.L, self.R, self.S, self.Kw, self.E
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, t):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = t.reshape(-1)

        put("init_w", xav((Kw, 1, R), Kw, Kw * R))
        put("WF", xav((L, Kw, R, R), Kw * R, Kw * R))
        put("WR", xav((L, R, R), R, R))
        put("WC", xav((L, E, R), E, R))
        put("flow_w", xav((R, 2), R, 2))
        self.params.copy_(host)
        self.dead_gate["WG"].copy_(xav((L, Kw, R, R), Kw * R, Kw * R))
        self.dead_skip["WS"].copy_(xav((L, R, S), R, S))

    def load_oracle_params(self, sp):
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, arr):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = torch.tensor(np.asarray(arr), dtype=torch.float32).reshape(-1)

        put("init_w", sp.init_w); put("init_b", sp.init_b)
        for nm, f in (("WF", "wf"), ("BF", "bf"), ("WR", "wr"), ("BR", "br"), ("WC", "wc"), ("BC", "bc")):
            put(nm, np.stack([getattr(l, f) for l in sp.layers]))
        put("flow_w", sp.head_w2); put("flow_b", sp.head_b2)
        self.params.copy_(host)
        self.repack()

    def named_tensors(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        out = {"init_w": self.view("init_w", buf), "init_b": self.view("init_b", buf)}
        for i in range(self.L):
            for nm, f in (("WF", "wf"), ("BF", "bf"), ("WR", "wr"), ("BR", "br"), ("WC", "wc"), ("BC", "bc")):
                out[f"l{i}.{f}"] = self.view(nm, buf)[i]
        out["head_w2"] = self.view("flow_w", buf); out["head_b2"] = self.view("flow_b", buf)
        return out

    def tf_variables(self, scope: str, decoder: bool = True) -> Dict[str, torch.Tensor]:
        """Reference names inside ``<scope>`` = 'ParallelWaveNet/Flow{i}/Flow{i}' (model.py:417,468): per layer
        cond = conv1d_{3i}, residual = conv1d_{3i+1}, skip = conv1d_{3i+2}; the head is conv1d_{3L}."""
        cname = lambda j: "conv1d" if j == 0 else "conv1d_%d" % j
        n = self.named_tensors()
        out = {f"{scope}/causal_conv_Kernel": n["init_w"], f"{scope}/causal_conv_Bias": n["init_b"].view(1, 1, -1)}
        for i in range(self.L):
            nm = f"dilated_conv_{i}"
            out[f"{scope}/{nm}_filter/{nm}_Kernel"] = n[f"l{i}.wf"]
            out[f"{scope}/{nm}_filter/{nm}_Bias"] = n[f"l{i}.bf"].view(1, 1, -1)
            out[f"{scope}/{nm}_gate/{nm}_Kernel"] = self.dead_gate["WG"][i]
            out[f"{scope}/{nm}_gate/{nm}_Bias"] = self.dead_gate["BG"][i].view(1, 1, -1)
            out[f"{scope}/{cname(3 * i)}/kernel"] = n[f"l{i}.wc"].unsqueeze(0)
            out[f"{scope}/{cname(3 * i)}/bias"] = n[f"l{i}.bc"]
            out[f"{scope}/{cname(3 * i + 1)}/kernel"] = n[f"l{i}.wr"].unsqueeze(0)
            out[f"{scope}/{cname(3 * i + 1)}/bias"] = n[f"l{i}.br"]
            out[f"{scope}/{cname(3 * i + 2)}/kernel"] = self.dead_skip["WS"][i].unsqueeze(0)
            out[f"{scope}/{cname(3 * i + 2)}/bias"] = self.dead_skip["BS"][i]
        out[f"{scope}/{cname(3 * self.L)}/kernel"] = n["head_w2"].unsqueeze(0)
        out[f"{scope}/{cname(3 * self.L)}/bias"] = n["head_b2"]
        return out

    # -- images / buffers ----------------------------------------------------------------------------
    def _pack_head(self, pk):
        pass   # the R->2 head is a streaming dot product on fp32 weights (srwn_flow_affine_*)

    def _alloc_head_buffers(self):
        N, L, R = self.N, self.L, self.R
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=self.dev)
        self.pooled = self.mol = False
        self.prm = z(N, 2)
        self.x_out = z(N)
        self.dx_in = z(N)
        self.ent_parts = z(K.flow_partials(N))
        self.flow_parts = z(K.flow_partials(N), 2 * R + 2)
        if not self.use_wl:   # legacy per-product weight gradients (dilation_channels = 32)
            self.wg_parts = z(self.nslabs * L * R * R)
            self.wg_bparts = z(self.nslabs * L * R)

    def set_cond(self, cond: torch.Tensor):
        self.cond_in.zero_()
        self.cond_in[:, :self.E].copy_(cond.reshape(self.B * self.frames, self.E))

    # -- forward / backward ------------------------------------------------------------------------------
    def forward(self, x_in: Optional[torch.Tensor] = None):
        """x_in [B,T] fp32 (default: the staged ``self.audio``) -> self.x_out [B*T], self.prm [B*T,2]."""
        B, T, N, L, R = self.B, self.T, self.N, self.L, self.R
        v = self.view
        if x_in is not None:
            self.audio = x_in
        x = self.audio
        K.causal_conv1d_fwd(x.view(B, T, 1), v("init_w"), v("init_b"), 1, 1, out=self.xs[0])   # model.py:423-424
        self._cond_bias_to_input()                                                               # model.py:431-435
        with _Span(self, "flow_fwd_layers"):
            self._stack_fwd(self.cond_all)
        K.flow_affine_fwd(self.xs[L].view(N, R), v("flow_w"), v("flow_b"), x.view(N), self.prm, self.x_out,
                          self.ent_parts)                                                       # model.py:451-483

    def backward(self, dx_out: torch.Tensor, ent_grad: float, join: bool = True):
        """dx_out [B*T] = d loss / d x_out; leaves d loss / d x_in in self.dx_in and parameter gradients.
        join=False leaves the weight-gradient work running on the side stream (the caller joins ``self.side`` later):
        the flow below only needs dx_in, so its dgrad chain starts while this flow's weight gradients finish."""
        B, T, N, L, R, Kw = self.B, self.T, self.N, self.L, self.R, self.Kw
        dt = self.dt
        gp, sec = self.grads.data_ptr(), self.sections
        x = self.audio
        K.flow_affine_bwd(self.xs[L].view(N, R), self.view("flow_w"), self.prm, x.view(N), dx_out, ent_grad,
                          self.gs[L].view(N, R), self.dx_in, self.flow_parts)
        K.reduce_partials(self.flow_parts, self.flow_parts.shape[0], 2 * R + 2, 1, True, 1.0,
                          gp + 4 * sec["flow_w"].offset, 0)
        main = torch.cuda.current_stream()
        overlap = self.overlap and not self.timing and self.use_wl
        side = self.side if overlap else main
        groups = self._wl_groups() if self.use_wl else []
        group_lo = {g[0]: g for g in groups}
        with _Span(self, "flow_bwd_layers"):
            for l0, l1 in (reversed(self.groups) if self.fused_bwd else ()):   # one launch per layer group
                if self.fused_wt:       # ... that also sums the group's weight gradients (srwn_residual_group_bwd_wt)
                    self._group_bwd_wt(l0, l1)
                    continue
                self._group_bwd(l0, l1)
                if overlap:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._wgrad_layers_group(l0, l1)
            for l in (() if self.fused_bwd else range(L - 1, -1, -1)):
                top = l == L - 1
                K.residual_layer_bwd(None if top else self.gs[l + 2], None if top else self.dfs[l + 1],
                                     None if top else self.wptr(self.o_convT[l + 1]), self.gs[l + 1],
                                     self.wptr(self.o_resT[l]), None, None, self.zs[l], self.dfs[l], B, T, R, 0, Kw,
                                     1 if top else self.dil[l + 1], 2 if top else 1, True, dt)
                if l in group_lo:
                    if overlap:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                    with torch.cuda.stream(side):
                        self._wgrad_layers_group(*group_lo[l])
            if not self.fused_bwd:
                K.residual_layer_bwd(self.gs[1], self.dfs[0], self.wptr(self.o_convT[0]), self.gs[0],
                                     None, None, None, None, None, B, T, R, 0, Kw, self.dil[0], 1, False, dt)
        if overlap:   # gs[0] (and every G_l for the conditioning gradients) is complete
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        with torch.cuda.stream(side):
            self._wgrad_layers_finish()
            self._wgrad_input_and_cond()
        # through the input conv and RightShift to the flow input (model.py:423-424)
        K.causal_conv1d_dgrad(self.gs[0], self.view("init_w"), self.dx_in.view(B, T, 1), 1, shift=1, accumulate=True)
        if overlap and join:
            main.wait_stream(side)


class StudentEngine:
    """``ParallelWaveNet``'s graph and training step (model.py:290-401, 490-535) over pre-allocated buffers."""

    def __init__(self, teacher: WaveNetEngine, flow_cfg: StackConfig, num_flows: int, alpha: float = 1.0,
                 beta: float = 1.0, gamma: float = 1.0, learning_rate: float = 1e-3, seed: int = 0,
                 process_group=None):
        if not teacher.mol or not teacher.E or not teacher.cfg.shift_input:
            raise ValueError("the teacher must be the conditioned mixture-of-logistics decoder (model.py:158-200)")
        if flow_cfg.cond_channels != teacher.E or flow_cfg.pool_stride != teacher.cfg.pool_stride:
            raise ValueError("flows and teacher share encoding_w_condition (model.py:318-324): cond_channels/pool_stride differ")
        self.teacher = teacher
        self.B, self.T, self.N = teacher.B, teacher.T, teacher.N
        self.loss_div = float(self.B)      # the loss divides by the number of noise rows fed (model.py:379)
        self.dev = teacher.dev
        if K.stft_frames(self.T) < 1:
            raise ValueError("clips must hold at least one 512-sample STFT frame (model.py:360)")
        self.alpha, self.beta, self.gamma, self.lr = float(alpha), float(beta), float(gamma), float(learning_rate)
        self.pg = process_group
        self.world = dp.world_size(process_group)
        self.F = int(num_flows)
        per = FlowStack.param_count(replace(flow_cfg, head_mode="flow"))
        self.storage = FlowStorage(per * self.F, self.dev)
        self.flows: List[FlowStack] = [FlowStack(flow_cfg, self.B, self.T, device=self.dev, seed=seed + 101 * i,
                                                 storage=self.storage, slot=i) for i in range(self.F)]
        B, T, N = self.B, self.T, self.N
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self.noise, self.truth = z(B, T), z(B, T)
        self.out = z(N)                    # clip(x_F, -1, 1)  (model.py:535)
        self.dx = z(N)                     # d loss / d out, then d loss / d x_F
        nf = K.stft_frames(T)
        self.spec = z(B, nf, 257, 2); self.fpow = z(B, nf, 257)
        self.pow_truth, self.pow_out, self.dpow = z(B, 257), z(B, 257), z(B, 257)
        self.ce_parts = z((N + 255) // 256)
        self.ce, self.power, self.logs = z(1), z(1), z(1)
        self.sq_parts = z(K.sumsq_partials(self.storage.grads.numel()))
        self.clip = z(2)                   # [combined gradient scale, global norm]
        import os as _os
        self.tstream = torch.cuda.Stream() if _os.environ.get("SRWN_OVERLAP", "1") != "0" else None
        self.ent_all = z(self.F, K.flow_partials(N))
        for i, f in enumerate(self.flows):  # flow i reads flow i-1's output in place
            f.audio = self.noise if i == 0 else self.flows[i - 1].x_out.view(B, T)
            f.ent_parts = self.ent_all[i]

    # ------------------------------------------------------------------------------------------------
    def set_inputs(self, noise: torch.Tensor, truth: Optional[torch.Tensor], cond: torch.Tensor):
        """noise [B,T] logistic samples (student.py:100), truth [B,T], cond = encoding_w_condition [B,frames,E]."""
        self.noise.copy_(noise.reshape(self.B, self.T))
        if truth is not None:
            self.truth.copy_(truth.reshape(self.B, self.T))
            self.teacher.set_inputs(self.truth, None, cond)
        for f in self.flows:
            f.set_cond(cond)

    def forward_flows(self):
        """model.py:490-535: the flows and the clipped output; also sum(log s_tot) for the entropy."""
        for f in self.flows:
            f.forward()
        call("srwn_clamp", self.flows[-1].x_out.data_ptr(), self.out.data_ptr(), self.N, -1.0, 1.0, K._stream())
        # sum over flows and rows of prm0 = sum(log s_tot)  (model.py:517-519, 356)
        K.reduce_loss(self.ent_all, self.ent_all.numel(), 1.0, self.logs)

    def forward(self):
        """Teacher logits on the truth clip, the flows, and the three loss terms (model.py:356-379)."""
        B, T, N = self.B, self.T, self.N
        tch = self.teacher
        main = torch.cuda.current_stream()
        if self.tstream is not None:   # the frozen teacher depends on (truth, encoding) only: run it beside the flows
            self.tstream.wait_stream(main)
            with torch.cuda.stream(self.tstream):
                tch.forward(with_loss=False, train=False)              # logits32 [N, 4M] on RightShift(truth); forward only
        else:                                                          # (stop_gradient, model.py:334): no weight-gradient tiles
            tch.forward(with_loss=False, train=False)
        self.forward_flows()
        K.stft_power(self.truth, None, self.fpow, self.pow_truth)      # model.py:360,367
        K.stft_power(self.out.view(B, T), self.spec, self.fpow, self.pow_out)
        K.power_loss(self.pow_truth, self.pow_out, self.gamma, 1.0 / self.loss_div, self.dpow, self.power)
        if self.tstream is not None:
            main.wait_stream(self.tstream)
        K.mol_loss_dx(tch.logits32, self.out, tch.C // 4, self.ce_parts, self.dx, self.beta / self.loss_div)   # model.py:374
        K.reduce_loss(self.ce_parts, self.ce_parts.numel(), 1.0, self.ce)

    def losses(self) -> Dict[str, float]:
        """Host-side combination of the device scalars (model.py:356,371,375-379)."""
        ce, power, logs = float(self.ce.item()), float(self.power.item()), float(self.logs.item())
        entropy = logs + 2.0 * self.N
        return dict(loss=(self.beta * ce - self.alpha * entropy + power) / self.loss_div, power_loss=power,
                    entropy=entropy, cross_entropy=self.beta * ce)

    def backward(self):
        B, T, N = self.B, self.T, self.N
        K.stft_power_bwd(self.spec, self.dpow, self.dx.view(B, T), accumulate=True)
        # tf.minimum/maximum (model.py:535) pass the gradient where the pre-clip value lies in [-1, 1]
        call("srwn_clamp_bwd", self.flows[-1].x_out.data_ptr(), self.dx.data_ptr(), self.dx.data_ptr(), N, -1.0, 1.0,
             K._stream())
        g = self.dx
        for f in reversed(self.flows):
            f.backward(g, -self.alpha / self.loss_div, join=False)
            g = f.dx_in
        main = torch.cuda.current_stream()
        for f in self.flows:   # every flow's weight gradients must be in before the norm / update
            if f.side is not None and f.overlap and not f.timing and f.use_wl:
                main.wait_stream(f.side)

    def allreduce_grads(self):
        dp.allreduce_sum_(self.storage.grads, self.pg)

    def optimizer_step(self):
        """tf.clip_by_global_norm(grads, 1.0) + Adam over every flow's variables (model.py:382-385, 401); under data
        parallelism the flat buffer holds the SUM of the ranks' (loss / local B) gradients -> mean, then clip."""
        st = self.storage
        K.sumsq(st.grads, self.sq_parts)
        K.clip_scale(self.sq_parts, 1.0, 1.0 / self.world, self.clip)
        K.adam_step_scaled(st.params, st.grads, st.adam_m, st.adam_v, st.adam_step, self.lr, self.clip, True)
        for f in self.flows:
            f.repack()

    def train_step(self):
        self.forward()
        self.backward()
        self.allreduce_grads()
        self.optimizer_step()

    def train_per_sample(self):
        """``ParallelWaveNet.train`` (model.py:599-632), the slow path: for every noise row i the graph is run with
        ``inputs = [noise_i]`` against the WHOLE batch of encodings and truths (the 1-row noise broadcasts over them,
        ``h + upsampled`` at model.py:435), its loss divided by 1 row, its gradient clipped to norm 1 on its own;
        the clipped gradients are averaged and applied without further clipping.  Returns (mean loss, mean power)."""
        st = self.storage
        if not hasattr(self, "_acc"):
            self._acc = torch.zeros_like(st.grads)
            self._noise_all = torch.zeros_like(self.noise)
        self._noise_all.copy_(self.noise)
        self._acc.zero_()
        losses, powers = [], []
        keep = self.loss_div
        self.loss_div = 1.0
        try:
            for i in range(self.B):
                self.noise.copy_(self._noise_all[i:i + 1].expand(self.B, -1))
                self.forward()
                self.backward()
                K.sumsq(st.grads, self.sq_parts)
                K.clip_scale(self.sq_parts, 1.0, 1.0, self.clip)
                call("srwn_axpy_dev", self._acc.data_ptr(), st.grads.data_ptr(), self.clip.data_ptr(), 1.0 / self.B,
                     st.grads.numel(), K._stream())
                l = self.losses()
                losses.append(l["loss"]); powers.append(l["power_loss"])
        finally:
            self.loss_div = keep
            self.noise.copy_(self._noise_all)
        st.grads.copy_(self._acc)
        self.allreduce_grads()
        K.adam_step(st.params, st.grads, st.adam_m, st.adam_v, st.adam_step, self.lr, grad_scale=1.0 / self.world)
        for f in self.flows:
            f.repack()
        return float(np.mean(losses)), float(np.mean(powers))

    def capture_graphs(self):
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self.forward()
            self.backward()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool()):
            self.optimizer_step()
        torch.cuda.synchronize()

    def train_step_graphed(self):
        self._g_fb.replay()
        self.allreduce_grads()
        self._g_opt.replay()
