"""Training/inference engine for the WaveNet residual stack on one MI355X (one process per GPU).

Mirrors the graph bodies of the reference's model classes -- ``WaveNet.createNetwork``
(model.py:33-62) and ``WaveNetAutoEncoder.createDecoder`` (model.py:158-200) -- as a fixed
sequence of libsrwn.so kernel launches over pre-allocated HBM buffers:

  forward   input conv -> L x fused residual layer (h, z saved) -> skip sum as ONE K = L*R
            contraction over the saved z -> relu -> 1x1 -> relu -> last 1x1 + softmax-CE (fused)
  backward  head data gradients -> L x fused layer data gradient (top down) -> batched weight
            gradients (time-contraction MFMA GEMMs + deterministic slab reduction)
  update    [RCCL all-reduce of the flat fp32 gradient buffer] -> TF-Adam -> re-pack bf16 weights

HBM layout (channels-last, compute dtype): xs [L+1,B,T,R] layer inputs, zs [L,B,T,R] tanh
outputs, dfs [L,B,T,R], gs [L+1,B,T,R] residual-stream gradients (gs[L] stays zero), r0/r1/da1/
dtotal [B*T,S], dlogits [B*T,Cp].  Parameters, gradients and Adam moments are single flat fp32
buffers laid out struct-of-arrays across layers so every batched kernel sees a constant stride.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dp
from . import kernels as K
from . import packing as P

SQRT_HALF = 0.7071067811865476


@dataclass
class StackConfig:
    dilations: Sequence[int]
    filter_width: int = 2
    dilation_channels: int = 32   # R
    skip_channels: int = 256      # S
    output_channels: int = 256    # C (softmax classes / head width)
    cond_channels: int = 0        # channels of encoding_w_condition (model.py:161-167); 0 = no conditioning
    pool_stride: int = 1
    shift_input: bool = False     # RightShift(truth) teacher forcing (model.py:172)
    head_mode: str = "per_timestep"  # "per_timestep": mu-law softmax CE per sample (model.py:100-112);
    #                                  "pooled": class WaveNet's clip-level softmax (model.py:56-60, 24-29);
    #                                  "mol": discretised mixture of logistics, the live teacher's loss
    #                                         (model.py:114,196; ops.py:124-175): output_channels = 4*mixtures,
    #                                         loss SUMMED over batch and time
    #                                  "flow": no skip path, head relu -> 1x1 R->2 + affine transform: one flow of
    #                                         ParallelWaveNet (model.py:415-487), see student.FlowStack
    dtype: torch.dtype = torch.bfloat16
    learning_rate: float = 1e-3


class Section:
    __slots__ = ("name", "offset", "shape", "numel")

    def __init__(self, name, offset, shape):
        self.name, self.offset, self.shape = name, offset, tuple(shape)
        self.numel = int(np.prod(shape))


def _os_environ_flag(name: str, default: bool) -> bool:
    import os
    return os.environ.get(name, "1" if default else "0") != "0"


class _Span:
    """Optional HIP-event bracket around a group of launches on the current stream (bench.py roofline)."""

    def __init__(self, eng, name):
        self.eng, self.name = eng, name

    def __enter__(self):
        if self.eng.timing or self.eng.timing_overlap:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if self.eng.timing or self.eng.timing_overlap:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.eng.spans.setdefault(self.name, []).append((self.s, e))
        return False


class WaveNetEngine:
    def __init__(self, cfg: StackConfig, batch: int, length: int, device="cuda", seed: int = 0,
                 process_group=None, share_from: Optional["WaveNetEngine"] = None, frozen: bool = False):
        if cfg.filter_width != 2:
            raise NotImplementedError("filter_width %d: only 2 is built (reference default, model.py:9)" % cfg.filter_width)
        if cfg.dilation_channels not in (32, 64):
            raise NotImplementedError("dilation_channels %d: built for 32 and 64" % cfg.dilation_channels)
        if cfg.skip_channels % 32 or cfg.skip_channels < 32:
            raise NotImplementedError("skip_channels must be a multiple of 32")
        if cfg.output_channels < 1 or cfg.output_channels > 256:
            raise NotImplementedError("output_channels must be in [1, 256]")
        if cfg.head_mode == "mol" and (cfg.output_channels % 4 or not 4 <= cfg.output_channels <= 64):
            raise ValueError("mol head: output_channels = 4 * num_mixtures (<= 16 mixtures)")
        if cfg.head_mode not in ("per_timestep", "pooled", "mol", "flow"):
            raise ValueError("head_mode %r" % cfg.head_mode)
        if cfg.cond_channels and (length % cfg.pool_stride):
            raise ValueError("length %d is not a multiple of pool_stride %d" % (length, cfg.pool_stride))
        self.cfg = cfg
        self.timing = False           # HIP-event spans with every launch alone on the chip (no side stream)
        self.timing_overlap = False   # the same spans inside the real schedule (side-stream work left running)
        self.spans: Dict[str, list] = {}
        import os as _os
        # multi-layer kernels (csrc/srwn_group.hip): SRWN_FUSE=0 keeps one launch per layer (the parity twin)
        fuse = _os.environ.get("SRWN_FUSE", "1")
        self.fuse_fwd = fuse not in ("0", "bwd")
        self.fuse_bwd = fuse not in ("0", "fwd")
        # SRWN_FUSE_WT=1 (default): layer weight gradients inside the 8-wave backward group kernel, split by output over
        # the waves; the forward group kernel writes the transposed operands ("weight-gradient tiles") they need
        # (csrc/srwn_group.hip, _wt entry points).  0: chain kernel + separate weight-gradient pass (the parity twin)
        self.fuse_wt = _os.environ.get("SRWN_FUSE_WT", "1") != "0"
        self.wt_store_x = _os.environ.get("SRWN_WT_STORE_X", "0") != "0"
        # SRWN_FUSE_IC (default on): the input conv inside the first layer group's forward kernel (unconditioned stacks)
        self.fuse_ic = _os.environ.get("SRWN_FUSE_IC", "1") != "0"
        # frozen: a stack that is never trained (a distillation teacher, model.py:334): fixed before allocation, so no
        # weight-gradient tiles / per-workgroup partial slabs are allocated for it and backward() refuses to run.  A
        # trainable stack can still be run forward-only (forward(train=False): the student does that to its teacher).
        self.frozen = bool(frozen)
        self._tiles_valid = False
        # weight-gradient passes on a side stream beside the data-gradient chain: worth 11 % with one launch per layer
        # (short latency-bound chain kernels), but with the group kernels every kernel of the backward phase is
        # bandwidth-bound and running two at once is slower than one after the other (2.15 vs 2.13 ms; the skip data
        # gradient beside the skip weight gradient: 446 us together, 347 us in turn) -> off by default there.
        # (decided on whether the grouped backward actually RUNS for this stack -- a 64/128 stack, say, keeps the per-layer
        # chain even with SRWN_FUSE=1 and wants the overlap)
        will_group = (self.fuse_bwd and cfg.dilation_channels in (32, 64) and cfg.filter_width == 2 and
                      ((cfg.dilation_channels, cfg.skip_channels) in ((64, 256), (32, 128)) or cfg.head_mode == "flow"))
        self.overlap = _os.environ.get("SRWN_OVERLAP", "0" if will_group else "1") != "0"
        self.seg_rows = int(_os.environ.get("SRWN_SEG_ROWS", "0"))
        # head 1x1 + softmax-CE + head data gradients as one launch (SRWN_HEAD_CHAIN=0: the four separate ones)
        self.head_chain = (_os.environ.get("SRWN_HEAD_CHAIN", "1") != "0" and cfg.head_mode == "per_timestep"
                           and cfg.dtype == torch.bfloat16 and cfg.skip_channels == 256)
        self._head_bwd_done = False
        self._ic_job = None
        self._loss_job = None
        self.defer_loss = _os.environ.get("SRWN_DEFER_LOSS", "1") != "0"      # (0: the loss sum keeps a launch of its own)
        self.side = None
        if torch.cuda.is_available() and self.overlap:
            # weight-gradient passes are throughput work: lowest priority, so the latency-critical dgrad
            # chain on the main stream gets CU slots first
            self.side = torch.cuda.Stream(priority=0)
        self.B, self.T = int(batch), int(length)
        self.N = self.B * self.T
        self.L = len(cfg.dilations)
        self.dil = [int(d) for d in cfg.dilations]
        self.R, self.S, self.C = cfg.dilation_channels, cfg.skip_channels, cfg.output_channels
        self.Kw = cfg.filter_width
        self.Cp = (self.C + 31) // 32 * 32
        self.E = cfg.cond_channels
        self.Ep = (self.E + 15) // 16 * 16
        self.frames = self.T // cfg.pool_stride if self.E else 0
        self.dev = torch.device(device)
        self.dt = cfg.dtype
        gl = int(_os.environ.get("SRWN_GROUP_LAYERS", "8"))
        # longest runs with a halo of at most one tile.  (SRWN_GROUP_PLAN=auto: the cost-model cut of
        # srwn_group_plan_auto -- measured slower on the benchmark shape: every extra launch costs ~14 us of launch,
        # segment prologue and store drain, more than the fuller tile rounds of shorter groups give back.)
        if _os.environ.get("SRWN_GROUP_PLAN", "greedy") == "auto" and self.R in (32, 64):
            self.groups = K.group_plan_auto(self.dil, self.B, self.T, self.R, self.dt, gl)
        else:
            self.groups = K.group_plan(self.dil, 31, gl)
        self.pg = process_group
        self.world = dp.world_size(process_group)
        if share_from is None:
            self._build_params(seed)
            self._build_packing()
        else:   # another (batch, length) view of the same model: parameters, moments, images are shared
            for a in ("sections", "nparams", "params", "grads", "adam_m", "adam_v", "adam_step", "bs_sum", "dead_gate",
                      "packer", "packed", "pack_train_elems", "o_conv", "o_res", "o_convT", "o_resT", "o_skipT", "o_skipT_all", "o_skip", "o_gen", "o_skip_gen",
                      "o_w1", "o_w2",
                      "o_w1T", "o_w2T", "o_w2p", "o_w2Tp", "o_w1Tp"):
                setattr(self, a, getattr(share_from, a))
            if self.E:
                self.o_wc = share_from.o_wc
        self._alloc_buffers()
        self.repack()

    # ------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------
    def _build_params(self, seed):
        L, R, S, Kw, Cp, E = self.L, self.R, self.S, self.Kw, self.Cp, self.E
        secs: Dict[str, Section] = {}
        off = 0

        def add(name, shape):
            nonlocal off
            secs[name] = Section(name, off, shape)
            off += secs[name].numel

        add("init_w", (Kw, 1, R)); add("init_b", (R,))
        add("WF", (L, Kw, R, R)); add("BF", (L, R))
        add("WR", (L, R, R)); add("BR", (L, R))
        if E:
            add("WC", (L, E, R)); add("BC", (L, R))
        # the skip and head kernels come last: their gradients are final early in the backward pass and form the
        # first all-reduce bucket (everything from WS to the end), the per-layer gradients above form the second
        add("WS", (L, R, S)); add("BS", (L, S))
        add("head_w1", (S, S)); add("head_b1", (S,))
        add("head_w2", (S, Cp)); add("head_b2", (Cp,))   # padded to Cp columns (pad stays exactly zero)
        self.sections = secs
        self.nparams = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.adam_m = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.adam_v = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.adam_step = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.bs_sum = torch.zeros(S, dtype=torch.float32, device=self.dev)      # sum_l BS[l] (kept current by repack())
        # the dead gate conv variables of ops.py:31-33 exist in reference checkpoints; they take no
        # part in the graph (TF reports None gradients) so they live outside the trained buffer.
        self.dead_gate = {"WG": torch.zeros((L, Kw, R, R), dtype=torch.float32, device=self.dev),
                          "BG": torch.zeros((L, R), dtype=torch.float32, device=self.dev)}
        self.init_parameters(seed)

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.sections[name]
        buf = self.params if buf is None else buf
        return buf[s.offset:s.offset + s.numel].view(s.shape)

    def init_parameters(self, seed: int):
        """Xavier-uniform kernels, zero biases (ops.py:15,18; tf.layers.conv1d defaults)."""
        rng = np.random.default_rng(seed)

        def xav(shape, fan_in, fan_out):
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return torch.tensor(rng.uniform(-lim, lim, size=shape), dtype=torch.float32)

        L, R, S, Kw, C, E = self.L, self.R, self.S, self.Kw, self.C, self.E
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, t):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = t.reshape(-1)

        put("init_w", xav((Kw, 1, R), Kw * 1, Kw * R))
        put("WF", xav((L, Kw, R, R), Kw * R, Kw * R))
        put("WR", xav((L, R, R), R, R))
        put("WS", xav((L, R, S), R, S))
        if E:
            put("WC", xav((L, E, R), E, R))
        put("head_w1", xav((S, S), S, S))
        w2 = torch.zeros((S, self.Cp))
        w2[:, :C] = xav((S, C), S, C)
        put("head_w2", w2)
        self.params.copy_(host)
        self.dead_gate["WG"].copy_(xav((L, Kw, R, R), Kw * R, Kw * R))
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_step.zero_()

    def load_oracle_params(self, sp):
        """Copies an oracle ``StackParams`` (tests) into the flat buffer."""
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, arr):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = torch.tensor(np.asarray(arr), dtype=torch.float32).reshape(-1)

        put("init_w", sp.init_w); put("init_b", sp.init_b)
        put("WF", np.stack([l.wf for l in sp.layers])); put("BF", np.stack([l.bf for l in sp.layers]))
        put("WR", np.stack([l.wr for l in sp.layers])); put("BR", np.stack([l.br for l in sp.layers]))
        put("WS", np.stack([l.ws for l in sp.layers])); put("BS", np.stack([l.bs for l in sp.layers]))
        if self.E:
            put("WC", np.stack([l.wc for l in sp.layers])); put("BC", np.stack([l.bc for l in sp.layers]))
        put("head_w1", sp.head_w1); put("head_b1", sp.head_b1)
        w2 = np.zeros((self.S, self.Cp)); w2[:, :self.C] = sp.head_w2
        b2 = np.zeros(self.Cp); b2[:self.C] = sp.head_b2
        put("head_w2", w2); put("head_b2", b2)
        self.params.copy_(host)
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_step.zero_()
        self.repack()

    def named_tensors(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Oracle-style names (l{i}.wf ...) -> views of `buf` (params by default)."""
        out = {"init_w": self.view("init_w", buf), "init_b": self.view("init_b", buf)}
        for i in range(self.L):
            out[f"l{i}.wf"] = self.view("WF", buf)[i]; out[f"l{i}.bf"] = self.view("BF", buf)[i]
            out[f"l{i}.wr"] = self.view("WR", buf)[i]; out[f"l{i}.br"] = self.view("BR", buf)[i]
            out[f"l{i}.ws"] = self.view("WS", buf)[i]; out[f"l{i}.bs"] = self.view("BS", buf)[i]
            if self.E:
                out[f"l{i}.wc"] = self.view("WC", buf)[i]; out[f"l{i}.bc"] = self.view("BC", buf)[i]
        out["head_w1"] = self.view("head_w1", buf); out["head_b1"] = self.view("head_b1", buf)
        out["head_w2"] = self.view("head_w2", buf)[:, :self.C]; out["head_b2"] = self.view("head_b2", buf)[:self.C]
        return out

    def tf_variables(self, scope: str, decoder: bool) -> Dict[str, torch.Tensor]:
        """Reference variable names -> tensors in TF shapes, for checkpoint interchange."""
        per = 3 if decoder else 2

        def cname(j):
            return "conv1d" if j == 0 else "conv1d_%d" % j

        n = self.named_tensors()
        out = {f"{scope}/causal_conv_Kernel": n["init_w"], f"{scope}/causal_conv_Bias": n["init_b"].view(1, 1, -1)}
        for i in range(self.L):
            nm = f"dilated_conv_{i}"
            out[f"{scope}/{nm}_filter/{nm}_Kernel"] = n[f"l{i}.wf"]
            out[f"{scope}/{nm}_filter/{nm}_Bias"] = n[f"l{i}.bf"].view(1, 1, -1)
            out[f"{scope}/{nm}_gate/{nm}_Kernel"] = self.dead_gate["WG"][i]
            out[f"{scope}/{nm}_gate/{nm}_Bias"] = self.dead_gate["BG"][i].view(1, 1, -1)
            j = per * i
            if decoder:
                out[f"{scope}/{cname(j)}/kernel"] = n[f"l{i}.wc"].unsqueeze(0)
                out[f"{scope}/{cname(j)}/bias"] = n[f"l{i}.bc"]
                j += 1
            out[f"{scope}/{cname(j)}/kernel"] = n[f"l{i}.wr"].unsqueeze(0)
            out[f"{scope}/{cname(j)}/bias"] = n[f"l{i}.br"]
            out[f"{scope}/{cname(j + 1)}/kernel"] = n[f"l{i}.ws"].unsqueeze(0)
            out[f"{scope}/{cname(j + 1)}/bias"] = n[f"l{i}.bs"]
        out[f"{scope}/{cname(per * self.L)}/kernel"] = n["head_w1"].unsqueeze(0)
        out[f"{scope}/{cname(per * self.L)}/bias"] = n["head_b1"]
        out[f"{scope}/{cname(per * self.L + 1)}/kernel"] = n["head_w2"].unsqueeze(0)
        out[f"{scope}/{cname(per * self.L + 1)}/bias"] = n["head_b2"]
        return out

    # ------------------------------------------------------------------------------------------
    # MFMA weight images
    # ------------------------------------------------------------------------------------------
    def _build_packing(self):
        L, R, S, Kw, Cp, E, Ep = self.L, self.R, self.S, self.Kw, self.Cp, self.E, self.Ep
        sec = self.sections
        pk = K.Packer(self.dev)
        self._pack_stack(pk)
        self.pack_train_elems = None
        self._pack_head(pk)
        if self.pack_train_elems is None:      # (a head without generation-only images)
            self.pack_train_elems = pk.total
        pk.finalize()
        self.packer = pk
        self.packed = torch.zeros(max(pk.total, 1), dtype=self.dt, device=self.dev)

    def _pack_stack(self, pk):
        """Per-layer images of the residual stack: conv, 1x1 residual, their transposes, conditioning 1x1s."""
        L, R, Kw, E, Ep = self.L, self.R, self.Kw, self.E, self.Ep
        sec = self.sections
        self.o_conv, self.o_res, self.o_convT, self.o_resT = [], [], [], []
        for l in range(L):
            self.o_conv.append(P.pack_conv(pk, sec["WF"].offset + l * Kw * R * R, Kw, R))
            self.o_res.append(P.pack_res(pk, sec["WR"].offset + l * R * R, R))
            self.o_convT.append(P.pack_conv_T(pk, sec["WF"].offset + l * Kw * R * R, Kw, R))
            self.o_resT.append(P.pack_linear_T(pk, sec["WR"].offset + l * R * R, R, R, R, perm=True))
        if E:
            # conditioning 1x1 of every layer as one [Ep] -> [L*R] product (model.py:180)
            self.o_wc = pk.reserve(L * R // 32, Ep // 16)
            for l in range(L):
                P.fill_linear(pk, self.o_wc + l * (R // 32) * (Ep // 16) * 512, sec["WC"].offset + l * E * R, E, R,
                              R // 32, Ep // 16)

    def _pack_head(self, pk):
        L, R, S, Kw, Cp = self.L, self.R, self.S, self.Kw, self.Cp
        sec = self.sections
        self.o_skipT = []
        # transposed skip kernels of all layers back to back (srwn_skip_dgrad_all streams them in order)
        per = (R // 32) * (S // 16) * 512
        self.o_skipT_all = pk.reserve(L * (R // 32), S // 16)
        for l in range(L):
            P.fill_linear_T(pk, self.o_skipT_all + l * per, sec["WS"].offset + l * R * S, R, S, R // 32, S // 16)
            self.o_skipT.append(self.o_skipT_all + l * per)
        # all skip 1x1s as one image: rows = skip channel, k = layer*R + n
        self.o_skip = pk.reserve(S // 32, L * R // 16)
        for l in range(L):
            P.fill_linear(pk, self.o_skip, sec["WS"].offset + l * R * S, R, S, S // 32, L * R // 16,
                          ks_offset=l * R // 16, ks_count=R // 16)
        self.o_w1 = P.pack_linear(pk, sec["head_w1"].offset, S, S, S)
        self.o_w2 = P.pack_linear(pk, sec["head_w2"].offset, S, Cp, Cp)
        self.o_w1T = P.pack_linear_T(pk, sec["head_w1"].offset, S, S, S)
        self.o_w2T = P.pack_linear_T(pk, sec["head_w2"].offset, S, Cp, S)
        # the same three in the accumulator's k order, for the one-launch head (csrc/srwn_head.hip)
        self.o_w2p = self.o_w2Tp = self.o_w1Tp = None
        if S == 256 and Cp == 256:
            self.o_w2p = P.pack_linear(pk, sec["head_w2"].offset, S, Cp, Cp, perm=True)
            self.o_w2Tp = P.pack_linear_T(pk, sec["head_w2"].offset, S, Cp, S, perm=True)
            self.o_w1Tp = P.pack_linear_T(pk, sec["head_w1"].offset, S, S, S, perm=True)
        # ---- everything above is read by the training step and re-gathered after every optimizer step; the images below
        # serve generate() only and are re-gathered there (they are 45 % of the image: 3.7 of 8.1 MB for config 2)
        self.pack_train_elems = pk.total
        # generation images: per layer [conv (last tap permuted) | residual], back to back (srwn_generate)
        self.o_gen = self.o_skip_gen = None
        if R in (32, 64) and S in (128, 256) and Kw == 2:
            for l in range(L):
                o = P.pack_conv_gen(pk, sec["WF"].offset + l * Kw * R * R, Kw, R)
                P.pack_res(pk, sec["WR"].offset + l * R * R, R)
                if l == 0:
                    self.o_gen = o
            # skip kernels for generation: B operand is the gate tile in registers -> permuted k order
            self.o_skip_gen = pk.reserve(S // 32, L * R // 16)
            for l in range(L):
                P.fill_linear(pk, self.o_skip_gen, sec["WS"].offset + l * R * S, R, S, S // 32, L * R // 16,
                              ks_offset=l * R // 16, ks_count=R // 16, perm=True)
        # the latency-optimised generator's fragment images (csrc/srwn_gen16.hip)
        self.o_g16 = None
        if (self.o_gen is not None and R in (32, 64) and S in (128, 256) and self.dt == torch.bfloat16 and L <= 64
                and ((self.cfg.head_mode == "per_timestep" and not self.E) or self.cfg.head_mode == "mol")):
            self.o_g16 = pk.reserve_raw(np.concatenate([P.gen16_layer_index(sec["WF"].offset, sec["WR"].offset,
                                                                            sec["WS"].offset, l, R, S) for l in range(L)]))
            self.o_g16_h1 = pk.reserve_raw(P.gen16_head_index(sec["head_w1"].offset, S, S, S))
            self.o_g16_h2 = pk.reserve_raw(P.gen16_head_index(sec["head_w2"].offset, S, Cp, Cp, interleave=True))

    def wptr(self, off: int) -> int:
        return self.packed.data_ptr() + off * self.packed.element_size()

    def repack(self):
        """Rebuilds what the kernels derive from the parameters: the weight images the training / forward kernels read (the
        generation-only tail: `_repack_generation`) and, in the same launch, the sum of the layers' skip biases (the bias of
        the skip sum, model.py:50: it was a reduction launch in front of every forward pass)."""
        bs = getattr(self, "bs_sum", None)      # (the flows of the student carry no skip path)
        rs = (self.view("BS"), bs) if (bs is not None and "BS" in self.sections) else None
        self.packer.gather(self.params, self.packed, 0, getattr(self, "pack_train_elems", None), rowsum=rs)

    def _repack_generation(self):
        n = getattr(self, "pack_train_elems", None)
        if n is not None and n < self.packer.total:
            self.packer.gather(self.params, self.packed, n, None)

    # ------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------
    def _alloc_buffers(self):
        self._alloc_stack_buffers()
        self._alloc_head_buffers()

    def _alloc_stack_buffers(self):
        """Saved activations / gradients of the residual stack and its weight-gradient scratch."""
        B, T, N, L, R = self.B, self.T, self.N, self.L, self.R
        z = lambda *s, dt=self.dt: torch.zeros(s, dtype=dt, device=self.dev)
        self.audio = z(B, T, dt=torch.float32)
        self.xs = z(L + 1, B, T, R)
        self.zs = z(L, B, T, R)
        self.dfs = z(L, B, T, R)
        self.gs = z(L + 1, B, T, R)   # gs[L] is never written by the teacher: its last dense output is unused
        self.nslabs = K.wgrad_slabs(N)
        self.use_wl = (R in (32, 64) and self.Kw == 2)
        self.use_dcs = (R, self.S) in ((64, 256), (32, 128))
        import os as _os
        # decided ONCE, before anything is sized (the tiles, the partial slabs and the skip weight-gradient path follow it)
        self._fused_wt = (self.fuse_wt and self.fuse_fwd and self.fused_bwd and self.cfg.head_mode != "flow"
                          and not self.frozen)
        if self.fused_wt:
            # weight-gradient tiles of every layer (written by the forward group kernels, read by the backward ones) and one
            # partial slab per workgroup of the backward kernels (slabs a group does not reach stay zero)
            geo = [K.group_wt_geometry(self.dil[l0:l1], B, T, R, self.dt, self.seg_rows) for l0, l1 in self.groups]
            self.wt_seg_rows = [g[0] for g in geo]
            wt_elems = max(g[2] for g in geo)
            self.xTs = z(L, wt_elems)
            self.cTs = z(L, wt_elems)
            self.nslabs = max(g[3] for g in geo)
            # per layer: the stride and segment length of its group (what fixes the positions its tiles hold)
            self.wt_layer_st = [math.gcd(*self.dil[l0:l1]) for l0, l1 in self.groups for _ in range(l0, l1)]
            self.wt_layer_seg = [self.wt_seg_rows[i] for i, (l0, l1) in enumerate(self.groups) for _ in range(l0, l1)]
        elif self.use_wl and self.fuse_bwd and "SRWN_WG_SLAB_ROWS" not in _os.environ and self.groups:
            # the layer weight-gradient pass is launched per layer group with one workgroup per (layer, slab): cut the
            # rows so that the widest group's launch is one workgroup per CU (5-layer groups at 3072 rows per slab left
            # 46 of 256 CUs idle: 578 -> 503 us per step), slabs of at least 256 rows
            cus = torch.cuda.get_device_properties(self.dev).multi_processor_count if torch.cuda.is_available() else 256
            widest = max(l1 - l0 for l0, l1 in self.groups)
            self.nslabs = int(max(1, min(cus // widest, (N + 255) // 256, 256)))
        # SRWN_PART16 (default on; bf16 weight-gradient-tile mode only): the per-workgroup partial sums of the conv-tap and
        # residual 1x1 weight gradients are stored in the compute type (16 x 16 blocks in lane order) instead of fp32 --
        # 256 workgroups x 30 layers x 48 KB = 0.38 GB per step written by the backward group kernels and read back by the
        # reduction, halved, for one more bf16 rounding per partial (measured: +1.3e-4 .. 5.7e-4 relative L2 on those
        # gradients, whose bf16-mode error against the exact-fp32 mode is 7e-3: DESIGN.md 4c)
        self.part16 = (self.fused_wt and self.dt == torch.bfloat16 and _os.environ.get("SRWN_PART16", "1") != "0")
        if self.use_wl:
            ns = self.nslabs
            pdt = torch.bfloat16 if self.part16 else torch.float32
            self.pl_f = z(L * ns * 2 * R * R, dt=pdt); self.pl_r = z(L * ns * R * R, dt=pdt)
            self.pl_bf = z(L * ns * R, dt=torch.float32); self.pl_br = z(L * ns * R, dt=torch.float32)
        from . import _lib
        # the input conv's weight-gradient partials: srwn_init_conv_wgrad's stage-1 slabs, or -- default path, unconditioned
        # stacks with a skip path -- one slab per workgroup of the FIRST group's backward launch, which forms them from its
        # bottom gradient while it is on the chip (no launch of its own)
        self.fuse_icg = (self.fused_wt and self.fuse_ic and not self.E and self.Kw == 2 and self.use_dcs
                         and (self.part16 or self.dt == torch.float32))
        self.ic_ws = z(max(int(_lib.load().srwn_init_conv_wgrad_partials(B, T, R, self.Kw)),
                           self.nslabs * (8 // (R // 16)) * (self.Kw + 1) * R if self.fuse_icg else 0), dt=torch.float32)
        if self.E:
            self.cond_in = z(B * self.frames, self.Ep)
            self.cond_all = z(L, B * self.frames, R)      # cb of every layer, layer by layer: a layer's frame rows are dense
            self.dcb = z(L, B * self.frames, R)
            # (few rows: one slab would be L workgroups walking B*frames rows in 32-row steps -- 69 us for 31 MFLOP at
            # 1 024 rows; 128-row slabs fill the chip)
            rows_c = B * self.frames
            self.nslabs_c = max(K.wgrad_slabs(rows_c), min(max(1, 256 // L), max(1, rows_c // 128)))
            self.wgc_parts = z(self.nslabs_c * L * self.Ep * R, dt=torch.float32)
            self.wgc_bparts = z(self.nslabs_c * L * R, dt=torch.float32)
            self.wc_grad_pad = z(L, self.Ep, R, dt=torch.float32)

    def _alloc_head_buffers(self):
        B, T, N, L, R, S, Cp = self.B, self.T, self.N, self.L, self.R, self.S, self.Cp
        z = lambda *s, dt=self.dt: torch.zeros(s, dtype=dt, device=self.dev)
        self.targets = torch.zeros(N, dtype=torch.int32, device=self.dev)
        if self.use_dcs:
            self.dcs = z(L, B, T, R)  # Ws_l . dtotal of every layer (one output-streaming GEMM)
        self.r0 = z(N, S); self.r1 = z(N, S); self.da1 = z(N, S); self.dtotal = z(N, S)
        self.dlogits = z(N, Cp)
        self.loss_parts = z((N + 31) // 32, dt=torch.float32)
        self.loss = z(1, dt=torch.float32)
        self.pooled = self.cfg.head_mode == "pooled"
        self.mol = self.cfg.head_mode == "mol"
        if self.mol:
            self.logits32 = z(N, Cp, dt=torch.float32)
        if self.pooled:
            from . import _lib as _l
            self.labels = z(B, self.C, dt=torch.float32)
            self.probs = z(B, self.C, dt=torch.float32)
            self.mean_r1 = z(B, S, dt=torch.float32)
            self.dmean = z(B, S, dt=torch.float32)
            self.tm_parts = z(B * int(_l.load().srwn_time_mean_slabs(T)) * S, dt=torch.float32)
        big = max(L * R * S, S * S, S * Cp, L * self.Kw * R * R)
        self.use_w256 = (S in (128, 256) and R in (32, 64) and (L * R) % 64 == 0)
        if self.use_w256:
            self.ns_skip = K.wgrad256_slabs(N, L, R)
            self.ns_head = K.wgrad256_slabs(N, S // 64)
            big = max(big, -(-max(self.ns_skip * L * R * S, self.ns_head * S * 256) // self.nslabs))
        # skip weight gradients from the forward's transposed gate outputs (csrc/srwn_wgradt.hip)
        self.skip_wt = (self.use_w256 and self.fused_wt and self.dt == torch.bfloat16 and (R, S) == (64, 256)
                        and _os_environ_flag("SRWN_WGRAD_WT", True))
        self.skip_parts16 = None
        if self.skip_wt:
            self.ns_skip_wt = K.wgrad_skip_wt_slabs(self.wt_layer_st, self.wt_layer_seg, T)
            if self.part16:      # its partial slabs in the compute type too (a buffer of their own: wg_parts is fp32)
                self.skip_parts16 = z(self.ns_skip_wt * L * R * S)
            else:
                big = max(big, -(-self.ns_skip_wt * L * R * S // self.nslabs))
        self.wg_parts = z(self.nslabs * big, dt=torch.float32)
        self.wg_bparts = z(max(self.nslabs * max(L * S, Cp), 256 * 256), dt=torch.float32)
        # the two head products keep partials of their own, so that skip + head finish in ONE reduction launch
        self.batch_reduce = self.use_w256 and not self.pooled and Cp == 256 and _os_environ_flag("SRWN_BATCH_REDUCE", True)
        if self.batch_reduce:
            self.hd_parts = [z(self.ns_head * S * 256, dt=torch.float32) for _ in range(2)]
            self.hd_bparts = [z(self.ns_head * 256, dt=torch.float32) for _ in range(2)]

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def set_inputs(self, audio: torch.Tensor, targets: Optional[torch.Tensor] = None,
                   cond: Optional[torch.Tensor] = None):
        self.audio.copy_(audio.reshape(self.B, self.T))
        if targets is not None:
            if self.pooled:
                self.labels.copy_(targets.reshape(self.B, self.C))
            else:
                self.targets.copy_(targets.reshape(self.N))
        if self.E:
            if cond is None:
                raise ValueError("this stack was built with conditioning; pass cond [B, frames, cond_channels]")
            self.cond_in.zero_()
            self.cond_in[:, :self.E].copy_(cond.reshape(self.B * self.frames, self.E))

    def forward(self, want_logits: bool = False, with_loss: bool = True, train: bool = True,
                defer_loss: bool = False) -> Optional[torch.Tensor]:
        """Runs the stack on the staged inputs; leaves loss in self.loss and dlogits for backward.
        Returns fp32 per-time-step logits [B,T,C] when want_logits.  train=False: forward only (no weight-gradient tiles
        are written; backward() refuses to follow such a pass).  defer_loss (the training step): the final sum of the loss
        partials is left to the backward pass, where it is one more job of the skip / head reduction launch."""
        B, T, N, L, R, S = self.B, self.T, self.N, self.L, self.R, self.S
        es = self.packed.element_size()
        v = self.view
        # input conv (model.py:40 / 172-173); RightShift folded into the tap offset.  Where the first layer group runs as a
        # group kernel on an unconditioned stack, the conv is computed inside it (its output never reaches HBM: nothing else
        # reads xs[0] unless the inner layer inputs are kept for inspection)
        self._tiles_valid = bool(train) and self.fused_wt
        self._ic_fused = (self.fuse_ic and self._tiles_valid and not self.E and self.Kw == 2 and not self.wt_store_x
                          and bool(self.groups))
        if not self._ic_fused:
            K.causal_conv1d_fwd(self.audio.view(B, T, 1), v("init_w"), v("init_b"), 1,
                                1 if self.cfg.shift_input else 0, out=self.xs[0])
        if self.E:
            self._cond_bias_to_input()
        with _Span(self, "fwd_layers"):
            self._stack_fwd(self.cond_all if self.E else None, wt=self._tiles_valid)
        with _Span(self, "skip_sum"):      # model.py:50-51 (bs_sum = the sum of the layers' skip biases: formed by repack())
            K.pw_linear(self.zs.data_ptr(), R, N * R, R, L * R, self.wptr(self.o_skip), self.bs_sum, self.r0, S, S,
                        N, pro=K.PRO_GATE, epi=K.EPI_RELU)
        self._head_bwd_done = False
        if self.head_chain and self.o_w2p is not None and not want_logits and with_loss:
            # model.py:53-56 + softmax CE + the two head data gradients: rows never leave the registers in between
            with _Span(self, "head_chain"):
                K.head_chain(self.r0, self.wptr(self.o_w1), self.wptr(self.o_w2p), self.wptr(self.o_w2Tp),
                             self.wptr(self.o_w1Tp), v("head_b1"), v("head_b2"), self.targets, self.loss_parts,
                             self.r1, self.dlogits, self.da1, self.dtotal, self.C, 1.0 / N)
            self._head_bwd_done = True
            if defer_loss and train and self.batch_reduce and not self.frozen and self.defer_loss:
                self._loss_job = (self.loss_parts, self.loss_parts.numel(), 1, 1, True, 1.0 / N, self.loss.data_ptr(), 0, "sum")
            else:
                K.reduce_loss(self.loss_parts, self.loss_parts.numel(), 1.0 / N, self.loss)
            return None
        with _Span(self, "head_1x1"):
            K.pw_linear(self.r0.data_ptr(), S, 0, S, S, self.wptr(self.o_w1), v("head_b1"), self.r1, S, S, N,
                        epi=K.EPI_RELU)                                               # model.py:53-54
        if self.pooled:
            return self._forward_pooled_head(with_loss)
        if self.mol:
            # last 1x1 in fp32 (the mixture parameters need it), then the mixture-of-logistics NLL on the
            # UNshifted clip (labels = inputs, model.py:103,114) and its gradient
            with _Span(self, "head_mol"):
                K.pw_linear(self.r1.data_ptr(), S, 0, S, S, self.wptr(self.o_w2), v("head_b2"), self.logits32,
                            self.Cp, self.C, N, epi=K.EPI_F32, compute_dtype=self.dt)
                K.mol_loss(self.logits32, self.audio.view(N), self.C // 4, self.loss_parts, self.dlogits, 1.0)
            if with_loss:
                K.reduce_loss(self.loss_parts, (N + 255) // 256, 1.0, self.loss)
            return self.logits32[:, :self.C].reshape(B, T, self.C).clone() if want_logits else None
        logits = None
        if want_logits:
            logits = torch.empty((N, self.C), dtype=torch.float32, device=self.dev)
        with _Span(self, "head_softmax_ce"):
            K.head_softmax_ce(self.r1, self.wptr(self.o_w2), v("head_b2"), self.targets, self.loss_parts,
                              self.dlogits, logits, self.Cp, self.C, 1.0 / N)          # model.py:56 + softmax CE
        if with_loss:
            K.reduce_loss(self.loss_parts, self.loss_parts.numel(), 1.0 / N, self.loss)
        return None if logits is None else logits.view(B, T, self.C)

    def _forward_pooled_head(self, with_labels: bool):
        """model.py:56-60: last 1x1, average pool over the clip, softmax -- computed as the 1x1 of the
        time-mean (the pool commutes with it); leaves probs [B,C], loss, and dmean for backward."""
        from ._lib import call
        st = torch.cuda.current_stream().cuda_stream
        B, T, S, C, Cp = self.B, self.T, self.S, self.C, self.Cp
        g = self.grads
        call("srwn_time_mean", self.r1.data_ptr(), self.tm_parts.data_ptr(), self.mean_r1.data_ptr(), B, T, S,
             K.abi_dtype(self.dt), st)
        call("srwn_pooled_head", self.mean_r1.data_ptr(), self.view("head_w2").data_ptr(),
             self.view("head_b2").data_ptr(), self.labels.data_ptr() if with_labels else None, self.probs.data_ptr(),
             self.loss.data_ptr(), self.view("head_w2", g).data_ptr(), self.view("head_b2", g).data_ptr(),
             self.dmean.data_ptr(), B, S, C, Cp, st)
        return None

    def _cond_bias_to_input(self):
        """cb_l = 1x1(encoding_w_condition) of every layer as one product (model.py:180), and the first layer's bias
        onto the input conv's output (model.py:181-183).  Every later layer receives its bias from the layer below:
        xs[l] always holds layer l's complete input, so taps, residual base and weight gradients never re-add it."""
        from ._lib import call
        L, R = self.L, self.R
        rows_c = self.B * self.frames
        st = torch.cuda.current_stream().cuda_stream
        call("srwn_pw_linear_ychunks", self.cond_in.data_ptr(), self.Ep, self.Ep, self.wptr(self.o_wc),
             self.view("BC").reshape(-1).data_ptr(), self.cond_all.data_ptr(), R, R, rows_c * R, L * R, L * R, rows_c,
             K.abi_dtype(self.dt), st)
        call("srwn_add_frame_bias", self.xs[0].data_ptr(), self.cond_all.data_ptr(), R, self.B, self.T, R,
             self.frames, self.cfg.pool_stride, K.abi_dtype(self.dt), st)

    def _stack_fwd(self, cond_all: Optional[torch.Tensor], wt: Optional[bool] = None):
        """The residual layers (model.py:42-47 / 176-189 / 428-453): xs[0] -> xs[1..L], zs[0..L-1].
        wt: also write the weight-gradient tiles (default: whenever the backward pass reads them)."""
        wt = self.fused_wt if wt is None else (wt and self.fused_wt)
        if self.fuse_fwd:
            for l0, l1 in self.groups:      # runs of layers whose outputs travel between layers in LDS
                if l1 - l0 >= 2 or wt:
                    self._group_fwd(l0, l1, cond_all, wt)
                else:
                    self._layer_fwd(l0, cond_all)
        else:
            for l in range(self.L):
                self._layer_fwd(l, cond_all)

    def _group_fwd(self, l0: int, l1: int, cond_all: Optional[torch.Tensor], use_wt: bool = True):
        """Layers [l0, l1) in one launch (srwn_residual_group_fwd); same stored xs / zs as the per-layer path."""
        v = self.view
        cond3 = None
        if cond_all is not None:      # layer l adds the bias of layer l + 1 onto its output
            cond3 = [cond_all[l + 1].view(self.B, self.frames, self.R) if l + 1 < self.L else None for l in range(l0, l1)]
        wt = {}
        use_wt = use_wt and self.fused_wt
        if l0 == 0 and getattr(self, "_ic_fused", False):
            if use_wt:
                wt = dict(xT=self.xTs[l0:l1], cT=self.cTs[l0:l1], store_inner_x=self.wt_store_x)
            K.residual_group_fwd_ic(self.audio, v("init_w"), v("init_b"), 1 if self.cfg.shift_input else 0,
                                    self.xs[l0 + 1:l1 + 1], self.zs[l0:l1],
                                    [self.wptr(self.o_conv[l]) for l in range(l0, l1)],
                                    [self.wptr(self.o_res[l]) for l in range(l0, l1)],
                                    [v("BF")[l] for l in range(l0, l1)], [v("BR")[l] for l in range(l0, l1)],
                                    self.dil[l0:l1], self.Kw,
                                    seg_rows=self.wt_seg_rows[self.groups.index((l0, l1))] if use_wt else self.seg_rows, **wt)
            return
        if use_wt:
            # (xs of the layers inside a group is NOT written in this mode: only the weight gradients would read it, and they
            # take the transposed tiles; SRWN_WT_STORE_X=1 keeps it for inspection)
            wt = dict(xT=self.xTs[l0:l1], cT=self.cTs[l0:l1], store_inner_x=self.wt_store_x)
        K.residual_group_fwd(self.xs[l0], self.xs[l0 + 1:l1 + 1], self.zs[l0:l1],
                             [self.wptr(self.o_conv[l]) for l in range(l0, l1)],
                             [self.wptr(self.o_res[l]) for l in range(l0, l1)],
                             [v("BF")[l] for l in range(l0, l1)], [v("BR")[l] for l in range(l0, l1)],
                             self.dil[l0:l1], self.Kw, cond=cond3,
                             pool_stride=self.cfg.pool_stride,
                             seg_rows=self.wt_seg_rows[self.groups.index((l0, l1))] if use_wt else self.seg_rows, **wt)

    def _layer_fwd(self, l: int, cond_all: Optional[torch.Tensor]):
        v = self.view
        nxt = cond_all is not None and l + 1 < self.L      # the NEXT layer's conditioning bias goes onto the output
        cond3 = cond_all[l + 1].view(self.B, self.frames, self.R) if nxt else None
        K.residual_layer_fwd(self.xs[l], cond3, self.wptr(self.o_conv[l]), self.wptr(self.o_res[l]), v("BF")[l],
                             v("BR")[l], self.xs[l + 1], self.zs[l], self.Kw, self.dil[l], self.cfg.pool_stride)

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def backward(self, join: bool = True, part: int = 0):
        """join=False leaves the weight-gradient passes running on ``self.side`` (the caller joins it before the
        optimizer): work that only needs the data gradients can start right away.
        part: 0 = the whole pass; 1 = head + dgrad chain down to layer ``self.split_layer`` and, on return, the skip /
        head gradients are FINAL (first all-reduce bucket, ``grads[bucket_off:]``); 2 = the rest.  Splitting lets the
        data-parallel step all-reduce the first bucket while part 2 runs.
        Data gradients top-down on the current stream; weight gradients on a side stream as soon as their
        operands exist (skip/head kernels right after the head data gradients, per-layer kernels in groups
        behind the dgrad chain), so the bandwidth-bound wgrad passes fill the ramp/tail bubbles of the
        short per-layer dgrad kernels.  Joined before the optimizer.  SRWN_OVERLAP=0 serialises everything."""
        B, T, N, L, R, S, Kw = self.B, self.T, self.N, self.L, self.R, self.S, self.Kw
        dt = self.dt
        if self.frozen:
            raise RuntimeError("backward: this stack was built frozen (forward only)")
        if self.fused_wt and not self._tiles_valid:
            raise RuntimeError("backward: the last forward pass ran with train=False and wrote no weight-gradient tiles")
        main = torch.cuda.current_stream()
        overlap = self.overlap and not self.timing
        side = self.side if overlap else main
        if part in (0, 1):
            self._bwd_head()
            if overlap:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                self._wgrad_skip_and_head()
            if self.use_dcs:
                with _Span(self, "skip_dgrad_all"):
                    K.skip_dgrad_all(self.dtotal, self.wptr(self.o_skipT_all), self.dcs.view(L, N, R), R, S)
        # ---- residual stack, top down
        groups = self._wl_groups() if self.use_wl else []
        group_lo = {g[0]: g for g in groups}
        l_hi = self.split_layer - 1 if part == 2 else L - 1
        l_lo = self.split_layer if part == 1 else 0
        span = _Span(self, "bwd_layers" if part == 0 else "bwd_layers_part%d" % part).__enter__()
        if self.fused_bwd:
            # one launch per group of layers (srwn_residual_group_bwd), top group first; each group's weight-gradient
            # pass follows it on the side stream.  The bottom group writes gs[0]: no UP-only launch below layer 0.
            for l0, l1 in reversed(self.groups):
                if l0 > l_hi or l0 < l_lo:
                    continue
                if self.fused_wt:
                    self._group_bwd_wt(l0, l1)
                    continue
                self._group_bwd(l0, l1)
                if not self.timing:
                    if overlap:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                    with torch.cuda.stream(side):
                        self._wgrad_layers_group(l0, l1)
            span.__exit__()
            if part == 1:
                if overlap:
                    main.wait_stream(side)
                return
            if self.timing and not self.fused_wt:
                for g in groups:
                    self._wgrad_layers_group(*g)
            merged = self.use_wl      # the input conv's slab sum joins the final reduction launch
            if merged and overlap:
                side.wait_stream(main)      # (gs[0], which the input conv's gradient reads, is complete on the main stream)
            with torch.cuda.stream(side):
                if merged and self._ic_job is None:      # (not already left by the first group's backward launch)
                    self._wgrad_input_conv_partials()
                self._wgrad_layers_finish()
            self._wgrad_input_and_cond(input_conv=not merged)
            if overlap and join:
                main.wait_stream(side)
            return
        for l in range(l_hi, l_lo - 1, -1):
            has_up = l < L - 1
            g_in = self.gs[l + 2] if (has_up and l + 2 < L) else None
            K.residual_layer_bwd(g_in, self.dfs[l + 1] if has_up else None,
                                 self.wptr(self.o_convT[l + 1]) if has_up else None,
                                 self.gs[l + 1] if has_up else None,
                                 self.wptr(self.o_resT[l]) if has_up else None,
                                 None if self.use_dcs else self.wptr(self.o_skipT[l]),
                                 None if self.use_dcs else self.dtotal, self.zs[l], self.dfs[l], B, T, R, S, Kw,
                                 self.dil[l + 1] if has_up else 1, has_up, True, dt,
                                 dcs=self.dcs[l] if self.use_dcs else None)
            if l in group_lo and not self.timing:   # df_l.. and G_{l+1}.. of this group are complete
                if overlap:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._wgrad_layers_group(*group_lo[l])
        if part == 1:
            span.__exit__()
            if overlap:
                main.wait_stream(side)   # skip/head gradients (and the layer groups launched so far) are in
            return
        K.residual_layer_bwd(self.gs[1] if L > 1 else None, self.dfs[0], self.wptr(self.o_convT[0]), self.gs[0],
                             None, None, None, None, None, B, T, R, S, Kw, self.dil[0], True, False, dt)
        span.__exit__()
        if self.timing:   # (timed runs keep the dgrad chain's span free of the weight-gradient passes)
            for g in groups:
                self._wgrad_layers_group(*g)
        merged = self.use_wl      # (as in the grouped path: the same sums in the same order)
        if merged and overlap:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            if merged:
                self._wgrad_input_conv_partials()
            self._wgrad_layers_finish()
        self._wgrad_input_and_cond(input_conv=not merged)
        if overlap and join:
            main.wait_stream(side)

    def join_side(self):
        if self.side is not None and self.overlap and not self.timing:
            torch.cuda.current_stream().wait_stream(self.side)

    @property
    def fused_bwd(self) -> bool:
        """The data-gradient chain runs as one launch per layer group (needs the precomputed skip gradients `dcs`, or a
        stack without a skip path)."""
        return self.fuse_bwd and self.use_wl and (getattr(self, "use_dcs", False) or self.cfg.head_mode == "flow")

    @property
    def fused_wt(self) -> bool:
        """The layer weight gradients are summed inside the backward group kernel from the forward kernel's weight-gradient
        tiles (no df / G round trip through HBM, no separate weight-gradient pass)."""
        # (not for the flows of the student: they carry no skip path and write every layer's input gradient for the
        # conditioning 1x1 anyway -- measured: their backward gains nothing and their forward pays for the tiles,
        # 7.09 vs 6.77 ms per distillation step -- and not for a stack that is never trained: StudentEngine's teacher)
        return self._fused_wt

    def _group_bwd_wt(self, l0: int, l1: int):
        """Chain + weight-gradient partials of layers [l0, l1) in one launch (srwn_residual_group_bwd_wt)."""
        flow = self.cfg.head_mode == "flow"
        R, ns = self.R, self.nslabs
        g_top = self.gs[l1] if (flow or l1 < self.L) else None
        ic = None
        if l0 == 0 and self.fuse_icg:      # the stack's first group: the input conv's weight-gradient partials ride along
            ic = (self.audio, self.ic_ws, 1 if self.cfg.shift_input else 0)
            sec = self.sections
            self._ic_job = (self.ic_ws, ns * (8 // (R // 16)), (self.Kw + 1) * R, 1, True, 1.0,
                            self.grads.data_ptr() + 4 * sec["init_w"].offset, 0)
        with _Span(self, "group_bwd_wt"):
            K.residual_group_bwd_wt(g_top, self.gs[l0:l1], self.zs[l0:l1], None if flow else self.dcs[l0:l1],
                                    self.xTs[l0:l1], self.cTs[l0:l1],
                                    [self.wptr(self.o_convT[l]) for l in range(l0, l1)],
                                    [self.wptr(self.o_resT[l]) for l in range(l0, l1)], self.dil[l0:l1],
                                    self.pl_f[l0 * ns * 2 * R * R:], self.pl_r[l0 * ns * R * R:],
                                    self.pl_bf[l0 * ns * R:], self.pl_br[l0 * ns * R:], ns,
                                    self.wt_seg_rows[self.groups.index((l0, l1))], self.Kw, write_all_g=bool(self.E), ic=ic)

    def _group_bwd(self, l0: int, l1: int):
        flow = self.cfg.head_mode == "flow"
        g_top = self.gs[l1] if (flow or l1 < self.L) else None    # the teacher's last dense output is unused: G_L = 0
        K.residual_group_bwd(g_top, self.gs[l0:l1], self.dfs[l0:l1], self.zs[l0:l1],
                             None if flow else self.dcs[l0:l1],
                             [self.wptr(self.o_convT[l]) for l in range(l0, l1)],
                             [self.wptr(self.o_resT[l]) for l in range(l0, l1)], self.dil[l0:l1], self.Kw,
                             seg_rows=self.seg_rows)

    def _wl_groups(self):
        import os as _os
        if self.fused_bwd:
            return list(self.groups)
        per = int(_os.environ.get("SRWN_WL_GROUP", "6"))
        return [(l0, min(l0 + per, self.L)) for l0 in range(0, self.L, per)]

    @property
    def split_layer(self) -> int:
        """Where the two-part backward is cut: the layer-group boundary nearest 40 % of the depth (the part above it
        outlasts the skip/head weight-gradient kernels running beside it)."""
        los = [g[0] for g in self._wl_groups() if 0 < g[0] < self.L]
        return min(los, key=lambda v: abs(v - 0.4 * self.L)) if los else 0

    @property
    def bucket_off(self) -> int:
        return self.sections["WS"].offset

    @property
    def bucketed(self) -> bool:
        """Two all-reduce buckets, the first overlapped with the lower part of the backward pass (needs the grouped
        weight-gradient path and a deep enough stack)."""
        import os as _os
        forced = _os.environ.get("SRWN_FORCE_DIST") == "1"
        mode = _os.environ.get("SRWN_BUCKETS", "auto")   # "0" off, "1" on, "auto": on for RCCL only (gloo's
        if mode == "0" or not ((self.world > 1 or forced) and self.use_wl and self.split_layer > 0 and not self.pooled):
            return False                                 # asynchronous CUDA all-reduce stalls for tens of ms)
        if mode == "1":
            return True
        import torch.distributed as dist
        return dist.is_initialized() and dist.get_backend(self.pg) == "nccl"

    def _bwd_head(self):
        B, T, N, S, Cp = self.B, self.T, self.N, self.S, self.Cp
        if self._head_bwd_done:     # da1, dtotal came out of the forward's head launch
            return
        with _Span(self, "bwd_head"):   # relu masks against the saved activations
            if self.pooled:
                from ._lib import call
                call("srwn_bcast_mask", self.dmean.data_ptr(), self.r1.data_ptr(), self.da1.data_ptr(), B, T, S,
                     1.0 / T, K.abi_dtype(self.dt), torch.cuda.current_stream().cuda_stream)
            else:
                K.pw_linear(self.dlogits.data_ptr(), Cp, 0, Cp, Cp, self.wptr(self.o_w2T), None, self.da1, S, S, N,
                            aux=self.r1, epi=K.EPI_MASK)
            K.pw_linear(self.da1.data_ptr(), S, 0, S, S, self.wptr(self.o_w1T), None, self.dtotal, S, S, N,
                        aux=self.r0, epi=K.EPI_MASK)

    def _wgrad_layers_group(self, l0: int, l1: int):
        """conv taps + 1x1 residual of layers [l0, l1) in one pass over x, z, df, G."""
        N, L, R, T, ns = self.N, self.L, self.R, self.T, self.nslabs
        es = self.xs.element_size()
        NR = N * R
        ckw = {}   # (xs already holds the conditioned inputs)
        with _Span(self, "wgrad_layers"):
            K.wgrad_layers(self.xs.view(L + 1, N, R)[l0:l1], self.zs.view(L, N, R)[l0:l1],
                           self.dfs.view(L, N, R)[l0:l1], self.gs.data_ptr() + (l0 + 1) * NR * es, self.dil[l0:l1],
                           self.pl_f[l0 * ns * 2 * R * R:], self.pl_r[l0 * ns * R * R:], self.pl_bf[l0 * ns * R:],
                           self.pl_br[l0 * ns * R:], T, ns, **ckw)

    def _wgrad_layers_finish(self):
        L, R, S, Kw, N, T = self.L, self.R, self.S, self.Kw, self.N, self.T
        gp, sec, ns, dt = self.grads.data_ptr(), self.sections, self.nslabs, self.dt
        es = self.xs.element_size()
        NR = N * R
        xs_p, zs_p, dfs_p, gs_p = self.xs.data_ptr(), self.zs.data_ptr(), self.dfs.data_ptr(), self.gs.data_ptr()
        if self.use_wl:
            blk = (R,) if self.part16 else ()      # (bf16 partial blocks in lane order: SRWN_PARTIALS_BLK16, R columns)
            jobs = [(self.pl_f, ns, Kw * R * R, L, True, 1.0, gp + 4 * sec["WF"].offset, Kw * R * R) + blk,
                    (self.pl_bf, ns, R, L, True, 1.0, gp + 4 * sec["BF"].offset, R),
                    (self.pl_r, ns, R * R, L, True, SQRT_HALF, gp + 4 * sec["WR"].offset, R * R) + blk,
                    (self.pl_br, ns, R, L, True, SQRT_HALF, gp + 4 * sec["BR"].offset, R)]
            if self._ic_job is not None:      # the input conv's kernel + bias gradient (init_w | init_b are adjacent)
                jobs.append(self._ic_job)
                self._ic_job = None
            K.reduce_partials_multi(jobs)
            return
        for k in range(Kw):                                                          # dilated conv taps (legacy)
            shifts = [(Kw - 1 - k) * d for d in self.dil]
            last = k == Kw - 1
            K.wgrad(xs_p, NR, R, dfs_p, NR, R, shifts, L, self.wg_parts, self.wg_bparts if last else None, N, T, ns,
                    dt)                                  # (xs holds the conditioned conv inputs, model.py:183)
            K.reduce_partials(self.wg_parts, ns, R * R, L, True, 1.0, gp + 4 * (sec["WF"].offset + k * R * R),
                              Kw * R * R)
            if last:
                K.reduce_partials(self.wg_bparts, ns, R, L, True, 1.0, gp + 4 * sec["BF"].offset, R)
        K.wgrad(zs_p, NR, R, gs_p + NR * es, NR, R, None, L, self.wg_parts, self.wg_bparts, N, T, ns, dt,
                pro=K.PRO_GATE)                                                       # 1x1 residual
        K.reduce_partials(self.wg_parts, ns, R * R, L, True, SQRT_HALF, gp + 4 * sec["WR"].offset, R * R)
        K.reduce_partials(self.wg_bparts, ns, R, L, True, SQRT_HALF, gp + 4 * sec["BR"].offset, R)

    def _wgrad_skip_and_head(self):
        """Gradients of the skip 1x1s and the two head 1x1s: need only z, r0, r1, da1, dtotal, dlogits."""
        N, T, L, R, S, Cp = self.N, self.T, self.L, self.R, self.S, self.Cp
        gp, sec, ns, dt = self.grads.data_ptr(), self.sections, self.nslabs, self.dt
        NR = N * R
        zs_p = self.zs.data_ptr()
        if self.batch_reduce:
            ns_skip = self.ns_skip
            with _Span(self, "wgrad_skip"):
                if self.skip_wt and self.fused_wt:
                    ns_skip = self.ns_skip_wt
                    K.wgrad_skip_wt(self.cTs, self.wt_layer_st, self.wt_layer_seg, self.dtotal,
                                    self.wg_parts if self.skip_parts16 is None else self.skip_parts16,
                                    self.wg_bparts, ns_skip, self.B, T, R)
                else:
                    K.wgrad256(zs_p, NR, R, L, self.dtotal, self.wg_parts, self.wg_bparts, N, self.ns_skip,
                               pro=K.PRO_GATE, chunk_width=R)
            if S == Cp:   # both head 1x1s (S->S, S->C) as one launch
                K.wgrad256_pair(self.r0.data_ptr(), self.da1, self.hd_parts[0], self.hd_bparts[0],
                                self.r1.data_ptr(), self.dlogits, self.hd_parts[1], self.hd_bparts[1], 64, S, S // 64, N,
                                self.ns_head)
            else:
                K.wgrad256(self.r0.data_ptr(), 64, S, S // 64, self.da1, self.hd_parts[0], self.hd_bparts[0], N, self.ns_head)
                K.wgrad256(self.r1.data_ptr(), 64, S, S // 64, self.dlogits, self.hd_parts[1], self.hd_bparts[1], N,
                           self.ns_head)
            skip16 = self.skip_wt and self.fused_wt and self.skip_parts16 is not None
            jobs = [
                (self.skip_parts16, ns_skip, L * R * S, 1, True, 1.0, gp + 4 * sec["WS"].offset, 0, S) if skip16 else
                (self.wg_parts, ns_skip, L * R * S, 1, True, 1.0, gp + 4 * sec["WS"].offset, 0),
                (self.wg_bparts, ns_skip, S, L, False, 1.0, gp + 4 * sec["BS"].offset, S),
                (self.hd_parts[0], self.ns_head, S * S, 1, True, 1.0, gp + 4 * sec["head_w1"].offset, 0),
                (self.hd_bparts[0], self.ns_head, S, 1, True, 1.0, gp + 4 * sec["head_b1"].offset, 0),
                (self.hd_parts[1], self.ns_head, S * Cp, 1, True, 1.0, gp + 4 * sec["head_w2"].offset, 0),
                (self.hd_bparts[1], self.ns_head, Cp, 1, True, 1.0, gp + 4 * sec["head_b2"].offset, 0)]
            if self._loss_job is not None:      # the loss the forward pass deferred: one more (one-output) reduction
                jobs.append(self._loss_job)
                self._loss_job = None
            K.reduce_partials_multi(jobs)
            return
        if self.use_w256:
            # every skip 1x1 at once: out[L*R, S] = c_all^T . dtotal (dtotal re-read once per 4 layers)
            ns_skip = self.ns_skip
            with _Span(self, "wgrad_skip"):
                if self.skip_wt and self.fused_wt:
                    ns_skip = self.ns_skip_wt
                    K.wgrad_skip_wt(self.cTs, self.wt_layer_st, self.wt_layer_seg, self.dtotal,
                                    self.wg_parts if self.skip_parts16 is None else self.skip_parts16,
                                    self.wg_bparts, ns_skip, self.B, T, R)
                else:
                    K.wgrad256(zs_p, NR, R, L, self.dtotal, self.wg_parts, self.wg_bparts, N, self.ns_skip,
                               pro=K.PRO_GATE, chunk_width=R)
            if self.skip_wt and self.fused_wt and self.skip_parts16 is not None:
                K.reduce_partials_multi([(self.skip_parts16, ns_skip, L * R * S, 1, True, 1.0, gp + 4 * sec["WS"].offset, 0, S)])
            else:
                K.reduce_partials(self.wg_parts, ns_skip, L * R * S, 1, True, 1.0, gp + 4 * sec["WS"].offset, 0)
            K.reduce_partials(self.wg_bparts, ns_skip, S, L, False, 1.0, gp + 4 * sec["BS"].offset, S)
            K.wgrad256(self.r0.data_ptr(), 64, S, S // 64, self.da1, self.wg_parts, self.wg_bparts, N, self.ns_head)
            K.reduce_partials(self.wg_parts, self.ns_head, S * S, 1, True, 1.0, gp + 4 * sec["head_w1"].offset, 0)
            K.reduce_partials(self.wg_bparts, self.ns_head, S, 1, True, 1.0, gp + 4 * sec["head_b1"].offset, 0)
        else:
            with _Span(self, "wgrad_skip"):
                K.wgrad(zs_p, NR, R, self.dtotal.data_ptr(), 0, S, None, L, self.wg_parts, self.wg_bparts, N, T, ns,
                        dt, pro=K.PRO_GATE)                                           # 1x1 skip
            K.reduce_partials(self.wg_parts, ns, R * S, L, True, 1.0, gp + 4 * sec["WS"].offset, R * S)
            K.reduce_partials(self.wg_bparts, ns, S, L, True, 1.0, gp + 4 * sec["BS"].offset, S)
            K.wgrad(self.r0.data_ptr(), 0, S, self.da1.data_ptr(), 0, S, None, 1, self.wg_parts, self.wg_bparts, N, T,
                    ns, dt)                                                           # head 1x1 (S->S)
            K.reduce_partials(self.wg_parts, ns, S * S, 1, True, 1.0, gp + 4 * sec["head_w1"].offset, 0)
            K.reduce_partials(self.wg_bparts, ns, S, 1, True, 1.0, gp + 4 * sec["head_b1"].offset, 0)
        if self.use_w256 and not self.pooled and Cp == 256:
            K.wgrad256(self.r1.data_ptr(), 64, S, S // 64, self.dlogits, self.wg_parts, self.wg_bparts, N,
                       self.ns_head)                                                  # last 1x1 (S->C)
            K.reduce_partials(self.wg_parts, self.ns_head, S * Cp, 1, True, 1.0, gp + 4 * sec["head_w2"].offset, 0)
            K.reduce_partials(self.wg_bparts, self.ns_head, Cp, 1, True, 1.0, gp + 4 * sec["head_b2"].offset, 0)
        elif not self.pooled:   # (the pooled head wrote its own kernel/bias gradients in forward)
            K.wgrad(self.r1.data_ptr(), 0, S, self.dlogits.data_ptr(), 0, Cp, None, 1, self.wg_parts, self.wg_bparts,
                    N, T, ns, dt)                                                     # last 1x1 (S->C)
            K.reduce_partials(self.wg_parts, ns, S * Cp, 1, True, 1.0, gp + 4 * sec["head_w2"].offset, 0)
            K.reduce_partials(self.wg_bparts, ns, Cp, 1, True, 1.0, gp + 4 * sec["head_b2"].offset, 0)

    def _wgrad_input_conv_partials(self):
        """Stage 1 of the input conv's weight gradient (model.py:40): per-slab partial sums from gs[0]; its slab
        reduction rides in the final srwn_reduce_partials_multi launch (it was a launch of its own)."""
        sec = self.sections
        assert sec["init_b"].offset == sec["init_w"].offset + sec["init_w"].numel
        nsl = K.init_conv_wgrad(self.audio, self.gs[0], None, None, self.Kw, 1 if self.cfg.shift_input else 0, self.ic_ws)
        self._ic_job = (self.ic_ws, nsl, (self.Kw + 1) * self.R, 1, True, 1.0,
                        self.grads.data_ptr() + 4 * sec["init_w"].offset, 0)

    def _wgrad_input_and_cond(self, input_conv: bool = True):
        B, T, L, R, Kw = self.B, self.T, self.L, self.R, self.Kw
        g = self.grads
        gp, sec, dt = g.data_ptr(), self.sections, self.dt
        if input_conv:
            K.init_conv_wgrad(self.audio, self.gs[0], self.view("init_w", g).reshape(-1), self.view("init_b", g), Kw,
                              1 if self.cfg.shift_input else 0, self.ic_ws)
        if self.E:
            # conditioning 1x1 (model.py:180): dcb_l = adjoint of the NN upsample applied to G_l
            rows_c, Ep, E = B * self.frames, self.Ep, self.E
            from ._lib import call
            call("srwn_frame_sum_batched", self.gs.data_ptr(), B * T * R, self.dcb.data_ptr(), rows_c * R, L, B, T, R,
                 self.frames, self.cfg.pool_stride, 1.0, K.abi_dtype(dt), torch.cuda.current_stream().cuda_stream)
            K.wgrad(self.cond_in.data_ptr(), 0, Ep, self.dcb.data_ptr(), rows_c * R, R, None, L, self.wgc_parts,
                    self.wgc_bparts, rows_c, self.frames, self.nslabs_c, dt)
            if Ep == E:
                K.reduce_partials(self.wgc_parts, self.nslabs_c, Ep * R, L, True, 1.0, gp + 4 * sec["WC"].offset, E * R)
            else:
                K.reduce_partials(self.wgc_parts, self.nslabs_c, Ep * R, L, True, 1.0, self.wc_grad_pad.data_ptr(),
                                  Ep * R)
                self.view("WC", g).copy_(self.wc_grad_pad[:, :E, :])
            K.reduce_partials(self.wgc_bparts, self.nslabs_c, R, L, True, 1.0, gp + 4 * sec["BC"].offset, R)

    # ------------------------------------------------------------------------------------------
    # update
    # ------------------------------------------------------------------------------------------
    def allreduce_grads(self):
        """Data parallel: sum the flat gradient over ranks (RCCL over xGMI); Adam divides by world."""
        dp.allreduce_sum_(self.grads, self.pg)

    def optimizer_step(self):
        # mean losses: mean of shard gradients; the mixture-of-logistics loss is a SUM over batch and time
        # (ops.py:173-174), so shard gradients simply add
        K.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.adam_step, self.cfg.learning_rate,
                    grad_scale=1.0 if self.mol else 1.0 / self.world)
        self.repack()

    def _allreduce_bucket_a(self):
        """Skip + head gradients: issued while the lower part of the backward pass runs."""
        return dp.allreduce_sum_(self.grads[self.bucket_off:], self.pg, async_op=True)

    def _allreduce_bucket_b(self, pending):
        dp.allreduce_sum_(self.grads[:self.bucket_off], self.pg)
        if pending is not None:
            pending.wait()

    def train_step(self) -> torch.Tensor:
        """fwd + bwd + (all-reduce) + Adam on the staged inputs; returns the device loss scalar.
        With several ranks the gradient all-reduce runs in two buckets, the first (skip + head kernels, 65 % of the
        bytes) overlapped with the lower part of the backward pass."""
        self.forward(defer_loss=True)
        if self.bucketed and not self.timing:
            self.backward(part=1)
            h = self._allreduce_bucket_a()
            self.backward(part=2)
            self._allreduce_bucket_b(h)
        else:
            self.backward()
            self.allreduce_grads()
        self.optimizer_step()
        return self.loss

    def generate(self, nsteps: int, mode: str = "sample", seed: int = 0, forced: Optional[torch.Tensor] = None,
                 want_logits: bool = False, batch: Optional[int] = None, cond: Optional[torch.Tensor] = None):
        """Queue-cached autoregressive generation of `nsteps` samples for `batch` utterances.
        Softmax teacher: returns (audio [B,nsteps] f32, mu-law codes [B,nsteps] i32, logits [B,nsteps,C] f32 or None).
        Mixture-of-logistics decoder (head_mode "mol"; `cond` = encoding_w_condition [B, frames, cond_channels] when the
        stack is conditioned): returns (audio, selected mixture, logits [B,nsteps,4M])."""
        import ctypes as C
        from . import _lib
        if self.o_gen is None or self.pooled:
            raise NotImplementedError("generate: built for R=64 or 32, S=256 or 128, K=2 stacks with a per-time-step head")
        B = int(batch or self.B)
        self._repack_generation()      # (the generation-only images follow the parameters lazily: not part of a training step)
        dl = (C.c_int32 * self.L)(*self.dil)
        relems = int(_lib.load().srwn_generate_ring_elems(dl, self.L, self.R))
        ring = torch.zeros(relems * ((B + 31) // 32), dtype=self.dt, device=self.dev)
        audio = torch.zeros((B, nsteps), dtype=torch.float32, device=self.dev)
        codes = torch.zeros((B, nsteps), dtype=torch.int32, device=self.dev)
        logits = torch.zeros((B, nsteps, self.C), dtype=torch.float32, device=self.dev) if want_logits else None
        fp = None
        if forced is not None:
            forced = forced.to(device=self.dev, dtype=torch.float32).contiguous()
            if tuple(forced.shape) != (B, nsteps):
                raise ValueError("forced must be [batch, nsteps]")
            fp = forced.data_ptr()
        v = self.view
        common = (self.wptr(self.o_gen), self.wptr(self.o_skip_gen), self.wptr(self.o_w1), self.wptr(self.o_w2),
                  v("BF").data_ptr(), v("BR").data_ptr(), self.bs_sum.data_ptr(), v("head_b1").data_ptr(),
                  v("head_b2").data_ptr(), v("init_w").data_ptr(), v("init_b").data_ptr(), ring.data_ptr(),
                  audio.data_ptr(), codes.data_ptr(), None if logits is None else logits.data_ptr(), fp, dl, self.L, B,
                  nsteps, nsteps, self.R, self.S)
        md = {"argmax": 0, "mean": 0, "sample": 1}[mode]
        st = torch.cuda.current_stream().cuda_stream
        if self.mol:
            cond_all, frames = None, 1
            if self.E:
                if cond is None:
                    raise ValueError("this decoder is conditioned: pass cond [batch, frames, cond_channels]")
                cond = cond.to(device=self.dev, dtype=torch.float32)
                frames = cond.shape[1]
                if tuple(cond.shape) != (B, frames, self.E) or frames * self.cfg.pool_stride < nsteps:
                    raise ValueError("cond must be [batch, frames >= nsteps/pool_stride, %d]" % self.E)
                cin = torch.zeros((B * frames, self.Ep), dtype=self.dt, device=self.dev)
                cin[:, :self.E].copy_(cond.reshape(B * frames, self.E))
                cond_all = torch.empty((B * frames, self.L * self.R), dtype=self.dt, device=self.dev)
                K.pw_linear(cin.data_ptr(), self.Ep, 0, self.Ep, self.Ep, self.wptr(self.o_wc), v("BC").reshape(-1),
                            cond_all, self.L * self.R, self.L * self.R, B * frames)          # model.py:180
            elif cond is not None:
                raise ValueError("this decoder is not conditioned")
            import os as _os
            if self.o_g16 is not None and _os.environ.get("SRWN_GEN16", "1") != "0":
                _lib.call("srwn_generate16_mol", self.wptr(self.o_g16), self.wptr(self.o_g16_h1), self.wptr(self.o_g16_h2),
                          *common[4:21], self.R, self.S, self.C // 4, None if cond_all is None else cond_all.data_ptr(), frames,
                          self.cfg.pool_stride, self.L * self.R, md, int(seed), st)
            else:
                _lib.call("srwn_generate_mol", *common, self.Kw, self.C // 4,
                          None if cond_all is None else cond_all.data_ptr(), frames, self.cfg.pool_stride, self.L * self.R,
                          md, int(seed), K.abi_dtype(self.dt), st)
        else:
            if self.E or cond is not None:
                raise NotImplementedError("generate: the conditioned softmax teacher is not built (the conditioned "
                                          "decoder of the reference has the mixture-of-logistics head)")
            import os as _os
            if self.o_g16 is not None and _os.environ.get("SRWN_GEN16", "1") != "0":
                _lib.call("srwn_generate16", self.wptr(self.o_g16), self.wptr(self.o_g16_h1), self.wptr(self.o_g16_h2),
                          *common[4:21], self.R, self.S, self.C, md, int(seed), st)
            else:
                _lib.call("srwn_generate", *common, self.C, self.Kw, md, int(seed), K.abi_dtype(self.dt), st)
        return audio, codes, logits

    def capture_graphs(self):
        """Captures the step as two hipGraphs -- {forward, backward} and {Adam, re-pack} -- with the
        gradient all-reduce left between them as an ordinary RCCL call, so one and many GPUs replay the
        same launch-free kernel sequence.  Call after at least one eager train_step (warm-up)."""
        import os as _os
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        self._g_b2 = None
        if self.world == 1 and _os.environ.get("SRWN_FORCE_DIST") != "1":
            # no collective to leave between the graphs: the whole step is one replay
            with torch.cuda.graph(self._g_fb):
                self.forward(defer_loss=True)
                self.backward()
                self.optimizer_step()
            self._g_opt = None
            torch.cuda.synchronize()
            return
        with torch.cuda.graph(self._g_fb):
            self.forward(defer_loss=True)
            self.backward(part=1 if self.bucketed else 0)
        if self.bucketed:   # {forward, upper backward} | bucket A in flight | {lower backward} | bucket B | {Adam}
            self._g_b2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_b2, pool=self._g_fb.pool()):
                self.backward(part=2)
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool()):
            self.optimizer_step()
        torch.cuda.synchronize()

    def train_step_graphed(self) -> torch.Tensor:
        self._g_fb.replay()
        if self._g_opt is None:
            return self.loss
        if self._g_b2 is not None:
            h = self._allreduce_bucket_a()
            self._g_b2.replay()
            self._allreduce_bucket_b(h)
        else:
            self.allreduce_grads()
        self._g_opt.replay()
        return self.loss
