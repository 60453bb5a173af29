"""The auto-encoder's encoder and the joint encoder+decoder training step on one MI355X.

Mirrors ``WaveNetAutoEncoder.createEncoder`` / ``createNetwork`` (model.py:136-156, 203-216) and
``ResidualDilationLayerNC`` (ops.py:48-58):

  a_0     = relu(conv_K(relu(inputs)))                      1 -> EC channels, SAME padding ('nc_conv')
  r_0     = relu(a_0 W_r + b_r)                             (every consumer of a layer output applies relu first,
  a_{l+1} = relu(conv_K(r_l) + b_l);  r_{l+1} = relu(a_{l+1} Wr_l + br_l)       so relu(h) is what is stored)
  encoding = avgpool_T( (sum_l a_{l+1} Ws_l + bs_l) W_lat + b_lat )

The skip path is linear up to the pool, so the pool is applied FIRST: per-frame means of a_l ([B*frames, EC] per layer,
one batched pass), then the skip 1x1s of all layers as ONE K = L*EC product and the latent 1x1 on B*frames rows --
the [B,T,S] skip tensors of the reference are never formed, and their gradient returns to the layers as a
per-frame broadcast inside the backward GEMM epilogue (srwn_tap_linear's frame_add).

The K = 2 non-causal convolutions, their data gradients and the 1x1s run as time-tap MFMA GEMMs (srwn_tap_linear);
weight gradients are time-contraction GEMMs (srwn_wgrad) batched over layers.
"""
from __future__ import annotations

import os

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import kernels as K
from . import packing as P
from . import _lib
from ._lib import call
from .engine import Section, StackConfig, WaveNetEngine


class EncoderStack:
    def __init__(self, nlayers: int, batch: int, length: int, pool_stride: int, encoder_channels: int = 128,
                 skip_channels: int = 256, latent_channels: int = 16, filter_width: int = 2,
                 dtype: torch.dtype = torch.bfloat16, learning_rate: float = 1e-3, device="cuda", seed: int = 0):
        if encoder_channels != 128:
            raise NotImplementedError("encoder_channels %d: the encoder kernels are built for 128 (the reference "
                                      "default, model.py:76)" % encoder_channels)
        if filter_width != 2:
            raise NotImplementedError("filter_width %d: only 2 is built" % filter_width)
        if skip_channels % 32 or skip_channels < 32:
            raise NotImplementedError("skip_channels must be a multiple of 32")
        if length % pool_stride:
            raise ValueError("length %d is not a multiple of pool_stride %d" % (length, pool_stride))
        self.L, self.B, self.T, self.pool = int(nlayers), int(batch), int(length), int(pool_stride)
        self.N = self.B * self.T
        self.frames = self.T // self.pool
        self.rows_c = self.B * self.frames
        self.EC, self.S, self.lat, self.Kw = encoder_channels, skip_channels, latent_channels, filter_width
        self.dt, self.dev, self.lr = dtype, torch.device(device), learning_rate
        self._build_params(seed)
        self._build_packing()
        self._alloc()
        self.repack()

    # -- parameters ----------------------------------------------------------------------------------
    def _build_params(self, seed):
        L, EC, S, Kw, lat = self.L, self.EC, self.S, self.Kw, self.lat
        secs: Dict[str, Section] = {}
        off = 0
        for name, shape in (("nc_w", (Kw, 1, EC)), ("nc_b", (EC,)), ("nc_wr", (EC, EC)), ("nc_br", (EC,)),
                            ("EW", (L, Kw, EC, EC)), ("EB", (L, EC)), ("EWR", (L, EC, EC)), ("EBR", (L, EC)),
                            ("EWS", (L, EC, S)), ("EBS", (L, S)), ("lat_w", (S, lat)), ("lat_b", (lat,))):
            secs[name] = Section(name, off, shape)
            off += secs[name].numel
        self.sections, self.nparams = secs, off
        z = lambda: torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.params, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z()
        self.adam_step = torch.zeros(1, dtype=torch.int64, device=self.dev)
        # 'nc_conv' has a skip 1x1 whose output is discarded (model.py:141): a variable without gradient
        self.dead = {"nc_ws": torch.zeros((EC, S), device=self.dev), "nc_bs": torch.zeros(S, device=self.dev)}
        self.init_parameters(seed)

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.sections[name]
        buf = self.params if buf is None else buf
        return buf[s.offset:s.offset + s.numel].view(s.shape)

    def init_parameters(self, seed: int):
        rng = np.random.default_rng(seed)

        def xav(shape, fan_in, fan_out):
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            return torch.tensor(rng.uniform(-lim, lim, size=shape), dtype=torch.float32)

        L, EC, S, Kw, lat = self.L, self.EC, self.S, self.Kw, self.lat
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, t):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = t.reshape(-1)

        put("nc_w", xav((Kw, 1, EC), Kw, Kw * EC)); put("nc_wr", xav((EC, EC), EC, EC))
        put("EW", xav((L, Kw, EC, EC), Kw * EC, Kw * EC)); put("EWR", xav((L, EC, EC), EC, EC))
        put("EWS", xav((L, EC, S), EC, S)); put("lat_w", xav((S, lat), S, lat))
        self.params.copy_(host)
        self.dead["nc_ws"].copy_(xav((EC, S), EC, S))
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_step.zero_()

    def load_oracle_params(self, ep):
        host = torch.zeros(self.nparams, dtype=torch.float32)

        def put(name, arr):
            s = self.sections[name]
            host[s.offset:s.offset + s.numel] = torch.tensor(np.asarray(arr), dtype=torch.float32).reshape(-1)

        put("nc_w", ep.nc.w); put("nc_b", ep.nc.b); put("nc_wr", ep.nc.wr); put("nc_br", ep.nc.br)
        for nm, f in (("EW", "w"), ("EB", "b"), ("EWR", "wr"), ("EBR", "br"), ("EWS", "ws"), ("EBS", "bs")):
            put(nm, np.stack([getattr(p, f) for p in ep.layers]))
        put("lat_w", ep.lat_w); put("lat_b", ep.lat_b)
        self.params.copy_(host)
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_step.zero_()
        self.repack()

    def named_tensors(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        v = lambda n: self.view(n, buf)
        out = {"nc.w": v("nc_w"), "nc.b": v("nc_b"), "nc.wr": v("nc_wr"), "nc.br": v("nc_br")}
        for i in range(self.L):
            out[f"e{i}.w"] = v("EW")[i]; out[f"e{i}.b"] = v("EB")[i]
            out[f"e{i}.wr"] = v("EWR")[i]; out[f"e{i}.br"] = v("EBR")[i]
            out[f"e{i}.ws"] = v("EWS")[i]; out[f"e{i}.bs"] = v("EBS")[i]
        out["lat_w"] = v("lat_w"); out["lat_b"] = v("lat_b")
        return out

    def tf_variables(self, scope: str) -> Dict[str, torch.Tensor]:
        """Reference names under ``<scope>`` = 'WaveNetAutoEncoder/Encoder': the K-tap conv lives in '<name>_NC/conv1d'
        (ops.py:50-51); the unnamed 1x1s number per scope -- residual conv1d_{2j}, skip conv1d_{2j+1} for the j-th
        NC layer ('nc_conv' is j = 0), the latent 1x1 is conv1d_{2(L+1)} (model.py:141-152)."""
        cname = lambda j: "conv1d" if j == 0 else "conv1d_%d" % j
        n = self.named_tensors()
        out = {f"{scope}/nc_conv_NC/conv1d/kernel": n["nc.w"], f"{scope}/nc_conv_NC/conv1d/bias": n["nc.b"],
               f"{scope}/{cname(0)}/kernel": n["nc.wr"].unsqueeze(0), f"{scope}/{cname(0)}/bias": n["nc.br"],
               f"{scope}/{cname(1)}/kernel": self.dead["nc_ws"].unsqueeze(0), f"{scope}/{cname(1)}/bias": self.dead["nc_bs"]}
        for i in range(self.L):
            nm = f"dilated_conv_{i}_NC"
            out[f"{scope}/{nm}/conv1d/kernel"] = n[f"e{i}.w"]
            out[f"{scope}/{nm}/conv1d/bias"] = n[f"e{i}.b"]
            out[f"{scope}/{cname(2 * i + 2)}/kernel"] = n[f"e{i}.wr"].unsqueeze(0)
            out[f"{scope}/{cname(2 * i + 2)}/bias"] = n[f"e{i}.br"]
            out[f"{scope}/{cname(2 * i + 3)}/kernel"] = n[f"e{i}.ws"].unsqueeze(0)
            out[f"{scope}/{cname(2 * i + 3)}/bias"] = n[f"e{i}.bs"]
        out[f"{scope}/{cname(2 * self.L + 2)}/kernel"] = n["lat_w"].unsqueeze(0)
        out[f"{scope}/{cname(2 * self.L + 2)}/bias"] = n["lat_b"]
        return out

    # -- MFMA weight images --------------------------------------------------------------------------
    def _build_packing(self):
        L, EC, S, Kw = self.L, self.EC, self.S, self.Kw
        sec = self.sections
        pk = K.Packer(self.dev)
        self.o_nc_wr = P.pack_linear(pk, sec["nc_wr"].offset, EC, EC, EC)
        self.o_nc_wrT = P.pack_linear_T(pk, sec["nc_wr"].offset, EC, EC, EC)
        self.o_conv, self.o_convT, self.o_wr, self.o_wrT = [], [], [], []
        # fused layer kernels (srwn_nc_layer_fwd/_bwd: bf16, 128 channels, K = 2; SRWN_NC_FUSED=0 keeps the two-launch
        # path): the 1x1's B operand is the conv's accumulator tile -> the 1x1 images in permuted k order
        self.fused = (self.dt == torch.bfloat16 and EC == 128 and Kw == 2 and
                      os.environ.get("SRWN_NC_FUSED", "1") != "0")
        self.o_wr_p, self.o_wrT_p = [], []
        for l in range(L):
            self.o_conv.append(P.pack_conv(pk, sec["EW"].offset + l * Kw * EC * EC, Kw, EC))
            self.o_convT.append(P.pack_conv_T(pk, sec["EW"].offset + l * Kw * EC * EC, Kw, EC))
            self.o_wr.append(P.pack_linear(pk, sec["EWR"].offset + l * EC * EC, EC, EC, EC))
            self.o_wrT.append(P.pack_linear_T(pk, sec["EWR"].offset + l * EC * EC, EC, EC, EC))
            if self.fused:
                self.o_wr_p.append(P.pack_linear(pk, sec["EWR"].offset + l * EC * EC, EC, EC, EC, perm=True))
                self.o_wrT_p.append(P.pack_linear_T(pk, sec["EWR"].offset + l * EC * EC, EC, EC, EC, perm=True))
        if self.fused:
            self.o_nc_wrT_p = P.pack_linear_T(pk, sec["nc_wr"].offset, EC, EC, EC, perm=True)
        # every skip 1x1 as one image: rows = skip channel, k = layer*EC + n (applied to the frame means)
        self.o_ws = pk.reserve(S // 32, L * EC // 16)
        for l in range(L):
            P.fill_linear(pk, self.o_ws, sec["EWS"].offset + l * EC * S, EC, S, S // 32, L * EC // 16,
                          ks_offset=l * EC // 16, ks_count=EC // 16)
        # and transposed: rows = layer*EC + n, k = skip channel (gradient back to the frame means)
        per = (EC // 32) * (S // 16) * 512
        self.o_wsT = pk.reserve(L * (EC // 32), S // 16)
        for l in range(L):
            P.fill_linear_T(pk, self.o_wsT + l * per, sec["EWS"].offset + l * EC * S, EC, S, EC // 32, S // 16)
        pk.finalize()
        self.packer = pk
        self.packed = torch.zeros(max(pk.total, 1), dtype=self.dt, device=self.dev)

    def wptr(self, off: int) -> int:
        return self.packed.data_ptr() + off * self.packed.element_size()

    def repack(self):
        self.packer.gather(self.params, self.packed)

    # -- buffers -------------------------------------------------------------------------------------
    def _alloc(self):
        B, T, N, L, EC, S = self.B, self.T, self.N, self.L, self.EC, self.S
        z = lambda *s, dt=self.dt: torch.zeros(s, dtype=dt, device=self.dev)
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self.x = f(B, T); self.xr = f(B, T)
        self.a = z(L + 1, N, EC)          # a[0] from 'nc_conv', a[l+1] from layer l (post-relu)
        self.r = z(L, N, EC)              # r[l] = relu(residual output feeding layer l)
        self.dpre = z(L + 1, N, EC)       # gradient at the conv pre-activations
        self.dh = z(L + 1, N, EC)         # gradient at the residual pre-activations; dh[L] stays 0 (unused output)
        if self.fused:                    # relu masks as bit words in the fused kernels' tile/lane layout
            nw = int(_lib.load().srwn_nc_mask_words(B, T))
            self.abits = torch.zeros(L + 1, nw, dtype=torch.int64, device=self.dev)
            self.rbits = torch.zeros(L, nw, dtype=torch.int64, device=self.dev)
        self.a_mean = z(L, self.rows_c, EC)
        self.bs_sum = f(S)
        self.s_mean = f(self.rows_c, S)
        self.s_parts = f(L * self.rows_c * S)
        self.enc = f(self.rows_c, self.lat)
        self.ds_mean = z(self.rows_c, S)
        self.da_all = f(self.rows_c, L * EC)
        self.nslabs = K.wgrad_slabs(N)
        # (few frame rows: 128-row slabs, or the skip 1x1 gradients are L workgroups walking them in 32-row steps: 139 us)
        self.nslabs_c = max(K.wgrad_slabs(self.rows_c), min(max(1, 256 // max(L, 1)), max(1, self.rows_c // 128)))
        self.wg_parts = f(max(self.nslabs * EC * EC, self.nslabs_c * L * EC * S))
        self.wg_bparts = f(max(self.nslabs * EC, self.nslabs_c * L * S))
        # fused per-layer weight-gradient pass: about two 8-wave workgroups per CU over (slab, layer)
        self.ns_enc = max(1, min(self.nslabs, max(4, 512 // max(L, 1))))
        ne = self.ns_enc
        self.pe_w = f(L * ne * self.Kw * EC * EC); self.pe_r = f(L * ne * EC * EC)
        self.pe_b = f(L * ne * EC); self.pe_br = f(L * ne * EC)
        self.ic_ws = f(int(_lib.load().srwn_init_conv_wgrad_partials(B, T, EC, self.Kw)))

    # -- forward -------------------------------------------------------------------------------------
    def _tap(self, x, ntaps, step, wp, bias, y, epi, aux=None, fadd_ptr=None):
        EC, L = self.EC, self.L
        call("srwn_tap_linear", x.data_ptr(), EC, ntaps, step, self.T, EC, wp, None if bias is None else bias.data_ptr(),
             y.data_ptr(), EC, EC, self.N, None if aux is None else aux.data_ptr(), EC, fadd_ptr, L * EC, self.frames,
             self.pool, 1.0 / self.pool, epi, K.abi_dtype(self.dt), K._stream())

    def forward(self, inputs: Optional[torch.Tensor] = None) -> torch.Tensor:
        """inputs [B,T] fp32 -> self.enc [B*frames, latent] fp32 (model.py:136-156)."""
        B, T, N, L, EC, S = self.B, self.T, self.N, self.L, self.EC, self.S
        v = self.view
        if inputs is not None:
            self.x.copy_(inputs.reshape(B, T))
        st = K._stream()
        call("srwn_nc_input_fwd", self.x.data_ptr(), v("nc_w").data_ptr(), v("nc_b").data_ptr(), self.a[0].data_ptr(),
             B, T, EC, self.Kw, K.abi_dtype(self.dt), st)
        self._tap(self.a[0], 1, 0, self.wptr(self.o_nc_wr), v("nc_br"), self.r[0], K.EPI_RELU)
        for l in range(L):
            if self.fused:   # conv + relu + 1x1 + relu in one launch; the last layer's residual output is never used
                call("srwn_nc_layer_fwd", self.r[l].data_ptr(), self.wptr(self.o_conv[l]), self.wptr(self.o_wr_p[l]),
                     v("EB")[l].data_ptr(), v("EBR")[l].data_ptr(), self.a[l + 1].data_ptr(),
                     self.r[l + 1].data_ptr() if l < L - 1 else None, self.abits[l + 1].data_ptr(),
                     self.rbits[l + 1].data_ptr() if l < L - 1 else None, B, T, EC, self.Kw, K.abi_dtype(self.dt), st)
                continue
            self._tap(self.r[l], self.Kw, 1, self.wptr(self.o_conv[l]), v("EB")[l], self.a[l + 1], K.EPI_RELU)
            if l < L - 1:   # the last layer's residual output is never used (model.py:144-150)
                self._tap(self.a[l + 1], 1, 0, self.wptr(self.o_wr[l]), v("EBR")[l], self.r[l + 1], K.EPI_RELU)
        # pooled skip path: frame means, all skip 1x1s as one product, latent 1x1
        call("srwn_frame_sum_batched", self.a[1].data_ptr(), N * EC, self.a_mean.data_ptr(), self.rows_c * EC, L, B, T,
             EC, self.frames, self.pool, 1.0 / self.pool, K.abi_dtype(self.dt), st)
        K.reduce_partials(v("EBS").reshape(-1), L, S, 1, True, 1.0, self.bs_sum.data_ptr(), 0)
        # K = L*EC on few rows: one k-slice per layer, then a fixed-order sum of the slices
        call("srwn_pw_linear_ksplit", self.a_mean.data_ptr(), EC, self.rows_c * EC, EC, L * EC, self.wptr(self.o_ws),
             self.bs_sum.data_ptr(), self.s_parts.data_ptr(), S, S, S, self.rows_c, L, K.abi_dtype(self.dt), st)
        K.reduce_partials(self.s_parts, L, self.rows_c * S, 1, True, 1.0, self.s_mean.data_ptr(), 0)
        call("srwn_small_gemm", self.s_mean.data_ptr(), S, S, 0, K.F32, v("lat_w").data_ptr(), self.lat, 1, S, 0,
             v("lat_b").data_ptr(), self.enc.data_ptr(), self.lat, K.F32, self.rows_c, self.lat, S, 0, st)
        return self.enc

    # -- backward ------------------------------------------------------------------------------------
    def backward(self, denc: torch.Tensor):
        """denc [B*frames, latent] fp32 = d loss / d encoding; fills self.grads."""
        B, T, N, L, EC, S, Kw = self.B, self.T, self.N, self.L, self.EC, self.S, self.Kw
        v = self.view
        g = self.grads
        gp, sec, dt, st = g.data_ptr(), self.sections, self.dt, K._stream()
        rc = self.rows_c
        K._chk(denc, "denc", torch.float32, (rc, self.lat))
        # latent 1x1 and the skip 1x1s on the frame axis
        call("srwn_small_wgrad", self.s_mean.data_ptr(), S, denc.data_ptr(), self.lat, v("lat_w", g).data_ptr(),
             v("lat_b", g).data_ptr(), rc, S, self.lat, 1.0, st)
        call("srwn_small_gemm", denc.data_ptr(), self.lat, self.lat, 0, K.F32, v("lat_w").data_ptr(), 1, self.lat,
             self.lat, 0, None, self.ds_mean.data_ptr(), S, K.abi_dtype(dt), rc, S, self.lat, 0, st)
        K.wgrad(self.a_mean.data_ptr(), rc * EC, EC, self.ds_mean.data_ptr(), 0, S, None, L, self.wg_parts,
                self.wg_bparts, rc, self.frames, self.nslabs_c, dt)
        K.reduce_partials(self.wg_parts, self.nslabs_c, EC * S, L, True, 1.0, gp + 4 * sec["EWS"].offset, EC * S)
        K.reduce_partials(self.wg_bparts, self.nslabs_c, S, L, True, 1.0, gp + 4 * sec["EBS"].offset, S)
        K.pw_linear(self.ds_mean.data_ptr(), S, 0, S, S, self.wptr(self.o_wsT), None, self.da_all, L * EC, L * EC, rc,
                    epi=K.EPI_F32, compute_dtype=dt)
        # layer chain, top down: the pooled skip gradient enters as a per-frame broadcast (frame_add)
        if self.fused:
            # one launch per layer: the conv data gradient of layer l, then the 1x1 data gradient of the layer below it
            # (layer l-1's residual 1x1, or 'nc_conv''s under layer 0) on the tile still in registers
            self._tap(self.dh[L], 1, 0, self.wptr(self.o_wrT[L - 1]), None, self.dpre[L], K.EPI_MASK, aux=self.a[L],
                      fadd_ptr=self.da_all.data_ptr() + 4 * (L - 1) * EC)          # dh[L] = 0: the skip path only
            for t_, bits in ((self.a[0], self.abits[0]), (self.r[0], self.rbits[0])):   # written by the generic kernels
                call("srwn_nc_mask_bits", t_.data_ptr(), bits.data_ptr(), B, T, EC, K.abi_dtype(dt), st)
            for l in range(L - 1, -1, -1):
                call("srwn_nc_layer_bwd", self.dpre[l + 1].data_ptr(), self.wptr(self.o_convT[l]),
                     self.rbits[l].data_ptr(), self.dh[l].data_ptr(),
                     self.wptr(self.o_wrT_p[l - 1] if l else self.o_nc_wrT_p),
                     self.da_all.data_ptr() + 4 * (l - 1) * EC if l else None, L * EC, self.frames, self.pool,
                     1.0 / self.pool, self.abits[l].data_ptr(), self.dpre[l].data_ptr(), B, T, EC, Kw, K.abi_dtype(dt), st)
        else:
            for l in range(L - 1, -1, -1):
                self._tap(self.dh[l + 1], 1, 0, self.wptr(self.o_wrT[l]), None, self.dpre[l + 1], K.EPI_MASK,
                          aux=self.a[l + 1], fadd_ptr=self.da_all.data_ptr() + 4 * l * EC)
                self._tap(self.dpre[l + 1], Kw, -1, self.wptr(self.o_convT[l]), None, self.dh[l], K.EPI_MASK,
                          aux=self.r[l])
            self._tap(self.dh[0], 1, 0, self.wptr(self.o_nc_wrT), None, self.dpre[0], K.EPI_MASK, aux=self.a[0])
        # weight gradients of the L layers -- both conv taps, the 1x1 residual and the two biases -- in ONE pass over
        # r, a, dpre, dh (layer l: r[l], a[l+1], dpre[l+1], dh[l+1]; the last layer's residual 1x1 sees dh[L] = 0)
        NE = N * EC
        ne = self.ns_enc
        call("srwn_wgrad_nc_layers", self.r.data_ptr(), self.a[1].data_ptr(), self.dpre[1].data_ptr(),
             self.dh[1].data_ptr(), NE, L, self.pe_w.data_ptr(), self.pe_r.data_ptr(), self.pe_b.data_ptr(),
             self.pe_br.data_ptr(), N, T, ne, EC, Kw, K.abi_dtype(dt), st)
        K.reduce_partials(self.pe_w, ne, Kw * EC * EC, L, True, 1.0, gp + 4 * sec["EW"].offset, Kw * EC * EC)
        K.reduce_partials(self.pe_b, ne, EC, L, True, 1.0, gp + 4 * sec["EB"].offset, EC)
        K.reduce_partials(self.pe_r, ne, EC * EC, L, True, 1.0, gp + 4 * sec["EWR"].offset, EC * EC)
        K.reduce_partials(self.pe_br, ne, EC, L, True, 1.0, gp + 4 * sec["EBR"].offset, EC)
        # 'nc_conv': its residual 1x1 (a[0], dh[0]) ...
        ns = self.nslabs
        K.wgrad(self.a.data_ptr(), NE, EC, self.dh.data_ptr(), NE, EC, None, 1, self.wg_parts, self.wg_bparts, N, T, ns,
                dt)
        K.reduce_partials(self.wg_parts, ns, EC * EC, 1, True, 1.0, gp + 4 * sec["nc_wr"].offset, 0)
        K.reduce_partials(self.wg_bparts, ns, EC, 1, True, 1.0, gp + 4 * sec["nc_br"].offset, 0)
        # ... and its K-tap conv on the raw clip: taps relu(x)[t+k]
        call("srwn_clamp", self.x.data_ptr(), self.xr.data_ptr(), N, 0.0, 3.0e38, st)
        K.init_conv_wgrad(self.xr, self.dpre[0].view(B, T, EC), v("nc_w", g).reshape(-1), v("nc_b", g), Kw,
                          -(Kw - 1) + (Kw - 1) // 2, self.ic_ws)

    def optimizer_step(self, grad_scale: float = 1.0):
        K.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.adam_step, self.lr, grad_scale=grad_scale)
        self.repack()


class AutoEncoderEngine:
    """Encoder + conditioned mixture-of-logistics decoder trained jointly (model.py:103-116, 203-216)."""

    def __init__(self, dec_cfg: StackConfig, batch: int, length: int, encoder_channels: int, latent_channels: int,
                 condition_size: int, device="cuda", seed: int = 0, process_group=None):
        if dec_cfg.head_mode != "mol" or not dec_cfg.shift_input:
            raise ValueError("the decoder is the RightShift-ed mixture-of-logistics stack (model.py:158-200)")
        if dec_cfg.cond_channels != latent_channels + condition_size:
            raise ValueError("decoder cond_channels must be latent_channels + condition_size (model.py:161-167)")
        self.dec = WaveNetEngine(dec_cfg, batch, length, device, seed=seed, process_group=process_group)
        self.enc = EncoderStack(len(dec_cfg.dilations), batch, length, dec_cfg.pool_stride, encoder_channels,
                                dec_cfg.skip_channels, latent_channels, dec_cfg.filter_width, dec_cfg.dtype,
                                dec_cfg.learning_rate, device, seed + 1)
        self.B, self.T, self.N = self.dec.B, self.dec.T, self.dec.N
        self.lat, self.cs = latent_channels, condition_size
        self.denc = torch.zeros((self.enc.rows_c, self.lat), dtype=torch.float32, device=self.dec.dev)
        self.loss = self.dec.loss

    def set_inputs(self, inputs: torch.Tensor, conditions: Optional[torch.Tensor] = None):
        d = self.dec
        self.enc.x.copy_(inputs.reshape(self.B, self.T))
        d.audio.copy_(self.enc.x)
        d.cond_in.zero_()
        if self.cs:
            if conditions is None:
                raise ValueError("this auto-encoder was built with condition_size > 0; pass conditions [B, condition_size]")
            c = conditions.reshape(self.B, 1, self.cs).expand(-1, d.frames, -1)              # model.py:162-165
            d.cond_in.view(self.B, d.frames, d.Ep)[:, :, self.lat:self.lat + self.cs].copy_(c)

    def encode(self):
        """Runs the encoder and writes the encoding into the decoder's conditioning rows."""
        e, d = self.enc, self.dec
        e.forward()
        call("srwn_small_gemm", e.s_mean.data_ptr(), e.S, e.S, 0, K.F32, e.view("lat_w").data_ptr(), e.lat, 1, e.S, 0,
             e.view("lat_b").data_ptr(), d.cond_in.data_ptr(), d.Ep, K.abi_dtype(d.dt), e.rows_c, e.lat, e.S, 0,
             K._stream())
        return e.enc

    def forward(self, want_logits: bool = False):
        self.encode()
        return self.dec.forward(want_logits=want_logits)

    def backward(self):
        d, e = self.dec, self.enc
        d.backward(join=False)   # its weight-gradient tail runs beside the encoder's backward
        # d loss / d encoding = sum_l dcb_l Wc_l^T restricted to the latent columns (model.py:180)
        call("srwn_small_gemm", d.dcb.data_ptr(), d.R, d.R, e.rows_c * d.R, K.abi_dtype(d.dt),
             d.view("WC").data_ptr(), 1, d.R, d.R, d.E * d.R, None, self.denc.data_ptr(), self.lat, K.F32, e.rows_c,
             self.lat, d.L * d.R, 0, K._stream())
        e.backward(self.denc)
        d.join_side()

    def allreduce_grads(self):
        self.dec.allreduce_grads()
        from . import dp
        dp.allreduce_sum_(self.enc.grads, self.dec.pg)

    def optimizer_step(self):
        self.dec.optimizer_step()      # the mixture loss is a SUM over batch and time: shard gradients add
        self.enc.optimizer_step(1.0)

    def train_step(self):
        self.forward()
        self.backward()
        self.allreduce_grads()
        self.optimizer_step()
        return self.loss

    def capture_graphs(self):
        """{forward, backward} and {Adam, re-pack} as two hipGraphs with the all-reduce between them (see
        WaveNetEngine.capture_graphs).  Call after one eager train_step."""
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
        return self.loss
