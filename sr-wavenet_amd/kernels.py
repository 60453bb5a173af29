"""Torch-tensor front end of the C-ABI: argument validation + pointer/stream plumbing.

PyTorch here only owns device memory and streams; every computation is a HIP kernel of
libsrwn.so launched on torch's current stream.  All shape checks that the kernels assume are
done here on the host *before* a launch, so a wrong shape raises instead of faulting the GPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import ctypes as _ct

import torch

from . import _lib
from ._lib import BF16, EPI_F32, EPI_MASK, EPI_NONE, EPI_RELU, F32, PRO_GATE, PRO_NONE, call

_TORCH2ABI = {torch.float32: F32, torch.bfloat16: BF16}


def abi_dtype(dt: torch.dtype) -> int:
    try:
        return _TORCH2ABI[dt]
    except KeyError:
        raise TypeError("unsupported activation dtype %s (float32 or bfloat16)" % dt)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=None, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP tensor" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s: shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None, shape=None):
    return None if t is None else _chk(t, name, dtype, shape)


# ----------------------------------------------------------------------------------------------
# mu-law (ops.py:82-104)
# ----------------------------------------------------------------------------------------------
def mu_law_encode(audio: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    pa = _chk(audio, "audio", torch.float32)
    codes = torch.empty(audio.shape, dtype=torch.int32, device=audio.device)
    call("srwn_mu_law_encode", pa, codes.data_ptr(), audio.numel(), int(quantization_channels), _stream())
    return codes


def mu_law_decode(codes: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    pc = _chk(codes, "codes", torch.int32)
    out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    call("srwn_mu_law_decode", pc, out.data_ptr(), codes.numel(), int(quantization_channels), _stream())
    return out


# ----------------------------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------------------------
class Packer:
    """Collects the MFMA A-operand images of a model; one int32 index image, one gather per step."""

    def __init__(self, device):
        self.device = device
        self.total = 0
        self._jobs: List[Tuple] = []
        self._raw: List[Tuple] = []
        self.idx: Optional[torch.Tensor] = None

    def reserve(self, mt_count: int, ks_total: int) -> int:
        """Reserves an image of mt_count x ks_total fragments; returns its element offset."""
        off = self.total
        self.total += mt_count * ks_total * 512
        return off

    def fill(self, image_off: int, *, src_offset: int, rows_valid: int, k_valid: int, row_stride: int,
             k_stride: int, mt_count: int, ks_total: int, ks_offset: int = 0, ks_count: Optional[int] = None,
             perm_from_ks: int = 1 << 30):
        ks_count = ks_total - ks_offset if ks_count is None else ks_count
        self._jobs.append((image_off, src_offset, rows_valid, k_valid, row_stride, k_stride, mt_count, ks_total,
                           ks_offset, ks_count, min(perm_from_ks, 1 << 30)))

    def reserve_raw(self, index) -> int:
        """An image given by its gather index itself (int32 array, -1 = zero): for fragment layouts other than the 32-row
        A images `fill` describes (the 16x32 fragments of csrc/srwn_gen16.hip).  Returns its element offset."""
        off = self.total
        self.total += int(index.size)
        self._raw.append((off, index))
        return off

    def finalize(self):
        self.idx = torch.full((max(self.total, 1),), -1, dtype=torch.int32, device=self.device)
        base = self.idx.data_ptr()
        for (off, so, rv, kv, rs, ks_, mt, kst, kso, ksc, pf) in self._jobs:
            call("srwn_pack_a_index", base + 4 * off, so, rv, kv, rs, ks_, mt, kst, kso, ksc, pf, _stream())
        for off, index in self._raw:
            self.idx[off:off + index.size].copy_(torch.as_tensor(index.reshape(-1), dtype=torch.int32))
        return self

    def gather(self, params_flat: torch.Tensor, out: torch.Tensor, start: int = 0, stop: Optional[int] = None,
               rowsum: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """out[i] = (out.dtype) params_flat[idx[i]] (0 where idx<0) for the image elements [start, stop) (default: all).
        rowsum = (matrix [rows, cols] fp32, out [cols] fp32): its column sums are formed in the same launch."""
        _chk(params_flat, "params", torch.float32)
        _chk(out, "packed", None, (max(self.total, 1),))
        stop = self.total if stop is None else int(stop)
        start = int(start)
        if not 0 <= start <= stop <= self.total:
            raise ValueError("gather: range [%d, %d) of %d" % (start, stop, self.total))
        if rowsum is not None:
            mat, vec = rowsum
            _chk(mat, "rowsum matrix", torch.float32)
            _chk(vec, "rowsum out", torch.float32, (mat.shape[-1],))
            call("srwn_pack_gather_rowsum", params_flat.data_ptr(), self.idx.data_ptr() + 4 * start,
                 out.data_ptr() + start * out.element_size(), stop - start, abi_dtype(out.dtype), mat.data_ptr(),
                 int(mat.numel() // mat.shape[-1]), int(mat.shape[-1]), vec.data_ptr(), _stream())
        elif stop > start:
            call("srwn_pack_gather", params_flat.data_ptr(), self.idx.data_ptr() + 4 * start,
                 out.data_ptr() + start * out.element_size(), stop - start, abi_dtype(out.dtype), _stream())
        return out


# ----------------------------------------------------------------------------------------------
# generic causal conv (ops.py:6-20) and the input conv's weight gradient
# ----------------------------------------------------------------------------------------------
def causal_conv1d_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dilation: int = 1,
                      shift: int = 0, out_dtype: torch.dtype = torch.float32,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, T, Cin = x.shape
    K, Cin2, Cout = w.shape
    if Cin != Cin2:
        raise ValueError("conv: x has %d channels, filters expect %d" % (Cin, Cin2))
    px = _chk(x, "x", torch.float32)
    pw = _chk(w, "w", torch.float32)
    pb = _opt(bias, "bias", torch.float32, (Cout,))
    if out is None:
        y = torch.empty((B, T, Cout), dtype=out_dtype, device=x.device)
    else:
        y = out
        _chk(y, "out", None, (B, T, Cout))
    call("srwn_causal_conv1d_fwd", px, pw, pb, y.data_ptr(), B, T, Cin, Cout, K, int(dilation), int(shift),
         abi_dtype(y.dtype), _stream())
    return y


def init_conv_wgrad(audio: torch.Tensor, g: torch.Tensor, gw: Optional[torch.Tensor], gb: Optional[torch.Tensor], K: int,
                    shift: int, workspace: torch.Tensor) -> int:
    """gw = gb = None: only the per-slab partials are left in `workspace` ([slabs][(K+1)*R] = [gw | gb] per slab of
    rows); returns the number of slabs (the caller sums them, e.g. as one more job of reduce_partials_multi)."""
    B, T = audio.shape
    R = g.shape[-1]
    pa = _chk(audio, "audio", torch.float32)
    pg = _chk(g, "g", None, (B, T, R))
    need = _lib.load().srwn_init_conv_wgrad_partials(B, T, R, K)
    if workspace.numel() < need or workspace.dtype != torch.float32:
        raise ValueError("init_conv_wgrad: workspace needs %d floats" % need)
    if (gw is None) != (gb is None):
        raise ValueError("init_conv_wgrad: gw and gb go together")
    if gw is not None:
        _chk(gw, "gw", torch.float32)
        _chk(gb, "gb", torch.float32)
        if gw.numel() != K * R or gb.numel() != R:
            raise ValueError("init_conv_wgrad: gw/gb size")
    call("srwn_init_conv_wgrad", pa, pg, workspace.data_ptr(), None if gw is None else gw.data_ptr(),
         None if gb is None else gb.data_ptr(), B, T, R, K, int(shift), abi_dtype(g.dtype), _stream())
    return int(need // ((K + 1) * R))


# ----------------------------------------------------------------------------------------------
# fused residual layer forward (ops.py:23-46)
# ----------------------------------------------------------------------------------------------
def residual_layer_fwd(x: torch.Tensor, cond: Optional[torch.Tensor], wconv_ptr: int, wres_ptr: int,
                       bias_f: torch.Tensor, bias_r: torch.Tensor, h_out: torch.Tensor, z_out: torch.Tensor, K: int,
                       dilation: int, pool_stride: int = 1, cond_channel_offset: int = 0):
    """cond: [B, frames, >=R] tensor; the layer reads channels [cond_channel_offset, +R) of every frame row."""
    B, T, R = x.shape
    px = _chk(x, "x")
    dt = abi_dtype(x.dtype)
    frames, cstride = 1, R
    pc = None
    if cond is not None:
        _chk(cond, "cond", x.dtype)
        if cond.dim() != 3 or cond.shape[0] != B or cond.shape[2] < cond_channel_offset + R:
            raise ValueError("cond: shape %s for B=%d R=%d offset=%d" % (tuple(cond.shape), B, R, cond_channel_offset))
        frames, cstride = cond.shape[1], cond.shape[2]
        if frames * pool_stride < T:
            raise ValueError("cond: %d frames x pool %d < T=%d" % (frames, pool_stride, T))
        pc = cond.data_ptr() + cond_channel_offset * cond.element_size()
    pbf = _chk(bias_f, "bias_f", torch.float32, (R,))
    pbr = _chk(bias_r, "bias_r", torch.float32, (R,))
    ph = _chk(h_out, "h_out", x.dtype, (B, T, R))
    pz = _chk(z_out, "z_out", x.dtype, (B, T, R))
    call("srwn_residual_layer_fwd", px, pc, wconv_ptr, wres_ptr, pbf, pbr, ph, pz, B, T, R, K, int(dilation),
         frames, int(pool_stride), int(cstride), dt, _stream())


def group_plan(dilations, max_halo: int = 31, max_layers: int = 8):
    """Cuts a dilation list into runs the multi-layer kernels take: [(first, end), ...]."""
    import ctypes as C
    n = len(dilations)
    d = (C.c_int32 * max(n, 1))(*[int(v) for v in dilations])
    starts = (C.c_int32 * (n + 1))()
    k = _lib.load().srwn_group_plan(d, n, int(max_halo), int(max_layers), starts)
    return [(int(starts[i]), int(starts[i + 1])) for i in range(k)]


def group_plan_auto(dilations, B: int, T: int, R: int, dtype: torch.dtype, max_layers: int = 8):
    """The cut that minimises the estimated time of the group kernels for this problem size."""
    import ctypes as C
    n = len(dilations)
    d = (C.c_int32 * max(n, 1))(*[int(v) for v in dilations])
    starts = (C.c_int32 * (n + 1))()
    k = _lib.load().srwn_group_plan_auto(d, n, int(B), int(T), int(R), abi_dtype(dtype), int(max_layers), starts)
    return [(int(starts[i]), int(starts[i + 1])) for i in range(k)]


def _ptr_array(ptrs):
    import ctypes as C
    return (C.c_void_p * len(ptrs))(*[None if p is None else int(p) for p in ptrs])


def residual_group_fwd(x0: torch.Tensor, x_out: torch.Tensor, z_out: torch.Tensor, wconv_ptrs, wres_ptrs,
                       biases_f, biases_r, dilations, K: int = 2, cond: Optional[torch.Tensor] = None,
                       cond_channel_offsets=None, pool_stride: int = 1, seg_rows: int = 0,
                       xT: Optional[torch.Tensor] = None, cT: Optional[torch.Tensor] = None, store_inner_x: bool = True):
    """x_out / z_out: [n,B,T,R] stacks (views of the engine's xs[l0+1:], zs[l0:]); cond: [B, frames, C] whose channels
    [cond_channel_offsets[g], +R) hold the bias of the layer above layer g (None entries: no add), or a list of n
    [B, frames, R] tensors / None, one per layer (the layer-by-layer layout: dense rows).
    xT / cT ([>=n, elems] each, `group_wt_geometry`): also write the layers' weight-gradient tiles (seg_rows from the same
    geometry call); store_inner_x = False: only the group's top layer stores its output rows."""
    import ctypes as C
    B, T, R = x0.shape
    n = len(dilations)
    if not (len(wconv_ptrs) == len(wres_ptrs) == len(biases_f) == len(biases_r) == n):
        raise ValueError("residual_group_fwd: per-layer argument lists differ in length")
    px = _chk(x0, "x0")
    dt = abi_dtype(x0.dtype)
    for name, t in (("x_out", x_out), ("z_out", z_out)):
        _chk(t, name, x0.dtype)
        if t.dim() != 4 or t.shape[0] < n or tuple(t.shape[1:]) != (B, T, R):
            raise ValueError("%s: shape %s, expected [>=%d,%d,%d,%d]" % (name, tuple(t.shape), n, B, T, R))
    pbf = [_chk(b, "bias_f", torch.float32, (R,)) for b in biases_f]
    pbr = [_chk(b, "bias_r", torch.float32, (R,)) for b in biases_r]
    frames, cstride, pcs = 1, R, None
    if isinstance(cond, (list, tuple)):
        if len(cond) != n:
            raise ValueError("cond: %d per-layer tensors for %d layers" % (len(cond), n))
        ptrs = []
        for c in cond:
            if c is None:
                ptrs.append(None)
                continue
            _chk(c, "cond", x0.dtype)
            if c.dim() != 3 or c.shape[0] != B or c.shape[2] != R or c.shape[1] * pool_stride < T or (
                    ptrs and frames != c.shape[1] and any(p is not None for p in ptrs)):
                raise ValueError("cond: shape %s for B=%d T=%d R=%d pool=%d" % (tuple(c.shape), B, T, R, pool_stride))
            frames = c.shape[1]
            ptrs.append(c.data_ptr())
        pcs = _ptr_array(ptrs) if any(p is not None for p in ptrs) else None
    elif cond is not None:
        _chk(cond, "cond", x0.dtype)
        frames, cstride = cond.shape[1], cond.shape[2]
        if cond.dim() != 3 or cond.shape[0] != B or frames * pool_stride < T:
            raise ValueError("cond: shape %s for B=%d T=%d pool=%d" % (tuple(cond.shape), B, T, pool_stride))
        offs = list(cond_channel_offsets)
        for o in offs:
            if o is not None and (o < 0 or o + R > cstride):
                raise ValueError("cond channel offset %r outside %d channels" % (o, cstride))
        pcs = _ptr_array([None if o is None else cond.data_ptr() + o * cond.element_size() for o in offs])
    dl = (C.c_int32 * n)(*[int(d) for d in dilations])
    if xT is not None or cT is not None:
        for name, t in (("xT", xT), ("cT", cT)):
            _chk(t, name, x0.dtype)
            if t.dim() != 2 or t.shape[0] < n:
                raise ValueError("%s: shape %s, expected [>=%d, elems]" % (name, tuple(t.shape), n))
        if xT.shape[1] != cT.shape[1]:
            raise ValueError("xT / cT: layer strides differ")
        call("srwn_residual_group_fwd_wt", px, x_out.data_ptr(), z_out.data_ptr(), B * T * R, xT.data_ptr(), cT.data_ptr(),
             int(xT.shape[1]), 1 if store_inner_x else 0, _ptr_array(wconv_ptrs), _ptr_array(wres_ptrs), _ptr_array(pbf),
             _ptr_array(pbr), pcs,
             frames, int(pool_stride), int(cstride), dl, n, B, T, R, int(K), int(seg_rows), dt, _stream())
        return
    call("srwn_residual_group_fwd", px, x_out.data_ptr(), z_out.data_ptr(), B * T * R, _ptr_array(wconv_ptrs),
         _ptr_array(wres_ptrs), _ptr_array(pbf), _ptr_array(pbr), pcs, frames, int(pool_stride), int(cstride), dl, n,
         B, T, R, int(K), int(seg_rows), dt, _stream())


def group_wt_geometry(dilations, B: int, T: int, R: int, dtype: torch.dtype, seg_rows: int = 0):
    """(seg_rows, tiles per segment, elements of one layer's xT / cT buffer, partial slabs per layer) of a layer group in
    the weight-gradient-tile mode: the cut both of its _wt kernels must be given."""
    import ctypes as C
    n = len(dilations)
    dl = (C.c_int32 * n)(*[int(d) for d in dilations])
    sr, kt, ns, el = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
    call("srwn_group_wt_geometry", dl, n, int(B), int(T), int(R), abi_dtype(dtype), int(seg_rows), C.byref(sr), C.byref(kt),
         C.byref(el), C.byref(ns))
    return sr.value, kt.value, el.value, ns.value


def residual_group_fwd_ic(audio: torch.Tensor, init_w: torch.Tensor, init_b: torch.Tensor, shift: int, x_out: torch.Tensor,
                          z_out: torch.Tensor, wconv_ptrs, wres_ptrs, biases_f, biases_r, dilations, K: int = 2,
                          seg_rows: int = 0, xT: Optional[torch.Tensor] = None, cT: Optional[torch.Tensor] = None,
                          store_inner_x: bool = True):
    """The stack's FIRST layer group with the input conv fused in: `audio` [B,T] fp32 instead of the group's input rows
    (srwn_residual_group_fwd_ic).  Other arguments as residual_group_fwd (no conditioning)."""
    import ctypes as C
    n = len(dilations)
    B, T = audio.shape
    R = z_out.shape[-1]
    if not (len(wconv_ptrs) == len(wres_ptrs) == len(biases_f) == len(biases_r) == n):
        raise ValueError("residual_group_fwd_ic: per-layer argument lists differ in length")
    pa = _chk(audio, "audio", torch.float32)
    pw = _chk(init_w, "init_w", torch.float32)
    pb = _chk(init_b, "init_b", torch.float32, (R,))
    if init_w.numel() != K * R:
        raise ValueError("init_w: %d elements, expected %d" % (init_w.numel(), K * R))
    for name, t in (("x_out", x_out), ("z_out", z_out)):
        _chk(t, name, z_out.dtype)
        if t.dim() != 4 or t.shape[0] < n or tuple(t.shape[1:]) != (B, T, R):
            raise ValueError("%s: shape %s, expected [>=%d,%d,%d,%d]" % (name, tuple(t.shape), n, B, T, R))
    pbf = [_chk(b, "bias_f", torch.float32, (R,)) for b in biases_f]
    pbr = [_chk(b, "bias_r", torch.float32, (R,)) for b in biases_r]
    px = pc = None
    stride = 0
    if xT is not None or cT is not None:
        for name, t in (("xT", xT), ("cT", cT)):
            _chk(t, name, z_out.dtype)
            if t.dim() != 2 or t.shape[0] < n:
                raise ValueError("%s: shape %s, expected [>=%d, elems]" % (name, tuple(t.shape), n))
        px, pc, stride = xT.data_ptr(), cT.data_ptr(), int(xT.shape[1])
    dl = (C.c_int32 * n)(*[int(d) for d in dilations])
    call("srwn_residual_group_fwd_ic", pa, pw, pb, int(shift), x_out.data_ptr(), z_out.data_ptr(), B * T * R, px, pc, stride,
         1 if store_inner_x else 0, _ptr_array(wconv_ptrs), _ptr_array(wres_ptrs), _ptr_array(pbf), _ptr_array(pbr), dl, n,
         B, T, R, int(K), int(seg_rows), abi_dtype(z_out.dtype), _stream())


def residual_group_bwd_wt(g_top: Optional[torch.Tensor], g_out: torch.Tensor, z: torch.Tensor, dcs: Optional[torch.Tensor],
                          xT: torch.Tensor, cT: torch.Tensor, wconvT_ptrs, wresT_ptrs, dilations, part_f: torch.Tensor,
                          part_r: torch.Tensor, part_bf: torch.Tensor, part_br: torch.Tensor, nslabs: int, seg_rows: int,
                          K: int = 2, write_all_g: bool = False, ic: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None):
    """Backward chain of a layer group + its layer weight-gradient partials in one launch (8 waves, output-split; the A
    operands are the forward kernel's weight-gradient tiles xT / cT [>=n, elems]).  g_out / z / dcs: [n,B,T,R] stacks;
    part_*: partial buffers starting at the group's first layer, [n][nslabs][2RR | RR | R | R]; part_f / part_r fp32, or in
    the compute type bf16 -- the kernel then writes them as 16 x 16 blocks in lane order (reduce with layout BLK16).
    ic = (audio [B,T] fp32, partials fp32 [nslabs * 3R], shift): the stack's first group also leaves the input conv's
    kernel + bias gradient partials (8 / (R/16) slabs per workgroup, srwn_init_conv_wgrad's stage-1 layout)."""
    import ctypes as C
    n = len(dilations)
    _, B, T, R = z.shape
    if not (len(wconvT_ptrs) == len(wresT_ptrs) == n):
        raise ValueError("residual_group_bwd_wt: per-layer argument lists differ in length")
    dt = abi_dtype(z.dtype)
    for name, t in (("z", z),) + ((("dcs", dcs),) if dcs is not None else ()):
        _chk(t, name, z.dtype)
        if t.dim() != 4 or t.shape[0] < n or tuple(t.shape[1:]) != (B, T, R):
            raise ValueError("%s: shape %s, expected [>=%d,%d,%d,%d]" % (name, tuple(t.shape), n, B, T, R))
    _chk(g_out, "g_out", z.dtype)
    if g_out.dim() != 4 or g_out.shape[0] < (n if write_all_g else 1) or tuple(g_out.shape[1:]) != (B, T, R):
        raise ValueError("g_out: shape %s" % (tuple(g_out.shape),))
    for name, t in (("xT", xT), ("cT", cT)):
        _chk(t, name, z.dtype)
        if t.dim() != 2 or t.shape[0] < n:
            raise ValueError("%s: shape %s, expected [>=%d, elems]" % (name, tuple(t.shape), n))
    part16 = part_f.dtype == torch.bfloat16
    if part16 and (z.dtype != torch.bfloat16 or part_r.dtype != torch.bfloat16):
        raise ValueError("residual_group_bwd_wt: bf16 partial blocks go with bf16 activations (part_f and part_r alike)")
    for name, t, per in (("part_f", part_f, 2 * R * R), ("part_r", part_r, R * R), ("part_bf", part_bf, R),
                         ("part_br", part_br, R)):
        _chk(t, name, torch.bfloat16 if (part16 and name in ("part_f", "part_r")) else torch.float32)
        if t.numel() < n * nslabs * per:
            raise ValueError("%s: %d elements, needs %d" % (name, t.numel(), n * nslabs * per))
    if ic is not None and ic[1].numel() < nslabs * (8 // (R // 16)) * 3 * R:
        raise ValueError("ic partials: %d floats, needs %d" % (ic[1].numel(), nslabs * (8 // (R // 16)) * 3 * R))
    pg = _opt(g_top, "g_top", z.dtype, (B, T, R))
    dl = (C.c_int32 * n)(*[int(d) for d in dilations])
    call("srwn_residual_group_bwd_wt", pg, g_out.data_ptr(), 1 if write_all_g else 0, z.data_ptr(),
         None if dcs is None else dcs.data_ptr(), B * T * R, xT.data_ptr(), cT.data_ptr(), int(xT.shape[1]),
         _ptr_array(wconvT_ptrs), _ptr_array(wresT_ptrs), dl, n, part_f.data_ptr(), part_r.data_ptr(),
         part_bf.data_ptr(), part_br.data_ptr(), 1 if part16 else 0,
         None if ic is None else _chk(ic[0], "audio", torch.float32, (B, T)),
         None if ic is None else _chk(ic[1], "ic_partials", torch.float32), 0 if ic is None else int(ic[2]),
         int(nslabs), B, T, R, int(K), int(seg_rows), dt, _stream())


def residual_group_bwd(g_top: Optional[torch.Tensor], g_out: torch.Tensor, df_out: torch.Tensor, z: torch.Tensor,
                       dcs: Optional[torch.Tensor], wconvT_ptrs, wresT_ptrs, dilations, K: int = 2, seg_rows: int = 0):
    """g_out / df_out / z / dcs: [n,B,T,R] stacks (views of the engine's gs[l0:], dfs[l0:], zs[l0:], dcs[l0:])."""
    import ctypes as C
    n = len(dilations)
    _, B, T, R = z.shape
    if not (len(wconvT_ptrs) == len(wresT_ptrs) == n):
        raise ValueError("residual_group_bwd: per-layer argument lists differ in length")
    dt = abi_dtype(z.dtype)
    for name, t in (("g_out", g_out), ("df_out", df_out), ("z", z)) + ((("dcs", dcs),) if dcs is not None else ()):
        _chk(t, name, z.dtype)
        if t.dim() != 4 or t.shape[0] < n or tuple(t.shape[1:]) != (B, T, R):
            raise ValueError("%s: shape %s, expected [>=%d,%d,%d,%d]" % (name, tuple(t.shape), n, B, T, R))
    pg = _opt(g_top, "g_top", z.dtype, (B, T, R))
    dl = (C.c_int32 * n)(*[int(d) for d in dilations])
    call("srwn_residual_group_bwd", pg, g_out.data_ptr(), df_out.data_ptr(), z.data_ptr(),
         None if dcs is None else dcs.data_ptr(), B * T * R, _ptr_array(wconvT_ptrs), _ptr_array(wresT_ptrs), dl, n, B, T,
         R, int(K), int(seg_rows), dt, _stream())


# ----------------------------------------------------------------------------------------------
# pointwise linear and the fused softmax head
# ----------------------------------------------------------------------------------------------
def pw_linear(x_ptr: int, x_row_stride: int, x_chunk_stride: int, chunk_len: int, Cin: int, wpack_ptr: int,
              bias: Optional[torch.Tensor], y: torch.Tensor, cout_pad: int, cout_valid: int, rows: int,
              aux: Optional[torch.Tensor] = None, pro: int = PRO_NONE, epi: int = EPI_NONE,
              compute_dtype: Optional[torch.dtype] = None):
    """Raw-pointer input (the skip sum reads a [L,rows,R] stack through chunk strides); y is [rows, >=cout_valid]."""
    py = _chk(y, "y", torch.float32 if epi == EPI_F32 else None)
    if y.shape[0] != rows or y.shape[-1] < cout_valid:
        raise ValueError("pw_linear: y shape %s vs rows=%d cout_valid=%d" % (tuple(y.shape), rows, cout_valid))
    pa, astride = None, 0
    if aux is not None:
        pa = _chk(aux, "aux", y.dtype)
        if aux.shape[0] != rows or aux.shape[-1] < cout_valid:
            raise ValueError("pw_linear: aux shape %s" % (tuple(aux.shape),))
        astride = aux.shape[-1]
    pb = _opt(bias, "bias", torch.float32)
    if bias is not None and bias.numel() < cout_valid:
        raise ValueError("pw_linear: bias too short")
    call("srwn_pw_linear", x_ptr, int(x_row_stride), int(x_chunk_stride), int(chunk_len), int(Cin), wpack_ptr, pb, py,
         y.shape[-1], int(cout_pad), int(cout_valid), int(rows), pa, astride, pro, epi,
         abi_dtype(compute_dtype if epi == EPI_F32 else y.dtype), _stream())


def mol_loss(logits: torch.Tensor, x: torch.Tensor, M: int, loss_partials: torch.Tensor, dlogits: torch.Tensor,
             grad_scale: float = 1.0):
    """Mixture-of-logistics NLL (ops.py:124-175) + gradient: logits [rows, >=4M] fp32, x [rows] fp32."""
    rows = logits.shape[0]
    _chk(logits, "logits", torch.float32)
    if logits.shape[-1] < 4 * M or not (1 <= M <= 16):
        raise ValueError("mol_loss: logits %s for M=%d" % (tuple(logits.shape), M))
    _chk(x, "x", torch.float32, (rows,))
    _chk(loss_partials, "loss_partials", torch.float32)
    if loss_partials.numel() < (rows + 255) // 256:
        raise ValueError("mol_loss: loss_partials needs %d floats" % ((rows + 255) // 256))
    _chk(dlogits, "dlogits")
    if dlogits.shape[0] != rows or dlogits.shape[-1] < 4 * M:
        raise ValueError("mol_loss: dlogits shape %s" % (tuple(dlogits.shape),))
    call("srwn_mol_loss", logits.data_ptr(), logits.shape[-1], x.data_ptr(), int(M), loss_partials.data_ptr(),
         dlogits.data_ptr(), dlogits.shape[-1], rows, float(grad_scale), abi_dtype(dlogits.dtype), _stream())


def head_softmax_ce(x: torch.Tensor, wpack_ptr: int, bias: torch.Tensor, targets: torch.Tensor,
                    loss_partials: torch.Tensor, dlogits: Optional[torch.Tensor], logits_out: Optional[torch.Tensor],
                    cout_pad: int, cout_valid: int, grad_scale: float):
    rows, Cin = x.shape
    px = _chk(x, "x")
    pb = _chk(bias, "bias", torch.float32)
    if bias.numel() < cout_valid:
        raise ValueError("head: bias too short")
    pt = _chk(targets, "targets", torch.int32, (rows,))
    need = (rows + 31) // 32
    pl = _chk(loss_partials, "loss_partials", torch.float32)
    if loss_partials.numel() < need:
        raise ValueError("head: loss_partials needs %d floats" % need)
    pd = _opt(dlogits, "dlogits", x.dtype, (rows, cout_pad))
    plo = _opt(logits_out, "logits_out", torch.float32, (rows, cout_valid))
    call("srwn_head_softmax_ce", px, Cin, Cin, wpack_ptr, pb, pt, pl, pd, plo, int(cout_pad), int(cout_valid), rows,
         float(grad_scale), abi_dtype(x.dtype), _stream())


def head_chain(r0: torch.Tensor, w1_ptr: int, w2p_ptr: int, w2Tp_ptr: int, w1Tp_ptr: int, b1: torch.Tensor,
               b2: torch.Tensor, targets: torch.Tensor, loss_partials: torch.Tensor, r1: torch.Tensor,
               dlogits: torch.Tensor, da1: torch.Tensor, dtotal: torch.Tensor, cout_valid: int, grad_scale: float):
    """srwn_head_chain: head 1x1 + softmax-CE + both head data gradients for [rows, 256] bf16 rows."""
    rows, S = r0.shape
    px = _chk(r0, "r0")
    outs = [_chk(t, n, r0.dtype, (rows, S)) for t, n in ((r1, "r1"), (dlogits, "dlogits"), (da1, "da1"), (dtotal, "dtotal"))]
    pb1 = _chk(b1, "b1", torch.float32)
    pb2 = _chk(b2, "b2", torch.float32)
    if b1.numel() < S or b2.numel() < cout_valid:
        raise ValueError("head_chain: bias too short")
    pt = _chk(targets, "targets", torch.int32, (rows,))
    pl = _chk(loss_partials, "loss_partials", torch.float32)
    if loss_partials.numel() < (rows + 31) // 32:
        raise ValueError("head_chain: loss_partials needs %d floats" % ((rows + 31) // 32))
    call("srwn_head_chain", px, w1_ptr, w2p_ptr, w2Tp_ptr, w1Tp_ptr, pb1, pb2, pt, pl, *outs, int(S), int(S),
         int(cout_valid), rows, float(grad_scale), abi_dtype(r0.dtype), _stream())


def reduce_loss(loss_partials: torch.Tensor, n: int, scale: float, out: torch.Tensor):
    call("srwn_reduce_loss", _chk(loss_partials, "loss_partials", torch.float32), int(n), float(scale),
         _chk(out, "loss", torch.float32), _stream())


# ----------------------------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------------------------
def residual_layer_bwd(g_in: Optional[torch.Tensor], df_up: Optional[torch.Tensor], wconvT_up_ptr: Optional[int],
                       g_out: Optional[torch.Tensor], wresT_ptr: Optional[int], wskipT_ptr: Optional[int],
                       dtotal: Optional[torch.Tensor], z: Optional[torch.Tensor], df_out: Optional[torch.Tensor],
                       B: int, T: int, R: int, S: int, K: int, dilation_up: int, has_up: bool, has_down: bool,
                       dtype: torch.dtype, dcs: Optional[torch.Tensor] = None):
    shp = (B, T, R)
    pdc = _opt(dcs, "dcs", dtype, shp)
    pg = _opt(g_in, "g_in", dtype, shp)
    pdu = _opt(df_up, "df_up", dtype, shp)
    pgo = _opt(g_out, "g_out", dtype, shp)
    pz = _opt(z, "z", dtype, shp)
    pdo = _opt(df_out, "df_out", dtype, shp)
    pdt = None
    if dtotal is not None:
        pdt = _chk(dtotal, "dtotal", dtype)
        if dtotal.numel() != B * T * S:
            raise ValueError("dtotal: %d elements, expected %d" % (dtotal.numel(), B * T * S))
    call("srwn_residual_layer_bwd", pg, pdu, wconvT_up_ptr, pgo, wresT_ptr, wskipT_ptr, pdt, pdc, pz, pdo, B, T, R, S, K,
         int(dilation_up), int(has_up), int(has_down), abi_dtype(dtype), _stream())


def skip_dgrad_all(dtotal: torch.Tensor, wskipT_all_ptr: int, dcs: torch.Tensor, R: int, S: int):
    """dcs[l] = dtotal @ Ws_l^T for every layer (dcs: [L, rows, R])."""
    L, rows, R2 = dcs.shape
    if R2 != R or dtotal.numel() != rows * S:
        raise ValueError("skip_dgrad_all: dcs %s / dtotal %s" % (tuple(dcs.shape), tuple(dtotal.shape)))
    call("srwn_skip_dgrad_all", _chk(dtotal, "dtotal", dcs.dtype), wskipT_all_ptr, _chk(dcs, "dcs"), rows * R, L, rows,
         R, S, abi_dtype(dcs.dtype), _stream())


def wgrad_slabs(rows: int) -> int:
    return int(_lib.load().srwn_wgrad_slabs(int(rows)))


def wgrad(in_ptr: int, in_batch_stride: int, cin: int, dout_ptr: int, dout_batch_stride: int, cout: int,
          shifts: Optional[List[int]], nbatch: int, partials: torch.Tensor, bias_partials: Optional[torch.Tensor],
          rows: int, T: int, nslabs: int, dtype: torch.dtype, pro: int = PRO_NONE, cond_ptr: Optional[int] = None,
          cond_batch_stride: int = 0, cond_frames: int = 1, pool_stride: int = 1, cond_row_stride: int = 0):
    """Raw pointers for in/dout (they index stacks of per-layer tensors); partials are checked for size."""
    import ctypes as C
    pp = _chk(partials, "partials", torch.float32)
    if partials.numel() < nbatch * nslabs * cin * cout:
        raise ValueError("wgrad: partials needs %d floats" % (nbatch * nslabs * cin * cout))
    pb = None
    if bias_partials is not None:
        pb = _chk(bias_partials, "bias_partials", torch.float32)
        if bias_partials.numel() < nbatch * nslabs * cout:
            raise ValueError("wgrad: bias_partials needs %d floats" % (nbatch * nslabs * cout))
    sh = None
    if shifts is not None:
        if len(shifts) != nbatch:
            raise ValueError("wgrad: len(shifts) != nbatch")
        sh = (C.c_int32 * nbatch)(*[int(s) for s in shifts])
    call("srwn_wgrad", in_ptr, int(in_batch_stride), int(cin), dout_ptr, int(dout_batch_stride), int(cout), cond_ptr,
         int(cond_batch_stride), int(cond_frames), int(pool_stride), int(cond_row_stride or cin), sh, int(nbatch), pp,
         pb, int(rows), int(T),
         int(nslabs), int(pro), abi_dtype(dtype), _stream())


def wgrad_layers(x: torch.Tensor, z: torch.Tensor, df: torch.Tensor, g_ptr: int, dilations: List[int],
                 part_f: torch.Tensor, part_r: torch.Tensor, part_bf: torch.Tensor, part_br: torch.Tensor, T: int,
                 nslabs: int, cond_ptr: Optional[int] = None, cond_layer_stride: int = 0, cond_frames: int = 1,
                 pool_stride: int = 1, cond_row_stride: int = 64):
    """x, z, df: [L, rows, 64] stacks; g_ptr: pointer to the [L, rows, 64] stack of G_{l+1} (same strides)."""
    import ctypes as C
    L, rows, R = z.shape
    for t, nm in ((x, "x"), (df, "df")):
        _chk(t, nm, z.dtype)
        if t.shape[-1] != R or t.shape[-2] != rows or t.shape[0] < L:
            raise ValueError("wgrad_layers: %s shape %s" % (nm, tuple(t.shape)))
    _chk(z, "z")
    if len(dilations) != L:
        raise ValueError("wgrad_layers: %d dilations for %d layers" % (len(dilations), L))
    for t, n in ((part_f, L * nslabs * 2 * R * R), (part_r, L * nslabs * R * R), (part_bf, L * nslabs * R),
                 (part_br, L * nslabs * R)):
        _chk(t, "partials", torch.float32)
        if t.numel() < n:
            raise ValueError("wgrad_layers: partial buffer needs %d floats" % n)
    dl = (C.c_int32 * L)(*[int(d) for d in dilations])
    call("srwn_wgrad_layers", x.data_ptr(), z.data_ptr(), df.data_ptr(), g_ptr, rows * R, cond_ptr,
         int(cond_layer_stride), int(cond_frames), int(pool_stride), int(cond_row_stride), dl, L, part_f.data_ptr(),
         part_r.data_ptr(), part_bf.data_ptr(), part_br.data_ptr(), rows, int(T), int(nslabs), R, 2,
         abi_dtype(z.dtype), _stream())


def wgrad256_slabs(rows: int, m_chunks: int, chunk_width: int = 64) -> int:
    return int(_lib.load().srwn_wgrad_wide_slabs(int(rows), int(m_chunks), int(chunk_width)))


def wgrad256(a_ptr: int, a_chunk_stride: int, a_row_stride: int, m_chunks: int, d: torch.Tensor,
             partials: torch.Tensor, bias_partials: Optional[torch.Tensor], rows: int, nslabs: int,
             pro: int = PRO_NONE, chunk_width: int = 64):
    """d: [rows, 256 or 128] tensor; a: raw pointer to chunks of `chunk_width` (64 or 32) channels (see srwn.h)."""
    pd = _chk(d, "d")
    dw = d.shape[-1]
    if d.shape[0] != rows or dw not in (128, 256):
        raise ValueError("wgrad256: d must be [rows, 256 or 128], got %s" % (tuple(d.shape),))
    if chunk_width not in (32, 64) or (m_chunks * chunk_width) % 64:
        raise ValueError("wgrad256: %d chunks of %d channels" % (m_chunks, chunk_width))
    pp = _chk(partials, "partials", torch.float32)
    if partials.numel() < nslabs * m_chunks * chunk_width * dw:
        raise ValueError("wgrad256: partials needs %d floats" % (nslabs * m_chunks * chunk_width * dw))
    pb = None
    if bias_partials is not None:
        pb = _chk(bias_partials, "bias_partials", torch.float32)
        if bias_partials.numel() < nslabs * dw:
            raise ValueError("wgrad256: bias_partials needs %d floats" % (nslabs * dw))
    call("srwn_wgrad_wide", a_ptr, int(a_chunk_stride), int(a_row_stride), int(m_chunks), int(chunk_width), pd, dw, dw,
         pp, pb, int(rows), int(nslabs), int(pro), abi_dtype(d.dtype), _stream())


def wgrad256_pair(a0_ptr: int, d0: torch.Tensor, parts0: torch.Tensor, bparts0: torch.Tensor, a1_ptr: int,
                  d1: torch.Tensor, parts1: torch.Tensor, bparts1: torch.Tensor, a_chunk_stride: int, a_row_stride: int,
                  m_chunks: int, rows: int, nslabs: int, chunk_width: int = 64):
    """Two wgrad256 products of one shape (the head's two 1x1s) as one launch."""
    dw = d0.shape[-1]
    if d0.shape != d1.shape or d0.shape[0] != rows or dw not in (128, 256):
        raise ValueError("wgrad256_pair: d0 %s d1 %s" % (tuple(d0.shape), tuple(d1.shape)))
    need = nslabs * m_chunks * chunk_width * dw
    for t, n in ((parts0, need), (parts1, need), (bparts0, nslabs * dw), (bparts1, nslabs * dw)):
        if t.numel() < n:
            raise ValueError("wgrad256_pair: partial buffer needs %d floats" % n)
    call("srwn_wgrad_wide_pair", a0_ptr, _chk(d0, "d0"), _chk(parts0, "partials0", torch.float32),
         _chk(bparts0, "bias_partials0", torch.float32), a1_ptr, _chk(d1, "d1"), _chk(parts1, "partials1", torch.float32),
         _chk(bparts1, "bias_partials1", torch.float32), int(a_chunk_stride), int(a_row_stride), int(m_chunks),
         int(chunk_width), dw, dw, int(rows), int(nslabs), PRO_NONE, abi_dtype(d0.dtype), _stream())


def _i32_array(vals):
    import ctypes as C
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def wgrad_skip_wt_slabs(st, seg_rows, T: int) -> int:
    return int(_lib.load().srwn_wgrad_skip_wt_slabs(_i32_array(st), _i32_array(seg_rows), len(st), int(T)))


def wgrad_skip_wt(cT: torch.Tensor, st, seg_rows, d: torch.Tensor, partials: torch.Tensor,
                  bias_partials: Optional[torch.Tensor], nslabs: int, B: int, T: int, R: int):
    """Skip 1x1 weight gradients of every layer from the transposed gate outputs cT [L, elems] the forward group kernels
    wrote (st / seg_rows: per layer, the stride and segment length of its group); d: dskip [B*T, 256].  partials: fp32
    [slab][L*R][S], or bf16: the kernel then writes each slab's matrix as 16 x 16 blocks in lane order (BLK16, S columns)."""
    L = cT.shape[0]
    if len(st) != L or len(seg_rows) != L or d.shape[0] != B * T:
        raise ValueError("wgrad_skip_wt: %d layers, st %d, seg_rows %d, d %s" % (L, len(st), len(seg_rows), tuple(d.shape)))
    S = d.shape[-1]
    part16 = partials.dtype == torch.bfloat16
    pp = _chk(partials, "partials", torch.bfloat16 if part16 else torch.float32)
    if partials.numel() < nslabs * L * R * S:
        raise ValueError("wgrad_skip_wt: partials needs %d elements" % (nslabs * L * R * S))
    pb = None
    if bias_partials is not None:
        pb = _chk(bias_partials, "bias_partials", torch.float32)
        if bias_partials.numel() < nslabs * S:
            raise ValueError("wgrad_skip_wt: bias_partials needs %d floats" % (nslabs * S))
    call("srwn_wgrad_skip_wt", _chk(cT, "cT"), int(cT.stride(0)), _i32_array(st), _i32_array(seg_rows), L, _chk(d, "d"),
         int(d.stride(0)), pp, pb, 1 if part16 else 0, int(nslabs), int(B), int(T), int(R), int(S), abi_dtype(d.dtype),
         _stream())


def reduce_partials(partials: torch.Tensor, nslabs: int, n: int, nbatch: int, partials_batched: bool, scale: float,
                    out_ptr: int, out_batch_stride: int):
    call("srwn_reduce_partials", _chk(partials, "partials", torch.float32), int(nslabs), int(n), int(nbatch),
         int(bool(partials_batched)), float(scale), out_ptr, int(out_batch_stride), _stream())


class _ReduceJob(_ct.Structure):
    """include/srwn.h: SrwnReduceJob"""
    _fields_ = [("partials", _ct.c_void_p), ("nslabs", _ct.c_int32), ("n", _ct.c_int64), ("nbatch", _ct.c_int32),
                ("partials_batched", _ct.c_int32), ("scale", _ct.c_float), ("out", _ct.c_void_p),
                ("out_batch_stride", _ct.c_int64), ("layout", _ct.c_int32), ("blk_cols", _ct.c_int32)]


def reduce_partials_multi(jobs):
    """jobs: the argument tuples of `reduce_partials`, finished by one launch (at most 16).  A ninth entry `blk_cols`
    marks a bf16 partial buffer in 16 x 16-block lane order (SRWN_PARTIALS_BLK16; n = rows * blk_cols); the string "sum"
    there: one output, the sum of all nslabs * n values in reduce_loss's order (SRWN_PARTIALS_SUM)."""
    arr = (_ReduceJob * len(jobs))()
    for j, job in zip(arr, jobs):
        partials, nslabs, n, nbatch, batched, scale, out_ptr, out_stride = job[:8]
        blk_cols = int(job[8]) if len(job) > 8 and job[8] != "sum" else 0
        j.partials = _chk(partials, "partials", torch.bfloat16 if blk_cols else torch.float32)
        j.layout, j.blk_cols = (1, blk_cols) if blk_cols else ((2, 0) if (len(job) > 8 and job[8] == "sum") else (0, 0))
        j.nslabs, j.n, j.nbatch, j.partials_batched = int(nslabs), int(n), int(nbatch), int(bool(batched))
        j.scale, j.out, j.out_batch_stride = float(scale), int(out_ptr), int(out_stride)
    call("srwn_reduce_partials_multi", _ct.addressof(arr), len(jobs), _stream())


def frame_sum(g: torch.Tensor, frames: int, pool_stride: int) -> torch.Tensor:
    B, T, Cc = g.shape
    out = torch.empty((B, frames, Cc), dtype=g.dtype, device=g.device)
    call("srwn_frame_sum", _chk(g, "g"), out.data_ptr(), B, T, Cc, int(frames), int(pool_stride), abi_dtype(g.dtype),
         _stream())
    return out


def adam_step(params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: torch.Tensor,
              lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0):
    n = params.numel()
    for t, nm in ((params, "params"), (grads, "grads"), (m, "m"), (v, "v")):
        _chk(t, nm, torch.float32, (n,))
    _chk(step, "step", torch.int64, (1,))
    call("srwn_adam_step", params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), n, step.data_ptr(),
         float(lr), float(beta1), float(beta2), float(eps), float(grad_scale), _stream())


# ----------------------------------------------------------------------------------------------
# Parallel-WaveNet student kernels (model.py:290-537)
# ----------------------------------------------------------------------------------------------
def flow_partials(rows: int) -> int:
    return int(_lib.load().srwn_flow_partials(int(rows)))


def flow_affine_fwd(h: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, x_in: torch.Tensor, prm: torch.Tensor,
                    x_out: torch.Tensor, ent_partials: torch.Tensor):
    """prm = relu(h) @ w2 + b2; x_out = x_in*exp(prm0) + prm1 (model.py:451-452, 479-483)."""
    rows, R = h.shape
    _chk(w2, "w2", torch.float32, (R, 2)); _chk(b2, "b2", torch.float32, (2,))
    _chk(x_in, "x_in", torch.float32, (rows,)); _chk(prm, "prm", torch.float32, (rows, 2))
    _chk(x_out, "x_out", torch.float32, (rows,))
    _chk(ent_partials, "ent_partials", torch.float32, (flow_partials(rows),))
    call("srwn_flow_affine_fwd", _chk(h, "h"), w2.data_ptr(), b2.data_ptr(), x_in.data_ptr(), prm.data_ptr(),
         x_out.data_ptr(), ent_partials.data_ptr(), rows, R, abi_dtype(h.dtype), _stream())


def flow_affine_bwd(h: torch.Tensor, w2: torch.Tensor, prm: torch.Tensor, x_in: torch.Tensor, dx_out: torch.Tensor,
                    ent_grad: float, g: torch.Tensor, dx_in: torch.Tensor, w_partials: torch.Tensor):
    rows, R = h.shape
    _chk(w2, "w2", torch.float32, (R, 2)); _chk(prm, "prm", torch.float32, (rows, 2))
    _chk(x_in, "x_in", torch.float32, (rows,)); _chk(dx_out, "dx_out", torch.float32, (rows,))
    _chk(g, "g", h.dtype, (rows, R)); _chk(dx_in, "dx_in", torch.float32, (rows,))
    _chk(w_partials, "w_partials", torch.float32, (flow_partials(rows), 2 * R + 2))
    call("srwn_flow_affine_bwd", _chk(h, "h"), w2.data_ptr(), prm.data_ptr(), x_in.data_ptr(), dx_out.data_ptr(),
         float(ent_grad), g.data_ptr(), dx_in.data_ptr(), w_partials.data_ptr(), rows, R, abi_dtype(h.dtype), _stream())


def causal_conv1d_dgrad(dy: torch.Tensor, w: torch.Tensor, dx: torch.Tensor, dilation: int = 1, shift: int = 0,
                        accumulate: bool = False, scale: float = 1.0):
    """dx[b,u,i] (+)= scale * sum_{k,o} w[k,i,o] * dy[b, u+shift+(K-1-k)*d, o] (adjoint of ops.py:6-10 + RightShift)."""
    B, T, Cout = dy.shape
    K, Cin, Co2 = w.shape
    if Co2 != Cout:
        raise ValueError("causal_conv1d_dgrad: w %s vs dy %s" % (tuple(w.shape), tuple(dy.shape)))
    _chk(w, "w", torch.float32); _chk(dx, "dx", torch.float32, (B, T, Cin))
    call("srwn_causal_conv1d_dgrad", _chk(dy, "dy"), w.data_ptr(), dx.data_ptr(), B, T, Cin, Cout, K, int(dilation),
         int(shift), int(bool(accumulate)), float(scale), abi_dtype(dy.dtype), _stream())


def mol_loss_dx(logits: torch.Tensor, x: torch.Tensor, M: int, loss_partials: torch.Tensor, dx: torch.Tensor,
                grad_scale: float = 1.0):
    rows = x.numel()
    _chk(logits, "logits", torch.float32); _chk(x, "x", torch.float32); _chk(dx, "dx", torch.float32)
    if logits.shape[0] != rows or dx.numel() != rows or loss_partials.numel() < (rows + 255) // 256:
        raise ValueError("mol_loss_dx: shapes")
    call("srwn_mol_loss_dx", logits.data_ptr(), logits.stride(0), x.data_ptr(), int(M),
         _chk(loss_partials, "loss_partials", torch.float32), dx.data_ptr(), rows, float(grad_scale), _stream())


def stft_frames(T: int) -> int:
    return int(_lib.load().srwn_stft_frames(int(T)))


def stft_power(x: torch.Tensor, spec: Optional[torch.Tensor], frame_power: torch.Tensor, power: torch.Tensor):
    B, T = x.shape
    nf = stft_frames(T)
    _chk(x, "x", torch.float32); _chk(frame_power, "frame_power", torch.float32, (B, nf, 257))
    _chk(power, "power", torch.float32, (B, 257))
    ps = _opt(spec, "spec", torch.float32, (B, nf, 257, 2))
    call("srwn_stft_power", x.data_ptr(), ps, frame_power.data_ptr(), power.data_ptr(), B, T, _stream())


def power_loss(power_truth: torch.Tensor, power_out: torch.Tensor, gamma: float, grad_scale: float,
               dpower: Optional[torch.Tensor], loss: torch.Tensor):
    n = power_truth.numel()
    _chk(power_truth, "power_truth", torch.float32); _chk(power_out, "power_out", torch.float32, power_truth.shape)
    call("srwn_power_loss", power_truth.data_ptr(), power_out.data_ptr(), n, float(gamma), float(grad_scale),
         _opt(dpower, "dpower", torch.float32, power_truth.shape), _chk(loss, "loss", torch.float32), _stream())


def stft_power_bwd(spec: torch.Tensor, dpower: torch.Tensor, dx: torch.Tensor, accumulate: bool = False):
    B, T = dx.shape
    nf = stft_frames(T)
    _chk(spec, "spec", torch.float32, (B, nf, 257, 2)); _chk(dpower, "dpower", torch.float32, (B, 257))
    call("srwn_stft_power_bwd", spec.data_ptr(), dpower.data_ptr(), _chk(dx, "dx", torch.float32), B, T,
         int(bool(accumulate)), _stream())


def sumsq_partials(n: int) -> int:
    return int(_lib.load().srwn_sumsq_partials(int(n)))


def sumsq(g: torch.Tensor, partials: torch.Tensor):
    _chk(g, "g", torch.float32)
    if partials.numel() < sumsq_partials(g.numel()):
        raise ValueError("sumsq: partials too small")
    call("srwn_sumsq", g.data_ptr(), g.numel(), _chk(partials, "partials", torch.float32), _stream())


def clip_scale(partials: torch.Tensor, clip_norm: float, pre_scale: float, out: torch.Tensor):
    _chk(out, "out", torch.float32, (2,))
    call("srwn_clip_scale", _chk(partials, "partials", torch.float32), partials.numel(), float(clip_norm),
         float(pre_scale), out.data_ptr(), _stream())


def adam_step_scaled(params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: torch.Tensor,
                     lr: float, scale_dev: torch.Tensor, tick: bool, beta1: float = 0.9, beta2: float = 0.999,
                     eps: float = 1e-8):
    n = params.numel()
    for t, nm in ((params, "params"), (grads, "grads"), (m, "m"), (v, "v")):
        _chk(t, nm, torch.float32, (n,))
    _chk(step, "step", torch.int64, (1,)); _chk(scale_dev, "scale_dev", torch.float32)
    call("srwn_adam_step_scaled", params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), n, step.data_ptr(),
         float(lr), float(beta1), float(beta2), float(eps), scale_dev.data_ptr(), int(bool(tick)), _stream())
