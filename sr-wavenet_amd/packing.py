"""Which fp32 master weight goes where in the MFMA A-operand images (see csrc/srwn_common.h).

All functions register a job on a ``kernels.Packer`` and return the image's element offset in the
packed buffer.  ``src_offset`` is the weight's offset (in floats) inside the flat parameter buffer.
"""
from __future__ import annotations


def pack_conv(pk, src_offset: int, K: int, R: int) -> int:
    """Dilated conv kernel [K,R,R] (ops.py:14) as one (K*R)-deep contraction: rows = out channel,
    k = tap*R + in channel, natural k order (the activation fragments come from LDS row images)."""
    mt, ks_total = R // 32, K * R // 16
    off = pk.reserve(mt, ks_total)
    pk.fill(off, src_offset=src_offset, rows_valid=R, k_valid=K * R, row_stride=1, k_stride=R, mt_count=mt,
            ks_total=ks_total)
    return off


def pack_res(pk, src_offset: int, R: int) -> int:
    """1x1 residual kernel [1,R,R] (ops.py:39); its B operand is an accumulator tile -> permuted k."""
    mt, ks_total = R // 32, R // 16
    off = pk.reserve(mt, ks_total)
    pk.fill(off, src_offset=src_offset, rows_valid=R, k_valid=R, row_stride=1, k_stride=R, mt_count=mt,
            ks_total=ks_total, perm_from_ks=0)
    return off


def fill_linear(pk, image_off: int, src_offset: int, Cin: int, Cout: int, mt_count: int, ks_total: int,
                ks_offset: int = 0, ks_count=None, perm: bool = False):
    """[Cin,Cout] kernel into k-steps [ks_offset, +ks_count) of an existing image (rows = out channel)."""
    pk.fill(image_off, src_offset=src_offset, rows_valid=Cout, k_valid=Cin, row_stride=1, k_stride=Cout,
            mt_count=mt_count, ks_total=ks_total, ks_offset=ks_offset, ks_count=ks_count,
            perm_from_ks=0 if perm else (1 << 30))


def pack_linear(pk, src_offset: int, Cin: int, Cout: int, cout_pad: int, perm: bool = False) -> int:
    """y = x @ W for W [Cin,Cout] (tf.layers.conv1d kernel [1,Cin,Cout]); rows padded to cout_pad."""
    cin_pad = (Cin + 15) // 16 * 16
    off = pk.reserve(cout_pad // 32, cin_pad // 16)
    fill_linear(pk, off, src_offset, Cin, Cout, cout_pad // 32, cin_pad // 16, perm=perm)
    return off


def fill_linear_T(pk, image_off: int, src_offset: int, Cin: int, Cout: int, mt_count: int, ks_total: int,
                  ks_offset: int = 0, ks_count=None, perm: bool = False):
    """dx = dy @ W^T for W [Cin,Cout]: rows = Cin (output of the product), k = Cout."""
    pk.fill(image_off, src_offset=src_offset, rows_valid=Cin, k_valid=Cout, row_stride=Cout, k_stride=1,
            mt_count=mt_count, ks_total=ks_total, ks_offset=ks_offset, ks_count=ks_count,
            perm_from_ks=0 if perm else (1 << 30))


def pack_linear_T(pk, src_offset: int, Cin: int, Cout: int, cin_pad: int, perm: bool = False) -> int:
    cout_pad = (Cout + 15) // 16 * 16
    off = pk.reserve(cin_pad // 32, cout_pad // 16)
    fill_linear_T(pk, off, src_offset, Cin, Cout, cin_pad // 32, cout_pad // 16, perm=perm)
    return off


def pack_conv_T(pk, src_offset: int, K: int, R: int) -> int:
    """Data gradient of the dilated conv [K,R,R]: rows = in channel i, k = tap*R + out channel o
    (A[i][(k,o)] = Wf[k][i][o]); one fill per tap because the source is not affine across taps."""
    mt, ks_total, ks = R // 32, K * R // 16, R // 16
    off = pk.reserve(mt, ks_total)
    for k in range(K):
        pk.fill(off, src_offset=src_offset + k * R * R, rows_valid=R, k_valid=R, row_stride=R, k_stride=1,
                mt_count=mt, ks_total=ks_total, ks_offset=k * ks, ks_count=ks)
    return off


def pack_conv_gen(pk, src_offset: int, K: int, R: int) -> int:
    """Conv image for incremental generation: taps 0..K-2 natural, the last tap (whose B operand is the
    previous layer's accumulator tile) in permuted k order."""
    mt, ks_total = R // 32, K * R // 16
    off = pk.reserve(mt, ks_total)
    pk.fill(off, src_offset=src_offset, rows_valid=R, k_valid=K * R, row_stride=1, k_stride=R, mt_count=mt,
            ks_total=ks_total, perm_from_ks=(K - 1) * R // 16)
    return off


# ----------------------------------------------------------------------------------------------
# 16-row x 32-k A fragments of v_mfma_f32_16x16x32_bf16 (csrc/srwn_gen16.hip): lane l holds row (l & 15), k = 8 (l >> 4) + j
# ----------------------------------------------------------------------------------------------
def frag16_index(src_offset: int, row0: int, k0: int, row_stride: int, k_stride: int, rows_valid: int, k_valid: int):
    """Gather index [64 lanes, 8] of one fragment: element (l, j) = W[row0 + (l & 15)][k0 + 8 (l >> 4) + j] at
    src_offset + row * row_stride + k * k_stride of the flat parameter buffer; -1 (zero) outside rows_valid x k_valid."""
    import numpy as np
    lane = np.arange(64)[:, None]
    j = np.arange(8)[None, :]
    row = row0 + (lane & 15)
    k = k0 + 8 * (lane >> 4) + j
    idx = src_offset + row * row_stride + k * k_stride
    return np.where((row < rows_valid) & (k < k_valid), idx, -1).astype(np.int32)


def gen16_layer_index(sec_wf: int, sec_wr: int, sec_ws: int, l: int, R: int, S: int):
    """[4 waves][conv k-steps | residual k-steps | skip (row blocks) x (k-steps)] fragment images of layer l: wave w owns
    output channels 16w.. of the conv (contraction over [delayed tap | current tap] x R: R/32 k-steps each) and of the
    residual 1x1 (waves beyond R/16: zero fragments), and skip channels (S/4) w .. in S/64 blocks of 16."""
    import numpy as np
    KR, SRB = R // 32, S // 64
    out = []
    for w in range(4):
        for ks in range(2 * KR):  # conv: k < R: tap 0 (the delayed tap, ops.py:6-10), k >= R: tap 1; W[k][i][o] at ((l*2+k)*R + i)*R + o
            tap, i0 = ks // KR, 32 * (ks % KR)
            out.append(frag16_index(sec_wf + (l * 2 + tap) * R * R, 16 * w, i0, 1, R, R, R))
        for ks in range(KR):      # residual 1x1: W[i][o] at (l*R + i)*R + o
            out.append(frag16_index(sec_wr + l * R * R, 16 * w, 32 * ks, 1, R, R, R))
        for rb in range(SRB):     # skip 1x1: W[i][s] at (l*R + i)*S + s
            for ks in range(KR):
                out.append(frag16_index(sec_ws + l * R * S, 16 * SRB * w + 16 * rb, 32 * ks, 1, S, S, R))
    return np.stack(out)


def gen16_head_index(sec_w: int, Cin: int, Cout_ld: int, Cout_valid: int, interleave: bool = False):
    """[4 waves][row blocks][Cin/32 k-steps] fragments of a head 1x1 stored [Cin, Cout_ld] (rows >= Cout_valid zero): the
    first 1x1 (Cin -> Cin): wave w, block rb < Cin/64 = rows (Cin/4) w + 16 rb; the last one (`interleave`, up to 256
    outputs): four blocks per wave, rows 16 (4 rb + w) -- few outputs still split over the four waves."""
    import numpy as np
    out = []
    nrb = 4 if interleave else Cin // 64
    for w in range(4):
        for rb in range(nrb):
            row0 = 16 * (4 * rb + w) if interleave else 16 * nrb * w + 16 * rb
            for ks in range(Cin // 32):
                out.append(frag16_index(sec_w, row0, 32 * ks, 1, Cout_ld, Cout_valid, Cin))
    return np.stack(out)
