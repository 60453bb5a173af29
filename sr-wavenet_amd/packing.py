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
