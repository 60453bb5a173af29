"""CPU tests of the host-side index builders for the 16-row MFMA fragment images (sr-wavenet_amd/packing.py): the generators
(csrc/srwn_gen16.hip) read their weights through these gather indices, so a wrong index is a wrong model, not a crash.
Each image is rebuilt here from a plain numpy weight tensor and multiplied out the way v_mfma_f32_16x16x32_bf16 would
(lane l = row l & 15, k = 8 (l >> 4) + j), against the plain product."""
import numpy as np
import pytest

from tests._pkg import sub

P = sub("packing")


def _frag_matrix(flat, idx):
    """[64 lanes, 8] gather index -> the 16 x 32 matrix the fragment stands for (zeros where the index is -1)."""
    m = np.zeros((16, 32))
    lane = np.arange(64)[:, None]
    j = np.arange(8)[None, :]
    vals = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0)
    m[(lane & 15).repeat(8, 1), (8 * (lane >> 4) + j)] = vals
    return m


def test_frag16_index_is_a_16_by_32_window():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((40, 70))                    # [rows, k], row-major
    idx = P.frag16_index(5, 16, 32, 70, 1, 40, 70)       # source offset 5, rows 16.., k 32..
    flat = np.concatenate([np.zeros(5), W.ravel()])
    m = _frag_matrix(flat, idx)
    want = np.zeros((16, 32))
    want[:, :] = np.pad(W, ((0, 8), (0, 26)))[16:32, 32:64]      # rows 32..39 and k 64..69 exist, beyond: zero
    assert np.array_equal(m, want)
    # transposed source (row stride 1, k stride = leading dimension): W^T windows
    idx_t = P.frag16_index(0, 0, 0, 1, 70, 70, 40)
    assert np.array_equal(_frag_matrix(W.ravel(), idx_t), W.T[:16, :32])


@pytest.mark.parametrize("R,S", [(64, 256), (32, 128), (32, 256), (64, 128)])
def test_gen16_layer_image_reproduces_the_layer_products(R, S):
    """conv (two taps), residual 1x1 and skip 1x1 of one layer from its fragment image == the plain products."""
    rng = np.random.default_rng(1)
    L, l = 3, 1
    WF = rng.standard_normal((L, 2, R, R)); WR = rng.standard_normal((L, R, R)); WS = rng.standard_normal((L, R, S))
    flat = np.concatenate([WF.ravel(), WR.ravel(), WS.ravel()])
    o_wf, o_wr, o_ws = 0, WF.size, WF.size + WR.size
    img = P.gen16_layer_index(o_wf, o_wr, o_ws, l, R, S)
    KR, SRB = R // 32, S // 64
    per_wave = 2 * KR + KR + SRB * KR
    assert img.shape == (4 * per_wave, 64, 8)
    x0 = rng.standard_normal(R); x1 = rng.standard_normal(R); c = rng.standard_normal(R)     # delayed tap, current tap, gate output
    conv = np.zeros(R); res = np.zeros(R); skip = np.zeros(S)
    for w in range(4):
        fr = img[w * per_wave:(w + 1) * per_wave]
        for ks in range(2 * KR):          # k < R: delayed tap; k >= R: current tap
            tap, i0 = ks // KR, 32 * (ks % KR)
            xin = (x0 if tap == 0 else x1)[i0:i0 + 32]
            rows = _frag_matrix(flat, fr[ks]) @ xin
            if 16 * w < R:
                conv[16 * w:16 * w + 16] += rows
            else:
                assert not rows.any()     # waves without channels of the chain hold zero fragments
        for ks in range(KR):
            rows = _frag_matrix(flat, fr[2 * KR + ks]) @ c[32 * ks:32 * ks + 32]
            if 16 * w < R:
                res[16 * w:16 * w + 16] += rows
        for rb in range(SRB):
            for ks in range(KR):
                r0 = 16 * SRB * w + 16 * rb
                skip[r0:r0 + 16] += _frag_matrix(flat, fr[3 * KR + KR * rb + ks]) @ c[32 * ks:32 * ks + 32]
    assert np.allclose(conv, x0 @ WF[l, 0] + x1 @ WF[l, 1])
    assert np.allclose(res, c @ WR[l])
    assert np.allclose(skip, c @ WS[l])


@pytest.mark.parametrize("S,C", [(256, 256), (256, 40), (128, 256), (128, 100)])
def test_gen16_head_images_cover_every_output_row_once(S, C):
    rng = np.random.default_rng(2)
    Cp = (C + 31) // 32 * 32
    W1 = rng.standard_normal((S, S)); W2 = np.zeros((S, Cp)); W2[:, :C] = rng.standard_normal((S, C))
    flat = np.concatenate([W1.ravel(), W2.ravel()])
    h = rng.standard_normal(S)
    nks = S // 32
    i1 = P.gen16_head_index(0, S, S, S)
    nrb = S // 64
    assert i1.shape == (4 * nrb * nks, 64, 8)
    y1 = np.zeros(S)
    for w in range(4):
        for rb in range(nrb):
            for ks in range(nks):
                r0 = 16 * nrb * w + 16 * rb
                y1[r0:r0 + 16] += _frag_matrix(flat, i1[(w * nrb + rb) * nks + ks]) @ h[32 * ks:32 * ks + 32]
    assert np.allclose(y1, h @ W1)
    i2 = P.gen16_head_index(W1.size, S, Cp, Cp, interleave=True)
    assert i2.shape == (4 * 4 * nks, 64, 8)
    y2 = np.zeros(256)
    for w in range(4):
        for rb in range(4):
            for ks in range(nks):
                r0 = 16 * (4 * rb + w)                    # the interleaved row blocks of the last 1x1
                y2[r0:r0 + 16] += _frag_matrix(flat, i2[(w * 4 + rb) * nks + ks]) @ h[32 * ks:32 * ks + 32]
    assert np.allclose(y2[:Cp], h @ W2) and not y2[Cp:].any()
