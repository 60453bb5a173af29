"""Multi-layer (group) kernels vs their per-layer twins: the fused launches must reproduce what one launch per
layer stores (forward: bit for bit -- same MFMA order; backward: to accumulation-order round-off), for every
stride / halo / segment shape the grouping produces, in both dtypes.  The per-layer path itself is held to the
oracle in test_gpu_kernels / test_gpu_bwd / test_gpu_engine (and, fused path on by default, so is this one)."""
import numpy as np
import pytest
import torch

from oracle import wavenet_np as O
from tests._pkg import sub
from tests.test_gpu_kernels import DEV

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, dil, B, T, R, S, C, dt, E=0, pool=1, seg_rows=0, seed=3):
    """The same model twice: one launch per layer (SRWN_FUSE=0) and the multi-layer kernels."""
    EG = sub("engine")
    cfg = EG.StackConfig(dilations=list(dil), dilation_channels=R, skip_channels=S, output_channels=C,
                         cond_channels=E, pool_stride=pool, shift_input=True, dtype=dt)
    monkeypatch.setenv("SRWN_FUSE", "0")
    ref = EG.WaveNetEngine(cfg, B, T, DEV, seed=seed)
    monkeypatch.setenv("SRWN_FUSE", "1")
    monkeypatch.setenv("SRWN_SEG_ROWS", str(seg_rows))
    fus = EG.WaveNetEngine(cfg, B, T, DEV, seed=seed)
    assert not ref.fuse_fwd and fus.fuse_fwd
    # biases are zero at init (tf.layers.conv1d defaults): make every one of them count
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name in ("BF", "BR", "BS", "init_b", "head_b1"):
        ref.view(name).copy_(0.1 * torch.randn(ref.view(name).shape, generator=g))
    fus.params.copy_(ref.params)
    ref.repack(); fus.repack()
    rng = np.random.default_rng(seed)
    audio = torch.tensor(np.clip(0.5 * np.sin(np.arange(B * T).reshape(B, T) * 0.05) + 0.1 * rng.normal(size=(B, T)), -1, 1),
                         dtype=torch.float32, device=DEV)
    tg = torch.tensor(rng.integers(0, C, size=(B, T)), dtype=torch.int32, device=DEV)
    cond = None
    if E:
        cond = torch.tensor(rng.normal(size=(B, T // pool, E)), dtype=torch.float32, device=DEV)
    for e in (ref, fus):
        e.set_inputs(audio, tg, cond)
    return ref, fus


SHAPES = [
    # dilations,                      B, T,    R,  S,   seg_rows
    ([1, 2, 4, 8, 16],                2, 700,  64, 256, 0),      # one stride-1 group, several segments per clip
    ([1, 2, 4, 8, 16] * 2,            1, 333,  32, 128, 0),      # the reference scripts' width, ragged length
    ([32, 64, 128, 256, 512],         2, 2100, 64, 256, 0),      # stride 32: residue classes, lengths differ by one
    ([4, 8, 16, 32],                  2, 515,  64, 256, 96),     # stride 4, forced short segments (several per class)
    ([1, 2, 4, 8, 16, 32, 64, 128],   1, 1100, 64, 256, 64),     # cut into {1..16} and {32..128}
    ([3, 6, 1, 2],                    2, 257,  32, 128, 0),      # gcd 3 then gcd 1
    ([1, 2, 4],                       3, 1,    64, 256, 0),      # T = 1
    ([2, 2, 2],                       1, 31,   64, 256, 0),      # fewer steps than one tile, halo 3 at stride 2
    ([512, 1],                        1, 600,  64, 256, 0),      # singletons fall back to the per-layer kernel
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil,B,T,R,S,seg", SHAPES)
def test_group_forward_equals_per_layer(monkeypatch, dt, dil, B, T, R, S, seg):
    ref, fus = _pair(monkeypatch, dil, B, T, R, S, 64, dt, seg_rows=seg)
    assert any(l1 - l0 >= 2 for l0, l1 in fus.groups) or dil == [512, 1]
    ref.forward(); fus.forward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.zs[l], fus.zs[l]), "z of layer %d" % l
        assert torch.equal(ref.xs[l + 1], fus.xs[l + 1]), "x of layer %d" % (l + 1)
    assert torch.equal(ref.loss, fus.loss)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("E,pool", [(16, 8), (40, 50)])
def test_group_forward_conditioned(monkeypatch, dt, E, pool):
    dil = [1, 2, 4, 8, 16, 32, 64]
    ref, fus = _pair(monkeypatch, dil, 2, 400, 64, 256, 64, dt, E=E, pool=pool, seg_rows=128)
    ref.forward(); fus.forward()
    torch.cuda.synchronize()
    for l in range(len(dil)):
        assert torch.equal(ref.zs[l], fus.zs[l]), "z of layer %d" % l
        assert torch.equal(ref.xs[l + 1], fus.xs[l + 1]), "x of layer %d" % (l + 1)


def test_group_plan():
    Kn = sub("kernels")
    d = [2 ** i for i in range(10)] * 3
    assert Kn.group_plan(d, 31, 8) == [(0, 5), (5, 10), (10, 15), (15, 20), (20, 25), (25, 30)]
    assert Kn.group_plan(d, 63, 8) == [(0, 6), (6, 10), (10, 16), (16, 20), (20, 26), (26, 30)]
    assert Kn.group_plan([1, 2, 4, 8, 16] * 2, 31, 8) == [(0, 5), (5, 10)]
    assert Kn.group_plan([1, 2, 4, 8, 16] * 2, 31, 3) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert Kn.group_plan([512, 1], 31, 8) == [(0, 1), (1, 2)]
    assert Kn.group_plan([], 31, 8) == []
