"""The engine's data-parallel step with TWO real ranks: two fresh child processes share
the box's one GPU, the gradient all-reduce goes through gloo, each rank takes half of a fixed global batch.  After one
eager and two graph-replayed steps both ranks must hold the same parameters, equal to a single process stepping on the
whole batch: mean losses (model.py:29) average the shard gradients (Adam's 1/world), the mixture-of-logistics SUM loss
(ops.py:171-172) adds them, the student clips AFTER the mean (model.py:384-385).  Both all-reduce schedules of the
engine run: one bucket, and two buckets with the first overlapped with the lower backward pass."""
import os
import subprocess
import sys

import pytest
import torch

from tests._pkg import ROOT

pytestmark = pytest.mark.gpu

WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _run(mode, world, tmp_path, tag, buckets, port, backend="gloo", force_dist=False):
    outs, procs = [], []
    for r in range(world):
        out = str(tmp_path / ("%s_%s_w%d_r%d.pt" % (mode, tag, world, r)))
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SRWN_DIST_BACKEND=backend, SRWN_BUCKETS=buckets,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("SRWN_FORCE_DIST", None)
        if force_dist:
            env["SRWN_FORCE_DIST"] = "1"
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, out], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs.append(out)
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l
    return [torch.load(o, weights_only=True) for o in outs]


@pytest.mark.parametrize("mode,buckets", [("softmax", "0"), ("softmax", "1"), ("mol", "1"), ("student", "0"), ("deep", "1")])
def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path, mode, buckets):
    """"deep": BASELINE config 3's stack and dtype (30 layers, bf16) with two ranks through the schedule the scaling bench
    replays: backward cut at layer 10, first bucket all-reduced beside the lower part, three hipGraphs.  The shards are
    whole clips, every kernel works clip by clip, so even in bf16 two ranks and one process form the same per-clip
    gradients: only the order of the fp32 sums differs."""
    port = 29600 + (os.getpid() % 200) + {"softmax": 0, "mol": 1, "student": 2, "deep": 3}[mode] * 3 + int(buckets)
    ref = _run(mode, 1, tmp_path, "ref", "0", port)[0]
    r0, r1 = _run(mode, 2, tmp_path, "dp" + buckets, buckets, port + 400)
    assert r0["info"]["world"] == 2 and r1["info"]["world"] == 2
    if mode != "student":
        assert r0["info"]["bucketed"] == (buckets == "1"), r0["info"]
        assert r0["info"]["fused"]
    if mode == "deep":
        assert r0["info"]["layers"] == 30 and r0["info"]["split_layer"] == 10, r0["info"]
    assert torch.equal(r0["params"], r1["params"]), "ranks diverged"
    # the all-reduced gradient buffer of the FIRST step (identical parameters everywhere): SUM over ranks of the shard
    # gradients.  Mean losses: sum / world is the global-batch gradient; the mixture-of-logistics SUM loss: the sum
    # itself.  (Adam is invariant to the scale of the gradient, so the parameters alone would not show a wrong 1/world.)
    g, gref = r0["grads1"].double(), ref["grads1"].double()
    assert torch.equal(r0["grads1"], r1["grads1"])
    scale = 1.0 if mode == "mol" else 0.5
    gerr = float((g * scale - gref).abs().max() / gref.abs().max())
    assert gerr < (1e-4 if mode == "deep" else 1e-5), gerr
    # parameters after the first Adam step and after the two graph-replayed ones, where the gradient is not numerically
    # zero (Adam turns the sign of a ~1e-9 gradient entry into a 1e-3 step, and later steps amplify that)
    # ("deep": the bf16 gradients of two ranks and one process may differ by 1e-4 of the largest entry -- the bound
    # above -- so an entry has to be ten times that to keep its sign and size through Adam's normalisation)
    live = gref.abs() > (1e-3 if mode == "deep" else 1e-4) * gref.abs().max()
    err1 = float((r0["params1"].double() - ref["params1"].double())[live].abs().max())
    assert err1 < 1e-6, err1
    p, q = r0["params"].double(), ref["params"].double()
    err = float((p - q)[live].abs().max())
    # (two more Adam steps amplify what the first one left: one step is lr = 1e-3 per entry whatever the gradient's size, so
    # the bound asks that no live entry has gone a whole step apart; "deep", bf16 with bf16 partial blocks: 5.2e-4 measured)
    assert err < (1e-3 if mode == "deep" else 3e-4), err
    # losses: each rank reports its shard's loss; mean losses average to the global one, sum losses add
    l2 = float(r0["loss"]) + float(r1["loss"])
    lg = float(ref["loss"])
    if mode == "mol":
        assert abs(l2 - lg) < 1e-4 * abs(lg)
    else:
        assert abs(l2 / 2 - lg) < 1e-4 * abs(lg)


@pytest.mark.parametrize("mode", ["deep", "softmax"])
def test_rccl_one_rank_three_graph_schedule_equals_plain_step(tmp_path, mode):
    """The engine's bucketed data-parallel schedule over backend "nccl" (= RCCL) -- {forward, upper backward} | all-reduce of
    the skip + head bucket in flight on RCCL's stream | {lower backward} | all-reduce of the layer bucket | {Adam, re-pack},
    eager once and then as three hipGraphs -- with ONE rank, the only RCCL configuration a one-GPU box allows.  The
    collectives are identities, so parameters, first-step gradients and loss must equal the plain one-graph step bit for
    bit: what this holds is the stream ordering between RCCL's stream and the replayed graphs (a bucket reduced before
    its gradients are final, or Adam replayed before the second bucket landed, would show).  "deep" = the benchmark's
    stack and dtype (30 layers, bf16, cut at layer 10); the gloo twin of this test is
    tests/test_gpu_engine.py::test_bucketed_allreduce_schedule_matches_plain_step."""
    port = 29900 + (os.getpid() % 90) + (0 if mode == "deep" else 97)
    ref = _run(mode, 1, tmp_path, "plain", "0", port)[0]
    got = _run(mode, 1, tmp_path, "rccl1", "1", port + 200, backend="nccl", force_dist=True)[0]
    assert ref["info"]["backend"] == "none" and not ref["info"]["bucketed"] and ref["info"]["graphs"] == 1
    assert got["info"]["backend"] == "nccl" and got["info"]["bucketed"] and got["info"]["graphs"] == 3, got["info"]
    assert got["info"]["world"] == 1
    assert torch.equal(got["grads1"], ref["grads1"])
    assert torch.equal(got["params1"], ref["params1"])
    assert torch.equal(got["params"], ref["params"])
    assert float(got["loss"]) == float(ref["loss"])
