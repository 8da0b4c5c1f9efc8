"""N>1 path on CPU: world_size-2 gloo processes run the keyframe-sharded BA step (gaus_slam_amd/ba_shard.py) with
a stand-in differentiable render function; the all-reduced bucket must equal the serial sum over keyframes, and
world_size 1 must reproduce the local gradients bit for bit."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaus_slam_amd import ba_shard
from tests import util


def _params(P, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {name: torch.randn(P, k, generator=g).requires_grad_(True) for name, k in ba_shard.BUCKET_FIELDS.items()}


def _fake_render_loss(params, kf):
    """Any differentiable function of all five parameter groups that depends on the keyframe."""
    a = float(kf + 1)
    return ((params["means3D"] * a).sin().sum() + (params["opacities"] * a).sum() ** 2 * 1e-3 +
            (params["scales"] ** 2).sum() * a + (params["rotations"] * params["rotations"].roll(1, 1)).sum() * a +
            (params["colors"] * (a + params["means3D"])).sum())


def _serial(P, keyframes):
    out = None
    for kf in keyframes:
        p = _params(P)
        _fake_render_loss(p, kf).backward()
        flat = torch.cat([p[n].grad.reshape(-1) for n in ba_shard.BUCKET_FIELDS])
        out = flat if out is None else out + flat
    return out


def _worker(rank, world, port, P, keyframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ba = ba_shard.KeyframeShardedBA(_params(P), _fake_render_loss)
        ba.step(keyframes)
        if rank == 0:
            q.put(ba.bucket.flat.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("keyframes", [[0, 1], [0, 1, 2], [5]])
def test_world2_equals_serial_sum(keyframes):
    P, world, port = 257, 2, util.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, keyframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.testing.assert_close(got, _serial(P, keyframes), rtol=1e-5, atol=1e-5)


def test_world1_is_bitwise_local():
    P = 100
    ba = ba_shard.KeyframeShardedBA(_params(P), _fake_render_loss)
    views = ba.step([3])
    p = _params(P)
    _fake_render_loss(p, 3).backward()
    for n in ba_shard.BUCKET_FIELDS:
        assert torch.equal(views[n], p[n].grad)
    assert ba.bucket.flat.numel() == 13 * P


def test_bucket_layout_and_sharding():
    b = ba_shard.GradBucket(10, "cpu")
    assert [tuple(v.shape) for v in b.views.values()] == [(10, 3), (10, 1), (10, 2), (10, 4), (10, 3)]
    assert all(v.is_contiguous() for v in b.views.values()) and ba_shard.BUCKET_FLOATS == 13
    assert ba_shard.shard_keyframes(list(range(5)), 1, 2) == [1, 3]
    b.pack({"means3D": torch.ones(10, 3)})
    assert b.flat[:30].sum() == 30 and b.flat[30:].abs().sum() == 0


def _worker_chunks(rank, world, port, P, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = ba_shard.GradBucket(P, "cpu")
        g = torch.Generator().manual_seed(100 + rank)
        b.flat.copy_(torch.randn(b.flat.numel(), generator=g))
        # the chunked, coalesced, asynchronous reduction the overlapped BA step issues (three ragged chunks)
        for g0, g1 in ((0, 100), (100, 200), (200, P)):
            b.reduce_rows(g0, g1)
        b.wait()
        ba = ba_shard.KeyframeShardedBA(_params(P), _fake_render_loss)
        frames = ba.gather_frame_params(torch.arange(9, dtype=torch.float32) + 10 * rank)
        if rank == 0:
            q.put((b.flat.clone(), frames.clone()))
    finally:
        dist.destroy_process_group()


def test_chunked_reduce_rows_and_frame_param_gather():
    """GradBucket.reduce_rows / wait (per-chunk coalesced all-reduce of the five field slices) sums exactly like one
    all-reduce of the whole bucket, and gather_frame_params all-gathers the 9 rank-local pose / exposure scalars."""
    P, world, port = 257, 2, util.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_chunks, args=(r, world, port, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, frames = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sum(torch.randn(13 * P, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-6)
    assert frames.shape == (2, 9) and torch.equal(frames[1], torch.arange(9, dtype=torch.float32) + 10)


def test_second_backward_inside_one_grad_sink_raises():
    """A loss function that reaches the operator twice under direct_grads would silently lose the second gradient: the
    sink hands itself out once and raises on the second take."""
    from gaus_slam_amd import rasterizer
    with rasterizer.grad_sink({"means3D": None}):
        assert rasterizer._take_sink()[0] is not None
        with pytest.raises(RuntimeError, match="second rasterizer backward"):
            rasterizer._take_sink()
    assert rasterizer._take_sink() == (None, None, None)  # outside a context: plain backward
    with pytest.raises(RuntimeError, match="nested"):
        with rasterizer.grad_sink({}):
            with rasterizer.grad_sink({}):
                pass


def test_k_keyframe_schedule_and_batch_fn_is_a_gpu_path():
    """ba_shard.k_keyframe_schedule: steps / K (rounded up), learning rates x K; a batch_fn is only used on CUDA parameters
    (the batched operator call has no CPU form) -- on CPU the keyframes go through render_loss_fn one by one."""
    from gaus_slam_amd import ba_shard
    steps, lrs = ba_shard.k_keyframe_schedule(4, 1001, {"means3D": 1e-4, "colors": 2.5e-3})
    assert steps == 251 and lrs == {"means3D": 4e-4, "colors": 1e-2}
    assert ba_shard.k_keyframe_schedule(1, 7, {"a": 0.5}) == (7, {"a": 0.5})
    P = 6
    params = {n: torch.randn(P, k, requires_grad=True) for n, k in ba_shard.BUCKET_FIELDS.items()}
    calls = []

    def one(p, kf):
        calls.append(("one", kf))
        return sum((v * (kf + 1)).sum() for v in p.values())

    def batch(p, kfs):
        calls.append(("batch", tuple(kfs)))
        return sum(one(p, k) for k in kfs)

    ba = ba_shard.KeyframeShardedBA(params, one, batch_fn=batch)
    g = ba.step([0, 1, 2])
    assert [c[0] for c in calls] == ["one", "one", "one"]
    for n, k in ba_shard.BUCKET_FIELDS.items():
        assert torch.allclose(g[n], torch.full((P, k), 6.0))
