"""Two ranks through the REAL rasterizer on one GPU (rehearsal of the N > 1 path; the driver runs the true multi-GPU
bench): two fresh child processes, both on cuda:0, gloo backend, one keyframe each, gradients written straight into the
all-reduce bucket (direct_grads) and reduced chunk by chunk while the backward's per-Gaussian stage is still running.
Checked: the reduced bucket equals the serial sum of the two keyframes' gradients, rank 0's parameter gradients ALIAS
the bucket (no pack copies), one-chunk and four-chunk reductions agree, and the frame-parameter all-gather."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads, random_w2c, setup_camera
rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
P, W, H = 20000, 320, 240
sc = make_scene(P, W, H, seed=0, regime="mapping")
names = ("means3D", "opacities", "scales", "rotations", "colors")
dc, da = make_upstream_grads(W, H, seed=1, channels=(0, 1, 5, 6))
dc, da = (dc * W * H).to(dev), (da * W * H).to(dev)

def camera(kf):
    cam = sc["cam"]
    if kf > 0:
        cam = setup_camera(W, H, cam.K, random_w2c(np.random.default_rng(1000 + kf), 3.0, 0.1) @ cam.w2c)
    return gs_render.settings_from_camera(cam, dev, use_sa=True)

def make(chunks):
    params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
    def render_fn(p, kf):
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        pkg = gs_render.render(camera(kf), p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                               rotations=p["rotations"])
        return (pkg["render_color"], pkg["allmap"]), (dc, da)
    return params, ba_shard.KeyframeShardedBA(params, render_fn, direct_grads=True, overlap_chunks=chunks)

res = {}
for chunks in (4, 1):
    params, ba = make(chunks)
    views = ba.step([0, 1])
    torch.cuda.synchronize()
    res[f"flat{chunks}"] = ba.bucket.flat.cpu().numpy()
    if chunks == 4:
        # direct_grads: the gradients autograd handed to the leaves live in the bucket itself
        res["alias"] = np.array([params[n].grad is not None and params[n].grad.data_ptr() == ba.bucket.views[n].data_ptr() for n in names])
        fr = ba.gather_frame_params(torch.arange(9, dtype=torch.float32, device=dev) + 100 * rank)
        res["frames"] = fr.cpu().numpy()
# autotune: every rank ends with the same choice, and a step after it still reduces correctly
params, bat = make(4)
times = bat.autotune([0, 1], candidates=(1, 2), reps=1)
pick = torch.tensor([bat.overlap_chunks], device=dev)
dist.all_reduce(pick, op=dist.ReduceOp.MAX)
res["tuned_same"] = np.array([int(pick.item()) == bat.overlap_chunks and set(times) == {1, 2} and bat.overlap_chunks in (1, 2)])
bat.step([0, 1])
torch.cuda.synchronize()
res["flat_tuned"] = bat.bucket.flat.cpu().numpy()
# serial reference on this rank: both keyframes, plain autograd, summed
params, ba1 = make(1)
tot = None
for kf in (0, 1):
    g = ba1.local_backward(kf)
    flat = torch.cat([g[n].reshape(-1) for n in ba_shard.BUCKET_FIELDS])
    tot = flat if tot is None else tot + flat
res["serial"] = tot.cpu().numpy()
if rank == 0:
    np.savez(out, **res)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_real_rasterizer_direct_grads_chunked_allreduce():
    with tempfile.TemporaryDirectory() as td:
        script, out = os.path.join(td, "child.py"), os.path.join(td, "out.npz")
        with open(script, "w") as f:
            f.write(CHILD)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(util.free_port()), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, script, ROOT, out], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        logs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)
        d = np.load(out)
    scale = np.abs(d["serial"]).max()
    assert scale > 0
    assert np.abs(d["flat4"] - d["serial"]).max() <= 1e-4 * scale  # float atomics: equal up to summation order
    assert np.abs(d["flat1"] - d["serial"]).max() <= 1e-4 * scale
    assert d["tuned_same"].all()
    assert np.abs(d["flat_tuned"] - d["serial"]).max() <= 1e-4 * scale
    assert d["alias"].all(), d["alias"]
    assert d["frames"].shape == (2, 9) and d["frames"][1, 0] == 100.0 and d["frames"][0, 8] == 8.0


# The N > 1 bench runs over RCCL (backend "nccl"), which cannot put two ranks on one GPU -- so the gloo rehearsal above never
# executes the RCCL-specific calls (communicator bound to a device, the coalesced asynchronous all-reduce of the chunked
# reduction, all_gather_into_tensor).  A ONE-rank RCCL communicator accepts every one of them: with ba_shard's collective
# threshold lowered to 1 the exact code path of a multi-GPU step runs here on the real library.
RCCL_CHILD = r'''
import os, sys, warnings
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
out = sys.argv[2]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # as bench.py does
ba_shard.MIN_COLLECTIVE_WORLD = 1
P, W, H = 20000, 320, 240
sc = make_scene(P, W, H, seed=0, regime="mapping")
names = ("means3D", "opacities", "scales", "rotations", "colors")
dc, da = make_upstream_grads(W, H, seed=1, channels=(0, 1, 5, 6))
dc, da = (dc * W * H).to(dev), (da * W * H).to(dev)
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)

def make(chunks):
    params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
    def render_fn(p, kf):
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        pkg = gs_render.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                               rotations=p["rotations"])
        return (pkg["render_color"], pkg["allmap"]), (dc, da)
    return params, ba_shard.KeyframeShardedBA(params, render_fn, direct_grads=True, overlap_chunks=chunks)

res = {}
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    for chunks in (4, 1):
        params, ba = make(chunks)
        ba.step([0])
        torch.cuda.synchronize()
        res[f"flat{chunks}"] = ba.bucket.flat.cpu().numpy()
        if chunks == 4:
            res["overlapped"] = np.array([ba._overlap_ok and ba.overlap_chunks == 4])
            res["frames"] = ba.gather_frame_params(torch.arange(9, dtype=torch.float32, device=dev)).cpu().numpy()
    params, bat = make(4)
    times = bat.autotune([0], candidates=(1, 2), reps=1)
    res["tuned"] = np.array([set(times) == {1, 2} and bat.overlap_chunks in (1, 2)])
res["fallback_warnings"] = np.array([sum("all-reduce unavailable" in str(w.message) for w in caught)])
ba_shard.MIN_COLLECTIVE_WORLD = 2
params, ba1 = make(1)
g = ba1.local_backward(0)
res["serial"] = torch.cat([g[n].reshape(-1) for n in ba_shard.BUCKET_FIELDS]).cpu().numpy()
np.savez(out, **res)
dist.barrier()
dist.destroy_process_group()
'''


def test_one_rank_rccl_runs_the_multi_gpu_code_path():
    with tempfile.TemporaryDirectory() as td:
        script, out = os.path.join(td, "child.py"), os.path.join(td, "out.npz")
        with open(script, "w") as f:
            f.write(RCCL_CHILD)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(util.free_port()), WORLD_SIZE="1", RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, script, ROOT, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout
        d = np.load(out)
    scale = np.abs(d["serial"]).max()
    assert scale > 0
    assert d["overlapped"].all() and d["fallback_warnings"][0] == 0  # the coalesced asynchronous form was accepted
    assert np.abs(d["flat4"] - d["serial"]).max() <= 1e-4 * scale
    assert np.abs(d["flat1"] - d["serial"]).max() <= 1e-4 * scale
    assert d["tuned"].all()
    assert d["frames"].shape == (1, 9) and d["frames"][0, 8] == 8.0
