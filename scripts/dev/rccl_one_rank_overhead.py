"""Dev tool: what the N > 1 code path costs on ONE rank (RCCL communicator of size 1, so the collective itself is ~free):
ms per step of the bypass (plain single-GPU step), whole-bucket all-reduce and chunked/overlapped reductions, and the host
time spent inside the collective calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P, W, H = 500000, 640, 480
sc = make_scene(P, W, H, seed=0, regime="mapping")
names = ("means3D", "opacities", "scales", "rotations", "colors")
dc, da = make_upstream_grads(W, H, seed=1, channels=(0, 1, 5, 6))
dc, da = dc.to(dev), da.to(dev)
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
m2 = torch.zeros_like(params["means3D"], requires_grad=True)


def render_fn(p, kf):
    pkg = gs_render.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                           rotations=p["rotations"])
    return (pkg["render_color"], pkg["allmap"]), (dc, da)


def run(label, min_world, chunks, steps=60):
    ba_shard.MIN_COLLECTIVE_WORLD = min_world
    ba = ba_shard.KeyframeShardedBA(params, render_fn, direct_grads=True, overlap_chunks=chunks)
    host = [0.0]
    orig_rr, orig_ar, orig_wait = ba.bucket.reduce_rows, ba.bucket.all_reduce, ba.bucket.wait

    def timed(fn):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            host[0] += time.perf_counter() - t
            return r
        return w
    ba.bucket.reduce_rows, ba.bucket.all_reduce, ba.bucket.wait = timed(orig_rr), timed(orig_ar), timed(orig_wait)
    for _ in range(10):
        ba.step([0])
    torch.cuda.synchronize()
    host[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        ba.step([0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:34s} {ms:.4f} ms/step   host time inside collective calls {host[0] / steps * 1e3:.4f} ms/step", flush=True)


run("single-GPU step (no collective)", 2, 1)
run("whole bucket, one all-reduce", 1, 1)
run("2 chunks, overlapped", 1, 2)
run("4 chunks, overlapped", 1, 4)
run("single-GPU step (no collective)", 2, 1)
dist.destroy_process_group()
