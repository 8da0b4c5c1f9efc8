#!/usr/bin/env python3
"""bench.py -- fwd+bwd frames/s of the 2D-Gaussian-surfel rasterizer op on MI355X.

One "step" = one forward + backward of the operator (GaussianRasterizer surface) on one synthetic keyframe per
GPU, 640x480, 500k Gaussians, inputs resident in HBM; with --gpus N > 1 each rank renders its own keyframe of the
same replicated map and the [P,13] Gaussian-gradient bucket is all-reduced over RCCL (keyframe-sharded BA step,
gaus_slam_amd/ba_shard.py).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel: algorithmic bytes / hipEvent-measured launch duration vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (oracle/, C + OpenMP) timed on this box's host cores on the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gaus_slam_amd import _lib, ba_shard, render as gs_render  # noqa: E402
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads, random_w2c, setup_camera  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
STAGES = ["preprocess", "scan", "duplicate", "sort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "cull"]


def stage_bytes(P, R, HW):
    """Algorithmic bytes per launch of each stage (DESIGN.md 'Kernels'; SURVEY.md section 8(d) per-unit figures)."""
    return {
        "preprocess": 52 * P + 92 * P,         # SoA in; 80-B record + depth/radius/tiles out
        "scan": 8 * P,
        "duplicate": 20 * P + 12 * R,
        "sort": 24 * R,                        # one read + one write of the 12-B pairs (a 6-pass LSD sort moves 6x)
        "ranges": 8 * R,
        "cull": 60 * R,                        # id + 52 B of the record in, 4 B of sub-block bits out
        "blend_fwd": 84 * R + 68 * HW,         # 4-B id + 80-B record per instance; 40 B images + 28 B state per pixel
        "blend_bwd": 84 * R + 68 * HW + 72 * P,  # gather + per-pixel grads/state + accumulator write-back
        "preprocess_bwd": 152 * P + 80 * P,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=500000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-sa", action="store_true", help="use_sa=False (SLAM default is True)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--adam", nargs="?", const="fused", default=None, choices=["fused", "torch"],
                    help="also run a (lr=0) Adam step on the 13 floats per Gaussian inside the timed step -- BASELINE.md "
                    "section 5's BA-step definition; off for the fwd+bwd metric.  fused = gs2d_adam_step over the flat SoA "
                    "(one launch), torch = torch.optim.Adam(fused=True) over the five tensors")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a single GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--keyframes-per-gpu", type=int, default=1, help="keyframes each rank renders per step (default 1 = the "
                    "headline metric; >1 uses two HIP streams per rank, see ba_shard.KeyframeShardedBA)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: run the N ranks as a child torch.distributed.run job (nothing here has touched the
        # GPU yet) and hand back its exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    P, W, H = args.gaussians, args.width, args.height
    use_sa = not args.no_sa
    sc = make_scene(P, W, H, seed=0, regime="mapping")
    cam = sc["cam"]
    if rank > 0:  # every rank renders its own keyframe: a different small SE3 around the same map
        import numpy as np
        w2c = random_w2c(np.random.default_rng(1000 + rank), max_rot_deg=3.0, max_trans=0.1) @ cam.w2c
        cam = setup_camera(W, H, cam.K, w2c)
    params = {
        "means3D": sc["means3D"].to(dev).requires_grad_(True),
        "opacities": sc["opacities"].to(dev).requires_grad_(True),
        "scales": sc["scales"].to(dev).requires_grad_(True),
        "rotations": sc["rotations"].to(dev).requires_grad_(True),
        "colors": sc["colors"].to(dev).requires_grad_(True),
    }
    soa = None
    if args.adam == "fused":  # parameters live in ONE flat buffer in the all-reduce bucket layout (gaus_slam_amd/optim.py)
        from gaus_slam_amd import optim as gs_optim
        soa = gs_optim.GaussianSoA(params)
        params = dict(soa.leaves())
    dcolor, dallmap = make_upstream_grads(W, H, seed=1)
    dcolor, dallmap = dcolor.to(dev), dallmap.to(dev)
    settings = gs_render.settings_from_camera(cam, dev, use_sa=use_sa)
    last = {}

    def render_fn(p, _kf):
        means2D = torch.zeros_like(p["means3D"], requires_grad=True)
        pkg = gs_render.render(settings, p["means3D"], means2D, p["opacities"], colors_precomp=p["colors"],
                               scales=p["scales"], rotations=p["rotations"])
        last["radius"] = pkg["radius"]
        return (pkg["render_color"], pkg["allmap"]), (dcolor, dallmap)

    # gradients are produced directly in the all-reduce bucket whenever the bucket is consumed (N > 1, fused Adam)
    ba = ba_shard.KeyframeShardedBA(params, render_fn, direct_grads=(world > 1 or args.adam == "fused"))
    kpg = max(1, args.keyframes_per_gpu)
    keyframes = list(range(world * kpg))  # keyframe i goes to rank i % world
    opt = None
    # lr = 0: the full moment update runs, the scene (and so num_rendered) stays fixed (eps as scene/Gaussians.py:137)
    if args.adam == "torch":
        opt = torch.optim.Adam(list(params.values()), lr=0.0, eps=1e-15, fused=True)
    elif args.adam == "fused":
        opt = gs_optim.FusedGaussianAdam(soa, {})
    names = list(ba_shard.BUCKET_FIELDS)

    def one_step():
        g = ba.step(keyframes)
        if args.adam == "torch":
            for n in names:
                params[n].grad = g[n].reshape(params[n].shape)
            opt.step()
        elif args.adam == "fused":
            if world == 1:  # K=1 bypasses the bucket (bit-for-bit single-GPU path): pack it here
                ba.bucket.pack(g)
            opt.step(ba.bucket.flat)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = args.steps * world * kpg
    ms_per_step = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        L = _lib.lib()
        # ---- per-stage device time via hipEvents on the launch stream (separate, untimed leg)
        L.gs2d_stage_timing_enable(1)
        acc = [0.0] * len(STAGES)
        nrep = 10
        buf = (C.c_float * len(STAGES))()
        for _ in range(nrep):
            ba.local_backward(0)
            L.gs2d_stage_timing_read(buf)
            for i in range(len(STAGES)):
                acc[i] += max(buf[i], 0.0)
        L.gs2d_stage_timing_enable(0)
        stage_ms = {n: acc[i] / nrep for i, n in enumerate(STAGES)}
        visible = int((last["radius"] > 0).sum().item())
        from gaus_slam_amd import rasterizer
        with torch.no_grad():
            e = torch.empty(0, device=dev)
            R = rasterizer.rasterize_gaussians(settings.bg, params["means3D"], params["colors"], params["opacities"],
                                               params["scales"], params["rotations"], 1.0, e, settings.viewmatrix,
                                               settings.projmatrix, settings.tanfovx, settings.tanfovy, H, W, e, 0,
                                               settings.campos, use_sa, False, False)[0]
        sb = stage_bytes(P, R, H * W)
        dom = max(("blend_fwd", "blend_bwd", "sort", "preprocess", "preprocess_bwd"), key=lambda n: stage_ms[n])
        achieved = sb[dom] / (stage_ms[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        # measured streaming bandwidth of this box (1 GiB device-to-device copy), quoted beside the 8 TB/s spec peak
        x = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        for _ in range(2):
            y.copy_(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * x.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del x, y
        # secondary (honest) ceiling: the dominant kernel is VALU-issue bound; instruction counts come from the committed
        # rocprofv3 PMC summary, the 4-cycles-per-wave64-fp32-instruction peak from scripts/dev/valu_bench.hip
        valu = None
        vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if os.path.exists(vpath) and (P, W, H) == (500000, 640, 480):
            try:
                vi = json.load(open(vpath)).get(dom)
                peak_ginst = 1024 * 2.4 / 4.0  # G wave-instructions/s
                ach = vi["valu_wave_insts"] / (stage_ms[dom] * 1e-3) / 1e9
                valu = {"kernel": dom, "achieved_Gwaveinst_s": round(ach, 1), "peak_Gwaveinst_s": peak_ginst,
                        "frac": round(ach / peak_ginst, 3)}
            except Exception:
                valu = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "measured_copy_GBps": round(copy_gbs, 1), "valu_ceiling": valu,
                    "kernel_ms": round(stage_ms[dom], 4),
                    "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                    "frame_algorithmic_bytes": 492 * P + 196 * R + 136 * H * W,
                    "frame_frac_of_hbm_peak": round((492 * P + 196 * R + 136 * H * W) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = cpu_baseline_leg(sc, W, H, use_sa)
        result = {
            "metric": "fwd+bwd frames/sec @ 640x480, 500k Gaussians", "value": round(frames / elapsed, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {W}x{H} / {P} Gaussians (BASELINE.md config B), mapping regime, "
                                   f"use_sa={use_sa}, {kpg} keyframe{'s' if kpg > 1 else ''} per GPU" + (" on two HIP streams" if kpg > 1 else ""), "num_rendered": R, "visible": visible,
                       "step": "op forward+backward" + (" + all-reduce of the [P,13] grad bucket" if world > 1 else "")
                               + (f" + {args.adam} Adam (lr=0)" if args.adam else ""),
                       "parallelism": f"keyframe-sharded x{world}"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def cpu_baseline_leg(sc, W, H, use_sa):
    """The CPU oracle (C, OpenMP over tiles) on the same scene: 1 warm-up + 2 timed fwd+bwd frames."""
    from oracle import gs2d_oracle as orc
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    cam = sc["cam"]
    dc, da = make_upstream_grads(W, H, seed=1)
    dc, da = dc.numpy(), da.numpy()

    def frame():
        st = orc.forward(sc["means3D"].numpy(), sc["opacities"].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                         cam.campos.numpy(), W, H, cam.tanfovx, cam.tanfovy, scales=sc["scales"].numpy(),
                         rotations=sc["rotations"].numpy(), colors_precomp=sc["colors"].numpy(), use_sa=use_sa,
                         want_stability=False)
        orc.backward(st, dc, da)

    frame()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        frame()
    dt = (time.perf_counter() - t0) / reps
    # BASELINE.json configs[0]: 160x120 / 256 Gaussians forward through the pure-PyTorch CPU path (oracle/torch_ref.py)
    torch_a = None
    try:
        from oracle import torch_ref
        sa = make_scene(256, 160, 120, seed=0, regime="mapping")
        ca = sa["cam"]
        torch.set_num_threads(min(cores, 8))  # thousands of tiny ops: more threads only add synchronisation cost
        t1 = time.perf_counter()
        with torch.no_grad():
            torch_ref.render(sa["means3D"], sa["scales"], sa["rotations"], sa["opacities"], sa["colors"], ca.viewmatrix,
                             ca.projmatrix, 160, 120, use_sa=use_sa)
        torch_a = round(time.perf_counter() - t1, 4)
    except Exception as ex:  # the CPU reference is optional colour, never a reason to fail the bench
        torch_a = f"failed: {ex}"
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "pure_pytorch_cpu_160x120_256_fwd_s": torch_a, "pure_pytorch_threads": min(cores, 8),
            "sample": f"same workload ({W}x{H}, {sc['means3D'].shape[0]} Gaussians), oracle/gs2d_oracle.c fwd+bwd, "
                      f"OpenMP over tiles in the blend stages, 1 warm-up + {reps} timed frames"}


if __name__ == "__main__":
    main()
