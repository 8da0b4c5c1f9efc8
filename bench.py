#!/usr/bin/env python3
"""bench.py -- fwd+bwd frames/s of the 2D-Gaussian-surfel rasterizer op on MI355X.

One "step" = one forward + backward of the operator (GaussianRasterizer surface) on one synthetic keyframe per
GPU, 640x480, 500k Gaussians, inputs resident in HBM; with --gpus N > 1 each rank renders its own keyframe of the
same replicated map and the [P,13] Gaussian-gradient bucket is all-reduced over RCCL (keyframe-sharded BA step,
gaus_slam_amd/ba_shard.py).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel: algorithmic bytes / hipEvent-measured launch duration vs 8 TB/s HBM, the HBM bytes
                  of the committed rocprofv3 PMC profile, and the vector-issue ceiling the kernel actually runs against
  cpu_baseline -- the CPU oracle (oracle/, C + OpenMP) timed on this box's host cores on the same workload, and the
                  pure-PyTorch CPU render (oracle/torch_batched.py) fwd+bwd at 640x480 / 50k Gaussians.

--workload selects the other configurations of BASELINE.md / BASELINE.json (same JSON shape, metric named after the
workload): replica (1200x680 / 600k), scannetpp (1168x876 / 2M), scannetpp_ref (876x584 / 2M, the reference config's
render size), b200k (640x480 / 200k), tracking (BASELINE.json configs[2]: one Frontend tracking iteration = fused
pose-gradient render + tracking loss + Adam on the pose) and mapping (render + mapping loss + backward + fused Adam).
"""
import argparse
import ctypes as C
import glob
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
# Vector-issue ceiling, measured (profiles/issue_bench_r02.txt, profiles/select_bench_r02.txt): with 5-8 waves per SIMD a
# SIMD retires one wave64 v_fma_f32 every 1.15-1.24 ns (2.4-2.9 cycles at the 2.0-2.4 GHz the chip holds under this load;
# the guide's "2 cycles" row), v_cmp / v_min / v_max / v_cndmask ~1.75-1.9 ns, v_exp / v_rcp ~3.4 ns.
# The ceiling below prices every instruction as an FMA (optimistic): 1024 SIMDs / 1.19 ns.
VALU_PEAK_GINST = 1024 / 1.19
# The guide's nominal rate (MI355X_MICROARCH.md constants table): one wave64 v_fma_f32 per 2 cycles per SIMD at 2.4 GHz.
VALU_NOMINAL_GINST = 1024 * 2.4 / 2
STAGES = ["preprocess", "scan", "duplicate", "sort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "cull"]
WORKLOADS = {  # name -> (W, H, P, regime)
    "op": (640, 480, 500000, "mapping"), "b200k": (640, 480, 200000, "mapping"), "replica": (1200, 680, 600000, "tracking"),
    "scannetpp": (1168, 876, 2000000, "mapping"), "scannetpp_ref": (876, 584, 2000000, "mapping"),
    "tracking": (640, 480, 500000, "tracking"), "mapping": (640, 480, 500000, "mapping"),
    # the same two SLAM iterations at Replica's native frame size (configs/data/replica.yaml:3-4; BASELINE configs[2] / [3] run at it)
    "tracking_replica": (1200, 680, 600000, "tracking"), "mapping_replica": (1200, 680, 600000, "mapping"),
}


def stage_bytes(P, R, HW):
    """Algorithmic bytes per launch of each stage (DESIGN.md 'Kernels'; SURVEY.md section 8(d) per-unit figures)."""
    return {
        "preprocess": 52 * P + 92 * P,         # SoA in; 80-B record + depth/radius/tiles out
        "scan": 8 * P,
        "duplicate": 20 * P + 12 * R,
        "sort": 24 * R,                        # tile binning: one read + one write of the 12-B pairs (hist + row scan + scatter); the
                                               # per-tile depth sort runs inside blend_fwd when the lists fit its LDS (round 3)
        "ranges": 8 * R,
        "cull": 0,                             # (runs inside blend_fwd since round 2: id + 52 B of the record in, 4 B out)
        "blend_fwd": 12 * R + 68 * R + 84 * R + 68 * HW,  # depth-sort phase (8-B pair in, 4-B id out) + cull phase (id + 52 B of the
                                               # record in, 8 + 4 B of bits out) + 4-B id + 80-B record per instance; 40 B images +
                                               # 28 B state per pixel
        "blend_bwd": 84 * R + 68 * HW + 72 * P,  # gather + per-pixel grads/state + accumulator write-back
        "preprocess_bwd": 152 * P + 80 * P,
    }


def latest_profile(stem):
    """Newest committed profiles/<stem>_rNN.json (falls back to the round-1 name)."""
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", stem + "_r[0-9][0-9].json")))
    if c:
        return c[-1]
    p = os.path.join(ROOT, "profiles", stem + ".json")
    return p if os.path.exists(p) else None


def main():
    global PREWARM_STEPS, PREWARM_SECONDS, _ONE_RANK_RCCL
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="op", choices=sorted(WORKLOADS))
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
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
                    "headline metric); rendered one after the other unless --streams 2")
    ap.add_argument("--loss-autograd", action="store_true", help="tracking / mapping workloads: the fused loss as an autograd node "
                    "(loss.backward(): three loss kernels per iteration) instead of the one-call loss + gradients (two)")
    ap.add_argument("--no-batch", action="store_true", help="--keyframes-per-gpu > 1: render a rank's keyframes one operator call "
                    "after the other instead of one batched call (gs2d_forward_batch / gs2d_backward_batch)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams a rank spreads its keyframes over (ba_shard.KeyframeShardedBA)")
    ap.add_argument("--prewarm-steps", type=int, default=PREWARM_STEPS, help="upper bound on the untimed steps before the warm-up that take "
                    "the GPU out of its idle clocks (about 1.6 s worth are run; 0 = none); the count is reported in config.prewarm_steps")
    ap.add_argument("--prewarm-seconds", type=float, default=None, help="target duration of the untimed pre-warm (default %.1f s)" % PREWARM_SECONDS)
    ap.add_argument("--tune-allreduce", action="store_true", help="N > 1: time 1 / 2 / 4 all-reduce chunks overlapped with the "
                    "backward's per-Gaussian stage before the run and use the fastest (default: one whole-bucket all-reduce)")
    ap.add_argument("--rccl-one-rank", action="store_true", help="rehearsal on one GPU: create a ONE-rank RCCL communicator and "
                    "run the N > 1 code path (bucket, chunked all-reduce, autotune, collective timing) through the real library; "
                    "not a performance number")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the cold-start leg (config.cold_value) and the "
                    "reference-binning leg (config.reference_binning_value)")
    ap.add_argument("--json-out", default=None, help="also write the JSON line to this file (profiles/...)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: run the N ranks as a child torch.distributed.run job (nothing here has touched the
        # GPU yet) and hand back its exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.rccl_one_rank:
        _ONE_RANK_RCCL = True
        from gaus_slam_amd import ba_shard as _bs
        _bs.MIN_COLLECTIVE_WORLD = 1
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    PREWARM_STEPS = max(0, args.prewarm_steps)
    if args.prewarm_seconds is not None:
        PREWARM_SECONDS = max(0.0, args.prewarm_seconds)
    # stdout carries exactly ONE line, the JSON: whatever libraries print on file descriptor 1 while the job runs (RCCL writes
    # a five-line version banner there when a communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    if _dist_on(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    W0, H0, P0, regime = WORKLOADS[args.workload]
    P, W, H = args.gaussians or P0, args.width or W0, args.height or H0
    use_sa = not args.no_sa
    if args.workload.startswith(("tracking", "mapping")):
        result = slam_iteration_workload(args, dev, P, W, H, use_sa, rank, world)
    else:
        result = op_workload(args, dev, P, W, H, regime, use_sa, rank, world)
    if rank == 0 and result is not None:
        # which kernels produced this number: the hash compiled into the loaded library and the hash of the tree's sources
        # (equal unless GS2D_LIB_PATH points at an experiment build); tests/test_host.py holds the kept artifacts to it
        from gaus_slam_amd import build as _gs_build
        result["build"] = {"source_hash": _lib.lib_source_hash(), "tree_source_hash": _gs_build.source_hash(),
                           "info": _lib.build_info()}
        line = json.dumps(result)
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())
        if args.json_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if _dist_on(world):
        dist.barrier()
        dist.destroy_process_group()
    return result


_ONE_RANK_RCCL = False


def _dist_on(world):
    """Collectives in play: N > 1, or the one-rank RCCL rehearsal (--rccl-one-rank)."""
    return world > 1 or _ONE_RANK_RCCL


PREWARM_STEPS = 3000    # upper bound
PREWARM_SECONDS = 1.6   # target duration
PREWARM_DONE = 0        # steps actually run (reported as config.prewarm_steps)


HOST_CPU_FRACTION = None


def timed(one_step, steps, warmup, world, dev, prewarm=True):
    def sync():
        if _dist_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed, before the W warm-up steps: bring the GPU out of its idle power state.  A bench process that has just
    # finished its setup times its first ~30 ms at ramping clocks (scripts/dev/fill_cost.py: the same loop, 0.554 ms per
    # step when it starts 10 steps after idle, 0.539 ms once the card has been busy for a few hundred ms); the metric is
    # the steady-state rate.  About PREWARM_SECONDS of steps, at most PREWARM_STEPS; the count is derived from the slowest
    # rank's time for steps 11-60, so that every rank issues the same collectives.
    global PREWARM_DONE
    if prewarm:
        PREWARM_DONE = 0
    if prewarm and PREWARM_STEPS > 0:
        n0 = min(60, PREWARM_STEPS)
        for _ in range(min(10, n0)):  # first launches: code objects load, the allocator grows
            one_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(n0 - min(10, n0)):
            one_step()
        sync()
        dt = torch.tensor([(time.perf_counter() - t0) / max(1, n0 - 10)], dtype=torch.float64, device=dev)
        if _dist_on(world):
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        n1 = int(max(0, min(PREWARM_STEPS - n0, PREWARM_SECONDS / max(float(dt.item()), 1e-6) - n0)))
        for _ in range(n1):
            one_step()
        sync()
        PREWARM_DONE = n0 + n1
    # (Tried in round 3: taking Python's cyclic garbage collector out of this region.  gc.disable() alone changes nothing --
    # 2182 vs 2185 frames/s over six runs each -- and a gc.collect() in front of the warm-up costs 8 %: the tens of milliseconds it
    # takes are an idle gap on the stream, the card drops its clocks and W = 5 warm-up steps do not bring them back.)
    for _ in range(warmup):
        one_step()
    sync()
    global HOST_CPU_FRACTION
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    c1 = time.process_time()  # (before the final synchronize, which spins in the HIP runtime)
    t1 = time.perf_counter()
    sync()
    elapsed = time.perf_counter() - t0
    # CPU time of this process (all its threads) per wall second while it issued the timed steps: ~1.0 = a core busy
    # throughout (a spinning wait), less = the host slept through part of every step (gs2d_api.hip, wait_total)
    HOST_CPU_FRACTION = round((c1 - c0) / max(t1 - t0, 1e-9), 3)
    if _dist_on(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def build_roofline(args, dev, stage_ms, P, R, HW, ms_per_step, pose_only=False):
    """The `roofline` object of the JSON line: dominant kernel (by the hipEvent stage times recorded on the launch stream),
    its algorithmic bytes against the HBM peak as the bench contract asks, HBM traffic and VALU instruction counts from the
    committed rocprofv3 PMC summary of the same command (profiles/pmc_traffic[_<workload>]_rNN.json, pmc_valu...)."""
    sb = stage_bytes(P, R, HW)
    if pose_only:  # tracking: the backward keeps 16 B per Gaussian (dense dT[2], dT[5], dT[8]) instead of the 72-B write-back
        sb["blend_bwd"] = 84 * R + 68 * HW + 16 * P
        sb["preprocess_bwd"] = 40 * P
    dom = max(("blend_fwd", "blend_bwd", "sort", "preprocess", "preprocess_bwd"), key=lambda n: stage_ms[n])
    achieved = sb[dom] / (stage_ms[dom] * 1e-3) / 1e9
    W0, H0, P0, _ = WORKLOADS[args.workload]
    default_shape = not (args.gaussians or args.width or args.height) and (P0 * HW > 0)
    suffix = "" if args.workload == "op" else "_" + args.workload
    traffic, traffic_source = None, None
    tpath = latest_profile("pmc_traffic" + suffix)
    if tpath and default_shape:
        try:
            traffic = json.load(open(tpath)).get(dom)
            traffic_source = (f"{os.path.relpath(tpath, ROOT)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                              "committed with the kernels; not measured in this run")
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
    # The ceiling the dominant kernel actually runs against: vector-instruction issue.  Instruction counts come from the
    # committed rocprofv3 PMC profile of this command, the peak from the kept microbenchmark output (see VALU_PEAK_GINST).
    valu = None
    vpath = latest_profile("pmc_valu" + suffix)
    if vpath and default_shape:
        try:
            vi = json.load(open(vpath)).get(dom)
            ach = vi["valu_wave_insts"] / (stage_ms[dom] * 1e-3) / 1e9
            valu = {"kernel": dom, "achieved_Gwaveinst_s": round(ach, 1), "peak_Gwaveinst_s": round(VALU_PEAK_GINST, 1),
                    "frac": round(ach / VALU_PEAK_GINST, 3), "nominal_peak_Gwaveinst_s": round(VALU_NOMINAL_GINST, 1),
                    "frac_nominal": round(ach / VALU_NOMINAL_GINST, 3), "valu_wave_insts": vi["valu_wave_insts"],
                    "salu_wave_insts": vi.get("salu_wave_insts"),
                    "source": f"{os.path.relpath(vpath, ROOT)} (SQ_INSTS_VALU of the committed PMC profile); peak from "
                              "profiles/issue_bench_r02.txt, profiles/select_bench_r02.txt"}
        except Exception:
            valu = None
    closest = "valu-issue" if valu and valu["frac"] > achieved / HBM_PEAK_GBS else "hbm"
    frame_bytes = sum(sb[k] for k in ("preprocess", "scan", "duplicate", "sort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd")) if pose_only \
        else 492 * P + 196 * R + 136 * HW
    return {"bound": "valu" if closest == "valu-issue" else "hbm",
            "bound_note": "achieved / peak / frac below price the kernel against the HBM peak, as the bench contract asks for every "
                          "kernel; the ceiling it actually runs against is in valu_ceiling (frac = against the measured issue "
                          "rate, frac_nominal = against the guide's 2-cycle rate)",
            "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
            "closest_ceiling": closest,
            "measured_copy_GBps": round(copy_gbs, 1), "valu_ceiling": valu,
            "kernel_ms": round(stage_ms[dom], 4),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "frame_algorithmic_bytes": frame_bytes,
            "frame_frac_of_hbm_peak": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}


def op_workload(args, dev, P, W, H, regime, use_sa, rank, world):
    sc = make_scene(P, W, H, seed=0, regime=regime)
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
    kf_settings = {}

    def render_fn(p, _kf):
        # gradient carrier only: neither the reference's extension (its Python wrapper does not even pass it down,
        # RAST/gaus_2dgs_rasterization/__init__.py:59-80) nor this library reads its values, so no fill kernel is spent on it
        means2D = torch.empty_like(p["means3D"]).requires_grad_(True)
        pkg = gs_render.render(kf_settings.get(_kf) or settings, p["means3D"], means2D, p["opacities"], colors_precomp=p["colors"],
                               scales=p["scales"], rotations=p["rotations"])
        last["radius"] = pkg["radius"]
        return (pkg["render_color"], pkg["allmap"]), (dcolor, dallmap)

    kpg = max(1, args.keyframes_per_gpu)
    # several keyframes per rank: every rank renders ITS keyframes in one batched operator call (one blend grid over the
    # tiles of all its frames).  Keyframe i of the step goes to rank i % world; each gets its own camera.
    import numpy as np
    for i in range(world * kpg):
        if i < world:
            kf_settings[i] = settings if i == rank else None
        elif i % world == rank:
            w2c_i = random_w2c(np.random.default_rng(2000 + i), max_rot_deg=3.0, max_trans=0.1) @ sc["cam"].w2c
            kf_settings[i] = gs_render.settings_from_camera(setup_camera(W, H, sc["cam"].K, w2c_i), dev, use_sa=use_sa)

    def render_batch_fn(p, kfs):
        means2D = torch.empty_like(p["means3D"]).requires_grad_(True)
        pkgs = gs_render.render_batch([kf_settings[k] for k in kfs], p["means3D"], means2D, p["opacities"],
                                      colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
        last["radius"] = pkgs[0]["radius"]
        outs, ups = [], []
        for pk in pkgs:
            outs += [pk["render_color"], pk["allmap"]]
            ups += [dcolor, dallmap]
        return outs, ups

    # gradients are produced directly in the all-reduce bucket whenever the bucket is consumed (N > 1, fused Adam)
    ba = ba_shard.KeyframeShardedBA(params, render_fn, direct_grads=(_dist_on(world) or args.adam == "fused"), streams=args.streams,
                                    batch_fn=None if (args.no_batch or args.streams > 1) else render_batch_fn)
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

    # N > 1: one all-reduce of the whole bucket per step.  --tune-allreduce additionally times the chunked / overlapped forms
    # in an untimed setup phase and runs with the fastest; off by default because on one rank every extra chunk measured
    # +40 us against at most 26 us it can hide (profiles/rccl_one_rank_overhead_r02.txt), and the plain dist.all_reduce is
    # the form least likely to meet a surprise on a node this build never saw.
    tuned, tune_error = {}, None
    if _dist_on(world) and kpg == 1 and not args.adam and args.tune_allreduce:
        # No rank-local fallback here: whether the chunked collectives work at all is settled across ranks by
        # KeyframeShardedBA._probe_overlap (a MAX all-reduce of a failure flag); anything that still throws ends the job
        # non-zero rather than leaving one rank on a different collective pattern than the others.
        tuned = ba.autotune(keyframes)
    # The driver's protocol as stated (W warm-up steps right after setup, then K timed steps) -- what `--prewarm-steps 0`
    # measures -- runs FIRST, while the card is still in the clocks a fresh process finds it in; it is reported as
    # config.cold_value.  The headline leg then adds the untimed clock pre-warm (see timed()).
    cold_elapsed = None
    if PREWARM_STEPS > 0 and not args.no_extra_legs:
        cold_elapsed = timed(one_step, args.steps, args.warmup, world, dev, prewarm=False)
    elapsed = timed(one_step, args.steps, args.warmup, world, dev)
    host_cpu = HOST_CPU_FRACTION
    # The same steps in the contract's binning mode (gs2d_set_reference_binning(1): the reference's 3-sigma tile
    # rectangles, num_rendered and the sorted lists bit-identical to the reference's), on the now warm card.
    ref_elapsed = None
    if not args.no_extra_legs:
        from gaus_slam_amd import rasterizer as _rz
        _rz.set_reference_binning(True)
        try:
            ref_elapsed = timed(one_step, args.steps, args.warmup, world, dev, prewarm=False)
        finally:
            _rz.set_reference_binning(False)
    frames = args.steps * world * kpg
    ms_per_step = elapsed / args.steps * 1e3
    allreduce_ms = None
    if _dist_on(world):  # the collective alone, on the same bucket (reported separately)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            ba.bucket.all_reduce()
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t0) / 10 * 1e3
    if rank != 0:
        return None

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
        count = lambda: rasterizer.rasterize_gaussians(settings.bg, params["means3D"], params["colors"], params["opacities"],
                                                       params["scales"], params["rotations"], 1.0, e, settings.viewmatrix,
                                                       settings.projmatrix, settings.tanfovx, settings.tanfovy, H, W, e, 0,
                                                       settings.campos, use_sa, False, False)[0]
        R = count()  # instances the timed path handles (footprint binning, the library default)
        rasterizer.set_reference_binning(True)
        try:
            R_ref = count()  # instances of the reference's 3-sigma rectangles, for comparison only
        finally:
            rasterizer.set_reference_binning(False)
    roofline = build_roofline(args, dev, stage_ms, P, R, H * W, ms_per_step)
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_leg(sc, W, H, use_sa)
    return {
        "metric": f"fwd+bwd frames/sec @ {W}x{H}, {P // 1000}k Gaussians", "value": round(frames / elapsed, 3),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM_DONE,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic {W}x{H} / {P} Gaussians ({args.workload}; BASELINE.md section 2), {regime} regime, "
                               f"use_sa={use_sa}, {kpg} keyframe{'s' if kpg > 1 else ''} per GPU" + (f" on {args.streams} HIP streams" if kpg > 1 and args.streams > 1 else "")
                               + (" in one batched operator call" if kpg > 1 and ba.batch_fn is not None else ""), "num_rendered": R, "num_rendered_reference_rects": R_ref, "visible": visible,
                   "step": "op forward+backward" + (" + all-reduce of the [P,13] grad bucket" if _dist_on(world) else "")
                           + (f" + {args.adam} Adam (lr=0)" if args.adam else ""),
                   "parallelism": f"keyframe-sharded x{world}", "prewarm_steps": PREWARM_DONE,
                   "value_is": "steady-clock rate: prewarm_steps untimed steps take the card out of its idle clocks before the "
                               "warm-up (DESIGN.md section 5); cold_value is the same K steps after only the W warm-up steps "
                               "(what --prewarm-steps 0 measures)",
                   "cold_value": None if cold_elapsed is None else round(frames / cold_elapsed, 3),
                   "cold_ms_per_step": None if cold_elapsed is None else round(cold_elapsed / args.steps * 1e3, 4),
                   "reference_binning_value": None if ref_elapsed is None else round(frames / ref_elapsed, 3),
                   "reference_binning_ms_per_step": None if ref_elapsed is None else round(ref_elapsed / args.steps * 1e3, 4),
                   "binning_note": "value: library default (footprint binning, num_rendered instances); reference_binning_value: "
                                   "gs2d_set_reference_binning(1), the reference's rectangles (num_rendered_reference_rects "
                                   "instances, lists bit-identical to the reference's)", "allreduce_ms": None if allreduce_ms is None else round(allreduce_ms, 4),
                   "allreduce_chunks": ba.overlap_chunks if _dist_on(world) else None,
                   "allreduce_chunks_tuning_ms": {str(k): round(v, 4) for k, v in tuned.items()} or None,
                   "allreduce_tuning_error": tune_error,
                   "host_cpu_fraction": host_cpu,
                   "launch_ahead": bool(_lib.lib().gs2d_get_launch_ahead())},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }


def slam_iteration_workload(args, dev, P, W, H, use_sa, rank, world):
    """BASELINE.json configs[2] (tracking) / the backend mapping step: whole SLAM iterations around the operator, in the
    fused formulation this package offers (gaus_slam_amd/tracking.py, loss.py, optim.py)."""
    import numpy as np
    from gaus_slam_amd import loss as gl, optim as gs_optim, tracking
    if _dist_on(world):
        raise SystemExit("the tracking / mapping workloads are single-GPU iterations")
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    g = torch.Generator().manual_seed(0)
    gt_color = torch.rand(H, W, 3, generator=g).to(dev)
    gt_depth = (0.5 + 5 * torch.rand(H, W, 1, generator=g)).to(dev)
    is_tracking = args.workload.startswith("tracking")
    if is_tracking:
        sc = make_scene(P, W, H, seed=0, regime="tracking")
        settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=use_sa)
        p = {k: sc[k].to(dev) for k in names}
        w2c = random_w2c(np.random.default_rng(1), 2.0, 0.05).to(dev).requires_grad_(True)
        # lr = 0: the pose (and so the workload) stays fixed, the update runs; fused=True: one kernel instead of torch's
        # seven foreach launches for this single 4x4 tensor (33 us of a 0.66-ms iteration)
        opt = torch.optim.Adam([w2c], lr=0.0, fused=True)
        one = torch.ones((), dtype=torch.float32, device=dev)  # seed of loss.backward(): not re-filled in every iteration

        def one_step():
            with torch.autograd.set_multithreading_enabled(False):  # backward in the calling thread (ba_shard.local_backward)
                _tracking_step()

        def _tracking_step():
            opt.zero_grad(set_to_none=True)
            pkg = tracking.render_tracking(settings, w2c, p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
            if args.loss_autograd:
                gl.tracking_loss(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 0.5, 1.0).backward(one)
            else:  # loss value + gradients in one call, the rasterizer's backward seeded directly
                _loss, g_c, g_a = gl.tracking_loss_and_grads(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 0.5, 1.0)
                torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [g_c, g_a])
            opt.step()
        step_desc = "pose transform fused into the preprocess + render + fused tracking loss (value + gradients in one call) + pose-only backward + Adam on the pose"
        metric = f"tracking iterations/sec @ {W}x{H}, {P // 1000}k Gaussians"
    else:
        sc = make_scene(P, W, H, seed=0, regime="mapping")
        settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=use_sa)
        soa = gs_optim.GaussianSoA({k: sc[k].to(dev) for k in names})
        leaves = dict(soa.leaves())
        fopt = gs_optim.FusedGaussianAdam(soa, {})

        def rasterize(q):
            m2 = torch.empty_like(q["means3D"]).requires_grad_(True)  # gradient carrier, values never read
            return gs_render.render(settings, q["means3D"], m2, q["opacities"], colors_precomp=q["colors"], scales=q["scales"],
                                    rotations=q["rotations"])

        def loss_fn(q, _kf):
            pk = rasterize(q)
            if args.loss_autograd:
                return gl.mapping_loss(pk["render_color"], pk["allmap"], gt_color, gt_depth, 0.5, 1.0, 0.1)
            _loss, g_c, g_a = gl.mapping_loss_and_grads(pk["render_color"], pk["allmap"], gt_color, gt_depth, 0.5, 1.0, 0.1)
            return (pk["render_color"], pk["allmap"]), (g_c, g_a)
        ba = ba_shard.KeyframeShardedBA(leaves, loss_fn, direct_grads=True)

        def one_step():
            ba.step([0])
            fopt.step(ba.bucket.flat, leaves)
        step_desc = "render + fused mapping loss (value + gradients in one call) + backward (gradients written into the bucket) + fused Adam over the SoA"
        metric = f"mapping iterations/sec @ {W}x{H}, {P // 1000}k Gaussians"
    elapsed = timed(one_step, args.steps, args.warmup, world, dev)
    host_cpu = HOST_CPU_FRACTION
    # per-stage device time of the operator's kernels inside the iteration (hipEvents on the launch stream; separate, untimed leg)
    L = _lib.lib()
    L.gs2d_stage_timing_enable(1)
    acc = [0.0] * len(STAGES)
    buf = (C.c_float * len(STAGES))()
    for _ in range(10):
        one_step()
        L.gs2d_stage_timing_read(buf)
        for i in range(len(STAGES)):
            acc[i] += max(buf[i], 0.0)
    L.gs2d_stage_timing_enable(0)
    stage_ms = {n: round(acc[i] / 10, 4) for i, n in enumerate(STAGES)}
    from gaus_slam_amd import rasterizer
    with torch.no_grad():  # instances the iteration's render handles
        e = torch.empty(0, device=dev)
        q = p if is_tracking else leaves
        kw = dict(pose_Rt=w2c[:3, :4].detach().contiguous()) if is_tracking else {}
        R = rasterizer.rasterize_gaussians(settings.bg, q["means3D"], q["colors"], q["opacities"], q["scales"], q["rotations"], 1.0, e,
                                           settings.viewmatrix, settings.projmatrix, settings.tanfovx, settings.tanfovy, H, W, e, 0,
                                           settings.campos, use_sa, False, False, **kw)[0]
    roofline = build_roofline(args, dev, stage_ms, sc["means3D"].shape[0], R, H * W, elapsed / args.steps * 1e3,
                              pose_only=is_tracking)
    return {"metric": metric, "value": round(args.steps / elapsed, 3), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {W}x{H} / {P} Gaussians, {args.workload} iteration (BASELINE.json configs[2] loop shape), "
                                   f"use_sa={use_sa}", "step": step_desc, "parallelism": "single GPU", "prewarm_steps": PREWARM_DONE,
                       "num_rendered": R, "stage_ms": stage_ms, "host_cpu_fraction": host_cpu},
            "roofline": roofline, "cpu_baseline": None}


def cpu_baseline_leg(sc, W, H, use_sa):
    """(1) The CPU oracle (C, OpenMP over tiles) on the same scene: 1 warm-up + 2 timed fwd+bwd frames -> `value`.
    (2) The pure-PyTorch CPU render (oracle/torch_batched.py, all host threads): BASELINE.md section 4's cases, bounded
    so that the default run stays within minutes."""
    from oracle import gs2d_oracle as orc
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    cam = sc["cam"]
    dc, da = make_upstream_grads(W, H, seed=1)
    dcn, dan = dc.numpy(), da.numpy()

    def frame():
        st = orc.forward(sc["means3D"].numpy(), sc["opacities"].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                         cam.campos.numpy(), W, H, cam.tanfovx, cam.tanfovy, scales=sc["scales"].numpy(),
                         rotations=sc["rotations"].numpy(), colors_precomp=sc["colors"].numpy(), use_sa=use_sa,
                         want_stability=False)
        orc.backward(st, dcn, dan)

    frame()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        frame()
    dt = (time.perf_counter() - t0) / reps
    torch_leg = pure_pytorch_leg(W, H, use_sa, cores)
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"same workload ({W}x{H}, {sc['means3D'].shape[0]} Gaussians), oracle/gs2d_oracle.c fwd+bwd, "
                      f"OpenMP over tiles in the blend stages, 1 warm-up + {reps} timed frames",
            "pure_pytorch": torch_leg}


def pure_pytorch_leg(W, H, use_sa, cores):
    """oracle/torch_batched.py on the host cores: BASELINE.json configs[0] (160x120 / 256, forward and fwd+bwd), 640x480 /
    50k fwd+bwd (autograd), and 640x480 / 500k forward on a bounded number of list positions scaled to the full frame."""
    threads = min(cores, 64)  # [tiles, 256] elementwise ops: more threads than that only add synchronisation cost
    out = {"threads": threads, "renderer": "oracle/torch_batched.py (all tiles in lock-step, torch.autograd backward)"}
    try:
        from oracle import torch_batched
        torch.set_num_threads(threads)

        def run(P, w, h, grad, max_steps=None):
            s = make_scene(P, w, h, seed=0, regime="mapping")
            c = s["cam"]
            leaves = {k: (s[k].clone().requires_grad_(True) if grad else s[k]) for k in ("means3D", "scales", "rotations", "opacities", "colors")}
            dc, da = make_upstream_grads(w, h, seed=1)
            t0 = time.perf_counter()
            with torch.enable_grad() if grad else torch.no_grad():
                r = torch_batched.render(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"],
                                         leaves["colors"], c.viewmatrix, c.projmatrix, w, h, use_sa=use_sa, max_steps=max_steps)
                t1 = time.perf_counter()
                if grad:
                    ((r["color"] * dc).sum() + (r["allmap"] * da).sum()).backward()
            return t1 - t0, time.perf_counter() - t0, r["steps"], r["steps_run"]

        run(256, 160, 120, False)  # warm-up (thread pool, allocator)
        f, _, _, _ = run(256, 160, 120, False)
        _, fb, _, _ = run(256, 160, 120, True)
        out["160x120_256_fwd_s"], out["160x120_256_fwd_bwd_s"] = round(f, 4), round(fb, 4)
        f, fb, steps, _ = run(50000, 640, 480, True)
        out["640x480_50k_fwd_bwd_s"], out["640x480_50k_fwd_s_with_autograd_tape"], out["640x480_50k_list_steps"] = round(fb, 3), round(f, 3), steps
        out["640x480_50k_fwd_bwd_frames_per_s"] = round(1.0 / fb, 4)
        f, _, steps, ran = run(500000, 640, 480, False, max_steps=150)
        out["640x480_500k_fwd_s_scaled"] = round(f * steps / max(ran, 1), 2)
        out["640x480_500k_note"] = (f"forward only, {ran} of {steps} list positions timed ({round(f, 2)} s) and scaled; fwd+bwd by autograd "
                                    "would need ~10x the 50k run's 14 GB tape")
    except Exception as ex:  # the CPU reference is context, never a reason to fail the bench
        out["error"] = f"{type(ex).__name__}: {ex}"
    return out


if __name__ == "__main__":
    main()
