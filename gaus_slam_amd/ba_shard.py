"""Keyframe-sharded bundle adjustment step: one process per GPU, one keyframe per rank, one all-reduce of the
Gaussian gradients over RCCL/xGMI.

The reference is strictly one keyframe per optimizer step on one GPU (slam/Backend.py:101-128,174-194); this
mini-batch formulation is new.  Every rank holds a full replica of the Gaussian SoA; rank r renders
keyframe r of the batch with the single-GPU op (fwd+bwd), the five parameter gradients are packed into ONE flat
fp32 bucket of 13 floats per Gaussian (xyz 3 | opacity 1 | scaling 2 | rotation 4 | rgb 3 -- the param groups of
scene/Gaussians.py:124-135) and summed with a single all-reduce.  World size 1 skips the collective, so K=1
reproduces the single-GPU path bit for bit.
"""
import contextlib
import os
from collections import OrderedDict

import torch
import torch.distributed as dist

# name -> floats per Gaussian, in bucket order
BUCKET_FIELDS = OrderedDict([("means3D", 3), ("opacities", 1), ("scales", 2), ("rotations", 4), ("colors", 3)])
BUCKET_FLOATS = sum(BUCKET_FIELDS.values())  # 13


# World sizes below this skip every collective (a one-rank job is the single-GPU path).  tests/test_gpu_multirank.py sets it
# to 1 to drive the RCCL calls below on the one GPU of a test box -- a one-rank RCCL communicator accepts them all.
MIN_COLLECTIVE_WORLD = 2


def _collective(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= MIN_COLLECTIVE_WORLD



def _rasterizer():
    from . import rasterizer  # lazy: ba_shard's bucket logic is importable (and tested) without the HIP library
    return rasterizer

class GradBucket:
    """Flat [13*P] fp32 buffer; each field is a contiguous [P,k] view (SoA segments, so packing a gradient is one
    contiguous copy and the collective is one large message: xGMI is per-link bound, fewer/larger is better)."""

    def __init__(self, P, device):
        self.P = P
        self.flat = torch.zeros(BUCKET_FLOATS * P, dtype=torch.float32, device=device)
        self._pending = []
        self.views = OrderedDict()
        o = 0
        for name, k in BUCKET_FIELDS.items():
            self.views[name] = self.flat[o:o + k * P].view(P, k)
            o += k * P

    def pack(self, grads):
        for name, v in self.views.items():
            g = grads.get(name)
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():  # already written in place by the rasterizer (direct_grads)
                v.copy_(g.reshape(v.shape))

    def all_reduce(self, group=None, average=False):
        if _collective(group):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        return self.flat

    def reduce_rows(self, g0, g1, group=None):
        """Start the all-reduce(sum) of Gaussians [g0, g1): the five field slices as ONE coalesced, asynchronous collective
        (a single grouped launch under RCCL).  Called while later rows are still being computed; finish with wait()."""
        if not _collective(group) or g1 <= g0:
            return
        if dist.get_backend(group) == "nccl" or not self.flat.is_cuda:
            # no `device` argument: the slices are recorded and issued as ONE allreduce_coalesced when the context closes
            # (c10d's fast path; with a device it would additionally wrap them in start/end-coalescing calls)
            with dist._coalescing_manager(group=group, async_ops=True) as cm:
                for v in self.views.values():
                    dist.all_reduce(v[g0:g1], op=dist.ReduceOp.SUM, group=group)
            self._pending.append(cm)
        else:  # gloo on device tensors (single-GPU rehearsal): no coalesced form, five asynchronous collectives
            for v in self.views.values():
                self._pending.append(dist.all_reduce(v[g0:g1], op=dist.ReduceOp.SUM, group=group, async_op=True))

    def wait(self, group=None, average=False):
        """Complete every reduce_rows() issued so far."""
        for cm in self._pending:
            cm.wait()
        self._pending = []
        if average and _collective(group):
            self.flat.div_(dist.get_world_size(group))
        return self.flat


MAX_BATCH_KEYFRAMES = 8  # GS2D_MAX_FRAMES
_ENGINE_THREADS = os.environ.get("GS2D_AUTOGRAD_ENGINE_THREADS", "0") == "1"  # dev A/B switch, see local_backward


def k_keyframe_schedule(K, reference_steps, reference_lrs):
    """The optimisation policy for steps that see K keyframes.  The reference takes ONE randomly
    drawn keyframe per optimizer step (slam/Backend.py:101-128, Adam with eps 1e-15 at scene/Gaussians.py:137); a sharded
    step takes K and the bucket holds the SUM of their gradients -- under Adam the update does not depend on the scale of
    the gradient, so sum and mean give the same trajectory (measured: 0.14853 vs 0.14852, profiles/kpolicy_probe_r03.txt)
    and no division kernel is spent.  To spend the same number of keyframe visits as the reference in 1/K of its steps:
        steps = ceil(reference_steps / K),  learning rates = reference x K   (linear scaling)
    which on the synthetic mapping problem of tests/test_gpu_round3.py ends at or below the reference loop's loss for K = 2
    and 4 (0.372 / 0.354 vs 0.407 of the start loss; sqrt scaling: 0.424; unscaled: 0.494).  Keeping the reference's step
    count and learning rates instead (K times the visits) is never worse per step (0.361 / 0.326 vs 0.407).
    Returns (steps, {name: lr})."""
    K = max(1, int(K))
    return -(-int(reference_steps) // K), {n: lr * K for n, lr in reference_lrs.items()}


def shard_keyframes(keyframes, rank, world_size):
    """Keyframe k of the batch goes to rank k % world_size (independent units, no data-path exchange)."""
    return [kf for i, kf in enumerate(keyframes) if i % world_size == rank]


class KeyframeShardedBA:
    """params: dict name -> leaf tensor (requires_grad) with the BUCKET_FIELDS names.
    render_loss_fn(params, keyframe) -> scalar loss OR (outputs, upstream_grads) pair for torch.autograd.backward.
    """

    def __init__(self, params, render_loss_fn, group=None, average=False, direct_grads=False, streams=1, overlap_chunks=1,
                 batch_fn=None):
        """direct_grads: let the rasterizer's backward write the parameter gradients straight into the bucket (no pack
        copies).  Safe in every case -- a gradient that did not land in the bucket (the op was not fed the leaf itself,
        e.g. activations in between) is packed by copy as before.
        overlap_chunks: with direct_grads and one keyframe per rank, the per-Gaussian stage of the backward runs in that
        many chunks of Gaussian indices and each chunk's gradients are all-reduced while the next chunk is computed
        (SURVEY.md section 8(e)); 1 (default) = one all-reduce of the whole bucket after the backward.  Every extra chunk
        costs ~40 us of a step (one more kernel of the per-Gaussian stage, a stream hand-over to and from RCCL's stream, five
        more collectives) and can hide at most that stage's 26 us, so chunking only pays on a slow interconnect: autotune()
        decides by measurement."""
        # batch_fn(params, [keyframes]) (optional): renders ALL the keyframes a rank holds in one batched operator call
        # (render.render_batch / GaussianRasterizerBatch: one blend grid over the tiles of all frames) and returns a scalar
        # loss or an (outputs, upstream_grads) pair like render_loss_fn; used whenever the rank holds 2..8 keyframes.  Same
        # gradients as keyframe-by-keyframe rendering (per-frame results are bit-identical, sums taken in the same order).
        self.batch_fn = batch_fn
        self.direct_grads = direct_grads
        self.overlap_chunks = max(1, int(overlap_chunks))
        self._overlap_ok = False  # set once a chunked reduction has completed
        # A rank that holds several keyframes of the batch renders them one after the other (streams=1).  streams=2 renders
        # them on two HIP streams (forwards first, then backwards): in round 1 a second frame's kernels filled the idle
        # tails of the first (1.34 vs 1.52 ms per pair); since the blend kernels keep the waves of a SIMD together by issue
        # priority there is no tail left to fill and two concurrent frames only fight for the SIMDs (round 2: 1.46 ms per
        # pair on two streams vs 1.25 ms sequentially, profiles/bench_r02_kpg2*.json).
        self.n_streams = max(1, int(streams))
        self._streams = None
        self._ones = {}
        self.params = params
        self.fn = render_loss_fn
        self.group = group
        self.average = average
        P = params["means3D"].shape[0]
        self.bucket = GradBucket(P, params["means3D"].device)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def local_backward(self, keyframe, sink=None, chunk_rows=None, on_chunk=None, fn=None):
        for p in self.params.values():
            p.grad = None
        res = (fn or self.fn)(self.params, keyframe)
        ctx = contextlib.nullcontext()
        if sink is not None:
            from . import rasterizer
            ctx = rasterizer.grad_sink(sink, chunk_rows=chunk_rows, on_chunk=on_chunk)
        # The autograd engine hands CUDA-device nodes to a worker thread and blocks on a future: two thread wake-ups per backward,
        # 100-180 us of host time per step on the GPU boxes (scripts/dev/host_overhead_mt.py: 329 -> 148 us per fwd+bwd step) --
        # more than the launches themselves.  One device, one graph: run the backward in the calling thread.
        with ctx, torch.autograd.set_multithreading_enabled(_ENGINE_THREADS):
            if isinstance(res, tuple):
                outs, ups = res
                torch.autograd.backward(list(outs), list(ups))
            else:
                # the seed gradient of a scalar loss: one cached 1.0 per (device, dtype) instead of autograd's ones_like fill
                # on the stream in every step
                key = (res.device, res.dtype)
                one = self._ones.get(key)
                if one is None or res.dim() != 0:
                    one = torch.ones_like(res)
                    if res.dim() == 0:
                        self._ones[key] = one
                res.backward(one)
        return {k: p.grad for k, p in self.params.items()}

    def gather_frame_params(self, local_params):
        """All-gather of the rank-local per-keyframe parameters after a step (pose quaternion 4 + translation 3 + exposure 2
        = 9 scalars in the reference, scene/Frame.py:84-92): [n] -> [world_size, n], row r = rank r's keyframe."""
        t = local_params.detach().reshape(-1).contiguous()
        if not _collective(self.group):
            return t.unsqueeze(0).clone()
        out = torch.empty(self.world_size * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.view(self.world_size, t.numel())

    def _multi_stream_grads(self, mine):
        """Per-keyframe gradient dicts, keyframes spread over the side streams; joined to the current stream on return."""
        dev = self.params["means3D"].device
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_streams)]
        cur = torch.cuda.current_stream(dev)
        names = list(self.params)
        leaves = [self.params[n] for n in names]
        pending = []
        for i, kf in enumerate(mine):
            st = self._streams[i % self.n_streams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                pending.append(self.fn(self.params, kf))
        out = []
        for i, res in enumerate(pending):
            st = self._streams[i % self.n_streams]
            sink = contextlib.nullcontext()
            if i == 0 and self.direct_grads:  # the first keyframe's gradients land in the bucket itself
                from . import rasterizer
                sink = rasterizer.grad_sink(self.bucket.views)
            with torch.cuda.stream(st), sink:
                if isinstance(res, tuple):
                    gs = torch.autograd.grad(list(res[0]), leaves, list(res[1]), allow_unused=True)
                else:
                    gs = torch.autograd.grad(res, leaves, allow_unused=True)
            for g in gs:
                if g is not None:
                    g.record_stream(cur)  # consumed on the current stream below
            out.append({n: g for n, g in zip(names, gs)})
        for st in self._streams:
            cur.wait_stream(st)
        return out

    def _step_overlapped(self, keyframe, P):
        """One keyframe on this rank, gradients written into the bucket, per-chunk all-reduce overlapped with the backward."""
        rows = -(-P // self.overlap_chunks)
        done = []

        def on_chunk(g0, g1):
            done.append((g0, g1))
            self.bucket.reduce_rows(g0, g1, self.group)
        g = self.local_backward(keyframe, self.bucket.views, chunk_rows=rows, on_chunk=on_chunk)
        stale = [n for n, v in self.bucket.views.items() if g.get(n) is None or g[n].data_ptr() != v.data_ptr()]
        if not done:  # the operator was not reached with the leaves themselves: nothing has been reduced yet
            self.bucket.pack(g)
            self.bucket.all_reduce(self.group, self.average)
            return self.bucket.views
        self.bucket.wait(self.group, self.average)
        self._overlap_ok = True
        for n in stale:  # a gradient that did not land in the bucket: copy it in and reduce that field on its own
            v = self.bucket.views[n]
            v.copy_(g[n].reshape(v.shape)) if g.get(n) is not None else v.zero_()
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                v.div_(self.world_size)
        return self.bucket.views

    def _probe_overlap(self):
        """The coalesced asynchronous collective is torch-internal API (dist._coalescing_manager).  Whether this torch build
        HAS it is checked first without any communication (a rank whose build lacks it must not leave its peers inside a
        collective it never joins); the ranks exchange that answer, and only if all have it is the coalesced form tried once
        on scratch rows and the outcome agreed on again (MAX all-reduce of a failure flag) before anyone changes its collective
        pattern.  What this cannot recover from: a rank-local failure INSIDE the coalesced collective itself (peers already
        in it) -- that hangs or aborts like any mismatched collective would; it covers API absence and failures that are the
        same on every rank."""
        have = 1 if hasattr(dist, "_coalescing_manager") else 0
        if _collective(self.group):
            hv = torch.tensor([1 - have], dtype=torch.int32, device=self.bucket.flat.device)
            dist.all_reduce(hv, op=dist.ReduceOp.MAX, group=self.group)
            have = 1 - int(hv.item())
        if not have:
            import warnings
            warnings.warn("overlapped chunk all-reduce unavailable (torch.distributed has no _coalescing_manager on some rank)")
            self.overlap_chunks = 1
            return
        failed = 0
        try:
            scratch = GradBucket(8, self.bucket.flat.device)
            scratch.reduce_rows(0, 8, self.group)
            scratch.wait(self.group)
        except Exception as ex:  # noqa: BLE001
            failed = 1
            import warnings
            warnings.warn(f"overlapped chunk all-reduce unavailable on rank {self.rank} ({type(ex).__name__}: {ex})")
        flag = torch.tensor([failed], dtype=torch.int32, device=self.bucket.flat.device)
        if _collective(self.group):
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            self.overlap_chunks = 1
        else:
            self._overlap_ok = True

    def autotune(self, keyframes, candidates=(1, 2, 4), reps=10, warmup=3, margin=0.02):
        """Pick overlap_chunks by measurement on the machine at hand: one whole-bucket all-reduce after the backward (1) against
        chunked reductions overlapped with the per-Gaussian stage.  Chunking hides at most the length of that stage (26 us at
        500k Gaussians) and pays for every extra chunk: one collective latency on the wire and ~50 us of host work in
        torch.distributed (scripts/dev/rccl_one_rank_overhead.py: on a one-rank communicator, where the collective itself is
        free, 2 / 4 chunks cost +43 / +100 us per step and the whole-bucket form nothing) -- so which one wins depends on the
        interconnect and the bucket size, and a chunked form has to beat the whole-bucket one by `margin` to be chosen.
        Every candidate runs `reps` timed steps after `warmup` untimed ones; ranks agree on the choice through a MAX
        all-reduce of their times.  Returns {candidate: ms per step}; a no-op (returns {}) without collectives."""
        if not _collective(self.group) or not self.direct_grads or not self.params["means3D"].is_cuda:
            return {}
        import time
        dev = self.params["means3D"].device
        times = {}
        for c in candidates:
            self.overlap_chunks = max(1, int(c))
            for _ in range(max(1, warmup)):
                self.step(keyframes)
            torch.cuda.synchronize(dev)
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(reps):
                self.step(keyframes)
            torch.cuda.synchronize(dev)
            t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            times[int(c)] = float(t.item())
        # identical on every rank (the times are the all-reduced maxima)
        best = min(times, key=lambda k: (times[k], k))
        if 1 in times and best != 1 and times[best] > (1.0 - margin) * times[1]:
            best = 1
        self.overlap_chunks = best
        return times

    def step(self, keyframes):
        """One BA step over a batch of keyframes (len == world_size in the bench; ragged batches allowed: ranks
        without a keyframe contribute zeros).  Returns the reduced bucket views (name -> [P,k])."""
        mine = shard_keyframes(keyframes, self.rank, self.world_size)
        if not _collective(self.group) and len(mine) == 1:
            # degenerate K=1 case == the single-GPU path, bit for bit: no bucket, no copies, no collective
            g = self.local_backward(mine[0], self.bucket.views if self.direct_grads else None)
            return OrderedDict((name, g[name].reshape(v.shape) if g.get(name) is not None else torch.zeros_like(v))
                               for name, v in self.bucket.views.items())
        P = self.bucket.P
        # Chunked, overlapped reduction: decided from GLOBAL facts only (every rank must issue the same collectives):
        # exactly one keyframe per rank, gradients written straight into the bucket.
        if self.direct_grads and self.overlap_chunks > 1 and len(keyframes) == self.world_size and self.params["means3D"].is_cuda:
            if not self._overlap_ok:
                self._probe_overlap()
            if self.overlap_chunks > 1:
                # after the probe every rank has agreed that the chunked form works: a failure from here on is re-raised
                # (the job ends non-zero) instead of letting one rank fall back alone while the others keep issuing
                # chunked collectives -- a mismatch that would deadlock
                return self._step_overlapped(mine[0], P)
        if not mine:
            self.bucket.flat.zero_()
        elif (self.batch_fn is not None and 1 < len(mine) <= MAX_BATCH_KEYFRAMES and self.params["means3D"].is_cuda
              and not _rasterizer().is_deterministic()):
            # one batched operator call for all of this rank's keyframes; frame 0's gradients land in the bucket itself
            # (the batched backward has no deterministic variant: in deterministic mode the keyframes go one by one below)
            self.bucket.pack(self.local_backward(mine, self.bucket.views if self.direct_grads else None, fn=self.batch_fn))
        elif len(mine) > 1 and self.n_streams > 1 and self.params["means3D"].is_cuda:
            per_kf = self._multi_stream_grads(mine)
            self.bucket.pack(per_kf[0])
            for g in per_kf[1:]:
                for name, v in self.bucket.views.items():
                    if g.get(name) is not None:
                        v.add_(g[name].reshape(v.shape))
        else:
            acc = None
            for kf in mine:
                g = self.local_backward(kf, self.bucket.views if (self.direct_grads and acc is None) else None)
                if acc is None:
                    self.bucket.pack(g)
                    acc = True
                else:
                    for name, v in self.bucket.views.items():
                        if g.get(name) is not None:
                            v.add_(g[name].reshape(v.shape))
        self.bucket.all_reduce(self.group, self.average)
        return self.bucket.views
