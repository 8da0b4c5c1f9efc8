#!/usr/bin/env python3
"""Instruction-mix table of the blend kernels from the compiler's ISA (hipcc -S with the product flags): VALU / SALU /
LDS / VMEM / waitcnt / nop per loop TRIP (one unrolled step of the software-pipelined trip loop), per staged CHUNK
(staging loop body) and per FLUSH pass (backward only), for the shipped source.  Output kept as profiles/isa_mix_rNN.txt.

Regions are located by anchors in the linear ISA: a backward trip step is the code between two consecutive queue
reads (`ds_read_u8`), the forward's trip loop (two unrolled trips) the smallest loop around its last three `v_exp_f32`; the staging chunk is the loop body around the cull-bit byte load
(`global_load_ubyte`); a flush pass is the loop body containing the gradient-record atomics (`global_atomic_add_f32`
after the last trip step).  usage: isa_mix.py [path/to/gs2d_blend.hip]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else os.path.join(ROOT, "gaus_slam_amd", "csrc", "gs2d_blend.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17"] + [a for a in sys.argv[1:] if a.startswith("-")]


def classify(op):
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log")):
        return "VALU-trans"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def mix(seg):
    c = collections.Counter(classify(o) for o in seg)
    c["VALU"] += c.pop("VALU-trans", 0) * 0  # keep transcendental count separately, also count them as VALU below
    return c


def fmt(name, seg):
    c = collections.Counter(classify(o) for o in seg)
    valu = c["VALU"] + c["VALU-trans"]
    return (f"  {name:34s} instr {len(seg):4d} | VALU {valu:4d} (transcendental {c['VALU-trans']:2d}) | SALU {c['SALU']:3d} | "
            f"branch {c['branch']:2d} | LDS {c['LDS']:2d} | VMEM {c['VMEM']:2d} | s_waitcnt {c['waitcnt']:2d} | s_nop {c['nop']:2d}")


def loop_body_around(ops, labels, idx):
    """Innermost [label ... backward branch] range containing instruction idx."""
    best = None
    for i, (op, arg) in enumerate(ops):
        if op.startswith(("s_cbranch", "s_branch")) and arg in labels and labels[arg] <= idx <= i:
            if best is None or (i - labels[arg]) < (best[1] - best[0]):
                best = (labels[arg], i)
    return best


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "blend.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, SRC],
                              stderr=subprocess.DEVNULL)
        src = open(out).read()
    print(f"# ISA instruction mix of {os.path.relpath(SRC, ROOT)} (hipcc {' '.join(FLAGS)})")
    for kname, title in (("blend_fwd_kernelILb1ELb0E", "blend_fwd_kernel<USE_SA=true, BATCH=false>"),
                         ("blend_bwd_kernelILb1ELb0ELb0ELb0ELi", "blend_bwd_kernel<USE_SA=true, DET=false, BATCH=false, POSE=false>"),
                         ("blend_bwd_kernelILb1ELb0ELb0ELb1ELi", "blend_bwd_kernel<USE_SA=true, DET=false, BATCH=false, POSE=true> (tracking)")):
        m = re.search(r"\n(_ZN\S*" + kname + r"\S*):[^\n]*\n(.*?)s_endpgm", src, re.S)
        body = m.group(2)
        ops, labels = [], {}
        for line in body.split("\n"):
            t = line.split(";")[0].strip()
            if not t:
                continue
            if t.endswith(":") and t.startswith(".LBB"):
                labels[t[:-1]] = len(ops)
                continue
            if re.match(r"^[a-z_0-9]+(\s|$)", t) and not t.startswith("."):
                parts = t.split()
                ops.append((parts[0], parts[-1] if len(parts) > 1 else ""))
        names = [o for o, _ in ops]
        meta = [e for e in src.split("\n  - .a") if re.search(r"\.name:\s+" + re.escape(m.group(1)) + r"\n", e)]
        info = ""
        if meta:
            f = lambda k: (re.search(r"\." + k + r":\s+(\d+)", meta[0]) or [None, "?"])[1]
            info = (f", {f('vgpr_count')} VGPRs, {f('vgpr_spill_count')} spilled, scratch {f('private_segment_fixed_size')} B, "
                    f"LDS {f('group_segment_fixed_size')} B")
        ng = re.search(r"Li(\d)E", m.group(1)[m.group(1).find(kname) + len(kname) - 2:])
        print(f"\n{title}{' [' + ng.group(1) + ' queues per wave]' if ng and 'bwd' in kname else ''}: {len(ops)} instructions in the kernel{info}")
        q = [i for i, o in enumerate(names) if o == "ds_read_u8"]
        # the body without normal-channel gradients comes last in the backward; the forward has one body.  Each body has
        # two prologue queue reads and one per unrolled step: step A = [read A, read B), step B = [read B, back edge]
        rA, rB = q[-2], q[-1]
        back = None
        for i in range(rB, len(ops)):
            op, arg = ops[i]
            if op.startswith(("s_cbranch", "s_branch")) and arg in labels and labels[arg] <= rA:
                back = i
                break
        if "bwd" in kname:  # one queue read (ds_read_u8) per step: the anchor of the backward's two unrolled steps
            print(fmt("trip step (between queue reads)", names[rA:rB]))
        else:
            # the forward's two unrolled steps share their queue reads' positions with the record reads, so the loop as a whole
            # is the unit: the smallest loop that contains the last three v_exp_f32 (two alpha exponentials + a confidence one)
            ex = [i for i, o in enumerate(names) if o.startswith("v_exp_f32")]
            body = None
            for i, (op, arg) in enumerate(ops):
                if op.startswith(("s_cbranch", "s_branch")) and arg in labels and labels[arg] < ex[-3] and ex[-1] < i:
                    if body is None or i - labels[arg] < body[1] - body[0]:
                        body = (labels[arg], i)
            print(fmt("trip loop body = TWO trips", names[body[0]:body[1] + 1]))
        hb = [i for i, o in enumerate(names) if o == "global_load_ubyte" and i < rA]
        stage = loop_body_around(ops, labels, hb[-1]) if hb else None
        if stage:
            print(fmt("staging loop body (64-instance chunk)", names[stage[0]:stage[1] + 1]))
        # the batch loop: innermost loop containing both the staging body and the trip steps
        batch = None
        for i, (op, arg) in enumerate(ops):
            if op.startswith(("s_cbranch", "s_branch")) and arg in labels and stage and labels[arg] <= stage[0] and i >= (back or rB):
                if batch is None or (i - labels[arg]) < (batch[1] - batch[0]):
                    batch = (labels[arg], i)
        if batch and stage and back:
            rest = names[batch[0]:stage[0]] + names[stage[1] + 1:rA] + names[back + 1:batch[1] + 1]
            if "bwd" not in kname:
                print(fmt("rest of a batch (queues, prologue)", rest))

if __name__ == "__main__":
    main()
