#!/usr/bin/env python3
"""Audit gfx950 ISA (hipcc -S output) for the slow form of VCC-reading VALU instructions.

Measured with scripts/dev/select_bench.hip on MI355X (profiles/select_bench_r02.txt): a VOP2 `v_cndmask_b32_e32 ..., vcc`
costs ~1.2-1.9 ns per wave-instruction when VCC was last written by a VALU compare, but ~9.8 ns (24 cycles, at any
occupancy) when VCC was last written by the scalar unit (s_mov/s_and/... vcc) or is stale; the VOP3 form with an SGPR
pair (or `_e64 ..., vcc`) is always fast.  This script walks each kernel linearly, tracks who wrote VCC last, and lists
the VCC-consuming e32 VALU instructions whose VCC came from SALU (linear scan: branches make it approximate).

usage: isa_vcc_audit.py file.s [kernel-substring]
"""
import re
import sys


def audit(path, want=None):
    src = open(path).read()
    kernels = re.split(r"\n(?=[_A-Za-z][\w$.]*:\s*;? *@?)", src)
    for k in kernels:
        name = k.split(":")[0].strip()
        if "\n" in name or not name.startswith("_Z"):
            continue
        if want and want not in name:
            continue
        last = None  # 'valu' | 'salu'
        slow, fast, total = [], 0, 0
        label = ""
        for ln, line in enumerate(k.split("\n")):
            t = line.strip()
            if t.startswith(".LBB"):
                label = t.split(":")[0]
                continue
            if not re.match(r"^[vs]_", t):
                continue
            t = t.split(";")[0].strip()
            op = t.split()[0]
            args = t[len(op):]
            dst = args.split(",")[0].strip() if args else ""
            reads_vcc = False
            if op.startswith("v_"):
                total += 1
                if op.startswith("v_cndmask") and op.endswith("_e32"):
                    reads_vcc = True
                elif op.startswith(("v_addc", "v_subb", "v_subbrev")) and "vcc" in args.split(",", 1)[-1]:
                    reads_vcc = True
                elif op.startswith("v_div_fmas"):
                    reads_vcc = True
                if reads_vcc:
                    if last == "salu" or last is None:
                        slow.append((label, ln, t))
                    else:
                        fast += 1
                # VALU writers of vcc: VOPC e32 compares (implicit vcc), or explicit vcc destination
                if (op.startswith("v_cmp") and (op.endswith("_e32") or dst == "vcc")) or dst == "vcc":
                    last = "valu"
                if re.match(r"v_(add|sub|subrev)_co_u32", op) and "vcc" in args:
                    last = "valu"
            else:
                if dst in ("vcc", "vcc_lo", "vcc_hi"):
                    last = "salu"
        print(f"{name[:70]}: VALU {total}, vcc-reading e32 fast {fast}, SLOW {len(slow)}")
        for lab, ln, t in slow:
            print(f"    {lab:10s} {t}")


if __name__ == "__main__":
    audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
