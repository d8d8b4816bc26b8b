#!/usr/bin/env python3
"""Instruction mix of the render kernel's sweep loop and of the rest of the kernel, priced with the measured issue
costs (profiles/r02_ubench_issue.txt, four waves per SIMD, nominal 2.4 GHz cycles per wave-instruction per SIMD):

    cheap   2.5   v_add/sub/mul_f32 (VOP2), and/or/xor/not, shifts, v_add/sub_u32, v_mov_b32
    trans   8.3   v_rcp/rsq/sqrt/log/exp_f32;  17 for v_rcp/rsq/sqrt_f64
    full    4.7   everything else: fma, min/max/med3, bfi/bitop3/perm, compares, cndmask, every packed f32 form, all of
                  fp64, conversions, readlane

usage: tools/valu_mix.py [ISA.s] [--pmc profiles/r03_pmc.json] [--entries N] [--out profiles/r03_valu_mix.json]
The ISA comes from tools/isa_meta.py (build/isa/<name>.s).  The sweep loop is found as the loop with the most
v_med3_i32 (the four-key insertion, unrolled over two entries); with --pmc the dynamic VALU count of a launch is split
into sweep (entries x instructions per entry) and rest, and the mix-weighted issue ceiling of the kernel is printed:
1024 SIMDs x 2.4 GHz / (mean cycles per instruction)."""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN3srh19k_render_binned_memILb0ELi1ELi0EEEvPU3AS4KNS_8FrameDevEPfS4_Pi"
CHEAP = re.compile(r"^v_(add_f32|sub_f32|subrev_f32|mul_f32|and_b32|or_b32|xor_b32|not_b32|ashrrev_i32|lshlrev_b32|lshrrev_b32|"
                   r"add_u32|sub_u32|subrev_u32|mov_b32|add_co_u32|addc_co_u32)(_e32)?$")
TRANS32 = re.compile(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos)_(f32|iflag_f32)")
TRANS64 = re.compile(r"^v_(rcp|rsq|sqrt)_f64")
COST = {"cheap": 2.5, "full": 4.7, "trans32": 8.3, "trans64": 17.0}


def klass(op):
    if TRANS64.match(op):
        return "trans64"
    if TRANS32.match(op):
        return "trans32"
    if CHEAP.match(op) and not op.endswith("_e64"):
        return "cheap"
    return "full"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("isa", nargs="?", default=os.path.join(ROOT, "build", "isa", "now.s"))
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--entries", type=float, default=None, help="(tile, primitive) entries swept per launch")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    text = open(args.isa).read()
    m = re.search(r"^" + re.escape(KERNEL) + r":(.*?)\.Lfunc_end", text, re.S | re.M)
    if not m:
        raise SystemExit(f"{KERNEL} not found in {args.isa}")
    lines = [l.strip() for l in m.group(1).splitlines()]
    labels = {}
    for i, l in enumerate(lines):
        lm = re.match(r"^(\.LBB\d+_\d+):", l)
        if lm:
            labels[lm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        b = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", l)
        if b and b.group(1) in labels and labels[b.group(1)] < i:
            loops.append((labels[b.group(1)], i))
    if not loops:
        raise SystemExit("no loop found")
    def med3(lo, hi):
        return sum("v_med3_i32" in x for x in lines[lo:hi])
    # the innermost loop with the most insertions; of the two instantiations (with / without the depth estimate: near > 0
    # or not) the larger body is the one BASELINE's frames run
    top = max(med3(*ab) for ab in loops)
    inner = [ab for ab in loops if med3(*ab) == top]
    inner = [ab for ab in inner if not any(o != ab and o[0] >= ab[0] and o[1] <= ab[1] and med3(*o) == top for o in inner)]
    best = max(inner, key=lambda ab: ab[1] - ab[0])
    lo, hi = best
    n_med3 = med3(lo, hi)
    per_unroll = 12                                          # 3 v_med3 per pixel x 4 pixels per entry (four keys)
    entries_in_body = max(1, round(n_med3 / per_unroll))

    def mix(seg):
        c = collections.Counter()
        s = collections.Counter()
        for l in seg:
            if not l or l.startswith((";", ".", "//")):
                continue
            op = l.split()[0]
            if op.startswith("v_"):
                c[klass(op)] += 1
            elif op.startswith(("s_load", "s_buffer_load")):
                s["smem"] += 1
            elif op.startswith("s_cbranch") or op == "s_branch":
                s["branch"] += 1
            elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
                s["salu"] += 1
        return c, s

    sweep_v, sweep_s = mix(lines[lo:hi + 1])
    rest_v, _ = mix(lines[:lo] + lines[hi + 1:])
    def mean_cost(c):
        n = sum(c.values())
        return sum(COST[k] * v for k, v in c.items()) / n if n else 0.0
    per_entry = {k: v / entries_in_body for k, v in sweep_v.items()}
    out = {"isa": os.path.relpath(args.isa, ROOT), "kernel": "k_render_binned<false, 1, SRH_PRIM_DISK>",
           "cost_cycles": COST, "cost_source": "profiles/r02_ubench_issue.txt (K = 4 waves per SIMD, wall)",
           "sweep_loop": {"entries_per_iteration": entries_in_body,
                          "valu_per_entry": {**{k: round(v, 2) for k, v in per_entry.items()}, "total": round(sum(per_entry.values()), 2)},
                          "scalar_side_per_entry": {k: round(v / entries_in_body, 2) for k, v in sweep_s.items()},
                          "mean_cycles_per_valu": round(mean_cost(sweep_v), 3),
                          "cycles_per_entry": round(mean_cost(sweep_v) * sum(per_entry.values()), 1)},
           "rest_of_kernel_static": {"valu": dict(rest_v), "mean_cycles_per_valu": round(mean_cost(rest_v), 3)}}
    if args.pmc:
        pmc = json.load(open(args.pmc))["kernels"]["k_render_binned"]
        total = float(pmc["SQ_INSTS_VALU"])
        entries = args.entries
        if entries is None:
            raise SystemExit("--pmc needs --entries (bench.py prints executed_pair_tests; entries = that / 256)")
        n_sweep = entries * sum(per_entry.values())
        n_rest = max(total - n_sweep, 0.0)
        cyc = (n_sweep * mean_cost(sweep_v) + n_rest * mean_cost(rest_v)) / total
        out["launch"] = {"valu_wave_instr": total, "entries_swept": entries, "sweep_share": round(n_sweep / total, 3),
                         "mean_cycles_per_valu": round(cyc, 3),
                         "peak_ginstr_s_mix_weighted": round(1024 * 2.4 / cyc, 1),
                         "peak_ginstr_s_all_4_cycle": 614.4,
                         "note": "the rest of the kernel is priced with its STATIC mix (every instruction once)"}
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
