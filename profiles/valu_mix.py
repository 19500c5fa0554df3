#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop -> what the VALU pipes can deliver for THAT mix.

`valu_pipe_busy_frac` ~ 1.0 (profiles/*_valu_utilization.json) says the VALU pipes never idle; it does not say the kernel runs at the
guide's peak ISSUE rate (one wave64 VALU instruction per SIMD every 2 cycles): the blend kernels average ~4 cycles per instruction because
of what they issue -- SGPR-operand forms, transcendentals, v_permlane swaps, compares and selects (profiles/r2_valu_model.txt: measured ns
per instruction form at 8 waves per SIMD).  This tool disassembles the kernel for gfx950 (hipcc -S, no GPU needed), takes the body of its
largest loop, classes every VALU instruction by form and prices the mix with the measured costs:

    mix_ns_per_instruction = sum over classes (share x measured ns per instruction)
    mix_floor_frac         = wave_instructions x mix_ns_per_instruction / SIMDs / kernel time      (bench.py, from the PMC count)

    python profiles/valu_mix.py > profiles/<tag>_valu_mix.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

from build_id import csrc_sha16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ns per wave64 instruction per SIMD at 8 waves per SIMD (profiles/r2_valu_model.txt, tools/microbench/valu_model.hip)
COST_NS = {"vgpr_fma": 1.31, "vgpr_other": 1.50, "sgpr_operand": 1.90, "packed": 1.90, "transcendental": 3.80, "permlane_swap": 3.67,
           "dpp": 1.92, "cmp_select": 1.95}
# (kernel name as the PMC summaries have it -> mangled prefix, what is histogrammed: the blend backward is one loop over the list entries;
#  the fused sort + forward blend is several -- bucket sort, compaction, the blend walk -- and is priced as a whole)
KERNELS = {"blend.hip": {"blend_backward_kernel": ("_ZN3hgs21blend_backward_kernelILi4EEEv", "largest loop")},
           "binning.hip": {"tile_sort_small_kernel": ("_ZN3hgs22tile_sort_small_kernelILb1ELb0EEEv", "whole kernel")}}
FLAGS = {"binning.hip": ["-mllvm", "-disable-machine-sink"]}


def disassemble(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-fno-slp-vectorize", "-S", "--cuda-device-only"] + FLAGS.get(src, []) + [os.path.join(ROOT, "ml-hugs_amd", "csrc", src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read().splitlines()


def function_body(lines, prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and re.match(r"\S+:\s*(;.*)?$", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
    return lines[start + 1:end + 1]


def classify(op, args):
    if not op.startswith("v_"):
        return None
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith("v_permlane"):
        return "permlane_swap"
    if "dpp" in op or re.search(r"quad_perm|row_(shl|shr|ror|mirror|half_mirror|bcast|share)|wave_(shl|shr|rol|ror)", args):
        return "dpp"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "cmp_select"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "sgpr_operand"
    srcs = args.split(",")[1:] if "," in args else []
    if any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp)", a) for a in srcs):
        return "sgpr_operand"
    return "vgpr_fma" if op.startswith(("v_fma", "v_fmac", "v_mad")) else "vgpr_other"


def hot_loop(body):
    """(first, last) line index of the largest natural loop: a branch to a label defined earlier in the function"""
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best


def mix_of(body, scope):
    lo, hi = hot_loop(body) if scope == "largest loop" else (0, len(body) - 1)
    hist, n_all = {}, 0
    for l in body[lo:hi + 1]:
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)(?:\s*;.*)?$", l)
        if not m or m.group(1).startswith("."):
            continue
        n_all += 1
        c = classify(m.group(1), m.group(2))
        if c:
            hist[c] = hist.get(c, 0) + 1
    n = sum(hist.values())
    return {"scope": scope, "loop_instructions": n_all, "loop_valu_instructions": n, "classes": {k: {"count": v, "share": round(v / n, 4), "ns_per_instruction": COST_NS[k]} for k, v in sorted(hist.items())},
            "mix_ns_per_instruction": round(sum(v / n * COST_NS[k] for k, v in hist.items()), 4)}


def main():
    out = {"what": "static VALU instruction mix of each kernel's largest loop (hipcc -S, gfx950) priced with the measured cost per instruction form at 8 waves per SIMD "
                   "(profiles/r2_valu_model.txt); bench.py turns it into roofline_valu.mix_floor_frac",
           "csrc_sha16": csrc_sha16(), "cost_ns": COST_NS, "issue_peak_ns_per_instruction": round(2 / 2.4, 4), "kernels": {}}
    for src, ks in KERNELS.items():
        lines = disassemble(src)
        for name, (prefix, scope) in ks.items():
            out["kernels"][name] = mix_of(function_body(lines, prefix), scope)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
