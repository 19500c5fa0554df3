#!/usr/bin/env python3
"""VALU issue utilisation of the kernels from the two SQ PMC summaries of a round:
    busy fraction = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
(SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).

    python profiles/valu_utilization.py profiles/<tag>_pmc_sq_set1.txt profiles/<tag>_pmc_sq_set2.txt > profiles/<tag>_valu_utilization.json"""
import json
import re
import sys

from build_id import csrc_sha16


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\w+)\s+\(dispatches", line)
        if m:
            cur = out.setdefault(m.group(1), {})
        elif cur is not None and line.startswith("    "):
            k, v = line.split()
            cur[k] = float(v)
    return out


def main(set1, set2, *workload):
    a, b = parse(set1), parse(set2)
    # the workload the counters were taken on: `gaussians height width sh_degree [profile]` for bench.py (which attaches the figures to its
    # line only on the same workload), else a free label (the side workloads: the script and its arguments)
    if len(workload) >= 4 and all(w.lstrip("-").isdigit() for w in workload[:4]):
        wl = {"gaussians": int(workload[0]), "height": int(workload[1]), "width": int(workload[2]), "sh_degree": int(workload[3]),
              "profile": workload[4] if len(workload) > 4 else "uniform"}
    else:
        wl = {"label": " ".join(workload)} if workload else {}
    res = {"source": [set1, set2], "csrc_sha16": csrc_sha16(), "workload": wl, "formula": "SQ_ACTIVE_INST_VALU*4 / (1024 * GRBM_GUI_ACTIVE/8)",
           "note": "numerator and denominator come from two separate rocprofv3 PMC passes (counter groups that cannot be collected "
                   "together): the ratio carries their run-to-run difference of a few per cent and can read slightly above 1",
           "kernels": {}}
    for k in sorted(set(a) & set(b)):
        if "SQ_ACTIVE_INST_VALU" in a[k] and "GRBM_GUI_ACTIVE" in b[k]:
            cyc = b[k]["GRBM_GUI_ACTIVE"] / 8.0
            instr = a[k].get("SQ_INSTS_VALU")
            res["kernels"][k] = {"valu_busy_frac": round(a[k]["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), 4),   # pipe-busy time: NOT a fraction of peak issue
                                 "valu_wave_instructions": instr, "kernel_cycles": round(cyc),
                                 # against the guide's issue rate (one wave64 VALU instruction per SIMD every 2 cycles)
                                 "issue_frac_of_peak": round(instr * 2.0 / (1024.0 * cyc), 4) if instr else None}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:])
