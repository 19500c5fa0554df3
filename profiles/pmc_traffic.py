#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) -> the JSON bench.py
reads for roofline.traffic.

    python profiles/pmc_traffic.py <fetch>_counter_collection.csv <write>_counter_collection.csv <round tag> \
        <gaussians> <height> <width> <sh_degree> > profiles/<tag>_pmc_traffic.json

Correction (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced streaming reads, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (an upper bound for kernels
whose reads are narrow/scalar, for which factor 1 -- "uncorrected" -- is the lower bound)."""
import json
import sys

import pandas as pd

from build_id import csrc_sha16


def per_kernel(path, counter):
    c = pd.read_csv(path)
    c = c[c["Counter_Name"] == counter]
    c["k"] = c["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.extract(r"(?:void )?(?:hgs::)?(\w+)")
    per_dispatch = c.groupby(["k", "Dispatch_Id"])["Counter_Value"].sum()   # summed over XCDs / instances
    return per_dispatch.groupby("k").mean()


def main(fetch_csv, write_csv, tag, P, H, W, D):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of "
                     f"`bench.py --steps 5 --warmup 2 --no-cpu-baseline` on MI355X, build {tag}; mean per dispatch, KiB as reported",
           "csrc_sha16": csrc_sha16(),
           "workload": {"gaussians": int(P), "height": int(H), "width": int(W), "sh_degree": int(D)},
           "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming "
                         "reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; most reads of the blend kernels are narrow or "
                         "scalar, for which the factor is uncalibrated, so this is an upper bound (factor 1 gives the lower bound).",
           "kernels": {}}
    for k in sorted(set(f.index) & set(w.index)):
        if not (k.endswith("_kernel") and not k.startswith(("vectorized", "elementwise"))):
            continue
        fk, wk = float(f[k]), float(w[k])
        out["kernels"][k] = {"FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                             "hbm_bytes_corrected": int((2 * fk + wk) * 1024),
                             "hbm_bytes_uncorrected": int((fk + wk) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:])
