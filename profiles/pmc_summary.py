#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (csv output) -> text table kept in profiles/.
    python profiles/pmc_summary.py gpurun_out/<dir>/<name>_counter_collection.csv [kernel substrings...]"""
import sys

import pandas as pd


def main(path, *subs):
    c = pd.read_csv(path)
    c["k"] = c["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.extract(r"(?:void )?(?:hgs::)?(\w+)")
    if subs:
        c = c[c.k.apply(lambda s: any(x in str(s) for x in subs))]
    g = c.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
    n = c.groupby("k")["Dispatch_Id"].nunique()
    print(f"# mean per dispatch, from {path}")
    for k, row in g.iterrows():
        print(f"{k}  (dispatches: {n[k]}, VGPR {int(c[c.k == k].VGPR_Count.iloc[0])}, SGPR {int(c[c.k == k].SGPR_Count.iloc[0])})")
        for name, v in row.items():
            if v == v:
                print(f"    {name:<28s} {v:.5g}")


if __name__ == "__main__":
    main(*sys.argv[1:])
