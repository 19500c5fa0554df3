#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) result DB into the plain-text kernel summary kept in profiles/.

    python profiles/summarize_rocprof.py gpurun_out/<dir>/<name>_results.db > profiles/<name>_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0]   # (kernels of the widening rows live in anonymous namespaces)
        if len(short) > 90:
            short = short[:87] + "..."
        print(f"  {calls:7d} {total:12.1f} {avg:10.3f} {pct:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1])
