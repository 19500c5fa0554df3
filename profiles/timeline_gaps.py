#!/usr/bin/env python3
"""Per-frame kernel timeline (start offset, duration, idle gap before each kernel) from a rocprofv3 --kernel-trace
result DB, averaged over the steady-state frames.  A frame starts at each preprocess_kernel launch.

    python profiles/timeline_gaps.py gpurun_out/<dir>/<name>_results.db [skip_frames]
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, skip=30):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    frames, cur_f = [], None
    for name, st, en in rows:
        short = name.split("(")[0].split("<")[0].replace("void ", "")
        if short.endswith("preprocess_kernel"):
            cur_f = []
            frames.append(cur_f)
        if cur_f is not None:
            cur_f.append((short, st, en))
    frames = frames[skip:-1]
    if not frames:
        print("no frames")
        return
    # (a step may hold several KINDS of frames -- the joint and the human-only render of a HUGS step: one table per kind,
    #  most frequent first)
    kinds = sorted(set(len(f) for f in frames), key=lambda k: -[len(f) for f in frames].count(k))
    for n in kinds:
        group = [f for f in frames if len(f) == n]
        if len(group) >= 3:
            report(path, group, n)


def report(path, frames, n):
    acc = defaultdict(lambda: [0.0, 0.0, 0.0])
    order = []
    span = 0.0
    for fi, f in enumerate(frames):
        t0 = f[0][1]
        prev_end = None
        for k, (name, st, en) in enumerate(f):
            key = (k, name)
            if fi == 0:
                order.append(key)
            a = acc[key]
            a[0] += (st - t0) / 1e3
            a[1] += (en - st) / 1e3
            a[2] += ((st - prev_end) / 1e3) if prev_end is not None else 0.0
            prev_end = en if prev_end is None else max(prev_end, en)
    for a, b in zip(frames[:-1], frames[1:]):
        span += (b[0][1] - a[0][1]) / 1e3
    m = len(frames)
    print(f"# {path}: {m} steady-state frames of {n} kernels; frame period {span / (m - 1):.1f} us")
    print(f"# {'start_us':>9} {'dur_us':>8} {'gap_before_us':>13}  kernel")
    busy = gaps = 0.0
    for key in order:
        a = acc[key]
        print(f"  {a[0] / m:9.1f} {a[1] / m:8.1f} {a[2] / m:13.1f}  {key[1]}")
        busy += a[1] / m
        gaps += a[2] / m
    print(f"# busy {busy:.1f} us, gaps inside frame {gaps:.1f} us, gap to next frame {span / (m - 1) - busy - gaps:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
