#!/usr/bin/env python3
"""Run a timing tool N times and print the MEDIAN run's JSON line with every run's figure beside it (`repeats`): the boxes' hosts are
shared and a run now and then lands in somebody else's burst (the same frame loop read 0.114 and 0.144 ms two minutes apart with
host-busy times of 94 and 134 us; profiles/r5c_boxes.jsonl).  The median of three is what a tracked number should be; all three stay visible.
    python3 profiles/median_of.py 3 python3 tools/bench_c3.py 6890 > out.json"""
import json
import subprocess
import sys

KEYS = ("ms_per_step", "ms_per_training_step_raster", "fused_rows_ms_per_step")


def main():
    n, cmd = int(sys.argv[1]), sys.argv[2:]
    runs = []
    for _ in range(n):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit(r.returncode)
        line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        key = next(k for k in KEYS if k in d)
        runs.append((d[key], d, key))
    runs.sort(key=lambda x: x[0])
    ms, d, key = runs[len(runs) // 2]
    d["repeats"] = {"key": key, "values": [r[0] for r in runs], "reported": "median"}
    print(json.dumps(d))


if __name__ == "__main__":
    main()
