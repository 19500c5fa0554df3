#!/usr/bin/env python3
"""Readable summary of a tools/shape_scan.py document: the points where a forced alternative beats the default, best first.
    python tools/scan_report.py gpurun_out/.../shape_scan.json [min_gain]"""
import json
import sys


def main(path, min_gain=0.03):
    d = json.load(open(path))
    rows = d["points"]
    print(f"{d['n_points']} points, {d['n_points_where_forced_wins_by_5pct']} where a forced alternative wins by more than 5 %")
    bad = [r for r in rows if r.get("gain_of_best_forced") and r["gain_of_best_forced"] > min_gain]
    bad.sort(key=lambda r: -r["gain_of_best_forced"])
    for r in bad:
        f, dm, L = r["forced_ms"], r["default_ms"], r.get("lists") or {}
        top = sorted(((dm / v - 1, k) for k, v in f.items() if v), reverse=True)[:5]
        ne = L.get("nonempty")
        stats = f" ne={ne} m={L['N'] / max(1, ne):.0f} E={L['sum_sq_over_N']:.0f} L={L['longest']} b256={L['beyond']['256']} b1k={L['beyond']['1024']} b2k={L['beyond']['2048']}" if L else ""
        print(f"{r['kind']:16s} {r['W']}x{r['H']} P={r['P']:8d} D={r['D']} {('d=%s' % r.get('dist')) if 'dist' in r else '':6s} def {dm:.4f} "
              f"{'S' if r['sparse_frame'] else 'D'}{'L' if r['has_long_tiles'] else ' '} ck={r['ckpt_MB']}{stats} | " + ", ".join(f"{k} {100 * g:+.1f}%" for g, k in top))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.03)
