#!/usr/bin/env python3
"""ONE table of the library's path-selection rules, generated from the constants in the kernels' sources (the header,
INTEGRATION.md and binning.hip had drifted apart twice).  The numbers are read out of ml-hugs_amd/csrc/hgs_common.h and binning.hip; the
table is written between the `BEGIN/END GENERATED: path selection` markers of include/hgs_rasterizer.h and INTEGRATION.md, and
tests/test_abi.py fails when either no longer equals what this tool generates.

    python tools/gen_thresholds.py            # rewrite the two blocks
    python tools/gen_thresholds.py --check    # exit 1 if they are stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("ml-hugs_amd/csrc/hgs_common.h", "ml-hugs_amd/csrc/binning.hip", "ml-hugs_amd/csrc/binning_walk.h")
TARGETS = (("include/hgs_rasterizer.h", " * "), ("INTEGRATION.md", ""))
BEGIN, END = "BEGIN GENERATED: path selection", "END GENERATED: path selection"


def constants():
    text = "\n".join(open(os.path.join(ROOT, f)).read() for f in SOURCES)
    out = {}
    for m in re.finditer(r"constexpr\s+(?:int|uint32_t|long|size_t)\s+([^;]+);", text):
        for part in m.group(1).split(","):
            mm = re.match(r"\s*([A-Z][A-Z0-9_]+)\s*=\s*(.+?)\s*$", part)
            if not mm:
                continue
            expr = mm.group(2).replace("u", "").replace("(uint32_t)", "").replace("(int)", "")
            try:
                out[mm.group(1)] = int(eval(expr, {}, dict(out)))
            except Exception:
                pass
    for m in re.finditer(r"#define\s+(HGS_[A-Z0-9_]+)\s+(\d+)u?\b", text):
        out.setdefault(m.group(1), int(m.group(2)))
    return out


ROWS = [
    ("decision", "rule (the frame's own numbers, taken by the tile scan unless said otherwise)", "constants"),
    ("frame kind: DENSE (one backward wave per tile) or SPARSE (a wave per 8x8 quad; from the forward's checkpoints: per 32-entry segment)",
     "n = non-empty tiles, E = sum(len^2) / N (the list length a random entry sits in), mean = N / n.  n >= {DENSE_ALWAYS_TILES}: dense unless E > {DENSE_ALWAYS_E_MAX} (+ up to {DENSE_ALWAYS_E_RISE} more below 8 192 tiles, linearly: {DENSE_ALWAYS_E_MAX} + {DENSE_ALWAYS_E_RISE} at {DENSE_ALWAYS_TILES}) AND the depth is the frame's own: E <= {DENSE_ALWAYS_TAIL_X10} / 10 mean (a heavy tail on a covered frame -- a person in front of a scene -- stays dense: its deep tiles take the checkpointed walk) and the frame is not flat (longest list <= 1.25 E, E <= {DENSE_E_FLAT_MAX}: dense).  "
     "{DENSE_MIN_TILES} <= n < {DENSE_ALWAYS_TILES}: dense while E <= min({DENSE_E_MAX}, 0.45 (n - {DENSE_E_ORIGIN})) -- up to {DENSE_E_FLAT_MAX} on a flat frame (longest list <= 1.25 E) -- and E <= 2.5 mean.  n < {DENSE_MIN_TILES}: sparse",
     "DENSE_ALWAYS_TILES, DENSE_ALWAYS_E_MAX, DENSE_ALWAYS_E_RISE, DENSE_ALWAYS_TAIL_X10, DENSE_MIN_TILES, DENSE_E_MAX, DENSE_E_ORIGIN, DENSE_E_FLAT_MAX (binning.hip, frame_is_sparse)"),
    ("checkpoints for the depth-segmented backward (when the caller offers a buffer)",
     "sparse frame: every tile -- none when n >= {NO_CKPT_MIN_TILES} and E < 1.6 mean (hgs_forward_state.ckpt_slots_used = -1).  dense frame: its tiles of >= {HGS_CKPT_DEEP_MIN} entries, "
     "and only when the shape's last frame held a list beyond {HGS_DEEP_BWD_MIN} entries (host, from the shape's record)",
     "CKPT_DEEP_MIN, DEEP_BWD_MIN, CKPT_SEG = {CKPT_SEG} entries per segment (hgs_common.h), NO_CKPT_MIN_TILES (binning.hip)"),
    ("LONG lists (sorted ahead of the fused kernel by the long tiles' kernels)",
     "sparse frame: beyond {LONG_MIN_SPARSE} entries when mean >= {HGS_DEEP_MEAN_MIN} and {LONG_MIN_SPARSE_TILES} .. {LONG_ONE_ROUND} lists are that long; else beyond {LONG_MIN_SPARSE_SHALLOW} when "
     "{LONG_MIN_SPARSE_TILES} .. {LONG_ONE_ROUND} lists are; with more than {LONG_ONE_ROUND} lists beyond {LONG_MIN_SPARSE_SHALLOW}: beyond {LONG_MIN_SPARSE_SHALLOW} if the longest list is <= {SORT_CAP_MID} (flat), else beyond {SORT_CAP_SMALL}.  "
     "dense frame: none unless the frame holds a list beyond {SORT_CAP_SMALL}; then beyond {LONG_MIN_DENSE}, or beyond {LONG_MIN_SPARSE_SHALLOW} when more than {DENSE_LONG_MANY} lists lie beyond {LONG_MIN_DENSE}",
     "LONG_MIN_SPARSE, DEEP_MEAN_MIN, LONG_MIN_SPARSE_TILES, LONG_ONE_ROUND, LONG_MIN_SPARSE_SHALLOW, LONG_MIN_DENSE, DENSE_LONG_MANY, SORT_CAP_SMALL, SORT_CAP_MID (binning.hip, tile_scan_body)"),
    ("long tiles blended split by depth (four waves per quad: the deep workers)",
     "dense frames; sparse frames with mean >= {HGS_DEEP_MEAN_MIN} -- except more than {LONG_ONE_ROUND} long lists none of which is beyond {SORT_CAP_MID} entries (flat: one wave per quad), and except sparse frames of >= {DEEP_EVEN_TILES} non-empty tiles whose longest list is <= {DEEP_EVEN_L_X10} / 10 E (full and even: a person filling the frame)",
     "n_total[8], DEEP_EVEN_TILES, DEEP_EVEN_L_X10 (binning.hip); HGS_DEEP_FORWARD=0 / HGS_DEEP_MIN override"),
    ("per-tile sort inside the fused kernel", "<= 256 entries: bitonic network in registers; <= 1 024: bucket sort in LDS; <= {SORT_CAP_SMALL}: bitonic network, eight keys per thread; "
     "long tiles' kernel: one workgroup per list of <= {SORT_CAP_MID} entries, longer lists split by depth into parts of {PLAN_PART} .. {SORT_CAP_MID}", "SORT_CAP_SMALL, SORT_CAP_MID, PLAN_PART (binning.hip)"),
    ("binning groups (host, before the first kernel)",
     "by screen cell (two more launches) when P >= 32 768, tiles >= 4 096, cells <= {BIN_MAX_CELLS} and the shape's last frame covered at least half the tiles; else in storage order.  "
     "Per-tile LDS counters: 32-bit up to {BIN_LDS_TILES} tiles, 16-bit up to {BIN_LDS16_TILES}, global atomics beyond.  Splats of more than {BIN_SPREAD_MIN} tiles: groups of their own, {BIG_PER_GROUP} each",
     "bin_mode_for, BIN_LDS_TILES, BIN_LDS16_TILES, BIN_SPREAD_MIN, BIG_PER_GROUP (hgs_common.h)"),
    ("tile scan folded into the emit launch (host)", "frames enqueued on a capacity guess with <= {EMIT_SCAN_TILES} x {EMIT_SCAN_MAX_CHUNKS} tiles and <= 1 024 binning groups", "EMIT_SCAN_TILES, EMIT_SCAN_MAX_CHUNKS (binning.hip)"),
]


def table(c, prefix):
    fmt = lambda s: re.sub(r"\{([A-Z0-9_]+)\}", lambda m: f"{c[m.group(1)]:,}".replace(",", " "), s)
    lines = [f"{BEGIN} (tools/gen_thresholds.py: do not edit by hand)"]
    if prefix == "":   # markdown
        lines.append("")
        lines.append("| " + " | ".join(ROWS[0]) + " |")
        lines.append("|---|---|---|")
        for r in ROWS[1:]:
            lines.append("| " + " | ".join(fmt(x) for x in r) + " |")
        lines.append("")
    else:
        for r in ROWS[1:]:
            lines.append(f"- {fmt(r[0])}:")
            lines.append(f"    {fmt(r[1])}")
            lines.append(f"    [{fmt(r[2])}]")
    lines.append(END)
    return [(prefix + l).rstrip() if prefix else l for l in lines]


def main(check):
    c = constants()
    stale = False
    for path, prefix in TARGETS:
        p = os.path.join(ROOT, path)
        text = open(p).read().splitlines()
        b = next(i for i, l in enumerate(text) if BEGIN in l)
        e = next(i for i, l in enumerate(text) if END in l)
        new = text[:b] + table(c, prefix) + text[e + 1:]
        if new != text:
            stale = True
            if not check:
                open(p, "w").write("\n".join(new) + "\n")
    if check and stale:
        print("path-selection tables are stale: run python tools/gen_thresholds.py", file=sys.stderr)
        raise SystemExit(1)


if __name__ == "__main__":
    main("--check" in sys.argv)
