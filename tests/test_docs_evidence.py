"""The documents cite evidence under profiles/ by file name; a citation of a file that is not tracked is a claim nobody can check
(evidence bugs).  CPU-only: file names, and the counts the shape-scan documents state about themselves."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("README.md", "DESIGN.md", "INTEGRATION.md", "profiles/README.md")


def _cited(text):
    names = set()
    for m in re.finditer(r"`(?:profiles/)?(r\d[a-z]\d?_[A-Za-z0-9_*{},.\-]+?\.(?:json|jsonl|txt))`", text):
        names.add(m.group(1))
    return names


def test_every_profile_a_current_document_cites_is_tracked():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for name in sorted(_cited(text)):
            if "{" in name:   # r4_trace_{step,trained}.txt
                head, alts, tail = re.match(r"(.*)\{(.*)\}(.*)", name).groups()
                cands = [head + a + tail for a in alts.split(",")]
            else:
                cands = [name]
            for c in cands:
                if not glob.glob(os.path.join(ROOT, "profiles", c)):
                    missing.append((doc, c))
    assert not missing, missing


def test_the_shape_scans_say_what_the_documents_say_about_them():
    want = {"r6z_shape_scan.json": (194, 1), "r6z_shape_scan_holdout.json": (60, 2), "r6z_shape_scan_person_grid.json": (72, 0),
            "r6z_shape_scan_holdout2.json": (36, 2), "r6z_shape_scan_holdout3.json": (36, 1), "r6a_shape_scan_before.json": (186, 37)}
    for name, (points, wins) in want.items():
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["n_points"] == points and d["n_points_where_forced_wins_by_5pct"] == wins, (name, d["n_points"], d["n_points_where_forced_wins_by_5pct"])
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for phrase in ("37 of 186", "1 of 194", "2 of 60", "0 of 72", "2 of 36"):
        assert phrase in design.replace("**", ""), phrase


def test_the_tracked_bench_line_is_of_the_committed_kernel_sources():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from build_id import csrc_sha16
    here = csrc_sha16()
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6*_pmc_traffic.json")))
    c2 = [f for f in newest if json.load(open(f)).get("workload", {}).get("gaussians") == 200_000]
    assert c2, "no PMC traffic summary of the bench workload is tracked"
    assert json.load(open(c2[-1]))["csrc_sha16"] == here, "the tracked PMC summaries were taken on other kernel sources: re-run profiles/collect_round.sh"
