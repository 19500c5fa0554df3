#!/usr/bin/env python3
"""The headline curve of BASELINE.json / SURVEY.md 8d: rasterizer FPS at 1080p against the number of Gaussians, forward
only and forward+backward, one MI355X -- plus the measured stream-copy bandwidth of the box next to the datasheet
figure.  Runs bench.py once per point (its JSON line is the measurement) and prints one JSON document.
    python tools/sweep.py > gpurun_out/sweep.json"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stream_copy_gbps():
    """the practical HBM ceiling of this GPU, now: the library's float4-per-thread copy kernel -- the ONE measured peak every tool reports
    (bench.py's roofline.peak_measured; round 5 had Tensor.copy_ here: 5 472 against 6 186 GB/s on the same box)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
    import bench
    return bench.measured_copy_peak(torch, torch.device("cuda:0"))[0]


def main():
    out = {"hbm_stream_copy_GBps_measured": round(stream_copy_gbps(), 1), "hbm_stream_copy_how": "hgs_copy_bandwidth: 1 GiB float4-per-thread copy, (read + write) / time",
           "hbm_peak_GBps_datasheet": 8000.0, "points": []}
    for P in (50_000, 100_000, 200_000, 300_000, 500_000, 1_000_000, 2_097_152):   # (the last: max_n_gaussians of hugs_scene.yaml:117)
        row = {"gaussians": P}
        for mode, flag in (("fwd", ["--forward-only"]), ("fwd_bwd", [])):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", str(P), "--steps", "300", "--warmup", "40",
                                "--no-cpu-baseline", "--no-two-streams"] + flag, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise SystemExit(f"bench.py failed for P={P} ({mode}), rc {r.returncode}:\n{r.stderr[-2000:]}")
            d = json.loads(r.stdout.strip().splitlines()[-1])
            row[mode] = {"fps": d["value"], "ms": d["ms_per_step"], "host_busy_us": d.get("host_busy_us_per_frame"), "stages_ms": d["stages_ms"]}
            row["num_rendered_N"], row["visible"] = d["config"]["num_rendered_N"], d["config"]["visible"]
            row[mode]["whole_frame_GBps"] = d["whole_frame"]["GB_per_s"]
        out["points"].append(row)
        print(f"P={P}: fwd {row['fwd']['fps']:.0f} FPS, fwd+bwd {row['fwd_bwd']['fps']:.0f} FPS, N={row['num_rendered_N']}", file=sys.stderr)
    # (round 5) frames under 4 096 non-empty tiles that are NOT a human alone -- 720p and below, every tile covered: "sparse" to the
    # library's heuristics, whose long-tile threshold such frames choose by how many lists it leaves (binning.hip, LONG_ONE_ROUND)
    out["smaller_frames"] = []
    for P, H, W in ((100_000, 720, 1280), (20_000, 540, 960), (60_000, 512, 512), (10_000, 256, 256)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", str(P), "--height", str(H), "--width", str(W), "--steps", "300",
                            "--warmup", "40", "--no-cpu-baseline", "--no-two-streams"], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise SystemExit(f"bench.py failed for P={P} {W}x{H}, rc {r.returncode}:\n{r.stderr[-2000:]}")
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out["smaller_frames"].append({"gaussians": P, "height": H, "width": W, "num_rendered_N": d["config"]["num_rendered_N"],
                                      "fwd_bwd": {"fps": d["value"], "ms": d["ms_per_step"], "host_busy_us": d.get("host_busy_us_per_frame"), "stages_ms": d["stages_ms"]}})
        print(f"P={P} {W}x{H}: fwd+bwd {d['value']:.0f} FPS", file=sys.stderr)
    # (round 4) the trained-scene profile: 200 000 scene + 110 210 human Gaussians, surfaces / heavy-tailed sizes / reset opacities
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--profile", "trained", "--steps", "300", "--warmup", "40",
                        "--no-cpu-baseline", "--no-two-streams"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise SystemExit(f"bench.py --profile trained failed, rc {r.returncode}:\n{r.stderr[-2000:]}")
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out["trained_profile"] = {"fwd_bwd": {"fps": d["value"], "ms": d["ms_per_step"], "stages_ms": d["stages_ms"]},
                              "gaussians": d["config"]["gaussians"], "num_rendered_N": d["config"]["num_rendered_N"], "visible": d["config"]["visible"]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
