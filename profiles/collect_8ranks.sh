#!/bin/bash
# The HOST side of BASELINE configs[4] (8 frames x 300 000 Gaussians, one rank per GPU) measured on a ONE-GPU lease: N ranks share the
# GPU over gloo, so the GPU is the common limit and what shows is whether N rank processes -- each spinning for its N and running an
# autograd thread -- fit the box's CPU quota ("next round" item 2).  Aggregate FPS against the one-rank figure, per-rank
# host-busy time, process CPU seconds, the cgroup's throttle counters; then the same with every rank's affinity cut to 8 CPUs in total.
#   gpurun --timeout 900 -- 'bash profiles/collect_8ranks.sh r6x'
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_8ranks
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# (the GPU's unique id, asked here -- no GPU process exists yet --: the python tools must not start rocm-smi once they have touched the GPU)
export HGS_GPU_UNIQUE_ID=${HGS_GPU_UNIQUE_ID:-$(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | sed -n 's/.*Unique ID: *0x\([0-9a-fA-F]*\).*/\1/p' | head -1)}
COMMON="--gaussians 300000 --steps 400 --warmup 30 --no-cpu-baseline --no-two-streams"
python3 bench.py $COMMON > "$OUT/ranks1.json" 2> "$OUT/ranks1.err"
for n in 2 4 8; do
  HGS_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus $n $COMMON > "$OUT/ranks$n.json" 2> "$OUT/ranks$n.err"
done
HGS_BENCH_BACKEND=gloo HGS_BENCH_CPUS_TOTAL=8 timeout 300 python3 bench.py --gpus 8 $COMMON > "$OUT/ranks8_cpus8.json" 2> "$OUT/ranks8_cpus8.err"
HGS_BENCH_BACKEND=gloo HGS_BENCH_CPUS_TOTAL=4 timeout 300 python3 bench.py --gpus 8 $COMMON > "$OUT/ranks8_cpus4.json" 2> "$OUT/ranks8_cpus4.err"
python3 - "$OUT" "$TAG" <<'PY'
import json, os, sys
out, tag = sys.argv[1], sys.argv[2]
rows = {}
for name in ("ranks1", "ranks2", "ranks4", "ranks8", "ranks8_cpus8", "ranks8_cpus4"):
    try:
        d = json.loads(open(os.path.join(out, name + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        rows[name] = {"error": str(e), "stderr_tail": open(os.path.join(out, name + ".err")).read()[-600:]}
        continue
    rows[name] = {"n_ranks": d["n_gpus"], "backend": d["backend"], "aggregate_fps": d["value"], "ms_per_step": d["ms_per_step"], "per_rank_fps": d["per_rank_fps"],
                  "rank_devices": d["rank_devices"], "ranks_host": d["ranks_host"], "workload": d["config"]["workload"]}
one = rows.get("ranks1", {}).get("aggregate_fps")
for r in rows.values():
    if one and "aggregate_fps" in r:
        r["aggregate_over_one_rank"] = round(r["aggregate_fps"] / one, 4)
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "ml-hugs_amd"))
box = None
try:
    import bench_common
    box = bench_common.box()
except Exception:
    pass
doc = {"what": "N ranks of bench.py --gaussians 300000 (configs[4]'s per-rank workload) SHARING ONE GPU over gloo: the host side of the 8-GPU job on one box's CPU quota; "
               "the GPU is the common limit, so aggregate_over_one_rank ~ 1 means the host keeps up", "box": box, "runs": rows}
json.dump(doc, open(os.path.join(out, f"{tag}_8ranks_shared_gpu.json"), "w"), indent=1)
print(json.dumps({k: {kk: v.get(kk) for kk in ("aggregate_fps", "aggregate_over_one_rank")} | {"cores_busy": v.get("ranks_host", {}).get("cores_busy"),
                      "nr_throttled": (v.get("ranks_host", {}).get("cgroup") or {}).get("nr_throttled"),
                      "host_busy": v.get("ranks_host", {}).get("host_busy_us_per_frame")} for k, v in rows.items()}, indent=1))
PY
