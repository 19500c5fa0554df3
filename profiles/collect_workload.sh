#!/bin/bash
# Kernel stats, timeline and VALU-issue PMC summaries of ONE of the side workloads (tools/bench_c3.py, tools/bench_c4.py, ...):
#   gpurun --timeout 900 -- 'bash profiles/collect_workload.sh r3a_c3 tools/bench_c3.py'
# then copy gpurun_out/<tag>/<tag>_* into profiles/.  The program goes directly after `--` (no env / bash -c hop), and
# every PMC set is its own rocprofv3 run with --kernel-trace only.
set -u
TAG=${1:?tag}
SCRIPT=${2:?python script}
shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# (the GPU's unique id, asked here -- no GPU process exists yet --: the python tools must not start rocm-smi once they have touched the GPU)
export HGS_GPU_UNIQUE_ID=${HGS_GPU_UNIQUE_ID:-$(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | sed -n 's/.*Unique ID: *0x\([0-9a-fA-F]*\).*/\1/p' | head -1)}
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o $TAG -- python3 "$SCRIPT" "$@" > "$OUT/${TAG}_under_rocprof.json" 2> "$OUT/trace.err"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
    name=$(echo $set | cut -d' ' -f1)
    HGS_BENCH_STEPS=6 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -o p -- python3 "$SCRIPT" "$@" \
        > /dev/null 2> "$OUT/pmc_$name.err"
done
python3 profiles/summarize_rocprof.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_kernel_stats.txt"
python3 profiles/timeline_gaps.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_timeline.txt"
python3 profiles/pmc_summary.py "$(find "$OUT/pmc_SQ_INSTS_VALU" -name '*counter_collection.csv' | head -1)" blend sort emit count scatter preprocess scan > "$OUT/${TAG}_pmc_sq_set1.txt"
python3 profiles/pmc_summary.py "$(find "$OUT/pmc_GRBM_GUI_ACTIVE" -name '*counter_collection.csv' | head -1)" blend sort emit count scatter preprocess scan > "$OUT/${TAG}_pmc_sq_set2.txt"
python3 profiles/valu_utilization.py "$OUT/${TAG}_pmc_sq_set1.txt" "$OUT/${TAG}_pmc_sq_set2.txt" "$SCRIPT" "$@" > "$OUT/${TAG}_valu_utilization.json"
F=$(find "$OUT/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
Wc=$(find "$OUT/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 profiles/pmc_traffic.py "$F" "$Wc" $TAG 0 0 0 0 > "$OUT/${TAG}_pmc_traffic.json" 2> "$OUT/traffic.err"
python3 profiles/median_of.py 3 python3 "$SCRIPT" "$@" > "$OUT/${TAG}.json" 2> "$OUT/run.err"   # (the median of three runs; all three in "repeats")
# the workload's own FETCH_SIZE / WRITE_SIZE passes -> roofline.traffic of its line (the dominant stage's kernel, by name)
python3 - "$OUT/${TAG}.json" "$OUT/${TAG}_pmc_traffic.json" <<'PY'
import json, sys
line, traffic = sys.argv[1], sys.argv[2]
try:
    d, t = json.load(open(line)), json.load(open(traffic))
    r = d.get("roofline") or {}
    k = t["kernels"].get(r.get("kernel"))
    if k:
        r["traffic"] = k["hbm_bytes_corrected"]
        r["traffic_uncorrected"] = k["hbm_bytes_uncorrected"]
        r["traffic_source"] = traffic.split("/")[-1] + " (FETCH_SIZE doubled per MI355X_MICROARCH.md: an upper bound for narrow / scalar reads)"
        json.dump(d, open(line, "w"))
except Exception as e:
    print("traffic not attached:", e, file=sys.stderr)
PY
find "$OUT" -name "*_agent_info.csv" -delete; find "$OUT" -name "*_kernel_trace.csv" -delete
rm -rf "$OUT/trace"/*/ 2>/dev/null
ls -la "$OUT"
cat "$OUT/${TAG}_kernel_stats.txt" "$OUT/${TAG}_timeline.txt" "$OUT/${TAG}_valu_utilization.json" "$OUT/${TAG}.json"
