#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a round tag:
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1h'      then copy gpurun_out/<tag>/<tag>_* into profiles/
# kernel-trace/stats and each PMC set are separate rocprofv3 runs (never --pmc together with other trace domains).
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# (the GPU's unique id, asked here -- no GPU process exists yet --: the python tools must not start rocm-smi once they have touched the GPU)
export HGS_GPU_UNIQUE_ID=${HGS_GPU_UNIQUE_ID:-$(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | sed -n 's/.*Unique ID: *0x\([0-9a-fA-F]*\).*/\1/p' | head -1)}
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o $TAG -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-two-streams \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/trace.err"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -o p -- python3 bench.py --steps 5 --warmup 2 \
        --no-cpu-baseline --no-two-streams > /dev/null 2> "$OUT/pmc_$name.err"
done
python3 profiles/summarize_rocprof.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_kernel_stats.txt"
python3 profiles/timeline_gaps.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_timeline.txt"
F=$(find "$OUT/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
Wc=$(find "$OUT/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 profiles/pmc_traffic.py "$F" "$Wc" $TAG 200000 1080 1920 3 > "$OUT/${TAG}_pmc_traffic.json"
python3 profiles/pmc_summary.py "$(find "$OUT/pmc_SQ_INSTS_VALU" -name '*counter_collection.csv' | head -1)" blend sort emit count scatter preprocess scan > "$OUT/${TAG}_pmc_sq_set1.txt"
python3 profiles/pmc_summary.py "$(find "$OUT/pmc_GRBM_GUI_ACTIVE" -name '*counter_collection.csv' | head -1)" blend sort emit count scatter preprocess scan > "$OUT/${TAG}_pmc_sq_set2.txt"
python3 profiles/valu_utilization.py "$OUT/${TAG}_pmc_sq_set1.txt" "$OUT/${TAG}_pmc_sq_set2.txt" 200000 1080 1920 3 uniform > "$OUT/${TAG}_valu_utilization.json"
python3 profiles/valu_mix.py > "$OUT/${TAG}_valu_mix.json" 2> "$OUT/valu_mix.err"   # (static: hipcc -S of the running sources)
# the bench line LAST, with this build's own PMC summaries in place (bench.py attaches traffic / VALU figures only from
# summaries whose recorded source hash is the running build's)
cp "$OUT/${TAG}_pmc_traffic.json" "$OUT/${TAG}_valu_utilization.json" "$OUT/${TAG}_valu_mix.json" "$ROOT/profiles/"
python3 profiles/median_of.py 3 python3 bench.py --no-cpu-baseline --no-two-streams > "$OUT/${TAG}_bench_median_of_3.json" 2> "$OUT/bench3.err"   # (host noise: see median_of.py)
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*_agent_info.csv" -delete; find "$OUT" -name "*_kernel_trace.csv" -delete
ls -la "$OUT"
cat "$OUT/${TAG}_timeline.txt"
