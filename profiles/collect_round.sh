#!/bin/bash
# Everything under profiles/ for one round tag, in one gpurun call:
#   gpurun --timeout 2400 -- 'bash profiles/collect_round.sh r3c'      then: cp gpurun_out/<tag>*/<tag>* profiles/
# = collect.sh (the bench workload: kernel stats, timeline, PMC traffic + VALU passes, the bench line) plus
# collect_workload.sh for the side workloads (C3 both sizes, C4 with the two renders concurrent and serial), the FPS
# sweep, the forward-only loops and the two soaks.
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export HGS_GPU_UNIQUE_ID=${HGS_GPU_UNIQUE_ID:-$(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | sed -n 's/.*Unique ID: *0x\([0-9a-fA-F]*\).*/\1/p' | head -1)}
bash profiles/collect.sh $TAG > /dev/null 2>&1
bash profiles/collect_workload.sh ${TAG}_c3_110210 tools/bench_c3.py > /dev/null 2>&1
bash profiles/collect_workload.sh ${TAG}_c3_6890 tools/bench_c3.py 6890 > /dev/null 2>&1
bash profiles/collect_workload.sh ${TAG}_c4 tools/bench_c4.py > /dev/null 2>&1
HGS_CONCURRENT_RENDERS=0 bash profiles/collect_workload.sh ${TAG}_c4_serial tools/bench_c4.py > /dev/null 2>&1
# (round 4) a trained-scene-like frame -- surfaces, heavy-tailed sizes, reset opacities, a human in front -- and the all-rows step
bash profiles/collect_workload.sh ${TAG}_trained bench.py --profile trained --no-cpu-baseline --no-two-streams > /dev/null 2>&1
bash profiles/collect_workload.sh ${TAG}_step tools/bench_step.py --only fused > /dev/null 2>&1
OUT=$ROOT/gpurun_out/$TAG
python3 tools/sweep.py > "$OUT/${TAG}_sweep.json" 2> "$OUT/sweep.err"
python3 tools/bench_fwd.py > "$OUT/${TAG}_fwd.jsonl" 2> "$OUT/fwd.err"
# the widening rows measured this round: K nearest neighbours through the grid (f-2), photometric loss (f-5), C4 with the loss
(python3 tools/bench_knn.py --shape body; python3 tools/bench_knn.py --shape body --no-grid; python3 tools/bench_knn.py --shape blob) > "$OUT/${TAG}_knn.jsonl" 2> "$OUT/knn.err"
python3 tools/bench_loss.py > "$OUT/${TAG}_loss.json" 2> "$OUT/loss.err"
HGS_C4_WITH_LOSS=1 python3 tools/bench_c4.py > "$OUT/${TAG}_c4_with_loss.json" 2> "$OUT/c4l.err"
python3 profiles/median_of.py 3 python3 tools/bench_step.py --only fused > "$OUT/${TAG}_step_median_of_3.json" 2> "$OUT/step3.err"
python3 tools/bench_step.py > "$OUT/${TAG}_step.json" 2> "$OUT/step.err"
python3 tools/bench_rotations.py > "$OUT/${TAG}_rotations.txt" 2> "$OUT/rot.err"
bash profiles/collect_rows.sh $TAG > /dev/null 2>&1; cp "$ROOT"/gpurun_out/${TAG}_rows/${TAG}_*_pmc.txt "$ROOT"/gpurun_out/${TAG}_rows/${TAG}_*_kernel_stats.txt "$OUT"/ 2>/dev/null
python3 tools/fuzz_rows.py --seconds 45 > "$OUT/${TAG}_fuzz_rows.txt" 2>&1
SOAK_SECONDS=40 python3 tools/soak.py > "$OUT/${TAG}_soak.txt" 2>&1
python3 tools/soak_churn.py > "$OUT/${TAG}_soak_churn.txt" 2>&1
# (round 5) the host side of a frame: where the host time goes level by level, what a HIP graph of the launches would buy, and the
# statement-by-statement adapter (two autograd nodes, torch's stream fences) next to the one-call path for C3 / C4
python3 tools/microbench/host_split.py > "$OUT/${TAG}_host_split.txt" 2>&1
make -C tools/microbench bin/graph_launch > /dev/null 2>&1
./tools/microbench/bin/graph_launch > "$OUT/${TAG}_graph_launch.txt" 2>&1
HGS_FRAME_CALL=0 python3 tools/bench_c3.py 6890 > "$OUT/${TAG}_c3_6890_statement_path.json" 2> /dev/null
HGS_FRAME_CALL=0 python3 tools/bench_c4.py > "$OUT/${TAG}_c4_statement_path.json" 2> /dev/null
HGS_EMIT_SCAN=0 python3 tools/bench_c3.py 6890 > "$OUT/${TAG}_c3_6890_scan_kernel.json" 2> /dev/null
HGS_EMIT_SCAN=0 python3 tools/bench_c3.py > "$OUT/${TAG}_c3_110210_scan_kernel.json" 2> /dev/null
rm -rf "$ROOT"/gpurun_out/${TAG}*/trace "$ROOT"/gpurun_out/${TAG}*/pmc_*
tail -n 3 "$OUT/${TAG}_soak.txt" "$OUT/${TAG}_soak_churn.txt"
cat "$OUT/${TAG}_bench.json"
