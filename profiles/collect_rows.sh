#!/bin/bash
# Kernel stats and PMC summaries (HBM traffic, VALU issue) of the widening rows' own kernels -- the photometric loss (f-5) and
# the neighbour search through the template grid (f-2):
#   gpurun --timeout 900 -- 'bash profiles/collect_rows.sh r3f'      then copy gpurun_out/<tag>_rows/<tag>_* into profiles/
# The program goes directly after `--`; every PMC set is its own rocprofv3 run with --kernel-trace only.
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_rows
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for w in loss knn; do
    case $w in loss) names="ssim_l1";; knn) names="knn_kernel lbsweight_top_k template_grid query_";; esac
    rocprofv3 --kernel-trace --stats -d "$OUT/trace_$w" -o $w -- python3 tools/bench_$w.py > "$OUT/${TAG}_${w}_under_rocprof.json" 2> "$OUT/trace_$w.err"
    python3 profiles/summarize_rocprof.py "$OUT/trace_$w/${w}_results.db" > "$OUT/${TAG}_${w}_kernel_stats.txt"
    : > "$OUT/${TAG}_${w}_pmc.txt"
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
               "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        name=$(echo $set | cut -d' ' -f1)
        HGS_BENCH_STEPS=6 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_${w}_$name" -o p -- python3 tools/bench_$w.py \
            > /dev/null 2> "$OUT/pmc_${w}_$name.err"
        python3 profiles/pmc_summary.py "$(find "$OUT/pmc_${w}_$name" -name '*counter_collection.csv' | head -1)" $names >> "$OUT/${TAG}_${w}_pmc.txt"
    done
done
rm -rf "$OUT"/trace_* "$OUT"/pmc_*/
ls -la "$OUT"; cat "$OUT/${TAG}_loss_pmc.txt" "$OUT/${TAG}_knn_pmc.txt"
