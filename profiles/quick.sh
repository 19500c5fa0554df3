#!/bin/bash
# Iteration helper (one gpurun call):  gpurun --timeout 900 -- 'bash profiles/quick.sh <tag> [pytest-args|-]'
# kernel-trace only (no PMC): per-kernel averages + the frame timeline of the bench workload, optionally the GPU tests.
set -u
TAG=${1:?tag}; TESTS=${2:--}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
if [ "$TESTS" != "-" ]; then (python3 -m pytest tests -m gpu -x -q $TESTS 2>&1 | tail -15) > "$OUT/tests.txt"; cat "$OUT/tests.txt"; fi
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o $TAG -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-two-streams \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/trace.err"
python3 profiles/summarize_rocprof.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_kernel_stats.txt"
python3 profiles/timeline_gaps.py "$OUT/trace/${TAG}_results.db" > "$OUT/${TAG}_timeline.txt"
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*_agent_info.csv" -delete; find "$OUT" -name "*_kernel_trace.csv" -delete; rm -f "$OUT"/trace/*.db
cat "$OUT/${TAG}_timeline.txt"; head -14 "$OUT/${TAG}_kernel_stats.txt"; python3 -c "
import json,sys; d=json.load(open('$OUT/${TAG}_bench.json')); print('FPS', d['value'], 'two-streams', d.get('two_frames_in_flight',{}).get('value'), d['stages_ms'])"
