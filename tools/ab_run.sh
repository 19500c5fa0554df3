#!/bin/bash
# On the GPU box: FPS and per-stage times of several builds (tools/ab_build.sh), uniform and clustered scene.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
CL=${AB_CLUSTERS:-"0 0.5"}
for c in $CL; do for v in "$@"; do
  HGS_RASTERIZER_LIB=scratch/lib_$v.so timeout 200 python3 bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-two-streams --cluster $c 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cluster=$c', '$v'.ljust(14), d['value'], d['config']['num_rendered_N'], d['stages_ms'])"
done; done
