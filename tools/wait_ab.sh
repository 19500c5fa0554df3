#!/bin/bash
# On the GPU box: the wait for N under injected oversleep (HGS_WAIT_TEST_OVERSLEEP_US), A/B library builds of tools/ab_build.sh:
#   tools/ab_build.sh waitfixed "-DHGS_WAIT_FIXED_MARGIN" waitadapt ""   then   gpurun -- bash tools/wait_ab.sh
cd $GRAFT_REPO_ROOT
for inj in 0 150 400; do for v in waitfixed waitadapt; do for P in 200000 300000; do
  HGS_WAIT_TEST_OVERSLEEP_US=$inj HGS_RASTERIZER_LIB=scratch/lib_$v.so python3 bench.py --gaussians $P --steps 400 --warmup 30 --no-cpu-baseline --no-two-streams 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inject $inj us', '$v'.ljust(10), $P, d['value'], 'host_busy', d.get('host_busy_us_per_frame'), 'cores', (d.get('ranks_host') or {}).get('cores_busy'))"
done; done; done
