#!/bin/bash
# On the GPU box: tools/ab_points.sh lib1 lib2 ...  -- the shape scan's default path on a few dense frames with deep tiles, per A/B library build (tools/ab_build.sh; "intree" = the built library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  L=""; [ "$v" != "intree" ] && L="scratch/lib_$v.so"
  HGS_RASTERIZER_LIB=$L python3 tools/shape_scan.py --out /tmp/ab_$v.json --person-grid --tracked --sizes 720x1280 --counts 300000 --humans "" --degrees 0 --variants default \
     --where "(kind==\"tracked_step_joint\" and ((H==720 and P in (630000,230000) and dist==4.0) or (H==1080 and P==310210 and dist==4.0))) or kind in (\"trained\",\"tracked_trained\",\"tracked_c4_joint\")" 2>/dev/null
  python3 - $v <<'PY'
import json,sys
d=json.load(open('/tmp/ab_%s.json'%sys.argv[1]))
print(sys.argv[1].ljust(8),' '.join('%s/%d/%d:%.4f(bwd %.4f)'%(p['kind'][-8:],p['H'],p['P'],p['default_ms'],p['stages_ms']['blend_backward']) for p in d['points']))
PY
done
