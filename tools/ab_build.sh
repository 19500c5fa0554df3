#!/bin/bash
# A/B builds of the library:  tools/ab_build.sh name "EXTRA hipcc flags" [name2 "flags2" ...]  -> scratch/lib_<name>.so
# (scratch/ is git-ignored but travels with gpurun); compare with  tools/ab_run.sh name1 name2 ...  on the GPU box.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/ml-hugs_amd/csrc"
mkdir -p "$ROOT/scratch"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  objs=""
  for f in hgs_api preprocess binning blend densify knn lbs loss scene_forward rotations; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $([ $f = binning ] && echo "-mllvm -disable-machine-sink") $flags -c $f.hip -o /tmp/ab_${name}_$f.o &
    objs="$objs /tmp/ab_${name}_$f.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$ROOT/scratch/lib_$name.so" || exit 1
done
ls -la "$ROOT"/scratch/lib_*.so
