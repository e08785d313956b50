#!/bin/bash
# A/B libraries of the plan / scan matcher's tuning knobs (round 5): the product sources + one set of -D flags each
# -> tools/variants/libmolahip_<name>.so.  bench.py / tests pick one with MOLAHIP_LIB_PATH=<file>.
# usage: tools/build_flat_variants.sh name1:"-Dflag ..." name2:"..."
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p $REPO/tools/variants
build() {  # name flags...
  local name=$1; shift
  local dir=/tmp/mh_var_$name
  rm -rf $dir; mkdir -p $dir
  for f in mh_api mh_map mh_preprocess mh_tile; do cp $f.o $dir/$f.o; done   # unchanged objects of the product build
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function -Wno-pass-failed --offload-arch=gfx950 -I../../include "$@" -c mh_icp.hip -o $dir/mh_icp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/variants/libmolahip_$name.so $dir/*.o
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
done
wait
ls -la $REPO/tools/variants
