#!/bin/bash
# A/B libraries of the match-kernel experiments: the product sources + one set of -D flags each
# -> tools/variants/libmolahip_<name>.so.  bench.py / tests pick one with MOLAHIP_LIB_PATH=<file>.
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p $REPO/tools/variants
build() {  # name flags...
  local name=$1; shift
  local dir=/tmp/mh_var_$name
  mkdir -p $dir
  for f in mh_api mh_map mh_preprocess mh_tile; do [ -f $dir/$f.o ] || cp $f.o $dir/$f.o; done   # unchanged objects of the product build
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --offload-arch=gfx950 -I../../include "$@" -c mh_icp.hip -o $dir/mh_icp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/variants/libmolahip_$name.so $dir/*.o
}
build carry4 -DMH_CARRY_WINNER &
build carry3 -DMH_CARRY_WINNER -DMH_QUAD_W=3 &
build carry4w7 -DMH_CARRY_WINNER -DMH_QUAD_WAVES=7 &
wait
ls -la $REPO/tools/variants
