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
# round 4: dword-per-lane I/O around the quad search against the product's 16-byte-per-lane reads / write (all measured slower)
build narrow4 -DMH_NARROW_IO &                                  # previous pairing, winner's record and pairing a dword per lane
build narrow3 -DMH_NARROW_IO -DMH_QUAD_W=3 &                    # ... with three records in flight per lane (no scratch)
build narrow4xyz -DMH_NARROW_IO -DMH_NARROW_XYZ &               # ... plus one coordinate load per lane through a lane-dependent base pointer
if [ -n "$WITH_CARRY" ]; then                # the winner's record carried in registers (measured slower: profiles/r04_match_kernel.md)
  build carry4 -DMH_CARRY_WINNER &
  build carry3 -DMH_CARRY_WINNER -DMH_QUAD_W=3 &
  build carry4w7 -DMH_CARRY_WINNER -DMH_QUAD_WAVES=7 &
fi
wait
ls -la $REPO/tools/variants
