#!/bin/bash
# debug library for tools/match_floor.py: the product sources + -DMH_DEBUG_FLOOR (script capture in nn_search_quad, the replay
# kernel k_match_floor_b, mh_debug_floor_*) -> tools/libmolahip_floor.so (objects under /tmp).  Nothing else differs.
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p /tmp/mh_floor
for f in mh_api mh_map mh_icp mh_preprocess mh_tile; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --offload-arch=gfx950 -I../../include -DMH_DEBUG_FLOOR $EXTRA_DBG_FLAGS -c $f.hip -o /tmp/mh_floor/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/libmolahip_floor.so /tmp/mh_floor/*.o
