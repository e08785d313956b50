#!/bin/bash
# debug library with wall_clock64 phase stamps (-DMH_DEBUG_WAVETRACE) -> tools/libmolahip_dbg.so (objects under /tmp)
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p /tmp/mh_dbg
for f in mh_api mh_map mh_icp mh_preprocess mh_tile; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --offload-arch=gfx950 -I../../include -DMH_DEBUG_WAVETRACE -DMH_DEV_VARIANTS $EXTRA_DBG_FLAGS -c $f.hip -o /tmp/mh_dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/libmolahip_dbg.so /tmp/mh_dbg/*.o
