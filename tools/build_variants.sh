#!/bin/bash
# The DEVELOPMENT library: the product sources + -DMH_DEV_VARIANTS (mh_dev_variants.h, mh_nn_dev_variants.h, mh_tile.hip: the tile,
# wave and sorted-scan matchers, MH_MATCH=t|w|o) -> tools/variants/libmolahip_dev.so.  Tests and tools pick it with
# MOLAHIP_LIB_PATH=<file> (+ LD_LIBRARY_PATH of a directory that holds it as libmolahip.so for the C++ host layer).
# Extra flags for A/B builds: EXTRA_FLAGS="-DMH_FLAT_W=6" NAME=w6 tools/build_variants.sh
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
NAME=${NAME:-dev}
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p $REPO/tools/variants /tmp/mh_var_$NAME
for f in mh_api mh_map mh_icp mh_preprocess mh_tile; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --offload-arch=gfx950 -I../../include \
    -DMH_DEV_VARIANTS $EXTRA_FLAGS -c $f.hip -o /tmp/mh_var_$NAME/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/variants/libmolahip_$NAME.so /tmp/mh_var_$NAME/*.o
ls -la $REPO/tools/variants/libmolahip_$NAME.so
