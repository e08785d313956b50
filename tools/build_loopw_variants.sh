#!/bin/bash
# A/B libraries of k_icpw's shape (waves per workgroup, points per wave, list sizes, register budget) -> tools/variants/libmolahip_<name>.so
set -e
REPO=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd $REPO/mola_lidar_odometry_amd/csrc
mkdir -p $REPO/tools/variants
build() {  # name flags...
  local name=$1; shift
  local dir=/tmp/mh_var_$name
  mkdir -p $dir
  for f in mh_api mh_map mh_preprocess mh_tile; do cp $f.o $dir/$f.o; done   # unchanged objects of the product build
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --offload-arch=gfx950 -I../../include "$@" -c mh_icp.hip -o $dir/mh_icp.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/variants/libmolahip_$name.so $dir/*.o
  $REPO/tools/kernel_resources.sh $dir/mh_icp.o "k_icpw" | sed "s/^/$name: /"
}
for v in "$@"; do
  case $v in
    w4p32r2) build $v -DMH_LW_MIN_WGS=2 & ;;
    w4p16)   build $v -DMH_LW_PTS=16 -DMH_LW_NCL=512 -DMH_LW_NCH=512 & ;;
    w4p16r2) build $v -DMH_LW_PTS=16 -DMH_LW_NCL=512 -DMH_LW_NCH=512 -DMH_LW_MIN_WGS=2 & ;;
    w8p16)   build $v -DMH_LW_WAVES=8 -DMH_LW_PTS=16 -DMH_LW_NCL=512 -DMH_LW_NCH=384 & ;;
    w2p32)   build $v -DMH_LW_WAVES=2 -DMH_LW_PTS=32 & ;;
    w2p16)   build $v -DMH_LW_WAVES=2 -DMH_LW_PTS=16 -DMH_LW_NCL=512 -DMH_LW_NCH=512 & ;;
  esac
done
wait
