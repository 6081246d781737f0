# Builds nori_amd/lib/libnori_hip_<name>.so from the working tree with extra compiler flags (experiments; tools/ab.sh times them):
#   bash tools/build_variant.sh v1 -DNORI_EXP_FOO=1
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
DEV=$ROOT/nori_amd/csrc/device
TMP=$(mktemp -d)
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-sink -fPIC -Wno-comment -Wno-unused-result"
for f in nori_hip.hip lbvh.hip wavefront.hip film.hip group.hip scene_prep.cpp; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $DEV/$f -o $TMP/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/*.o -ldl -o $ROOT/nori_amd/lib/libnori_hip_$NAME.so
rm -rf $TMP
echo built libnori_hip_$NAME.so
