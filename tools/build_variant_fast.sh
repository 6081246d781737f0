# Like tools/build_variant.sh, but recompiles only the translation units named in UNITS (default: wavefront.hip) with the extra
# flags and links them with cached objects of the others (build/obj, refreshed when a source is newer): ~15 s per variant.
#   UNITS="wavefront.hip film.hip" bash tools/build_variant_fast.sh v1 -DNORI_EXP_FOO=1
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
DEV=$ROOT/nori_amd/csrc/device
OBJ=$ROOT/build/obj
mkdir -p $OBJ
FLAGS="-DNORI_LAB -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-sink -fPIC -Wno-comment -Wno-unused-result"
UNITS=${UNITS:-wavefront.hip}
ALL="nori_hip.hip lbvh.hip wavefront.hip film.hip group.hip scene_prep.cpp"
NEWEST=$(ls -t $DEV/*.h $ROOT/include/nori_hip.h | head -1)
for f in $ALL; do
  if [ ! -f $OBJ/$f.o ] || [ $DEV/$f -nt $OBJ/$f.o ] || [ $NEWEST -nt $OBJ/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $DEV/$f -o $OBJ/$f.o &
  fi
done
TMP=$(mktemp -d)
for f in $UNITS; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $DEV/$f -o $TMP/$f.o & done
wait
LINK=""
for f in $ALL; do if [ -f $TMP/$f.o ]; then LINK="$LINK $TMP/$f.o"; else LINK="$LINK $OBJ/$f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LINK -ldl -o $ROOT/nori_amd/lib/libnori_hip_$NAME.so
rm -rf $TMP
echo built libnori_hip_$NAME.so
