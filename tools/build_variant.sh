#!/bin/bash
# A variant of the library for A/B runs (bench.py / tools/ab_bench.sh pick it up through CMB_LIB=...):
#   tools/build_variant.sh <name> <unit: columba_amd | move_backend> "<extra hipcc flags>"   ->  columba_amd/_variants/lib_<name>.so
# Only the named translation unit is rebuilt; the other objects are those of the last build_library().
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNIT=$2; shift; shift
mkdir -p $R/columba_amd/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-variable "$@" -c -o $R/columba_amd/_variants/${UNIT}_$NAME.o $R/columba_amd/csrc/$UNIT.hip 2>/dev/null
OBJS=""
for U in columba_amd move_backend pair_sam pair_best; do
  if [ $U = $UNIT ]; then OBJS="$OBJS $R/columba_amd/_variants/${UNIT}_$NAME.o"; else OBJS="$OBJS $R/columba_amd/_build/$U.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/columba_amd/_variants/lib_$NAME.so $OBJS
rm -f $R/columba_amd/_variants/${UNIT}_$NAME.o
echo $R/columba_amd/_variants/lib_$NAME.so
