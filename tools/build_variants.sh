#!/bin/bash
# Builds A/B variants of the library into tools/build/ab/<name>/lib<name>.so from -D switches of ONE translation unit (the others are
# taken from the in-tree build):   tools/build_variants.sh <unit.hip> name1:"-DX=1" name2:"-DX=2 -DY" ...
# (cross-compiles here; the .so files travel to the GPU box with the snapshot; tools/build is git-ignored)
set -e
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
UNIT=$1; shift
python $P/build.py > /dev/null
OUT=tools/build/ab; mkdir -p $OUT
# the unit's own switches, as build.py compiles it (ba.hip, for one, is built WITHOUT -ffp-contract=off: a variant built with other switches is
# not comparable with the in-tree library)
EXTRA=$(python -c "import sys; sys.path.insert(0, '$P'); import build; print(' '.join(build.UNITS['$UNIT']))")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $EXTRA $defs -c $P/csrc/$UNIT -o $OUT/$name.o
  objs=""
  for o in $P/build/*.o; do b=$(basename $o .o); if [ "$b.hip" = "$UNIT" ]; then objs="$objs $OUT/$name.o"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$name.so $objs
  rm $OUT/$name.o; echo built $OUT/lib$name.so
  ) &
done
wait
