#!/bin/bash
# Build a library variant for the A/B scripts: recompile the listed csrc files with extra flags, link with the default objects.
#   tools/build_variant.sh NAME "FLAGS" file1.hip [file2.hip ...]      -> tools/bin/libs/NAME.so
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/plonky2_goldibear_amd/build
TMP=$(mktemp -d)
mkdir -p $ROOT/tools/bin/libs
OBJS=""
for o in $OBJ/*.hip.o; do
    b=$(basename $o .o)
    if [[ " $* " == *" $b "* ]]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $FLAGS -c $ROOT/plonky2_goldibear_amd/csrc/$b -o $TMP/$b.o &
        OBJS="$OBJS $TMP/$b.o"
    else
        OBJS="$OBJS $o"
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libs/$NAME.so $OBJS
rm -rf $TMP
echo $ROOT/tools/bin/libs/$NAME.so
