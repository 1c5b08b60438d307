#!/bin/bash
# BUILD HOST: a copy of libymk.so in which ONE source file is compiled with extra defines — for interleaved A/B runs on the GPU box
# (tools/micro/calls_ab.sh).   tools/micro/lib_variant.sh <name> <file.hip> [-DFLAG ...]   ->  tools/micro/_dwab/libymk_<name>.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
mkdir -p tools/micro/_dwab
OBJ=tools/micro/_dwab/${NAME}_$(basename $SRC .hip).o
EXTRA=""
case $SRC in nms.hip|elementwise.hip|post.hip|preproc.hip) EXTRA="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 $EXTRA -std=c++17 -fPIC "$@" -c yolo_master_amd/csrc/$SRC -o $OBJ
OBJS=""
for o in yolo_master_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" == "$(basename $SRC .hip).o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o tools/micro/_dwab/libymk_${NAME}.so $OBJS
ls -la tools/micro/_dwab/libymk_${NAME}.so
