#!/bin/bash
# Build libadacodec.so from a SNAPSHOT of the sources (the kernel translation unit takes ~2.5 min and hipcc reads it
# twice, device pass then host pass): the working tree can be edited meanwhile.  usage: tools/snapbuild.sh <name>
# -> duckdb-adaptive-compression_amd/build/libadacodec_<name>.so (+ .resources.txt); select it with ADAC_LIB=...
set -e
NAME=${1:-snap}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SNAP=/tmp/adac_snap_$NAME
rm -rf "$SNAP" && mkdir -p "$SNAP"
cp -r "$ROOT/include" "$SNAP/include"
mkdir -p "$SNAP/pkg" && cp -r "$ROOT/duckdb-adaptive-compression_amd/csrc" "$ROOT/duckdb-adaptive-compression_amd/Makefile" "$SNAP/pkg/"
# reuse the objects of the translation units that did not change
mkdir -p "$SNAP/pkg/build"
make -C "$SNAP/pkg" all > "$SNAP/build.log" 2>&1 || { tail -30 "$SNAP/build.log"; grep -E "error" "$SNAP/pkg/build/adac_kernels.resources.txt" | head; exit 1; }
cp "$SNAP/pkg/libadacodec.so" "$ROOT/duckdb-adaptive-compression_amd/build/libadacodec_$NAME.so"
cp "$SNAP/pkg/build/adac_kernels.resources.txt" "$ROOT/duckdb-adaptive-compression_amd/build/adac_kernels_$NAME.resources.txt"
echo "built $ROOT/duckdb-adaptive-compression_amd/build/libadacodec_$NAME.so"
# --install as 2nd argument: the snapshot's objects and library become the in-tree build (only if the sources did not
# change since the snapshot was taken)
if [ "$2" = "--install" ]; then
  if diff -rq "$SNAP/pkg/csrc" "$ROOT/duckdb-adaptive-compression_amd/csrc" > /dev/null && diff -rq "$SNAP/include" "$ROOT/include" > /dev/null; then
    cp "$SNAP/pkg/build/"*.o "$SNAP/pkg/build/adac_kernels.resources.txt" "$ROOT/duckdb-adaptive-compression_amd/build/"
    cp "$SNAP/pkg/libadacodec.so" "$ROOT/duckdb-adaptive-compression_amd/libadacodec.so"
    make -q -C "$ROOT/duckdb-adaptive-compression_amd" all && echo "installed (make: up to date)" || echo "installed, but make wants to rebuild"
  else
    echo "sources changed since the snapshot: not installed"
  fi
fi
