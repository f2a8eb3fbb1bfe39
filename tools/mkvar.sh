#!/bin/bash
# Developer tool (build container): a variant build of the library for same-box A / B runs.
#   tools/mkvar.sh <name> <file-stem> [extra hipcc flags ...]
# compiles gluon-nlp_amd/csrc/<file-stem>.hip with the extra flags and links it with the other,
# already built objects of gluon-nlp_amd/build into tools/build/var/lib_<name>.so (git-ignored,
# travels to the GPU box); select it there with GNLP_LIB_PATH (tools/ab_lib.sh).
set -e
name=$1; stem=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gluon-nlp_amd/csrc; bld=$root/gluon-nlp_amd/build; var=$root/tools/build/var
mkdir -p "$var"
extra=""
[ "$stem" == attn_bf16 ] && extra="-fno-honor-nans -fno-signed-zeros"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" \
  -c "$src/$stem.hip" -o "$var/${name}_$stem.o"
objs=""
for o in "$bld"/*.o; do
  [ "$(basename "$o")" == "$stem.o" ] || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$var/lib_$name.so" $objs "$var/${name}_$stem.o"
echo "$var/lib_$name.so"
