#!/bin/bash
# A/B and ablation builds of the library: tools/build_variant.sh NAME "-DTSC_ABL_NODMA ..." -> exp/libtsc_NAME.so
# (select it with TSC_HIP_LIBRARY=exp/libtsc_NAME.so; exp/ is not tracked)
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/transcoder-sae-circuits_amd/csrc; OUT=$ROOT/exp/build_$NAME
mkdir -p $OUT
for f in capi encode_f32 encode_operands topk_merge sparse_ops screen_topk; do
  if [ $f = encode_f32 ] || [ ! -f $SRC/build/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-function $FLAGS -c $SRC/$f.hip -o $OUT/$f.o
  else cp $SRC/build/$f.o $OUT/$f.o; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/*.o -o $ROOT/exp/libtsc_$NAME.so
echo built exp/libtsc_$NAME.so
