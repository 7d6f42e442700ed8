#!/bin/bash
# Build a variant of the library into tools/ab/lib<name>.so with extra compiler flags (diagnostic / A-B builds only):
#   tools/ab/build_variant.sh stamps -DLOCO_GEMM_STAMPS
set -e
name=$1; shift
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../../loco-asr_amd/csrc
out=$here/_build_$name
mkdir -p $out
flags="$(sed -n 's/^CXXFLAGS ?= //p' $src/Makefile | sed 's/$(ARCH)/gfx950/') $(sed -n 's/^override CXXFLAGS += //p' $src/Makefile)"
for f in gemm_f32 gemm_f16x3 attention_f32 attention_f16x3 conv0_gn_gelu pos_conv norm_misc intent_head resample flac_decode loco_api; do
  hipcc $flags "$@" -c $src/$f.hip -o $out/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $here/lib$name.so $out/*.o
echo built $here/lib$name.so
