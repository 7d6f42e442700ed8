#!/bin/bash
# Builds tools/conv0_race/old_conv0.so: the packed-tap conv0 kernel of commit 491647a^ (the one that miscompared under
# concurrency in round 1), extracted from git history, behind one C entry point.  Run in the build container (needs .git);
# the .so travels to the GPU box with the snapshot, the extracted sources under _old/ do not need to.
set -e
cd "$(dirname "$0")"
mkdir -p _old/include
git -C ../.. show 491647a^:loco-asr_amd/csrc/conv0_gn_gelu.hip > _old/conv0_gn_gelu.hip
git -C ../.. show 491647a^:loco-asr_amd/csrc/loco_kernels.h > _old/loco_kernels.h
cat > _old/wrap.hip <<'EOC'
#include "conv0_gn_gelu.hip"
extern "C" int old_conv0_planes(const float* wav, int B, long L, const float* w, const float* gn_w, const float* gn_b, void* scratch,
                                void* hi, void* lo, void* stream) {
    return (int)loco::launch_conv0_gn_gelu(wav, B, L, w, gn_w, gn_b, nullptr, scratch, 1e-5f, (hipStream_t)stream, hi, lo);
}
extern "C" size_t old_conv0_scratch_bytes(int B) { return loco::conv0_scratch_bytes(B); }
EOC
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared _old/wrap.hip -o old_conv0.so
ls -la old_conv0.so
# libloco_oldconv0.so: the CURRENT library with only conv0_gn_gelu.hip taken from 491647a^ (compiled against the current header),
# for replaying round 1's two-stream scenario with per-layer outputs (probe_forward.py)
CS=../../loco-asr_amd/csrc
cp _old/conv0_gn_gelu.hip _old/conv0_old_for_current.hip
sed -i 's#"loco_kernels.h"#"../../../loco-asr_amd/csrc/loco_kernels.h"#' _old/conv0_old_for_current.hip
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c _old/conv0_old_for_current.hip -o _old/conv0_old.o
make -C $CS >/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o libloco_oldconv0.so $(for f in gemm_f32 gemm_f16x3 attention_f32 attention_f16x3 pos_conv norm_misc intent_head resample loco_api; do echo $CS/build/$f.o; done) _old/conv0_old.o
ls -la libloco_oldconv0.so
