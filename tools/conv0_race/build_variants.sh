#!/bin/bash
# Variants of the old (packed-tap) conv0 kernel, each changing ONE of the things the round-1 rewrite changed together, linked
# into today's library: which single change makes the two-stream miscompare disappear?  (needs _old/ from build.sh)
#   v1  the inline-asm hi/lo split (v_fma_mix*) replaced by plain casts                  -- packed taps, two frames per trip kept
#   v2  the ten taps as scalar FMAs                                                      -- asm split, two frames per trip kept
#   v3  one frame per trip (the second pair of the asm split is fed the first pair again) -- packed taps, asm split kept
#   v4  the unrolled tap loop kept from interleaving LDS reads with the packed FMAs: all ten taps of both frames are read into
#       registers first, then s_waitcnt lgkmcnt(0) (asm), then the arithmetic
set -e
cd "$(dirname "$0")"
CS=../../loco-asr_amd/csrc
OBJS=$(for f in gemm_f32 gemm_f16x3 attention_f32 attention_f16x3 pos_conv norm_misc intent_head resample loco_api; do echo $CS/build/$f.o; done)
python3 - <<'PY'
src = open("_old/conv0_old_for_current.hip").read()
def put(name, s):
    open(f"_old/conv0_{name}.hip", "w").write(s)
# v1: no asm split
v1 = src.replace("""            unsigned ha, la, hb, lb;
            split_f16_2pairs(ga.x, ga.y, gb.x, gb.y, ha, la, hb, lb);""", """            unsigned ha, la, hb, lb;
            {
                float a0 = ga.x, a1 = ga.y, b0 = gb.x, b1 = gb.y;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                h2_t h_a = {(_Float16)a0, (_Float16)a1}, h_b = {(_Float16)b0, (_Float16)b1};
                h2_t l_a = {(_Float16)(a0 - (float)h_a[0]), (_Float16)(a1 - (float)h_a[1])}, l_b = {(_Float16)(b0 - (float)h_b[0]), (_Float16)(b1 - (float)h_b[1])};
                ha = __builtin_bit_cast(unsigned, h_a); la = __builtin_bit_cast(unsigned, l_a);
                hb = __builtin_bit_cast(unsigned, h_b); lb = __builtin_bit_cast(unsigned, l_b);
            }""")
assert v1 != src
put("v1", v1)
# v2: scalar taps
v2 = src.replace("""            ya = __builtin_elementwise_fma(w01[k], f32x2_t{xa, xa}, ya);
            yb = __builtin_elementwise_fma(w01[k], f32x2_t{xb, xb}, yb);""", """            ya.x = fmaf(w01[k].x, xa, ya.x); ya.y = fmaf(w01[k].y, xa, ya.y);
            yb.x = fmaf(w01[k].x, xb, yb.x); yb.y = fmaf(w01[k].y, xb, yb.y);
            asm volatile("" : "+v"(ya.x), "+v"(ya.y), "+v"(yb.x), "+v"(yb.y));""")
assert v2 != src
put("v2", v2)
# v3: one frame per trip
v3 = src.replace("for (int t = 0; t < nt; t += 2) {", "for (int t = 0; t < nt; t += 1) {").replace("const int t1 = t + 1 < nt ? t + 1 : t;", "const int t1 = t;")
assert v3 != src
put("v3", v3)
# v4: all LDS reads first, explicit wait, then arithmetic
v4 = src.replace("""#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float xa = xs[5 * t + k], xb = xs[5 * t1 + k];
            ya = __builtin_elementwise_fma(w01[k], f32x2_t{xa, xa}, ya);
            yb = __builtin_elementwise_fma(w01[k], f32x2_t{xb, xb}, yb);
        }""", """        float xa_[10], xb_[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) { xa_[k] = xs[5 * t + k]; xb_[k] = xs[5 * t1 + k]; }
#pragma unroll
        for (int k = 0; k < 10; ++k) asm volatile("" : "+v"(xa_[k]), "+v"(xb_[k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            ya = __builtin_elementwise_fma(w01[k], f32x2_t{xa_[k], xa_[k]}, ya);
            yb = __builtin_elementwise_fma(w01[k], f32x2_t{xb_[k], xb_[k]}, yb);
        }""")
assert v4 != src
put("v4", v4)
PY
for v in v1 v2 v3 v4; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c _old/conv0_$v.hip -o _old/conv0_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o libloco_oldconv0_$v.so $OBJS _old/conv0_$v.o
done
ls -la libloco_oldconv0_v*.so
