#!/bin/bash
# Scan the gfx950 code objects of sionna_amd/csrc/build/*.o for the trap of DESIGN.md 4.0d: a v_cndmask_b32_e32 (lane mask =
# vcc, implicit) whose vcc was last written by a SCALAR instruction (s_cselect_b64 vcc / s_mov_b64 vcc ...): ~24 cycles of the
# SIMD's vector pipe on gfx950 against ~2.5 behind a vector comparison (profiles/r06w_valu_rate2.txt).  The compiler produces it
# when a wave-uniform condition selects between vector registers (uniformly indexed register arrays, `x = uniform ? a : x`).
# usage: bash tools/scan_scalar_vcc.sh [min_count]   (runs on the build machine, no GPU; prints kernels with >= min_count hits)
MIN=${1:-1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
LLVM=/opt/rocm/lib/llvm/bin
for o in $ROOT/sionna_amd/csrc/build/*.o; do
  b=$(basename $o .o)
  objcopy -O binary --only-section=.hip_fatbin $o $TMP/$b.fat 2>/dev/null || continue
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$TMP/$b.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/$b.co 2>/dev/null || continue
  [ -s $TMP/$b.co ] || continue
  $LLVM/llvm-objdump -d $TMP/$b.co 2>/dev/null | awk -v name=$b -v min=$MIN '
    /^[0-9a-f]+ <.*>:/ {fn=$2}
    /\ts_[a-z0-9_]+ vcc/ {sc=1}
    /\tv_cmp[a-z0-9_]* vcc|\tv_cmp[a-z0-9_]*_e32 |\ts_cbranch_vcc/ {sc=0}
    /\tv_(add|sub|subrev)c?_co_u32_e32 / {sc=0}
    /\tv_cndmask_b32_e32 / {if (sc) f[fn]++}
    END {for (k in f) if (f[k] >= min) printf "%5d  %s  %s\n", f[k], name, k}'
done | c++filt | sort -rn
rm -rf $TMP
