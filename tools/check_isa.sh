#!/bin/bash
# ISA screen of a built library: no kernel may contain the packed-f32 add with the halves of its SECOND source crossed,
#     v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]
# That form is what clang's SLP vectoriser makes of `ssum[e] += f[e]` next to `ssq[e] += f[e] * f[e]` in the convolution
# epilogues, and on MI355X it gives run-to-run different BatchNorm sums at the full-size visual shapes (round 4, DESIGN.md
# "Toolchain note": re-assembling the compiler's own output with that one instruction replaced by two v_add_f32 -- or by the same
# packed add with the operands commuted, op_sel:[1,0] op_sel_hi:[0,1] -- is bit-stable; wait states in front of / behind it are not).
# The library is built with -fno-slp-vectorize, so the count must be 0; a toolchain or flag change that brings it back fails here.
# usage: tools/check_isa.sh <build dir with the .o files> [number of .hip objects expected]     (exit 1 if the form is present)
set -e
B=${1:-iccv2025-gdl_amd/csrc/build}
ROCM=${ROCM_PATH:-$(hipconfig --rocmpath 2>/dev/null || echo /opt/rocm)}
LLVM=$ROCM/lib/llvm/bin
[ -x $LLVM/llvm-objdump ] || { echo "check_isa: no llvm-objdump under $LLVM (set ROCM_PATH)"; exit 1; }
T=$(mktemp -d)
bad=0; pk=0; seen=0; want=0
for o in "$B"/*.o; do
  # (objects built from .hip sources carry a fat binary; the plain C++ ones -- errors / api / encoder / prof / comm -- may hold none)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin "$o" 2>/dev/null || continue
  [ -s $T/fb.bin ] || continue
  want=$((want + 1))
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb.bin --output=$T/dev.co --unbundle 2>/dev/null || continue
  [ -s $T/dev.co ] || continue
  $LLVM/llvm-objdump -d $T/dev.co > $T/dev.s || continue
  grep -q 's_endpgm' $T/dev.s || continue
  seen=$((seen + 1))
  n=$(grep -c 'v_pk_add_f32.*op_sel:\[0,1\] op_sel_hi:\[1,0\]' $T/dev.s || true)
  p=$(grep -c 'v_pk_\(add\|mul\|fma\)_f32' $T/dev.s || true)
  [ "$n" != "0" ] && echo "$(basename $o): $n cross-half v_pk_add_f32"
  bad=$((bad + n)); pk=$((pk + p))
done
rm -rf $T
# Non-vacuous gate: every object that carries a fat binary must have been disassembled (seen == want), and -- when the caller says
# how many .hip sources the library has (the Makefile does) -- at least that many.  The build directory may be anywhere
# (`make BUILD=/tmp/x`): nothing here looks at the directory's neighbours.
expect=${2:-1}
if [ "$seen" = "0" ] || [ "$seen" != "$want" ] || [ "$seen" -lt "$expect" ]; then
  echo "check_isa: disassembled $seen device objects of $want with a fat binary (expected at least $expect) in $B -- the gate would be vacuous: failing"
  exit 1
fi
echo "check_isa: $bad cross-half v_pk_add_f32 (second source), $pk packed-f32 VALU instructions in $B"
[ "$bad" = "0" ]
