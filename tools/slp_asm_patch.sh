#!/bin/bash
# Round-4 experiment behind DESIGN.md's "Toolchain note": which instruction of the SLP build makes the BatchNorm sums of the
# convolution epilogues differ from run to run?  Compiles csrc/conv_igemm.hip WITH the SLP vectoriser, keeps hipcc's
# intermediate device assembly, rewrites every `v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]` in it one way per variant,
# re-runs the tail of hipcc's own pipeline (cc1as, lld, clang-offload-bundler, host compile) on the patched assembly and links
# csrc/build_slp_<variant>/libgdl_hip.so from it + the other objects of csrc/build_slp (make BUILD=build_slp CXXFLAGS=... first;
# that make stops at check_isa, after the objects exist).
#   p0  untouched re-assembly (control)                         -> stats differ run to run (6 of 16 shapes)
#   p1  s_nop 7 in FRONT of every such instruction              -> still differ
#   p3  s_nop 7 BEHIND every such instruction                   -> still differ
#   p5  the v_pk_fma_f32 behind it replaced by two v_fma_f32    -> still differ
#   p2  the instruction replaced by two v_add_f32               -> bit-identical
#   p4  operands commuted: D, B, A op_sel:[1,0] op_sel_hi:[0,1] -> bit-identical
# Then on the GPU box: GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_slp_<v>/libgdl_hip.so python3 tools/determinism_check.py
# usage: tools/slp_asm_patch.sh p0|p1|p2|p3|p4|p5
set -e
V=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/iccv2025-gdl_amd/csrc
W=/tmp/slp_asm_$V
rm -rf $W && mkdir -p $W && cd $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
hipcc $FLAGS -c $C/conv_igemm.hip -o conv_igemm.o -save-temps 2>/dev/null
hipcc $FLAGS -c $C/conv_igemm.hip -o conv_igemm.o -save-temps -### 2>&1 | grep '^ "' | sed 's/^ //' > cmds.txt
S=conv_igemm-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$V" "$S" <<'PY'
import re, sys
v, path = sys.argv[1], sys.argv[2]
src = open(path).read().split("\n")
pat = re.compile(r"^\s*v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[1,0\]\s*$")
fma = re.compile(r"^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]\s*$")
out, n, prev = [], 0, False
for l in src:
    m = pat.match(l)
    if m:
        n += 1
        d0, d1, a0, a1, b0, b1 = map(int, m.groups())
        assert {d0, d1}.isdisjoint({b0, b1})
        if v == "p1":
            out += ["\ts_nop 7", l]
        elif v == "p2":
            out += [f"\tv_add_f32_e32 v{d0}, v{a0}, v{b1}", f"\tv_add_f32_e32 v{d1}, v{a1}, v{b0}"]
        elif v == "p3":
            out += [l, "\ts_nop 7"]
        elif v == "p4":
            out += [f"\tv_pk_add_f32 v[{d0}:{d1}], v[{b0}:{b1}], v[{a0}:{a1}] op_sel:[1,0] op_sel_hi:[0,1]"]
        else:
            out += [l]
        prev = True
        continue
    f = fma.match(l)
    if v == "p5" and prev and f:
        d0, d1, x0, x1, y0, y1, c0, c1 = map(int, f.groups())
        out += [f"\tv_fma_f32 v{d0}, v{x0}, v{y0}, v{c0}", f"\tv_fma_f32 v{d1}, v{x1}, v{y1}, v{c1}"]
    else:
        out.append(l)
    prev = False
open(path, "w").write("\n".join(out))
print(n, "cross-half v_pk_add_f32 rewritten as", v)
PY
for i in 4 5 6 8 9 10; do eval "$(sed -n "${i}p" cmds.txt)" 2>&1 | grep -v warning || true; done
O=$C/build_slp_$V
mkdir -p $O && cp $C/build_slp/*.o $O/ && cp conv_igemm.o $O/conv_igemm.o
hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -ldl -o $O/libgdl_hip.so && ls -la $O/libgdl_hip.so
