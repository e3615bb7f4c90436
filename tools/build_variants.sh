#!/bin/bash
# Experimental: compile extra variants of a kernel source under suffixed symbols into libvt355_exp.so so they can be
# A/B-timed against the shipped build in one process (tools/kbench_variants.py).  Not part of build().
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../videotuna-dev_amd/csrc" && pwd)"
OUT="$HERE/../libvt355_exp.so"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics"
mkdir -p "$HERE/obj_exp"; rm -f "$HERE"/obj_exp/*.o
# attention backward A/B variants (see csrc/exp/README.md for what was measured with them)
hipcc $F -DVT_SUFFIX=_nolink -DVT_CHAIN=0 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_nolink.o" &
hipcc $F -DVT_W8=0 -DVT_SUFFIX=_w4 -mllvm -amdgpu-sched-strategy=max-ilp -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_w4.o" &
hipcc $F -DVT_DQ16=1 -DVT_SUFFIX=_dq16 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_dq16.o" &
hipcc $F -DVT_SUFFIX=_same -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_same.o" &
hipcc $F -DVT_SUFFIX=_prio -DVT_DQPRIO=1 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_prio.o" &
hipcc $F -DVT_SUFFIX=_ant -DVT_ATOM_AUX=2 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_ant.o" &
hipcc $F -DVT_SUFFIX=_asc1 -DVT_ATOM_AUX=16 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_asc1.o" &
# timing-only ablations of the eight-wave body (wrong results): which resource the step is bound by
for a in 1 2 5 6 7; do
  hipcc $F -DVT_SUFFIX=_abl$a -DVT_ABL=$a -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_abl$a.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj_exp/*.o
echo "built $OUT"
