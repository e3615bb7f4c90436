#!/bin/bash
# Experimental: compile extra variants of a kernel source under suffixed symbols into libvt355_exp.so so they can be
# A/B-timed against the shipped build in one process (tools/kbench_variants.py).  Not part of build().
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../videotuna-dev_amd/csrc" && pwd)"
OUT="$HERE/../libvt355_exp.so"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics"
mkdir -p "$HERE/obj_exp"; rm -f "$HERE"/obj_exp/*.o
hipcc $F -DVT_SUFFIX=_shift -DVT_DQSHIFT=1 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_shift.o" &
for a in 1 2; do hipcc $F -DVT_SUFFIX=_abl$a -DVT_ABL=$a -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_abl$a.o" & done
hipcc $F -DVT_SUFFIX=_nodma -DVT_GEMM_DMA=0 -c "$HERE/gemm_bf16.hip" -o "$HERE/obj_exp/gemm_nodma.o" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj_exp/*.o
echo "built $OUT"
