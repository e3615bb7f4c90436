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
hipcc $F -DVT_SUFFIX=_stat -DVT_STATMFMA=1 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_stat.o" &
for st in max-ilp max-memory-clause iterative-ilp iterative-minreg; do
  hipcc $F -mllvm -amdgpu-sched-strategy=$st -DVT_SUFFIX=_$(echo $st | tr - _) -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_$st.o" &
done
hipcc $F -DVT_SUFFIX=_abl2 -DVT_ABL=2 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_abl2.o" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj_exp/*.o
echo "built $OUT"
