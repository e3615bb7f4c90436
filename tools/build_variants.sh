#!/bin/bash
# Experimental: compile extra variants of a kernel source under suffixed symbols into libvt355_exp.so so they can be
# A/B-timed against the shipped build in one process (tools/kbench_variants.py).  Not part of build().
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../videotuna-dev_amd/csrc" && pwd)"
OUT="$HERE/../libvt355_exp.so"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics"
mkdir -p "$HERE/obj_exp"; rm -f "$HERE"/obj_exp/*.o
# dQ hand-off chains: ring depth / consumer hysteresis / cache policy of the tile exchange
hipcc $F -DVT_SUFFIX=_nolink -DVT_CHAIN=0 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_nolink.o" &
for v in "4 1 17 17" "6 2 17 17" "8 0 17 17" "8 2 17 17" "8 2 0 16" "4 1 0 16" "8 2 0 17"; do set -- $v
  hipcc $F -DVT_SUFFIX=_r$1h$2s$3l$4 -DCH_R=$1 -DCH_HYST=$2 -DCH_ST_AUX=$3 -DCH_LD_AUX=$4 -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd_r$1h$2s$3l$4.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj_exp/*.o
echo "built $OUT"
