#!/bin/bash
# Experimental: compile variants of csrc/attn_bwd.hip under suffixed symbols into libvt355_exp.so (A/B timing in one process with
# tools/kbench_variants.py / tools/kbench_stamp.py).  Not part of build().   usage: build_exp.sh suffix="-Dflags ..." ...
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../videotuna-dev_amd/csrc" && pwd)"
OUT="$HERE/../libvt355_exp.so"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics"
mkdir -p "$HERE/obj_exp"; rm -f "$HERE"/obj_exp/*.o
for a in "$@"; do
  suf="${a%%=*}"; fl="${a#*=}"
  hipcc $F -DVT_SUFFIX=$suf $fl -c "$HERE/attn_bwd.hip" -o "$HERE/obj_exp/bwd$suf.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj_exp/*.o
echo "built $OUT"
