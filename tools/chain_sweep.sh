#!/bin/bash
# attention-backward chain-length sweep (one process per setting; VT_BWD_CHAIN is read once per process)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for L in "$@"; do
  VT_BWD_CHAIN=$L timeout -k 10 120 python tools/kbench.py attn > gpurun_out/chain_L$L.log 2>&1
  echo "L=$L: $(grep -h 'attn bwd' gpurun_out/chain_L$L.log) | $(grep -h 'chain ws' gpurun_out/chain_L$L.log)"
done
