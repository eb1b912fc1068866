#!/bin/bash
# time the self-attention shapes with each A/B library of tools/ab_attn.sh
cd "$(dirname "$0")/.."
echo "== shipped"; timeout 120 python tools/bench_attn.py 2>&1 | grep "Tk=4096\|Tk=1024"
for lib in unirestore_amd/ab/libur_attn_*.so; do
  echo "== $lib"; UR_LIB=$PWD/$lib timeout 120 python tools/bench_attn.py 2>&1 | grep "Tk=4096\|Tk=1024"
done
