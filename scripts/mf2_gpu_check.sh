#!/bin/bash
# correctness + timing of the tile-resident min-cut against the oracle and the level-synchronous path (run on the GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== fuzz (default: single-workgroup kernel up to 8192 sites)"; timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -k "expansion" 2>&1 | tail -5
echo "== soak, forced tiles of 1024"; PGX_TILE_SINGLE=0 PGX_TILE_SITES=1024 timeout 600 python tests/soak_expansion.py 7 150 2>&1 | tail -5
echo "== soak, forced tiles of 4096"; PGX_TILE_SINGLE=0 PGX_TILE_SITES=4096 timeout 600 python tests/soak_expansion.py 8 100 2>&1 | tail -5
echo "== labelling bench, tile path"; timeout 900 python scripts/bench_labelling.py C2 C3 2>&1 | grep config
timeout 900 python scripts/bench_labelling.py C5 C4 --no-oracle 2>&1 | grep config
echo "== tiles of 1024"; PGX_TILE_SITES=1024 timeout 900 python scripts/bench_labelling.py C3 C5 C4 --no-oracle 2>&1 | grep config
echo "== labelling bench, level-synchronous path"; PGX_MF_TILE=0 timeout 900 python scripts/bench_labelling.py C2 C3 C5 C4 --no-oracle 2>&1 | grep config
} > gpurun_out/mf2_check.log 2>&1
cut -c1-420 gpurun_out/mf2_check.log | sed -e 's/"cycles.*"mincuts"/"mincuts"/' | tail -40
