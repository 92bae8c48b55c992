#!/bin/bash
# The CPU-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY 5 tooling):
#   oracle/pgx_oracle.c + bk_maxflow.c + progx_replay.c (the checkers of every parity test) and tests/emu/mf_emu.cpp (the max-flow bodies + host driver of
#   maxflow_body.hip.h / maxflow_driver.inl compiled for the host), driven by their own test files (+ tests/test_rng.py: the generator and the samplers of the oracle),
#   and csrc/sampler_host.hip (the one piece of host code in libpgx.so with its own data structures) under tests/emu/pnapsac_driver.cpp.
# usage: bash scripts/sanitize.sh        (from the repo root; ~2 min)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -C oracle asan
mkdir -p tests/emu/_san
g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -shared -fPIC \
    -o tests/emu/_san/libmf_emu.so tests/emu/mf_emu.cpp
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
# python itself is not instrumented: leaks of the interpreter are not ours (detect_leaks=0); everything else aborts on the first report
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export PGX_ORACLE_SO="$ROOT/oracle/_san/libpgx_oracle.so" PGX_EMU_SO="$ROOT/tests/emu/_san/libmf_emu.so"
# (round 5: + the Boykov-Kolmogorov solver and the control-flow replay, oracle/bk_maxflow.c / progx_replay.c, under their tests)
LD_PRELOAD="$ASAN:$UBSAN" python -m pytest tests/test_oracle.py tests/test_emu.py tests/test_rng.py tests/test_oracle_bk.py tests/test_replay.py -x -q -m "not gpu" "$@"
# csrc/sampler_host.hip (Progressive NAPSAC: host code of libpgx.so) compiled host-only with the same sanitizers and driven on
# 200 random problems (duplicates, points outside the image, subset sizes 0 / m .. n): must come back clean, leaks included
mkdir -p build/san
/opt/rocm/lib/llvm/bin/clang++ -x hip --cuda-host-only --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -std=c++17 -I/opt/rocm/include progressive-x_amd/csrc/sampler_host.hip -x c++ tests/emu/pnapsac_driver.cpp -o build/san/pnapsac_san \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 build/san/pnapsac_san
