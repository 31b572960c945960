#!/bin/bash
# Runs the CPU tests that exercise the oracle with the oracle built under AddressSanitizer +
# UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available on the pool).
# The instrumented libraries replace oracle/libgg_oracle_*.so for the run and the originals are restored.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
trap 'cp "$tmp"/bak/*.so oracle/ 2>/dev/null; rm -rf "$tmp"' EXIT
mkdir -p "$tmp/bak" "$tmp/san"
make -C oracle >/dev/null
cp oracle/libgg_oracle_*.so "$tmp/bak/"
CF="-O1 -g -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math -march=x86-64-v3 -fsanitize=address,undefined -fno-omit-frame-pointer"
COMPAT="-DGG_VJP_GSPLAT_COMPAT=1 -DGG_ALPHA_MAX_BWD=0.99f"
gcc $CF -o "$tmp/san/libgg_oracle_f32.so" oracle/gg_oracle.c -lm
gcc $CF -DGGO_F64 -o "$tmp/san/libgg_oracle_f64.so" oracle/gg_oracle.c -lm
gcc $CF $COMPAT -o "$tmp/san/libgg_oracle_compat_f32.so" oracle/gg_oracle.c -lm
gcc $CF $COMPAT -DGGO_F64 -o "$tmp/san/libgg_oracle_compat_f64.so" oracle/gg_oracle.c -lm
cp "$tmp"/san/*.so oracle/
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" OMP_NUM_THREADS=4 \
python -m pytest tests/test_oracle_analytic.py tests/test_oracle_grad.py tests/test_densify_adam.py \
    tests/test_mlp_losses.py tests/test_compat_variant.py tests/test_plugin.py -q -m "not gpu" -p no:cacheprovider
