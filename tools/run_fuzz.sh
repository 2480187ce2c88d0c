#!/bin/bash
# Sanitizer (ASan + UBSan, CPU only) fuzzing of the host-side code that faces untrusted input.
#   tools/run_fuzz.sh ITERS seed1 [seed2 ...]     mutation-fuzz the file parsers (PNG / JPEG / BMP / GIF / WebP) from seed files
#   tools/run_fuzz.sh compile ITERS               random + hostile op lists through the op-list compiler, invariants checked
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE"); C=$ROOT/imagestitching_amd/csrc
OUT=${IST_FUZZ_BIN:-/tmp/ist_fuzz}
FLAGS="-std=c++17 -O1 -g -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -ffp-contract=off -DIST_FUZZ_NO_CRC -I$ROOT/include -I$C"
export ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1
if [ "$1" = compile ]; then
  shift
  g++ $FLAGS "$HERE/fuzz_compile.cpp" "$C/ist_compile.cpp" "$C/ist_plan.cpp" -o "$OUT"
else
  # (IST_FUZZ_REUSE=1: keep a binary that is already there - the tests build the harness once per session)
  if [ -z "$IST_FUZZ_REUSE" ] || [ ! -x "$OUT" ]; then
    g++ $FLAGS "$HERE/fuzz_decoders.cpp" "$C/ist_png_decode.cpp" "$C/ist_image_misc.cpp" "$C/ist_jpeg.cpp" "$C/ist_webp.cpp" "$C/ist_webp_vp8.cpp" "$C/ist_plan.cpp" -lz -o "$OUT"
  fi
fi
"$OUT" "$@"
