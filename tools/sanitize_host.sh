#!/bin/bash
# Host code of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no GPU needed).
#
# Builds the product's HOST sources (api.cpp, the general path's orchestration, BGZF host side, pipeline.cpp — with them the host+device
# scalar cores canon_core.h, inflate_core.h, deflate_core.h, methylation_core.h, bamrec.h) with clang -fsanitize=address,undefined into
# tests/hostemu/_build/libfgumi_host_san.so — the device launchers it references are stubbed to abort() — and runs the CPU tests that
# exercise host code through the C ABI against it (FGX_LIB), plus the host emulation of the general path (tests/hostemu) built the same way.
# Any out-of-bounds access, use after free, misaligned / overflowing arithmetic in those sources fails the run.
#   tools/sanitize_host.sh [pytest args...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/fgumi_amd/csrc"
OUT="$ROOT/tests/hostemu/_build"
CL=/opt/rocm/lib/llvm/bin/clang++
mkdir -p "$OUT"
SAN="-O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -DFGX_HAVE_CODEC -I/opt/rocm/include -w -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan -fno-omit-frame-pointer"
LINK="-L/opt/rocm/lib -lamdhip64 -lz -pthread -Wl,-rpath,/opt/rocm/lib"
# 1. first link leaves the device launchers undefined; list them and define each as a jump to abort()
$CL $SAN "$CS"/{api,simplex_host,duplex_host,codec_host,bgzf_host,pipeline}.cpp -o "$OUT/libhost_nostub.so" $LINK
{
  echo '.text'
  ldd -r "$OUT/libhost_nostub.so" 2>&1 | sed -n 's/^undefined symbol: \(_ZN3fgx[^ \t]*\).*/\1/p' | sort -u | while read -r s; do
    case "$s" in *release*) body="ret" ;; *) body="jmp abort@PLT" ;; esac      # (buffer releases are called from fgx_destroy: nothing to free here)
    printf '.globl %s\n.type %s,@function\n%s:\n  %s\n' "$s" "$s" "$s" "$body"
  done
} > "$OUT/device_stubs.S"
$CL $SAN "$CS"/{api,simplex_host,duplex_host,codec_host,bgzf_host,pipeline}.cpp "$OUT/device_stubs.S" -o "$OUT/libfgumi_host_san.so" $LINK
rm -f "$OUT/libhost_nostub.so"
RT="$($CL -print-file-name=libclang_rt.asan-x86_64.so)"
export FGX_LIB="$OUT/libfgumi_host_san.so" HOSTEMU_SANITIZE=1
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
cd "$ROOT"
TESTS="tests/test_canon_core.py tests/test_canon_codec.py tests/test_reject_core.py tests/test_devemu.py tests/test_apiemu.py tests/test_inflate_core.py tests/test_deflate_core.py tests/test_methylation_core.py tests/test_bgzf.py tests/test_general_path_hostemu.py tests/test_general_path_fuzz.py"
LD_PRELOAD="$RT" python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider "$@"
rm -f "$OUT/libfgumi_host_san.so" "$OUT/libhostemu_san.so" "$OUT/libdevemu_san.so" "$OUT/libapiemu_san.so"     # (large; they would travel to the GPU box with the snapshot)
# ThreadSanitizer over the threaded host paths (the general path sharded over helper callers, the host threads of the canonical pass) — opt-in, slow:
#   HOSTEMU_SANITIZE=thread LD_PRELOAD="$($CL -print-file-name=libclang_rt.tsan-x86_64.so)" TSAN_OPTIONS=report_signal_unsafe=0 \
#     python -m pytest tests/test_apiemu.py -q -k "canonical_second_pass or sharded or hostile or rejects_side" -p no:cacheprovider
