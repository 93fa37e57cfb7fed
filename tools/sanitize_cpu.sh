#!/bin/bash
# The CPU suite against sanitizer builds of the library's HOST half (the device code is compiled as usual: gfx950 without xnack takes no
# sanitizer).  ASAN then UBSAN; reports go to /tmp/bzk_asan.* / /tmp/bzk_ubsan.* - no file = nothing found.  ~15 minutes on 8 cores.
#   tools/sanitize_cpu.sh            asan + ubsan        tools/sanitize_cpu.sh asan | ubsan | tsan   (tsan: the threaded host tests only)
# Round 6 outcome (with mg_exchange.h's process-to-process exchange, the staging entries and the deferred-program changes in the suite): ASAN 294 tests, UBSAN 294 tests,
# TSAN 51 tests: no reports.
# Round 5 outcome (with the deferred-witness paths in the suite): ASAN 268 tests, UBSAN 268 tests, TSAN 49 tests: no reports.
# Round 4 outcome: ASAN clean (162 tests); UBSAN one finding, fixed (memcpy from a null pointer with length 0 in sha3_256 of the empty
# message); TSAN on the threaded host tests (witness generator workers, worker pipeline, device-group CPU paths): no reports.  The native worker was checked the same way (g++ -fsanitize=address,undefined; tests/test_worker_native_cpu.py with BZK_WORKER_BIN).
set -e
cd "$(dirname "$0")/.."
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
want=${1:-both}
TESTS="tests/"
run() {  # name, compile flags, runtime library, options variable
  tools/build_variant.sh $1 $2 -fno-omit-frame-pointer -g1 -shared-libsan 2>&1 | grep -E "^built|error" || true
  rm -f /tmp/bzk_$1.*
  env BZK_LIBBZK=$PWD/bazuka_amd/libbzk.so.$1 LD_PRELOAD=$RT/$3 $4 python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider | tail -3
  ls /tmp/bzk_$1.* 2>/dev/null && grep -h "ERROR\|runtime error" /tmp/bzk_$1.* | sort | uniq -c || echo "$1: no reports"
  rm -rf bazuka_amd/libbzk.so.$1 bazuka_amd/csrc/_obj_$1
}
if [ $want = both ] || [ $want = asan ]; then run asan "-fsanitize=address" libclang_rt.asan-x86_64.so "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:log_path=/tmp/bzk_asan"; fi
if [ $want = both ] || [ $want = ubsan ]; then run ubsan "-fsanitize=undefined -fno-sanitize=vptr,function" libclang_rt.ubsan_standalone-x86_64.so "UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/bzk_ubsan"; fi
if [ $want = tsan ]; then TESTS="tests/test_host_mpn_cpu.py tests/test_defer_cpu.py tests/test_worker_pipeline_cpu.py tests/test_mg_cpu.py tests/test_pycircuit_cpu.py -k not(production)"; run tsan "-fsanitize=thread" libclang_rt.tsan-x86_64.so "TSAN_OPTIONS=log_path=/tmp/bzk_tsan:report_signal_unsafe=0"; fi
