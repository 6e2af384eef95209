#!/bin/bash
# Run a pytest selection with the HOST code of libxrl_amd under AddressSanitizer + UBSan (make -C pecos_amd/csrc asan).
#   scripts/asan_tests.sh -m "not gpu" tests
# CPU suite only: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and aborts in a process that also runs HIP on a
# stock (non-xnack, non-ASan) ROCm stack -- tried on the GPU box, the first HIP allocation fails.
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/pecos_amd/csrc asan >/dev/null || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $R
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1 \
  PECOS_XRL_AMD_SO=$R/pecos_amd/lib/libxrl_amd_asan.so python -m pytest -x -q -s "$@" 2>&1 | tee /tmp/asan_tests.log | grep -v "^$" | tail -5
if grep -q "runtime error\|AddressSanitizer" /tmp/asan_tests.log; then echo "SANITIZER FINDINGS:"; grep -n "runtime error\|AddressSanitizer\|SUMMARY" /tmp/asan_tests.log | head -20; exit 1; fi
echo "sanitizers: clean"
