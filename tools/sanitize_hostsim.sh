#!/bin/bash
# The host simulation of the library (tests/hostsim: api*.cpp, plan.cpp and every kernel file compiled for the CPU) under
# AddressSanitizer: tests/helpers.py builds it into tests/_hostsim[_real]_address/ when MI355Q_HOSTSIM_SANITIZE is set.
# usage: tools/sanitize_hostsim.sh [pytest arguments]      (default: the flow, projection and expression-fuzz tests)
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ -f "$RT" ] || { echo "no ASan runtime next to the ROCm clang"; exit 3; }
[ $# -gt 0 ] || set -- tests/test_hostsim_flow.py tests/test_projection.py tests/test_expr_fuzz.py
MI355Q_HOSTSIM_SANITIZE=address LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
  python -m pytest -q -x -p no:cacheprovider "$@"
