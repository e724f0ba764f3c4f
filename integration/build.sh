#!/bin/bash
# Builds integration/glue_check: the HeavyDB binding (Mi355qExecutor.cpp) against the mock HeavyDB headers, the
# native driver, linked with libmi355q.so (the product) and liboracle.so (the checker).  hipcc only for the HIP
# runtime API the driver uses (hipMalloc / hipMemcpy); there is no device code here.
set -e
cd "$(dirname "$0")"
ROOT=..
[ -f $ROOT/heavydb_amd/lib/libmi355q.so ] || (cd $ROOT && python -m heavydb_amd._build)
[ -f $ROOT/oracle/liboracle.so ] || make -C $ROOT/oracle
/opt/rocm/bin/hipcc -O2 -std=c++17 -Wall -Wno-unused-function -DMI355Q_GLUE_MOCK_HEADERS -I. -I$ROOT/include \
  Mi355qExecutor.cpp glue_check.cpp mock/heavydb_mock.cpp \
  -L$ROOT/heavydb_amd/lib -lmi355q -L$ROOT/oracle -loracle \
  -Wl,-rpath,'$ORIGIN/../heavydb_amd/lib' -Wl,-rpath,'$ORIGIN/../oracle' -o glue_check
echo built integration/glue_check
