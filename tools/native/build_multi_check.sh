#!/bin/bash
# Builds tools/native/multi_check (RCCL called directly over the C-ABI; see the file's header).
set -e
cd "$(dirname "$0")"
ROOT=../..
[ -f $ROOT/heavydb_amd/lib/libmi355q.so ] || (cd $ROOT && python -m heavydb_amd._build)
/opt/rocm/bin/hipcc -O2 -std=c++17 -Wall -Wno-unused-function -I$ROOT/include multi_check.cpp \
  -L$ROOT/heavydb_amd/lib -lmi355q -L/opt/rocm/lib -lrccl -lpthread \
  -Wl,-rpath,'$ORIGIN/../../heavydb_amd/lib' -Wl,-rpath,/opt/rocm/lib -o multi_check
echo built tools/native/multi_check
