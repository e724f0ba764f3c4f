// emu_atomics.h — TEST-ONLY: single-threaded stand-ins for the slot-update macro set of heavydb_amd/csrc/rowfunc.h, so that
// tests/emu/emu.cpp can compile the product's row logic for the host with plain g++.  Included BEFORE rowfunc.h
// (MQ_SLOT_ATOMICS tells that header not to define the device versions).  Never part of libmi355q.so.
#pragma once
#include <cstdint>
#define MQ_SLOT_ATOMICS 1
namespace mq {
template <typename T>
inline T emu_cas(T* p, T expect, T desired) {
  T old = *p;
  if (old == expect) *p = desired;
  return old;
}
#define MQ_CAS64(p, e, d) mq::emu_cas<unsigned long long>((unsigned long long*)(p), (unsigned long long)(e), (unsigned long long)(d))
#define MQ_CAS32(p, e, d) mq::emu_cas<unsigned int>((unsigned int*)(p), (unsigned int)(e), (unsigned int)(d))
}  // namespace mq
#define MQ_ADD64(p, v) (*(unsigned long long*)(p) += (unsigned long long)(v))
#define MQ_ADD32(p, v) (*(unsigned int*)(p) += (unsigned int)(v))
#define MQ_ADDF64(p, v) (*(double*)(p) += (v))
#define MQ_ADDF32(p, v) (*(float*)(p) += (v))
#define MQ_MIN64(p, v) (*(long long*)(p) = (*(long long*)(p) < (long long)(v) ? *(long long*)(p) : (long long)(v)))
#define MQ_MAX64(p, v) (*(long long*)(p) = (*(long long*)(p) > (long long)(v) ? *(long long*)(p) : (long long)(v)))
#define MQ_LOAD64(p) (*(volatile int64_t*)(p))
#define MQ_STORE64(p, v) (*(volatile int64_t*)(p) = (v))
#define MQ_FENCE() ((void)0)
#define MQ_PUBLISH_ORDER() ((void)0)
#define MQ_LOAD32(p) (*(volatile int32_t*)(p))
#define MQ_STORE32(p, v) (*(volatile int32_t*)(p) = (v))
#define MQ_FN inline
