// TEST INFRASTRUCTURE (tests/hostsim): the device side of the stand-in.  A kernel is an ordinary function; a launch
// runs it once per (block, thread), one block at a time (hip_host.cpp), so that __syncthreads(), LDS (`__shared__`
// = one static object, shared by the threads of the block that is running), wave shuffles / ballots and the atomics
// behave as the kernels expect.  Blocks never run concurrently: kernels that make workgroups wait for each other
// cannot be simulated this way (the fast families are not compiled here, see kernels_host.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>

#include "hip_runtime_api.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0

namespace hipsim {
struct Idx {
  unsigned x = 0, y = 0, z = 0;
};
extern thread_local Idx t_thread, t_block, t_block_dim, t_grid_dim;
void launch(const char* kernel_name, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void sync_threads();
uint64_t wave_exchange(uint64_t v, int src_lane_delta);   // value of lane (lane + delta), own value past the end
unsigned long long wave_ballot(bool pred);
uint64_t wave_exchange_abs(uint64_t v, int src_lane);     // value of lane src_lane (own value when out of range)
uint64_t wave_permute_push(int dst_lane, uint64_t v);     // ds_permute: lane dst_lane receives v; 0 where nobody wrote
void wave_sync();                                         // all live lanes of the wave (lockstep points of the device code)
void fiber_yield();                                       // lets the other fibers of the block run (s_sleep in a polling loop)
int lane_id();
void* dynamic_shared();
}  // namespace hipsim

#define threadIdx hipsim::t_thread
#define blockIdx hipsim::t_block
#define blockDim hipsim::t_block_dim
#define gridDim hipsim::t_grid_dim
constexpr int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipsim::launch(#kernel, (grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hipsim::sync_threads(); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __builtin_amdgcn_s_waitcnt(int) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
  (void)width;
  uint64_t bits = 0;
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  std::memcpy(&bits, &v, sizeof(T));
  bits = hipsim::wave_exchange(bits, (int)delta);
  T out;
  std::memcpy(&out, &bits, sizeof(T));
  return out;
}
inline unsigned long long __ballot(int pred) { return hipsim::wave_ballot(pred != 0); }
inline int __any(int pred) { return hipsim::wave_ballot(pred != 0) != 0ull; }
template <typename T>
inline T __shfl(T v, int src_lane, int width = 64) {
  (void)width;
  uint64_t bits = 0;
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  std::memcpy(&bits, &v, sizeof(T));
  bits = hipsim::wave_exchange_abs(bits, src_lane);
  T out;
  std::memcpy(&out, &bits, sizeof(T));
  return out;
}
// polling loops of the device code sleep between two looks: here the other fibers of the block get to run
inline void __builtin_amdgcn_s_sleep(int) { hipsim::fiber_yield(); }
// a wave runs in lockstep on the device; where the code relies on it, the fibers of a wave meet
inline void __builtin_amdgcn_wave_barrier() { hipsim::wave_sync(); }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // (only ever applied to wave-uniform values)
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned v) {
  const int l = hipsim::lane_id();
  return v + (unsigned)__builtin_popcount(mask & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned v) {
  const int l = hipsim::lane_id();
  return v + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}
inline int __builtin_amdgcn_ds_permute(int addr, int val) {
  return (int)(uint32_t)hipsim::wave_permute_push((addr >> 2) & 63, (uint64_t)(uint32_t)val);
}
inline long long clock64() { return (long long)__builtin_ia32_rdtsc(); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

// atomics: the threads of a block really run concurrently
template <typename T>
inline T atomicCAS(T* p, T expect, T desired) {
  __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expect;
}
template <typename T, typename V>
inline T hipsim_fetch_add(T* p, V v) {
  if constexpr (std::is_floating_point<T>::value) {
    T old = *p, want;
    do {
      want = old + (T)v;
    } while (!__atomic_compare_exchange(p, &old, &want, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return old;
  } else {
    return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST);
  }
}
inline int atomicAdd(int* p, int v) { return hipsim_fetch_add(p, v); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return hipsim_fetch_add(p, v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return hipsim_fetch_add(p, v); }
inline float atomicAdd(float* p, float v) { return hipsim_fetch_add(p, v); }
inline double atomicAdd(double* p, double v) { return hipsim_fetch_add(p, v); }
inline unsigned atomicSub(unsigned* p, unsigned v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
inline int atomicSub(int* p, int v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <typename T>
inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <typename T>
inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T>
inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return old;
}
template <typename T>
inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return old;
}
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
