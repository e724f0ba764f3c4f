// lds_atomics.hip — per-CU throughput of LDS read-modify-write flavours at random addresses
// (tools only): decides how phase 2 of the partitioned GROUP BY updates its LDS table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kT = 1024;
constexpr int kE = 6144;     // entries (8-byte) -> 48 KB
constexpr int kIters = 2048;

__device__ inline uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(kT) void k_lds(unsigned long long* sink) {
  __shared__ unsigned long long tab[kE];
  for (int i = threadIdx.x; i < kE; i += kT) tab[i] = 0;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
  unsigned long long acc = 0;
  for (int it = 0; it < kIters; ++it) {
    const uint32_t e = rnd(s) % kE;
    if (MODE == 0) atomicAdd((unsigned int*)&tab[e], 1u);                       // ds_add_u32
    else if (MODE == 1) atomicAdd(&tab[e], 1ull);                               // ds_add_u64
    else if (MODE == 2) atomicAdd((double*)&tab[e], 1.5);                       // ds_add_f64
    else if (MODE == 3) acc += atomicCAS(&tab[e], 0ull, (unsigned long long)e); // ds_cmpst_rtn_b64
    else if (MODE == 4) acc += *(volatile unsigned long long*)&tab[e];          // ds_read_b64
    else if (MODE == 5) {                                                       // plain RMW f64 (racy)
      double v = *(volatile double*)&tab[e];
      *(volatile double*)&tab[e] = v + 1.5;
    } else if (MODE == 6) acc += atomicAdd((unsigned int*)&tab[e], 1u);         // ds_add_rtn_u32
    else if (MODE == 7) atomicAdd((float*)&tab[e], 1.5f);                       // ds_add_f32
    else if (MODE == 8) {                                                       // f64 add by CAS loop
      unsigned long long old = *(volatile unsigned long long*)&tab[e];
      for (;;) {
        const unsigned long long nv = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)old) + 1.5);
        const unsigned long long seen = atomicCAS(&tab[e], old, nv);
        if (seen == old) break;
        old = seen;
      }
    } else if (MODE == 9) atomicMax((long long*)&tab[e], (long long)e);         // ds_max_i64
  }
  if (acc == 0x123456789ull) atomicAdd(sink, 1ull);
}

template <typename F>
float time_ms(F&& f, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError()); return ms / reps;
}

int main() {
  unsigned long long* sink; CK(hipMalloc(&sink, 8)); CK(hipMemset(sink, 0, 8));
  const char* names[] = {"ds_add_u32", "ds_add_u64", "ds_add_f64", "ds_cmpst_rtn_b64", "ds_read_b64", "plain rmw f64",
                         "ds_add_rtn_u32", "ds_add_f32", "f64 add via CAS loop", "ds_max_i64"};
#define RUN(M) { float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<M>, dim3(256), dim3(kT), 0, 0, sink); }); \
    const double ops = 256.0 * kT * kIters; \
    printf("%-22s %8.3f ms  %8.1f lane-ops/us/CU  (%.2f cycles/lane-op/CU at 2.4 GHz)\n", names[M], ms, ops / 256 / ms / 1e3, 2400.0 * ms * 1e3 * 256 / ops); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  return 0;
}
