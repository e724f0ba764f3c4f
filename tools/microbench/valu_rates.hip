// valu_rates.hip — issue cost of the integer / fp64 VALU ops the hash + home-slot arithmetic is
// made of (tools only).  Each kernel runs a long chain of ONE op kind, 8 independent chains per
// lane so the chain latency does not limit issue; 16 waves per CU (4 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(1024) void k_ops(uint32_t seed, unsigned long long* sink) {
  uint32_t a[8];
  double d[8];
  unsigned long long w[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; d[i] = 1.0 + a[i] * 1e-9; w[i] = a[i] * 0x9E3779B97F4A7C15ull; }
  const uint32_t m = seed | 0x85ebca6bu;
  const double md = 1.0000001 + seed * 1e-12;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = a[i] * m;                                        // v_mul_lo_u32
      else if (MODE == 1) a[i] = __umulhi(a[i], m) + 1u;                      // v_mul_hi_u32 (+add)
      else if (MODE == 2) w[i] = w[i] * (unsigned long long)m + 1ull;         // 64x32 mul (v_mad_u64_u32 chain)
      else if (MODE == 3) d[i] = __builtin_fma(d[i], md, 1e-9);               // v_fma_f64
      else if (MODE == 4) a[i] = (a[i] << 13) | (a[i] >> 19);                 // v_alignbit_b32
      else if (MODE == 5) a[i] = a[i] ^ (a[i] >> 16);                         // v_lshrrev + v_xor
      else if (MODE == 6) d[i] = (double)(uint32_t)((uint32_t)d[i] + a[i]);   // v_cvt_u32_f64 + add + v_cvt_f64_u32
      else if (MODE == 7) d[i] = __builtin_floor(d[i] * md);                  // v_mul_f64 + v_floor_f64
      else if (MODE == 8) w[i] = __umul64hi(w[i], 0x9E3779B97F4A7C15ull + m); // 64x64 high
      else if (MODE == 9) a[i] = a[i] * 5u + 0xe6546b64u;                     // mul-add by constant
      else if (MODE == 10) a[i] = __umul24(a[i], m) + 1u;                    // v_mul_u32_u24
      else if (MODE == 11) a[i] = a[i] + m;                                   // v_add_u32 (baseline)
    }
  }
  unsigned long long acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + (unsigned long long)d[i] + w[i];
  if (acc == 0x1234567ull) atomicAdd(sink, 1ull);
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
  const char* names[] = {"v_mul_lo_u32", "v_mul_hi_u32 + add", "u64 x u32 + 1 (mad_u64_u32)", "v_fma_f64", "rotl (v_alignbit)",
                         "x ^ (x >> 16)", "cvt f64->u32, add, cvt u32->f64", "v_mul_f64 + v_floor_f64", "umul64hi 64x64", "x*5 + c",
                         "v_mul_u32_u24 + add", "v_add_u32"};
#define RUN(M) { float ms = time_ms([&] { hipLaunchKernelGGL(k_ops<M>, dim3(256), dim3(1024), 0, 0, 12345u, sink); }); \
    const double steps = (double)kIters * 8;  /* per lane */ \
    /* 4 waves per SIMD: cycles per wave-level step on one SIMD */ \
    printf("%-34s %8.3f ms  %6.2f cycles per wave-op per SIMD (2.4 GHz, 4 waves/SIMD)\n", names[M], ms, ms * 1e-3 * 2.4e9 / (steps * 4)); }
  RUN(11) RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
  return 0;
}
