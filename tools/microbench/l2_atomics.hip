// l2_atomics.hip — rate of no-return global atomics into a small per-workgroup array that should stay in the
// XCD's L2 (tools only).  Question: can phase 2 of the partitioned GROUP BY keep only
// tags + counts in LDS and accumulate the fp64 sums with global_atomic_add_f64 into an L2-resident array, so that
// one LDS table covers a whole partition (R = 1)?  That needs 5 x 10^9 atomics per step next to the 80 GB record
// stream: >= ~140 G atomics/s chip-wide just to match today's phase 2, >= 400 G/s to make R = 1 pay.
//
//   mode 0  global_atomic_add_f64 (no return)        mode 1  global_atomic_add_u32 (no return)
//   mode 2  global_atomic_add_u64 (no return)        mode 3  plain 8-byte gather from the same array (reference)
//   stream = 1: every atomic is paired with a 16-byte non-temporal load of a record stream (what phase 2 does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

typedef int v4i32_t __attribute__((ext_vector_type(4)));

// One 1024-lane workgroup per CU; workgroup b owns win_elems 8-byte words at tab + b * win_elems.
template <int MODE, int STREAM>
__global__ __launch_bounds__(1024) void k_atom(unsigned long long* __restrict__ tab, uint32_t win_elems, int iters,
                                               const v4i32_t* __restrict__ recs, size_t recs_per_wg,
                                               unsigned long long* out) {
  unsigned long long* my = tab + (size_t)blockIdx.x * win_elems;
  const v4i32_t* rs = recs + (size_t)blockIdx.x * recs_per_wg;
  unsigned long long s = mix((unsigned long long)blockIdx.x * 1315423911ull + threadIdx.x);
  long long acc = 0;
  for (int i = 0; i < iters; i += 4) {
    v4i32_t r[4];
    if (STREAM) {
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = __builtin_nontemporal_load(rs + (size_t)(i + u) * 1024 + threadIdx.x);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s = mix(s);
      uint32_t e = (uint32_t)(((unsigned __int128)s * win_elems) >> 64);
      if (STREAM) e = (e + (uint32_t)r[u].x) % win_elems;  // the address depends on the record, like a table lookup
      if (MODE == 0) atomicAdd((double*)my + e, 1.5);
      else if (MODE == 1) atomicAdd((unsigned int*)(my + e), 1u);
      else if (MODE == 2) atomicAdd(my + e, 1ull);
      else acc += (long long)__builtin_nontemporal_load(my + e);
    }
  }
  if (acc == 0x7fffffffffffffffll) atomicAdd(out, 1ull);
}

template <int MODE, int STREAM>
void run(const char* name, unsigned long long* tab, uint32_t win_elems, const v4i32_t* recs, unsigned long long* out) {
  const int iters = 2048;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_atom<MODE, STREAM>), dim3(256), dim3(1024), 0, 0, tab, win_elems, 64, recs, (size_t)iters * 1024, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_atom<MODE, STREAM>), dim3(256), dim3(1024), 0, 0, tab, win_elems, iters, recs, (size_t)iters * 1024, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = 256.0 * 1024 * iters;
  printf("%-28s window/WG %6.1f KB  stream %d  %8.3f ms  %7.1f G ops/s%s\n", name, win_elems * 8 / 1024.0, STREAM, ms,
         ops / ms / 1e6, STREAM ? "" : "");
  if (STREAM) printf("%-28s   (record stream alone would be %.2f TB/s at this time)\n", "", ops * 16 / ms / 1e9);
}

int main() {
  const uint32_t max_win = 16384;  // 128 KB per workgroup
  unsigned long long *tab, *out;
  v4i32_t* recs;
  const size_t n_recs = (size_t)256 * 2048 * 1024;  // 8 GB of 16-byte records
  CK(hipMalloc(&tab, (size_t)256 * max_win * 8));
  CK(hipMalloc(&out, 8));
  CK(hipMalloc(&recs, n_recs * 16));
  CK(hipMemset(tab, 0, (size_t)256 * max_win * 8));
  CK(hipMemset(out, 0, 8));
  CK(hipMemset(recs, 3, n_recs * 16));
  for (uint32_t win : {2048u, 5120u, 10240u, 16384u}) {
    run<3, 0>("gather 8 B (reference)", tab, win, recs, out);
    run<0, 0>("global_atomic_add_f64", tab, win, recs, out);
    run<1, 0>("global_atomic_add_u32", tab, win, recs, out);
    run<2, 0>("global_atomic_add_u64", tab, win, recs, out);
    run<0, 1>("global_atomic_add_f64", tab, win, recs, out);
    run<1, 1>("global_atomic_add_u32", tab, win, recs, out);
  }
  return 0;
}
