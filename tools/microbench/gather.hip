// gather.hip — random 8-byte gather rate by working-set size and placement (tools only).
// Question it answers (DESIGN section on join probes): how many random reads per second does the chip
// sustain when the table being probed is (a) HBM-sized, (b) Infinity-Cache-sized, (c) small enough for
// one XCD's L2, with every XCD gathering from its OWN window (what an XCD-cooperative probe of a
// key-range slice would do) or all from one shared window?  Also the LDS rate for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// every lane: `iters` independent random 8-byte reads, 8 in flight; window w of the XCD (block % 8) when per_xcd
__global__ __launch_bounds__(256) void k_gather(const long long* __restrict__ tab, unsigned long long win_elems,
                                                int per_xcd, int iters, unsigned long long* out) {
  const unsigned long long base = per_xcd ? (unsigned long long)(blockIdx.x & 7) * win_elems : 0ull;
  unsigned long long s = mix((unsigned long long)blockIdx.x * 1315423911ull + threadIdx.x);
  long long acc = 0;
  for (int i = 0; i < iters; i += 8) {
    unsigned long long idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = mix(s);
      idx[u] = base + (unsigned long long)(((unsigned __int128)s * win_elems) >> 64);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += tab[idx[u]];
  }
  if (acc == 0x7fffffffffffffffll) atomicAdd(out, 1ull);
}

__global__ __launch_bounds__(1024) void k_lds_gather(int iters, unsigned long long* out) {
  extern __shared__ long long s_tab[];
  const int n = 16 * 1024;  // 128 KB
  for (int i = threadIdx.x; i < n; i += blockDim.x) s_tab[i] = i * 7;
  __syncthreads();
  unsigned long long s = mix((unsigned long long)blockIdx.x * 1315423911ull + threadIdx.x);
  long long acc = 0;
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = mix(s);
      acc += s_tab[s & (n - 1)];
    }
  }
  if (acc == 0x7fffffffffffffffll) atomicAdd(out, 1ull);
}

int main() {
  const size_t total = (size_t)4 << 30;  // 4 GB table
  long long* tab;
  unsigned long long* out;
  CK(hipMalloc(&tab, total));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(tab, 1, total));
  CK(hipMemset(out, 0, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 2048;
  for (int bpc : {4, 8}) {
    const int grid = 256 * bpc;
    for (int per_xcd = 0; per_xcd < 2; ++per_xcd) {
      for (size_t win_bytes : {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)16 << 20,
                               (size_t)64 << 20, (size_t)400 << 20}) {
        if (per_xcd && win_bytes * 8 > total) continue;
        hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, tab, (unsigned long long)(win_bytes / 8), per_xcd, 64, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, tab, (unsigned long long)(win_bytes / 8), per_xcd, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double reads = (double)grid * 256 * iters;
        printf("gather 8 B  blocks/CU=%d  window=%8.2f MB %s  %8.3f ms  %7.1f G reads/s\n", bpc, win_bytes / 1048576.0,
               per_xcd ? "per XCD " : "shared  ", ms, reads / ms / 1e6);
      }
    }
  }
  CK(hipFuncSetAttribute((const void*)k_lds_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipLaunchKernelGGL(k_lds_gather, dim3(256), dim3(1024), 128 * 1024, 0, 64, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_lds_gather, dim3(256), dim3(1024), 128 * 1024, 0, 4096, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("LDS gather 8 B (128 KB table per CU)  %8.3f ms  %7.1f G reads/s\n", ms, 256.0 * 1024 * 4096 / ms / 1e6);
  return 0;
}
