// groupby_small.hip — what bounds the small perfect-hash GROUP BY (cfg2: int32 key, SUM(int64), 1 K groups,
// 12 B/row) — the two-stream read pattern or the per-row LDS atomic?  Variants of one kernel over 1 B rows:
//   mode 0  both streams read, values summed in registers (no LDS)           -> the streaming ceiling of the pattern
//   mode 1  + key store + 64-bit ds_add per row (what k_perfect_lds does)
//   mode 2  + ds_add only (no key store)
//   mode 3  mode 1 with the sum split in two 32-bit ds_add (lo, hi + carry via a second add when lo wraps)
// each with 256-lane and 1024-lane workgroups (same lanes per CU), with and without software pipelining
// (next tile's loads issued before the current tile is consumed), and UQ = 1 / 2 / 4 quads per lane per stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define GLOBAL __attribute__((address_space(1)))

struct K4 { int32_t v[4]; };
struct V4 { int64_t v[4]; };

__device__ inline K4 ldk(const int32_t* p, int64_t q) {
  typedef int32_t i4 __attribute__((ext_vector_type(4)));
  const i4 x = __builtin_nontemporal_load((const GLOBAL i4*)(p) + q);
  return K4{{x.x, x.y, x.z, x.w}};
}
__device__ inline V4 ldv(const int64_t* p, int64_t q) {
  typedef int64_t l2 __attribute__((ext_vector_type(2)));
  const l2 a = __builtin_nontemporal_load((const GLOBAL l2*)(p) + 2 * q);
  const l2 b = __builtin_nontemporal_load((const GLOBAL l2*)(p) + 2 * q + 1);
  return V4{{a.x, a.y, b.x, b.y}};
}

template <int MODE>
__device__ inline void row(int64_t* tab, int ne, int32_t k, int64_t v, int64_t& acc) {
  if (MODE == 0) { acc += v + k; return; }
  const uint32_t idx = (uint32_t)k;
  if (idx >= (uint32_t)ne) { acc = -1; return; }
  if (MODE == 1 || MODE == 3) *(volatile int64_t*)(tab + idx) = k;
  if (MODE == 1 || MODE == 2) atomicAdd((unsigned long long*)(tab + ne + idx), (unsigned long long)v);
  if (MODE == 3) {
    uint32_t* t32 = (uint32_t*)(tab + ne);
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
    const uint32_t old = atomicAdd(t32 + idx, lo);
    const uint32_t carry = (old + lo < old) ? 1u : 0u;
    if (hi + carry) atomicAdd(t32 + ne + idx, hi + carry);
  }
}

template <int BLOCK, int MODE, int UQ, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_gb(const int32_t* __restrict__ keys, const int64_t* __restrict__ vals, int64_t n,
                                              int ne, int64_t* __restrict__ out) {
  extern __shared__ int64_t tab[];
  for (int i = threadIdx.x; i < 3 * ne; i += BLOCK) tab[i] = 0;
  __syncthreads();
  int64_t acc = 0;
  const int64_t tile_q = (int64_t)BLOCK * UQ, n_tiles = (n >> 2) / tile_q;
  K4 kq[UQ], kn[UQ];
  V4 vq[UQ], vn[UQ];
  int64_t t = blockIdx.x;
  if (PIPE && t < n_tiles) {
#pragma unroll
    for (int u = 0; u < UQ; ++u) kn[u] = ldk(keys, t * tile_q + threadIdx.x + (int64_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < UQ; ++u) vn[u] = ldv(vals, t * tile_q + threadIdx.x + (int64_t)u * BLOCK);
  }
  for (; t < n_tiles; t += gridDim.x) {
    if (PIPE) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) { kq[u] = kn[u]; vq[u] = vn[u]; }
      int64_t t2 = t + gridDim.x;
      if (t2 >= n_tiles) t2 = t;  // clamped: unconditional loads
#pragma unroll
      for (int u = 0; u < UQ; ++u) kn[u] = ldk(keys, t2 * tile_q + threadIdx.x + (int64_t)u * BLOCK);
#pragma unroll
      for (int u = 0; u < UQ; ++u) vn[u] = ldv(vals, t2 * tile_q + threadIdx.x + (int64_t)u * BLOCK);
    } else {
#pragma unroll
      for (int u = 0; u < UQ; ++u) kq[u] = ldk(keys, t * tile_q + threadIdx.x + (int64_t)u * BLOCK);
#pragma unroll
      for (int u = 0; u < UQ; ++u) vq[u] = ldv(vals, t * tile_q + threadIdx.x + (int64_t)u * BLOCK);
    }
#pragma unroll
    for (int u = 0; u < UQ; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) row<MODE>(tab, ne, kq[u].v[i], vq[u].v[i], acc);
  }
  __syncthreads();
  if (MODE == 0) {
    if (acc == 0x7fffffffffffffffll) atomicAdd((unsigned long long*)out, 1ull);
  } else {
    for (int i = threadIdx.x; i < ne; i += BLOCK) {
      int64_t s = tab[ne + i];
      if (MODE == 3) {
        const uint32_t* t32 = (const uint32_t*)(tab + ne);
        s = (int64_t)(((uint64_t)t32[ne + i] << 32) + t32[i]);
      }
      if (s) atomicAdd((unsigned long long*)(out + i), (unsigned long long)s);
    }
    if (acc < 0) out[ne] = -1;
  }
}

__global__ void k_fill(int32_t* keys, int64_t* vals, int64_t n, int ne) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    keys[i] = (int32_t)(x % (uint64_t)ne);
    vals[i] = (int64_t)(x >> 40) - 5000;
  }
}

template <int BLOCK, int MODE, int UQ, bool PIPE>
void run(const char* name, const int32_t* keys, const int64_t* vals, int64_t n, int ne, int64_t* out, int lanes_per_cu, int64_t* ref) {
  const int grid = 256 * lanes_per_cu / BLOCK;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipMemset(out, 0, (ne + 1) * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gb<BLOCK, MODE, UQ, PIPE>), dim3(grid), dim3(BLOCK), 3 * ne * 8, 0, keys, vals, n, ne, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  int64_t h[4] = {0, 0, 0, 0};
  CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
  const char* okay = "";
  if (MODE != 0) {
    if (ref[0] == 0 && ref[1] == 0) { ref[0] = h[0]; ref[1] = h[1]; }
    okay = (ref[0] == h[0] && ref[1] == h[1]) ? " sums ok" : " SUMS DIFFER";
  }
  printf("%-44s block %4d lanes/CU %4d UQ %d pipe %d   %7.3f ms  %6.0f GB/s (%.2f of 8 TB/s)%s\n", name, BLOCK, lanes_per_cu, UQ,
         (int)PIPE, best, 12.0 * n / best / 1e6, 12.0 * n / best / 1e6 / 8000.0, okay);
}

int main() {
  const int64_t n = 1000000000ll / 4096 * 4096;
  const int ne = 1000;
  int32_t* keys;
  int64_t *vals, *out;
  CK(hipMalloc(&keys, n * 4));
  CK(hipMalloc(&vals, n * 8));
  CK(hipMalloc(&out, (ne + 1) * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, keys, vals, n, ne);
  CK(hipDeviceSynchronize());
  int64_t ref[2] = {0, 0};
#define R(B, M, U, P, L, NAME) run<B, M, U, P>(NAME, keys, vals, n, ne, out, L, ref)
  R(256, 0, 2, false, 1024, "read only (register sums)");
  R(256, 0, 2, true, 1024, "read only (register sums)");
  R(1024, 0, 2, false, 1024, "read only (register sums)");
  R(1024, 0, 2, true, 1024, "read only (register sums)");
  R(256, 0, 1, true, 2048, "read only (register sums)");
  R(256, 0, 4, false, 1024, "read only (register sums)");
  R(256, 0, 4, true, 512, "read only (register sums)");
  R(256, 1, 2, false, 1024, "key store + ds_add_u64 (as k_perfect_lds)");
  R(256, 1, 2, true, 1024, "key store + ds_add_u64");
  R(1024, 1, 2, false, 1024, "key store + ds_add_u64");
  R(1024, 1, 2, true, 1024, "key store + ds_add_u64");
  R(256, 1, 1, true, 2048, "key store + ds_add_u64");
  R(256, 1, 1, false, 2048, "key store + ds_add_u64");
  R(256, 1, 4, true, 1024, "key store + ds_add_u64");
  R(256, 1, 4, true, 512, "key store + ds_add_u64");
  R(1024, 1, 1, true, 2048, "key store + ds_add_u64");
  R(256, 2, 2, false, 1024, "ds_add_u64 only");
  R(256, 2, 2, true, 1024, "ds_add_u64 only");
  R(1024, 2, 2, true, 1024, "ds_add_u64 only");
  R(256, 3, 2, false, 1024, "key store + 2 x ds_add_u32");
  R(256, 3, 2, true, 1024, "key store + 2 x ds_add_u32");
  R(1024, 3, 2, true, 1024, "key store + 2 x ds_add_u32");
  return 0;
}
