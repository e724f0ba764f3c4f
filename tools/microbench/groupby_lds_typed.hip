// groupby_lds_typed.hip — what would k_groupby_lds cost per row if every role were known at compile time?
//
// The product kernel (heavydb_amd/csrc/kernels_lds.hip) keeps roles at run time — key / value types, which accumulators a
// value column has, NULL-awareness, perfect vs baseline — and its row loop is 5 941 static instructions per 8 rows, 3 899 of
// them uniform branches (profiles/r03_isa_k_groupby_lds_static.txt); measured it runs one 64-row wave step per ~530 cycles
// and SIMD (PHS001-004: 0.10 - 0.26 of the roofline at 8 B/row, profiles/r03_refbench_1b_call6.jsonl).  This is the same
// work for the reference benchmark's PerfectHashSingleCol shape — int32 key = entry index, nullable int32 value, per entry
// rows (u32) + non-NULL count (u32) + sum (i64) + min (i64) + max (i64), K replicas in LDS, optional windows — with all of it
// as template arguments, 32-bit index arithmetic and the next tile's loads in flight:
//   ACC   bit 0 non-NULL count, 1 sum, 2 min, 3 max   (rows is always kept)
//   UQ    quads (4 rows) per lane and column per step
//   T     windows: workgroup b keeps the rows of window b % T (T = 1: the whole table)
// Prints ms, rows/s and the fraction of 8 TB/s on 8 B/row for 1 B rows.  (Round 3 wrote it with no GPU time left: the static
// instruction count is in profiles/, the timings are round 4's first job.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define GLOBAL __attribute__((address_space(1)))

typedef int32_t i4 __attribute__((ext_vector_type(4)));
constexpr int BLOCK = 1024;
constexpr int32_t kNullI32 = INT32_MIN;

__device__ inline i4 ld4(const int32_t* p, int64_t q) { return __builtin_nontemporal_load((const GLOBAL i4*)(p) + q); }

// one replica: rows[E] u32 | cnt[E] u32 | sum[E] i64 | min[E] i64 | max[E] i64 (only what ACC keeps), 8-byte arrays first
template <int ACC>
__host__ __device__ constexpr uint32_t replica_bytes(uint32_t E) {
  return E * (8u * (((ACC >> 1) & 1) + ((ACC >> 2) & 1) + ((ACC >> 3) & 1)) + 4u * (1 + (ACC & 1)));
}

template <int ACC>
__device__ inline void one_row(char* rep, uint32_t E, uint32_t e_lo, int32_t key, int32_t val, int32_t key_min) {
  const uint32_t e = (uint32_t)(key - key_min) - e_lo;  // 32-bit: the entry index of a perfect-hash table fits
  if (e >= E) return;                                   // another window's row (or out of range)
  constexpr uint32_t n8 = ((ACC >> 1) & 1) + ((ACC >> 2) & 1) + ((ACC >> 3) & 1);
  int64_t* a8 = (int64_t*)rep;
  uint32_t* a4 = (uint32_t*)(rep + (size_t)E * 8 * n8);
  atomicAdd(a4 + e, 1u);
  if (val == kNullI32) return;
  if (ACC & 1) atomicAdd(a4 + E + e, 1u);
  uint32_t k = 0;
  if (ACC & 2) { atomicAdd((unsigned long long*)(a8 + (size_t)E * k + e), (unsigned long long)(int64_t)val); ++k; }
  if (ACC & 4) { atomicMin((long long*)(a8 + (size_t)E * k + e), (long long)val); ++k; }
  if (ACC & 8) { atomicMax((long long*)(a8 + (size_t)E * k + e), (long long)val); ++k; }
}

template <int ACC, int UQ>
__global__ __launch_bounds__(BLOCK) void k_typed(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                 uint32_t E, int copies_lg, int T, int32_t key_min, unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const uint32_t K = 1u << copies_lg, rb = replica_bytes<ACC>(E);
  constexpr uint32_t n8 = ((ACC >> 1) & 1) + ((ACC >> 2) & 1) + ((ACC >> 3) & 1);
  for (uint32_t r = 0; r < K; ++r) {
    int64_t* a8 = (int64_t*)(smem + (size_t)r * rb);
    uint32_t* a4 = (uint32_t*)(smem + (size_t)r * rb + (size_t)E * 8 * n8);
    for (uint32_t e = t; e < E; e += BLOCK) {
      uint32_t k = 0;
      if (ACC & 2) a8[(size_t)E * k++ + e] = 0;
      if (ACC & 4) a8[(size_t)E * k++ + e] = INT64_MAX;
      if (ACC & 8) a8[(size_t)E * k++ + e] = INT64_MIN;
      a4[e] = 0;
      if (ACC & 1) a4[E + e] = 0;
    }
  }
  __syncthreads();
  char* const rep = smem + (size_t)((uint32_t)t & (K - 1)) * rb;
  const uint32_t win = T > 1 ? blockIdx.x % T : 0u, stripe = T > 1 ? blockIdx.x / T : blockIdx.x;
  const uint32_t n_stripes = T > 1 ? gridDim.x / T : gridDim.x, e_lo = win * E;
  const int64_t tile_q = (int64_t)BLOCK * UQ, n_tiles = (n >> 2) / tile_q;
  i4 kq[UQ], vq[UQ], kn[UQ], vn[UQ];
  int64_t tl = stripe;
  if (tl < n_tiles) {
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      kn[u] = ld4(keys, tl * tile_q + t + (int64_t)u * BLOCK);
      vn[u] = ld4(vals, tl * tile_q + t + (int64_t)u * BLOCK);
    }
  }
  for (; tl < n_tiles; tl += n_stripes) {
#pragma unroll
    for (int u = 0; u < UQ; ++u) { kq[u] = kn[u]; vq[u] = vn[u]; }
    const int64_t nx = tl + n_stripes;
    if (nx < n_tiles) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        kn[u] = ld4(keys, nx * tile_q + t + (int64_t)u * BLOCK);
        vn[u] = ld4(vals, nx * tile_q + t + (int64_t)u * BLOCK);
      }
    }
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      one_row<ACC>(rep, E, e_lo, kq[u].x, vq[u].x, key_min);
      one_row<ACC>(rep, E, e_lo, kq[u].y, vq[u].y, key_min);
      one_row<ACC>(rep, E, e_lo, kq[u].z, vq[u].z, key_min);
      one_row<ACC>(rep, E, e_lo, kq[u].w, vq[u].w, key_min);
    }
  }
  __syncthreads();
  // fold into one checksum per workgroup (the product flushes into the output table; same order of cost)
  unsigned long long acc = 0;
  for (uint32_t r = 0; r < K; ++r) {
    const uint32_t* a4 = (const uint32_t*)(smem + (size_t)r * rb + (size_t)E * 8 * n8);
    const int64_t* a8 = (const int64_t*)(smem + (size_t)r * rb);
    for (uint32_t e = t; e < E; e += BLOCK) {
      acc += a4[e];
      if (ACC & 2) acc += (unsigned long long)a8[e];
    }
  }
  if (acc) atomicAdd(out, acc);
}

__global__ void k_fill(int32_t* keys, int32_t* vals, int64_t n, uint32_t card) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    keys[i] = 1 + (int32_t)((x >> 20) % card);
    vals[i] = 1 + (int32_t)((x >> 7) % 10u);
  }
}

template <int ACC, int UQ>
void run(const char* tag, const int32_t* keys, const int32_t* vals, int64_t n, uint32_t card, int n_cus, unsigned long long* d_out) {
  // windows: the fewest whose share of the table fits 152 KB; replicas: as many as fit (<= 64)
  const uint32_t budget = 152 * 1024;
  int T = 1;
  while (replica_bytes<ACC>((card + T - 1) / T) > budget && T < 8) ++T;
  const uint32_t E = (card + T - 1) / T;
  if (replica_bytes<ACC>(E) > budget) { printf("%-28s card %8u: does not fit 8 windows\n", tag, card); return; }
  int lg = 0;
  while (lg < 6 && ((size_t)replica_bytes<ACC>(E) << (lg + 1)) <= budget) ++lg;
  const size_t lds = (size_t)replica_bytes<ACC>(E) << lg;
  auto k = k_typed<ACC, UQ>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (n_cus / T) * T;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, 0, keys, vals, n, E, lg, T, 1, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (it && ms < best) best = ms;
  }
  // bytes the member really loads per row: the key column, and the value column only when an accumulator reads it (ACC = 0,
  // "rows only", never touches `vals`: crediting it with 8 B/row printed fractions above 1 in round 3)
  const double bpr = ACC ? 8.0 : 4.0;
  printf("%-28s card %8u  T %d  replicas %2d  %7.3f ms  %6.1f G rows/s  %.3f of 8 TB/s (%.0f B/row)\n", tag, card, T, 1 << lg, best,
         n / best / 1e6, n * bpr / (best * 1e-3) / 8e12, bpr);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000000ll;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int n_cus = p.multiProcessorCount;
  int32_t *keys, *vals; unsigned long long* d_out;
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&vals, n * 4)); CK(hipMalloc(&d_out, 8)); CK(hipMemset(d_out, 0, 8));
  for (uint32_t card : {10u, 1000u, 10000u}) {
    hipLaunchKernelGGL(k_fill, dim3(n_cus * 8), dim3(256), 0, 0, keys, vals, n, card);
    CK(hipDeviceSynchronize());
    run<15, 1>("count,sum,min,max  UQ1", keys, vals, n, card, n_cus, d_out);
    run<15, 2>("count,sum,min,max  UQ2", keys, vals, n, card, n_cus, d_out);
    run<15, 4>("count,sum,min,max  UQ4", keys, vals, n, card, n_cus, d_out);
    run<3, 2>("count,sum          UQ2", keys, vals, n, card, n_cus, d_out);
    run<0, 2>("rows only (key col) UQ2", keys, vals, n, card, n_cus, d_out);
  }
  return 0;
}
