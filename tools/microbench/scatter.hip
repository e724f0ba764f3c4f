// scatter.hip — memory-system envelope for the partition (scatter) phase (tools only).
//
// Simulates the phase-1 traffic of the partitioned GROUP BY without its staging logic: every
// 1024-lane workgroup (one per CU) streams 20 B/row (int32 + int64 + fp64 columns, 16-byte
// non-temporal loads, contiguous tiles) and writes 16 B per *surviving* row (50 %) as whole
// L-byte lines to P x B scattered run fronts.  Sweeps L (bytes per flushed line), P (partitions),
// the scratch layout (partition-major runs vs slab-interleaved) and the store flavour, next to
// tuned copy / read-only baselines, so the kernel design can be chosen from measurements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kT = 1024;

template <bool NTS>
__device__ inline void st16(v4i32* p, v4i32 v) {
  if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

// mode 0: read only; mode 1: read + contiguous write of the surviving half; mode 2: scattered lines
template <int LINE_LANES, bool NTS, int LAYOUT>
__global__ __launch_bounds__(kT) void k_scatter_sim(const v4i32* __restrict__ fcol, const v4i32* __restrict__ kcol,
                                                     const v4i32* __restrict__ vcol, long long n_quads,
                                                     v4i32* __restrict__ scratch, int P, long long run_vec,
                                                     int mode, unsigned long long* sink) {
  const int b = blockIdx.x, B = gridDim.x, t = threadIdx.x;
  const long long n_tiles = n_quads / kT;
  int acc = 0;
  long long round = 0;
  for (long long tile = b; tile < n_tiles; tile += B, ++round) {
    const long long q = tile * kT + t;
    const v4i32 f = __builtin_nontemporal_load(fcol + q);
    const v4i32 k0 = __builtin_nontemporal_load(kcol + 2 * q);
    const v4i32 k1 = __builtin_nontemporal_load(kcol + 2 * q + 1);
    const v4i32 v0 = __builtin_nontemporal_load(vcol + 2 * q);
    const v4i32 v1 = __builtin_nontemporal_load(vcol + 2 * q + 1);
    acc += f.x + f.y + f.z + f.w;
    if (mode == 0) { acc += k0.x + k1.y + v0.z + v1.w; continue; }
    // 2 records (32 B) per lane per round = 2048 records per workgroup-round
    const v4i32 r0 = v4i32{k0.x, k0.y, v0.x, v0.y};
    const v4i32 r1 = v4i32{k1.x, k1.y, v1.x, v1.y};
    if (mode == 1) {
      v4i32* dst = scratch + ((long long)b * (n_tiles / B + 1) + round) * (2 * kT);
      st16<NTS>(dst + t, r0);
      st16<NTS>(dst + kT + t, r1);
      continue;
    }
    if (mode == 3) {
      // lane-owned lines: lanes 0..255 each write one whole 128-byte line (8 x 16-byte stores)
      if (t < 256) {
        const long long g = round * 256 + t;
        const unsigned pid = ((unsigned)g * 0x9E3779B1u) & (unsigned)(P - 1);
        const long long addr = ((long long)pid * B + b) * run_vec + (g / P) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) st16<NTS>(scratch + addr + k, (k & 1) ? r1 : r0);
      }
      continue;
    }
    // scattered: the workgroup-round emits 2048/LINE_LANES lines; line g -> partition perm(g % P)
    const int lines_per_round = 2 * kT / LINE_LANES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int vec = h * kT + t;               // 16-byte vector index within the round
      const long long g = round * lines_per_round + vec / LINE_LANES;
      const unsigned pid = ((unsigned)g * 0x9E3779B1u) & (unsigned)(P - 1);
      const long long off_vec = (g / P) * LINE_LANES + (vec % LINE_LANES);  // position in the run
      long long addr;
      if (LAYOUT == 0) {
        addr = ((long long)pid * B + b) * run_vec + off_vec;     // [p][b][run]
      } else if (LAYOUT == 1) {
        addr = ((long long)b * P + pid) * run_vec + off_vec;     // [b][p][run]
      } else {
        // slab-interleaved: [b][slab][p][16 vec = 256 B]
        const long long slab = off_vec >> 4;
        addr = (((long long)b * (run_vec >> 4) + slab) * P + pid) * 16 + (off_vec & 15);
      }
      st16<NTS>(scratch + addr, h ? r1 : r0);
    }
  }
  if (acc == 0x7fffffff) atomicAdd(sink, 1ull);
}

// Lines that are not a power of two long (round 4: VERDICT r03 lever (b), 15-byte records = 120-byte lines of eight; and
// 96-byte lines, which would let 1536 partitions share the LDS staging area).  Same reads; the round's records leave as
// 8-byte words: line g = word / W of the workgroup-round, packed runs (pitch W * 8 bytes, so 120-byte lines straddle
// the 64-byte write granules).  `keep16` = sixteenths of the record bytes that are written at all (15 for 15-byte records).
template <int W, bool NTS>
__global__ __launch_bounds__(kT) void k_scatter_words(const v4i32* __restrict__ fcol, const v4i32* __restrict__ kcol,
                                                       const v4i32* __restrict__ vcol, long long n_quads,
                                                       long long* __restrict__ scratch, int P, long long run_words, int keep16,
                                                       unsigned long long* sink) {
  const int b = blockIdx.x, B = gridDim.x, t = threadIdx.x;
  const long long n_tiles = n_quads / kT;
  int acc = 0;
  long long round = 0;
  const int words_per_round = 4 * kT * keep16 / 16;         // 8-byte words this workgroup-round emits
  const int lines_per_round = (words_per_round + W - 1) / W;
  for (long long tile = b; tile < n_tiles; tile += B, ++round) {
    const long long q = tile * kT + t;
    const v4i32 f = __builtin_nontemporal_load(fcol + q);
    const v4i32 k0 = __builtin_nontemporal_load(kcol + 2 * q);
    const v4i32 k1 = __builtin_nontemporal_load(kcol + 2 * q + 1);
    const v4i32 v0 = __builtin_nontemporal_load(vcol + 2 * q);
    const v4i32 v1 = __builtin_nontemporal_load(vcol + 2 * q + 1);
    acc += f.x + f.y + f.z + f.w;
    const long long w8[4] = {(long long)k0.x << 32 | (unsigned)k0.y, (long long)v0.x << 32 | (unsigned)v0.y,
                             (long long)k1.x << 32 | (unsigned)k1.y, (long long)v1.x << 32 | (unsigned)v1.y};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int word = h * kT + t;
      if (word >= words_per_round) continue;
      const long long g = round * lines_per_round + word / W;
      const unsigned pid = ((unsigned)g * 0x9E3779B1u) & (unsigned)(P - 1);
      const long long off = (g / P) * W + (word % W);
      long long* dst = scratch + ((long long)pid * B + b) * run_words + off;
      if (NTS) __builtin_nontemporal_store(w8[h], dst); else *dst = w8[h];
    }
  }
  if (acc == 0x7fffffff) atomicAdd(sink, 1ull);
}

// tuned copy: tile-contiguous, nt loads, U vectors in flight per lane
template <int U, bool NTS>
__global__ __launch_bounds__(256) void k_copy_tiled(const v4i32* __restrict__ a, v4i32* __restrict__ bdst, long long nvec) {
  const long long tile = (long long)blockDim.x * U;
  for (long long t0 = (long long)blockIdx.x * tile; t0 + tile <= nvec; t0 += (long long)gridDim.x * tile) {
    v4i32 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + t0 + u * blockDim.x + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) st16<NTS>(bdst + t0 + u * blockDim.x + threadIdx.x, v[u]);
  }
}

// phase-2 side: a workgroup per partition reads its B runs (wave per run, 16 B per lane)
__global__ __launch_bounds__(kT) void k_gather_sim(const v4i32* __restrict__ scratch, int P, int B, long long run_vec,
                                                   long long used_vec, int layout, unsigned long long* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int acc = 0;
  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    for (int b = wave; b < B; b += kT / 64) {
      if (layout == 2) {
        for (long long i = lane; i < used_vec; i += 64) {
          const long long slab = i >> 4;
          const v4i32 v = __builtin_nontemporal_load(scratch + (((long long)b * (run_vec >> 4) + slab) * P + p) * 16 + (i & 15));
          acc += v.x + v.w;
        }
      } else {
        const v4i32* run = scratch + (layout == 0 ? ((long long)p * B + b) : ((long long)b * P + p)) * run_vec;
        long long i = lane;
        for (; i + 192 < used_vec; i += 256) {
          const v4i32 a0 = __builtin_nontemporal_load(run + i), a1 = __builtin_nontemporal_load(run + i + 64);
          const v4i32 a2 = __builtin_nontemporal_load(run + i + 128), a3 = __builtin_nontemporal_load(run + i + 192);
          acc += a0.x + a1.y + a2.z + a3.w;
        }
        for (; i < used_vec; i += 64) { const v4i32 a0 = __builtin_nontemporal_load(run + i); acc += a0.x; }
      }
    }
  }
  if (acc == 0x7fffffff) atomicAdd(sink, 1ull);
}

template <typename F>
float time_ms(F&& f, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError()); return ms / reps;
}

int main(int argc, char** argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 1000000000LL;
  const long long nq = rows / 4;
  const int B = 256;
  v4i32 *fc, *kc, *vc, *scratch; unsigned long long* sink;
  CK(hipMalloc(&fc, rows * 4)); CK(hipMalloc(&kc, rows * 8)); CK(hipMalloc(&vc, rows * 8));
  const long long scratch_bytes = rows * 8 + (4LL << 30);   // 16 B x rows/2 + slack
  CK(hipMalloc(&scratch, scratch_bytes)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(fc, 1, rows * 4)); CK(hipMemset(kc, 2, rows * 8)); CK(hipMemset(vc, 3, rows * 8)); CK(hipMemset(sink, 0, 8));
  CK(hipMemset(scratch, 0, scratch_bytes));
  const double in_gb = rows * 20.0 / 1e9, out_gb = rows * 8.0 / 1e9;
  printf("rows %lld  in %.1f GB  out %.1f GB\n", rows, in_gb, out_gb);
  {
    const long long nvec = rows * 8 / 16;
#define COPY(U, NTS, bpc) { float ms = time_ms([&] { hipLaunchKernelGGL((k_copy_tiled<U, NTS>), dim3(256 * bpc), dim3(256), 0, 0, kc, scratch, nvec); }); \
      printf("copy_tiled U=%d nts=%d bpc=%d   %8.3f ms  %7.1f GB/s (r+w)\n", U, (int)NTS, bpc, ms, 2.0 * nvec * 16 / ms / 1e6); }
    COPY(4, false, 2) COPY(4, true, 2) COPY(8, false, 2) COPY(8, true, 2) COPY(4, true, 4) COPY(8, true, 4) COPY(4, false, 8) COPY(4, true, 8)
  }
  auto run = [&](auto kern, const char* label, int P, int mode) {
    // run capacity in 16-byte vectors: mean records per run x 1.0 (deterministic round-robin), rounded to 256 B
    long long run_vec = ((rows / 2) / ((long long)P * B) + 16 + 15) & ~15LL;
    if ((long long)P * B * run_vec * 16 > scratch_bytes) { printf("%s P=%d: scratch too small\n", label, P); return; }
    float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(B), dim3(kT), 0, 0, fc, kc, vc, nq, scratch, P, run_vec, mode, sink); });
    const double gb = in_gb + (mode ? out_gb : 0.0);
    printf("%-28s P=%5d  %8.3f ms  %7.1f GB/s total  (%.1f GB/s algorithmic)\n", label, P, ms, gb / ms * 1e3, in_gb / ms * 1e3);
  };
  run(k_scatter_sim<1, false, 0>, "read-only", 1, 0);
  run(k_scatter_sim<1, false, 0>, "read+contig write", 1, 1);
  run(k_scatter_sim<1, true, 0>, "read+contig write nt", 1, 1);
  run(k_scatter_sim<8, false, 0>, "lane-owned 128 B line plain", 1024, 3);
  run(k_scatter_sim<8, true, 0>, "lane-owned 128 B line nt", 1024, 3);
  run(k_scatter_sim<8, false, 0>, "L=128 [p][b] plain (8 lanes)", 1024, 2);
  run(k_scatter_sim<8, true, 0>, "L=128 [p][b] nt (8 lanes)", 1024, 2);
  {
    // odd line lengths, 8-byte stores (the 16-word line is the 128-byte line again, as the control for the store width)
    auto runw = [&](auto kern, const char* label, int P, int W, int keep16) {
      long long run_words = (((rows / 2) * 2 * keep16 / 16) / ((long long)P * B) + 4 * W + 15) & ~15LL;
      if ((long long)P * B * run_words * 8 > scratch_bytes) { printf("%s P=%d: scratch too small\n", label, P); return; }
      float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(B), dim3(kT), 0, 0, fc, kc, vc, nq, (long long*)scratch, P, run_words, keep16, sink); });
      const double gb = in_gb + out_gb * keep16 / 16.0;
      printf("%-44s P=%5d  %8.3f ms  %7.1f GB/s total  (%.1f GB/s algorithmic)\n", label, P, ms, gb / ms * 1e3, in_gb / ms * 1e3);
    };
    runw(k_scatter_words<16, true>, "128 B lines, 8-byte nt stores (control)", 1024, 16, 16);
    runw(k_scatter_words<15, true>, "120 B lines = 8 x 15-byte records, nt", 1024, 15, 15);
    runw(k_scatter_words<15, false>, "120 B lines = 8 x 15-byte records, plain", 1024, 15, 15);
    runw(k_scatter_words<12, true>, "96 B lines (1536 partitions), nt", 1536, 12, 16);
    runw(k_scatter_words<12, false>, "96 B lines (1536 partitions), plain", 1536, 12, 16);
    runw(k_scatter_words<8, true>, "64 B lines (2048 partitions), nt", 2048, 8, 16);
  }
  if (argc > 2) return 0;
  const int Ps[] = {512, 2048, 4096};
  for (int P : Ps) {
    run(k_scatter_sim<1, false, 0>, "L=16  [p][b] plain", P, 2);
    run(k_scatter_sim<2, false, 0>, "L=32  [p][b] plain", P, 2);
    run(k_scatter_sim<4, false, 0>, "L=64  [p][b] plain", P, 2);
    run(k_scatter_sim<8, false, 0>, "L=128 [p][b] plain", P, 2);
    run(k_scatter_sim<16, false, 0>, "L=256 [p][b] plain", P, 2);
    run(k_scatter_sim<2, true, 0>, "L=32  [p][b] nt", P, 2);
    run(k_scatter_sim<4, true, 0>, "L=64  [p][b] nt", P, 2);
    run(k_scatter_sim<8, true, 0>, "L=128 [p][b] nt", P, 2);
    run(k_scatter_sim<4, false, 1>, "L=64  [b][p] plain", P, 2);
    run(k_scatter_sim<2, false, 2>, "L=32  slab plain", P, 2);
    run(k_scatter_sim<4, false, 2>, "L=64  slab plain", P, 2);
    run(k_scatter_sim<8, false, 2>, "L=128 slab plain", P, 2);
    run(k_scatter_sim<2, true, 2>, "L=32  slab nt", P, 2);
    run(k_scatter_sim<4, true, 2>, "L=64  slab nt", P, 2);
  }
  // phase-2 read side
  for (int P : Ps) {
    long long run_vec = ((rows / 2) / ((long long)P * B) + 16 + 15) & ~15LL;
    long long used = (rows / 2) / ((long long)P * B);
    for (int layout = 0; layout < 3; ++layout) {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_gather_sim, dim3(256), dim3(kT), 0, 0, scratch, P, B, run_vec, used, layout, sink); });
      printf("gather layout=%d P=%5d run=%lld rec  %8.3f ms  %7.1f GB/s\n", layout, P, used, ms, (double)P * B * used * 16 / ms / 1e6);
    }
  }
  return 0;
}
