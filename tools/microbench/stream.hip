// stream.hip — read-bandwidth microbenchmark for the scan skeleton (tools only).
// Sweeps: plain vs nontemporal 16-byte loads, loads in flight per lane, block size, blocks/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int UNROLL, bool NT>
__global__ void k_read(const v4i32* __restrict__ p, long long nvec, unsigned long long* out) {
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gsize = (long long)gridDim.x * blockDim.x;
  int acc = 0;
  long long i = gtid;
  for (; i + (UNROLL - 1) * gsize < nvec; i += UNROLL * gsize) {
    v4i32 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * gsize) : p[i + u * gsize];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += (v[u].x < 1073741824) + (v[u].y < 1073741824) + (v[u].z < 1073741824) + (v[u].w < 1073741824);
  }
  for (; i < nvec; i += gsize) { v4i32 v = p[i]; acc += (v.x < 1073741824) + (v.y < 1073741824) + (v.z < 1073741824) + (v.w < 1073741824); }
  unsigned long long a = acc;
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(out, a);
}

// block-contiguous tiling: each block walks contiguous TILE-byte tiles
template <int UNROLL, bool NT>
__global__ void k_read_tiled(const v4i32* __restrict__ p, long long nvec, unsigned long long* out) {
  const int tile = blockDim.x * UNROLL;  // vectors per tile
  int acc = 0;
  for (long long t = (long long)blockIdx.x * tile; t + tile <= nvec; t += (long long)gridDim.x * tile) {
    v4i32 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + t + u * blockDim.x + threadIdx.x) : p[t + u * blockDim.x + threadIdx.x];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += (v[u].x < 1073741824) + (v[u].y < 1073741824) + (v[u].z < 1073741824) + (v[u].w < 1073741824);
  }
  unsigned long long a = acc;
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(out, a);
}

__global__ void k_fill(v4i32* p, long long nvec) {
  const long long gsize = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gsize) {
    unsigned x = (unsigned)(i * 2654435761u);
    p[i] = v4i32{(int)(x >> 1), (int)((x * 3) >> 1), (int)((x * 5) >> 1), (int)((x * 7) >> 1)};
  }
}
__global__ void k_copy(const v4i32* __restrict__ a, v4i32* __restrict__ b, long long nvec) {
  const long long gsize = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gsize) b[i] = a[i];
}

template <typename F>
float time_ms(F&& f, int reps = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
  const long long bytes = (argc > 1 ? atoll(argv[1]) : 8LL) << 30;
  const long long nvec = bytes / 16;
  v4i32 *a, *b; unsigned long long* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, nvec); CK(hipDeviceSynchronize());
  printf("buffer %lld GiB\n", bytes >> 30);
  { float ms = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpy D2D        %8.3f ms  %7.1f GB/s (r+w)\n", ms, 2.0 * bytes / ms / 1e6); }
  { float ms = time_ms([&] { CK(hipMemsetAsync(b, 1, bytes, 0)); }); printf("hipMemset            %8.3f ms  %7.1f GB/s (w)\n", ms, 1.0 * bytes / ms / 1e6); }
  { float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, nvec); }); printf("k_copy 2048x256      %8.3f ms  %7.1f GB/s (r+w)\n", ms, 2.0 * bytes / ms / 1e6); }
  const int bpcs[] = {2, 4, 8, 16};
  const int tpbs[] = {256, 512, 1024};
#define RUN(KN, U, NT, label) for (int tpb : tpbs) for (int bpc : bpcs) { if (tpb * bpc > 2048 * 2) continue; int grid = 256 * bpc; \
    float ms = time_ms([&] { hipLaunchKernelGGL((KN<U, NT>), dim3(grid), dim3(tpb), 0, 0, a, nvec, out); }); \
    printf("%-12s U=%d nt=%d tpb=%4d bpc=%2d  %8.3f ms  %7.1f GB/s\n", label, U, (int)NT, tpb, bpc, ms, 1.0 * bytes / ms / 1e6); }
  RUN(k_read, 1, false, "stride") RUN(k_read, 2, false, "stride") RUN(k_read, 4, false, "stride") RUN(k_read, 8, false, "stride")
  RUN(k_read, 2, true, "stride") RUN(k_read, 4, true, "stride") RUN(k_read, 8, true, "stride")
  RUN(k_read_tiled, 4, false, "tiled") RUN(k_read_tiled, 8, false, "tiled") RUN(k_read_tiled, 4, true, "tiled") RUN(k_read_tiled, 8, true, "tiled")
  return 0;
}
