// TEST INFRASTRUCTURE (tests/hostsim): a stand-in for the HIP runtime API so that the library's host code
// (heavydb_amd/csrc/api.cpp, plan.cpp) and the barrier-light kernels of kernels_generic.hip can be compiled with a
// plain C++ compiler and run on the CPU — "device" memory is host memory, a stream is a no-op, a kernel launch runs
// the kernel body on a pool of host threads (hip_host.cpp).  Only what those files use is declared.
#pragma once
#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotSupported = 801,
  hipErrorUnknown = 999
} hipError_t;

typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;

enum { hipStreamNonBlocking = 1 };
enum { hipEventDisableTiming = 2 };
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                             hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 } hipDeviceAttribute_t;
typedef enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;

typedef struct hipDeviceProp_t {
  char name[256];
  size_t totalGlobalMem;
  int multiProcessorCount;
  char gcnArchName[256];
  size_t sharedMemPerBlock;
  int maxSharedMemoryPerMultiProcessor;
  int l2CacheSize;
  int clockRate;
  int warpSize;
  int memoryClockRate;
  int memoryBusWidth;
} hipDeviceProp_t;

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern "C" {
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
enum { hipHostMallocDefault = 0 };
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int device);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
}
// (kernel attributes mean nothing here)
template <typename F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }
