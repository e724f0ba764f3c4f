// TEST INFRASTRUCTURE (tests/hostsim): the host stand-in for the HIP runtime (shim/hip/*.h).
//   memory   hipMalloc = aligned host memory FILLED WITH 0xA5 (a kernel that relies on fresh memory being zero, or
//            on a buffer somebody forgot to initialise, fails here the way it eventually would on the device);
//            hipFree poisons before releasing
//   streams  no-ops: every "launch" has finished when the call returns
//   launch   one block at a time on the calling thread: kernels without barriers run their threads as plain calls one
//            after the other; kernels with __syncthreads / shuffles / ballots / LDS run the threads of a block as
//            cooperative fibers (ucontext) that yield at a barrier — deterministic, no data races, atomics trivially
//            atomic; per-wave (64 lanes) exchange buffers for shuffles / ballots, a dynamic LDS area
#include <ucontext.h>
#include <unistd.h>

#include <csignal>

#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "hip/hip_runtime.h"

namespace hipsim {

thread_local Idx t_thread, t_block, t_block_dim, t_grid_dim;

namespace {

// Kernels of kernels_generic.hip that contain a barrier, a shuffle / ballot or LDS (found by scanning the source when
// the simulation is built, tests/helpers.py hostsim_lib): their blocks run as 256 cooperative fibers.  Every other
// kernel is a plain grid-stride loop: its threads run one after the other as ordinary calls.
const char* const kBarrierKernels[] = {
#include "barrier_kernels.inc"
    nullptr};

bool names_one_of(const char* name, const char* const* list) {
  for (int i = 0; list[i]; ++i)
    if (std::strstr(name, list[i])) {
      // (whole identifier: k_generic must not match k_generic_lds and vice versa)
      const char* at = std::strstr(name, list[i]);
      const char c = at[std::strlen(list[i])];
      if (!(std::isalnum((unsigned char)c) || c == '_')) return true;
    }
  return false;
}
#ifdef HOSTSIM_REAL_FAST
// the real kernels_fast.hip / kernels_lds.hip are compiled in (templates launched through local names): every kernel
// that is not a known barrier-free kernel of kernels_generic.hip runs as fibers
const char* const kPlainKernels[] = {
#include "plain_kernels.inc"
    nullptr};
bool needs_fibers(const char* name) { return !names_one_of(name, kPlainKernels); }
#else
bool needs_fibers(const char* name) { return names_one_of(name, kBarrierKernels); }
#endif

constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

// barrier among fibers: the party count shrinks when a fiber leaves the kernel
struct Barrier {
  int expected = 0, arrived = 0;
  uint64_t gen = 0;
  void reset(int n) {
    expected = n;
    arrived = 0;
  }
  void drop() {
    --expected;
    if (expected > 0 && arrived >= expected) {
      arrived = 0;
      ++gen;
    }
  }
};

struct Sched {
  ucontext_t main_ctx;
  ucontext_t ctx[kMaxThreads];
  char* stacks[kMaxThreads] = {};
  bool finished[kMaxThreads];
  int cur = -1, n = 0;
  const std::function<void()>* body = nullptr;
  Idx block, grid_dim, block_dim;
  Barrier block_barrier, wave_barrier[kMaxThreads / kWave];
  uint64_t xchg[kMaxThreads], recv[kMaxThreads];
  unsigned char state[kMaxThreads];   // HOSTSIM_WATCHDOG: where each fiber waits (0 runs, 1 block barrier, 2 wave barrier, 3 yield, 4 done)
  const Barrier* wait_b[kMaxThreads];  // the barrier a fiber is parked at (null: runnable) and the generation it waits to pass:
  uint64_t wait_gen[kMaxThreads];      // the scheduler does not switch to a fiber whose barrier has not moved
  const char* kernel_name = "";
  unsigned long long ballot[kMaxThreads / kWave];
  std::vector<char> dyn;
  bool in_fibers = false;
};
Sched& sched() {
  static Sched* s = new Sched();
  return *s;
}
std::mutex g_launch_mu;  // one launch at a time

void set_ids(Sched& s, int id) {
  const unsigned bx = s.block_dim.x, by = s.block_dim.y;
  t_thread.x = id % bx;
  t_thread.y = (id / bx) % by;
  t_thread.z = id / (bx * by);
  t_block = s.block;
  t_block_dim = s.block_dim;
  t_grid_dim = s.grid_dim;
}

void trampoline() {
  Sched& s = sched();
  const int id = s.cur;
  (*s.body)();
  s.finished[id] = true;
  // (returns to main_ctx through uc_link)
}

void yield() {
  Sched& s = sched();
  const int id = s.cur;
  swapcontext(&s.ctx[id], &s.main_ctx);
  set_ids(s, id);   // (thread-locals are shared by all fibers of this OS thread)
}

void wait_at(Barrier& b) {
  if (++b.arrived >= b.expected) {
    b.arrived = 0;
    ++b.gen;
    return;
  }
  Sched& s = sched();
  const int id = s.cur;
  s.state[id] = &b == &s.block_barrier ? 1 : 2;
  const uint64_t g = b.gen;
  s.wait_b[id] = &b;
  s.wait_gen[id] = g;
  while (b.gen == g) yield();
  s.wait_b[id] = nullptr;
  s.state[id] = 0;
}

// HOSTSIM_WATCHDOG=<seconds>: a launch that is still running then reports where its fibers wait, wave by wave, and aborts
void watchdog(int) {
  Sched& s = sched();
  char buf[256];
  int n = std::snprintf(buf, sizeof(buf), "[hipsim] watchdog: %s block %u, %d threads, current fiber %d\n", s.kernel_name, s.block.x, s.n, s.cur);
  (void)!write(2, buf, (size_t)n);
  for (int w = 0; w * kWave < s.n; ++w) {
    int c[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < kWave && w * kWave + l < s.n; ++l) ++c[s.state[w * kWave + l] > 4 ? 0 : s.state[w * kWave + l]];
    n = std::snprintf(buf, sizeof(buf), "  wave %2d: running %d, block barrier %d, wave barrier %d (arrived %d of %d), yielding %d, done %d\n", w,
                      c[0], c[1], c[2], s.wave_barrier[w].arrived, s.wave_barrier[w].expected, c[3], c[4]);
    (void)!write(2, buf, (size_t)n);
    if (c[4] && c[4] < kWave) {   // which lanes have left
      unsigned long long m = 0;
      for (int l = 0; l < kWave && w * kWave + l < s.n; ++l) m |= (unsigned long long)(s.state[w * kWave + l] == 4) << l;
      n = std::snprintf(buf, sizeof(buf), "           lanes done: %016llx\n", m);
      (void)!write(2, buf, (size_t)n);
    }
  }
  _exit(97);
}

void run_block_fibers(Sched& s, int n_threads) {
  s.block_barrier.reset(n_threads);
  for (int w = 0; w * kWave < n_threads; ++w) s.wave_barrier[w].reset(std::min(kWave, n_threads - w * kWave));
  for (int t = 0; t < n_threads; ++t) {
    if (!s.stacks[t]) s.stacks[t] = (char*)std::malloc(kStack);
    getcontext(&s.ctx[t]);
    s.ctx[t].uc_stack.ss_sp = s.stacks[t];
    s.ctx[t].uc_stack.ss_size = kStack;
    s.ctx[t].uc_link = &s.main_ctx;
    makecontext(&s.ctx[t], trampoline, 0);
    s.finished[t] = false;
    s.state[t] = 0;
    s.wait_b[t] = nullptr;
  }
  s.n = n_threads;
  std::vector<char> gone((size_t)n_threads, 0);
  int alive = n_threads;
  s.in_fibers = true;
  while (alive > 0) {
    int resumed = 0;
    for (int t = 0; t < n_threads; ++t) {
      if (gone[t]) continue;
      if (s.wait_b[t] && s.wait_b[t]->gen == s.wait_gen[t]) continue;  // still parked: resuming it would only yield again
      ++resumed;
      s.cur = t;
      set_ids(s, t);
      swapcontext(&s.main_ctx, &s.ctx[t]);
      if (s.finished[t]) {
        s.state[t] = 4;
        gone[t] = 1;
        --alive;
        s.block_barrier.drop();
        s.wave_barrier[t / kWave].drop();
      }
    }
    if (!resumed) {  // every live fiber is parked at a barrier nobody will complete: the kernel deadlocked
      std::fprintf(stderr, "[hipsim] deadlock: all %d live fibers are parked at barriers\n", alive);
      watchdog(0);
    }
  }
  s.in_fibers = false;
}

int flat_tid() { return (int)(t_thread.x + t_block_dim.x * (t_thread.y + t_block_dim.y * t_thread.z)); }

}  // namespace

void launch(const char* name, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  std::lock_guard<std::mutex> lg(g_launch_mu);
  Sched& s = sched();
  const int n_threads = (int)(block.x * block.y * block.z);
  if (n_threads < 1 || n_threads > kMaxThreads) {
    std::fprintf(stderr, "hipsim: block of %d threads\n", n_threads);
    std::abort();
  }
  s.body = &body;
  s.block_dim = Idx{block.x, block.y, block.z};
  s.grid_dim = Idx{grid.x, grid.y, grid.z};
  if (s.dyn.size() < shmem + 64) s.dyn.resize(shmem + 64);
  const bool fibers = needs_fibers(name);
  static const bool trace = std::getenv("HOSTSIM_TRACE") != nullptr;   // (test infrastructure: which launch is running)
  s.kernel_name = name;
  static const int wd = std::getenv("HOSTSIM_WATCHDOG") ? std::atoi(std::getenv("HOSTSIM_WATCHDOG")) : 0;
  if (wd > 0) {
    std::signal(SIGALRM, watchdog);
    alarm((unsigned)wd);
  }
  if (trace) std::fprintf(stderr, "[hipsim] %s grid %u block %u lds %zu %s\n", name, grid.x, block.x, shmem, fibers ? "fibers" : "plain");
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.block = Idx{x, y, z};
        if (fibers) {
          run_block_fibers(s, n_threads);
        } else {
          for (int t = 0; t < n_threads; ++t) {
            s.cur = t;
            set_ids(s, t);
            body();
          }
        }
      }
  if (wd > 0) alarm(0);
}

static void not_in_fibers(const char* what) {
  std::fprintf(stderr, "hipsim: %s in a kernel that was classified barrier-free\n", what);
  std::abort();
}

void sync_threads() {
  Sched& s = sched();
  if (!s.in_fibers) not_in_fibers("__syncthreads");
  wait_at(s.block_barrier);
}

uint64_t wave_exchange(uint64_t v, int delta) {
  Sched& s = sched();
  if (!s.in_fibers) not_in_fibers("a wave shuffle");
  const int tid = flat_tid(), lane = tid % kWave, wave = tid / kWave;
  const int n = (int)(s.block_dim.x * s.block_dim.y * s.block_dim.z);
  const int wave_size = std::min(kWave, n - wave * kWave);
  s.xchg[tid] = v;
  wait_at(s.wave_barrier[wave]);
  const uint64_t r = (lane + delta < wave_size && lane + delta >= 0) ? s.xchg[tid + delta] : v;
  wait_at(s.wave_barrier[wave]);
  return r;
}

unsigned long long wave_ballot(bool pred) {
  Sched& s = sched();
  if (!s.in_fibers) not_in_fibers("__ballot");
  const int tid = flat_tid(), lane = tid % kWave, wave = tid / kWave;
  if (lane == 0) s.ballot[wave] = 0;
  wait_at(s.wave_barrier[wave]);
  if (pred) s.ballot[wave] |= 1ull << lane;
  wait_at(s.wave_barrier[wave]);
  const unsigned long long r = s.ballot[wave];
  wait_at(s.wave_barrier[wave]);
  return r;
}

uint64_t wave_exchange_abs(uint64_t v, int src_lane) {
  Sched& s = sched();
  if (!s.in_fibers) not_in_fibers("a wave shuffle");
  const int tid = flat_tid(), wave = tid / kWave;
  const int n = (int)(s.block_dim.x * s.block_dim.y * s.block_dim.z);
  const int wave_size = std::min(kWave, n - wave * kWave);
  s.xchg[tid] = v;
  wait_at(s.wave_barrier[wave]);
  const uint64_t r = (src_lane >= 0 && src_lane < wave_size) ? s.xchg[wave * kWave + src_lane] : v;
  wait_at(s.wave_barrier[wave]);
  return r;
}

uint64_t wave_permute_push(int dst_lane, uint64_t v) {
  Sched& s = sched();
  if (!s.in_fibers) not_in_fibers("ds_permute");
  const int tid = flat_tid(), wave = tid / kWave;
  const int n = (int)(s.block_dim.x * s.block_dim.y * s.block_dim.z);
  const int wave_size = std::min(kWave, n - wave * kWave);
  s.recv[tid] = 0;
  wait_at(s.wave_barrier[wave]);
  if (dst_lane >= 0 && dst_lane < wave_size) s.recv[wave * kWave + dst_lane] = v;
  wait_at(s.wave_barrier[wave]);
  const uint64_t r = s.recv[tid];
  wait_at(s.wave_barrier[wave]);
  return r;
}

void wave_sync() {
  Sched& s = sched();
  if (!s.in_fibers) return;
  wait_at(s.wave_barrier[flat_tid() / kWave]);
}

void fiber_yield() {
  Sched& s = sched();
  if (!s.in_fibers) return;
  const int id = s.cur;
  s.state[id] = 3;
  yield();
  s.state[id] = 0;
}

int lane_id() { return flat_tid() % kWave; }

void* dynamic_shared() {
  Sched& s = sched();
  return (void*)(((uintptr_t)s.dyn.data() + 63) & ~(uintptr_t)63);
}

}  // namespace hipsim

// ------------------------------------------------------------------------------------------- runtime API
namespace {
std::mutex g_mem_mu;
std::map<void*, size_t> g_allocs;
size_t g_used = 0;
constexpr size_t kTotal = (size_t)8 << 30;
size_t g_fail_after = ~(size_t)0;   // hostsim_fail_allocs_after: make hipMalloc fail (out-of-memory paths)
thread_local int t_device = 0;
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

struct ihipEvent_t {
  double t;
};

extern "C" {

void hostsim_fail_allocs_larger_than(size_t bytes) { g_fail_after = bytes; }
size_t hostsim_bytes_in_use() { return g_used; }
int hostsim_live_allocations() { return (int)g_allocs.size(); }

hipError_t hipMalloc(void** p, size_t bytes) {
  if (!p) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> lk(g_mem_mu);
  if (bytes > g_fail_after || g_used + bytes > kTotal) {
    *p = nullptr;
    return hipErrorOutOfMemory;
  }
  const size_t n = (bytes + 255) & ~(size_t)255;
  void* q = nullptr;
  if (posix_memalign(&q, 256, n ? n : 256)) return hipErrorOutOfMemory;
  std::memset(q, 0xA5, n ? n : 256);
  g_allocs[q] = n;
  g_used += n;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(g_mem_mu);
  auto it = g_allocs.find(p);
  if (it == g_allocs.end()) {
    std::fprintf(stderr, "hipsim: hipFree of an unknown pointer %p\n", p);
    std::abort();
  }
  std::memset(p, 0x5A, it->second);
  g_used -= it->second;
  g_allocs.erase(it);
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  if (bytes) std::memmove(dst, src, bytes);
  return hipSuccess;
}
static std::atomic<int> g_h2d_async{0};
extern "C" int hostsim_h2d_async_copies() { return g_h2d_async.load(); }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t) {
  if (k == hipMemcpyHostToDevice) ++g_h2d_async;
  return hipMemcpy(dst, src, bytes, k);
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) {
  if (bytes) std::memset(dst, value, bytes);
  return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes) { return hipMemsetAsync(dst, value, bytes, nullptr); }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  if (free_b) *free_b = kTotal - g_used;
  if (total_b) *total_b = kTotal;
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = (hipStream_t) new int(0);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete (int*)s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new ihipEvent_t{0.0};
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = std::malloc(bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
  *dev = host;
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
// a launch runs when it is enqueued, so whatever an event stands for has already happened
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  if (!e) return hipErrorInvalidValue;
  e->t = now_ms();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a || !b) return hipErrorInvalidValue;
  *ms = (float)(b->t - a->t);
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = t_device;
  return hipSuccess;
}
hipError_t hipSetDevice(int d) {
  if (d != 0) return hipErrorInvalidValue;
  t_device = d;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
  *v = 8;  // "compute units": keeps the grids of the simulated launches small (8: the L2 probes spread over 8 XCDs)
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  std::snprintf(p->name, sizeof(p->name), "hostsim");
  std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "x86-64");
  p->totalGlobalMem = kTotal;
  p->multiProcessorCount = 8;
  p->warpSize = 64;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hostsim error"; }
}
