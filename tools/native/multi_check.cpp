// multi_check.cpp — native multi-device driver: INTEGRATION.md section 5 verbatim over the C-ABI, RCCL called
// directly (ncclSend / ncclRecv grouped), no Python and no torch.  It shows that what the library exports per
// device — mi355q_execute_async, mi355q_shard_pads, mi355q_shard_merge_slices / _merge_range — is sufficient for
// the reference's multi-device step (one kernel per device over `fragment % n_devices`, Execute.cpp:3080-3102,
// then Executor::reduceMultiDeviceResultSets, Execute.cpp:1772-1792) with the merge done ON the devices:
//
//   one host thread per GPU (the process-per-GPU deployment uses the same calls with ncclCommInitRank across
//   processes):
//     generate the rank's fragments of the headline table (fragment f belongs to rank f % world,
//       InsertOrderFragmenter.cpp:435-443)
//     mi355q_execute_async(plan, my fragments)              -> my partial 640 MB-class table, stream-ordered
//     mi355q_shard_pads(table, world, 1024, pads, ok)       -> the rows after each slice's end
//     ncclGroupStart: for r: ncclSend(slice r of MY table -> r); ncclRecv(slice `me` of r's table)   (in place)
//     ncclGroupStart: for r: ncclSend(pad r -> r); ncclRecv(pad from r)
//     ncclAllReduce(ok, min)
//     mi355q_shard_merge_slices(fresh, slices[], pads[], world, 1024, lo, hi)   (UNSUPPORTED -> _merge_range)
//     mi355q_wait(pending)
//   checks (size-independent): every rank's table only holds keys whose home slot is in its range; the row counts
//   add up to the key cardinality; SUM over all tables of COUNT(*) == rows passing the filter (counted by a second,
//   non-grouped step per rank and all-reduced).
//
// `multi_check --sim N [rows] [keys]`: N ranks SIMULATED ON ONE DEVICE — every rank its own host thread, stream, fragments
//   (f % N), tables and receive buffers; the grouped ncclSend / ncclRecv of the slice and pad exchanges become device-to-device
//   copies enqueued on the receiver's stream behind an event of the sender's stream (LoopComm below), the all-reduces go
//   through the host.  So the N x N slices really move and every rank folds N sources — what a 1-GPU box degenerates to a
//   no-op with RCCL.  The library keeps ONE workspace per device, so the ranks take turns for the calls that use it
//   (the step, mi355q_shard_merge_slices); timing means nothing in this mode.
//
// Build (tools/native/build_multi_check.sh):  hipcc -O2 -std=c++17 multi_check.cpp -I../../include
//        -L../../heavydb_amd/lib -lmi355q -lrccl -lpthread ;  run:  multi_check [n_gpus] [total_rows]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mi355q.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("[rank %d] HIP error %s at line %d\n", rank, hipGetErrorString(e_), __LINE__); failures.fetch_add(1); return; } } while (0)
#define NCCK(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { std::printf("[rank %d] RCCL error %s at line %d\n", rank, ncclGetErrorString(e_), __LINE__); failures.fetch_add(1); return; } } while (0)
#define MQCK(x) do { int32_t c_ = (x); if (c_) { std::printf("[rank %d] mi355q error %d (%s) at line %d\n", rank, c_, mi355q_error_string(c_), __LINE__); failures.fetch_add(1); return; } } while (0)
#define REQ(c) do { if (!(c)) { std::printf("[rank %d] FAILED line %d: %s\n", rank, __LINE__, #c); failures.fetch_add(1); } } while (0)

namespace {

constexpr int64_t kFragRows = 32000000;
constexpr int kPadRows = 1024;
std::atomic<int> failures{0};

uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t murmur3_u64(uint64_t key) {  // MurmurHash3_x86_32 of the 8 key bytes, seed 0 (GroupByRuntime.cpp:20-23)
  uint32_t h1 = 0;
  for (int i = 0; i < 2; ++i) {
    uint32_t k1 = (uint32_t)(key >> (32 * i)) * 0xcc9e2d51u;
    k1 = rotl32(k1, 15) * 0x1b873593u;
    h1 ^= k1;
    h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  }
  h1 ^= 8u;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

// ---- the loopback transport of --sim ------------------------------------------------------------------------------
struct HostBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, waiting = 0;
  uint64_t gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    } else {
      // (a rank that failed has left: the others must not wait for it for ever)
      while (!cv.wait_for(lk, std::chrono::milliseconds(200), [&] { return gen != g; }))
        if (failures.load()) break;
    }
  }
};
struct LoopComm {
  int world = 0;
  HostBarrier bar;
  std::mutex device_turn;                      // one workspace per device in the library: ranks take turns
  std::vector<const char*> send_ptr;           // [from][to]
  std::vector<size_t> send_bytes;              // [from][to]
  std::vector<hipEvent_t> ready, done;         // [rank]: "my sends may be read", "I have read what I receive"
  std::vector<int32_t> flags;                  // [rank][n] for the host all-reduces
  int flags_n = 0;
};
LoopComm* g_loop = nullptr;

// rank `me` sends `bytes[r]` from send[r] to every r and receives recv_bytes into recv[r] from every r (one grouped
// ncclSend / ncclRecv round); returns false on a HIP error
bool loop_all_to_all(LoopComm& c, int me, const std::vector<const char*>& send, const std::vector<size_t>& bytes,
                     const std::vector<char*>& recv, hipStream_t s) {
  for (int r = 0; r < c.world; ++r) {
    c.send_ptr[(size_t)me * c.world + r] = send[r];
    c.send_bytes[(size_t)me * c.world + r] = bytes[r];
  }
  if (hipEventRecord(c.ready[me], s) != hipSuccess) return false;
  c.bar.wait();
  for (int r = 0; r < c.world; ++r) {
    if (hipStreamWaitEvent(s, c.ready[r], 0) != hipSuccess) return false;
    const size_t n = c.send_bytes[(size_t)r * c.world + me];
    if (n && hipMemcpyAsync(recv[r], c.send_ptr[(size_t)r * c.world + me], n, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;
  }
  if (hipEventRecord(c.done[me], s) != hipSuccess) return false;
  c.bar.wait();
  for (int r = 0; r < c.world; ++r)   // my send buffers are free again once every reader is done
    if (hipStreamWaitEvent(s, c.done[r], 0) != hipSuccess) return false;
  return true;
}
// element-wise MIN (or MAX) of n int32 over the ranks, through the host
bool loop_all_reduce(LoopComm& c, int me, int32_t* d, int n, bool take_max, hipStream_t s) {
  std::vector<int32_t> h((size_t)n);
  if (hipMemcpyAsync(h.data(), d, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess) return false;
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  c.bar.wait();            // (the previous round's readers are done with `flags`)
  if (me == 0) {
    c.flags.assign((size_t)c.world * n, 0);
    c.flags_n = n;
  }
  c.bar.wait();
  for (int i = 0; i < n; ++i) c.flags[(size_t)me * n + i] = h[i];
  c.bar.wait();
  for (int i = 0; i < n; ++i) {
    int32_t v = c.flags[i];
    for (int r = 1; r < c.world; ++r) {
      const int32_t x = c.flags[(size_t)r * n + i];
      v = take_max ? std::max(v, x) : std::min(v, x);
    }
    h[i] = v;
  }
  if (hipMemcpyAsync(d, h.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess) return false;
  return hipStreamSynchronize(s) == hipSuccess;
}
#define LOOPCK(x) do { if (!(x)) { std::printf("[rank %d] loopback transport failed at line %d\n", rank, __LINE__); failures.fetch_add(1); return; } } while (0)

mi355q_plan headline_plan(int64_t n_keys, bool grouped) {
  mi355q_plan p{};
  p.abi_version = MI355Q_ABI_VERSION;
  p.n_cols = 3;
  p.cols[0] = {MI355Q_INT64, 0, 0, 0};
  p.cols[1] = {MI355Q_DOUBLE, 0, 0, 0};
  p.cols[2] = {MI355Q_INT32, 0, 0, 0};
  p.col_ranges[0] = {1, 0, 7, (n_keys - 1) * 1000003 + 7, 0, 0, 0};
  p.col_ranges[1] = {1, 0, 0, 0, 0.0, 1000.0, 0};
  p.col_ranges[2] = {1, 0, 0, INT32_MAX, 0, 0, 0};
  p.n_quals = 1;
  p.quals[0] = {2, MI355Q_LT, 1 << 30, 0.0};
  p.join_outer_col = -1;
  p.max_groups_buffer_entry_guess = 2 * n_keys;
  if (grouped) {
    p.n_group_cols = 1;
    p.group_cols[0] = 0;
    p.n_targets = 3;
    p.targets[0] = {MI355Q_PROJECT_KEY, 0, 0, 0, {}};
    p.targets[1] = {MI355Q_COUNT, -1, 0, 0, {}};
    p.targets[2] = {MI355Q_AVG, 1, 0, 0, {}};
  } else {
    p.n_targets = 1;
    p.targets[0] = {MI355Q_COUNT, -1, 0, 0, {}};
  }
  return p;
}

void rank_main(int rank, int world, int64_t total_rows, int64_t n_keys, ncclUniqueId id, int64_t* group_counts,
               double* step_ms) {
  const bool sim = g_loop != nullptr;   // N ranks on device 0
  const int dev = sim ? 0 : rank;
  HIPCK(hipSetDevice(dev));
  ncclComm_t comm = nullptr;
  if (!sim) NCCK(ncclCommInitRank(&comm, world, id, rank));
  hipStream_t s;
  HIPCK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // ---- this rank's fragments of the table
  std::vector<int64_t> frag_off, frag_rows;
  for (int64_t f = 0, off = 0; off < total_rows; ++f, off += kFragRows)
    if (f % world == rank) {
      frag_off.push_back(off);
      frag_rows.push_back(std::min(kFragRows, total_rows - off));
    }
  int64_t local = 0;
  for (int64_t r : frag_rows) local += r;
  void *d_key = nullptr, *d_val = nullptr, *d_fil = nullptr;
  HIPCK(hipMalloc(&d_key, (size_t)std::max<int64_t>(local, 1) * 8));
  HIPCK(hipMalloc(&d_val, (size_t)std::max<int64_t>(local, 1) * 8));
  HIPCK(hipMalloc(&d_fil, (size_t)std::max<int64_t>(local, 1) * 4));
  std::vector<const void*> cols;
  int64_t lo_row = 0;
  for (size_t i = 0; i < frag_rows.size(); ++i) {
    char *k = (char*)d_key + lo_row * 8, *v = (char*)d_val + lo_row * 8, *fl = (char*)d_fil + lo_row * 4;
    MQCK(mi355q_generate_column(dev, k, frag_rows[i], frag_off[i], MI355Q_GEN_I64_MOD_MUL, 0xC0FFEE00, n_keys, 1000003, 7, 0.0, 0, s));
    MQCK(mi355q_generate_column(dev, v, frag_rows[i], frag_off[i], MI355Q_GEN_F64_UNIT, 0xC0FFEE01, 0, 0, 0, 1000.0, 0, s));
    MQCK(mi355q_generate_column(dev, fl, frag_rows[i], frag_off[i], MI355Q_GEN_I32_UNIFORM31, 0xC0FFEE02, 0, 0, 0, 0.0, 0, s));
    cols.insert(cols.end(), {k, v, fl});
    lo_row += frag_rows[i];
  }
  HIPCK(hipStreamSynchronize(s));
  mi355q_inputs in{};
  in.device_id = dev;
  in.n_frags = (int32_t)frag_rows.size();
  in.col_buffers = cols.data();
  in.num_rows = frag_rows.data();

  const mi355q_plan plan = headline_plan(n_keys, true);
  mi355q_qmd q;
  MQCK(mi355q_qmd_init(&plan, &q));
  const int64_t E = q.entry_count, row_bytes = q.row_size, rq = row_bytes / 8;
  std::vector<int64_t> bound(world + 1);
  for (int r = 0; r <= world; ++r) bound[r] = (int64_t)r * E / world;
  const int64_t my_lo = bound[rank], my_hi = bound[rank + 1], my_len = my_hi - my_lo;
  void *table = nullptr, *fresh_buf = nullptr, *recv = nullptr, *pads = nullptr, *recv_pads = nullptr;
  int32_t* ok = nullptr;
  HIPCK(hipMalloc(&table, (size_t)(E * row_bytes)));
  HIPCK(hipMalloc(&fresh_buf, (size_t)(E * row_bytes)));
  HIPCK(hipMalloc(&recv, (size_t)((int64_t)world * my_len * row_bytes)));
  HIPCK(hipMalloc(&pads, (size_t)((int64_t)world * kPadRows * row_bytes)));
  HIPCK(hipMalloc(&recv_pads, (size_t)((int64_t)world * kPadRows * row_bytes)));
  HIPCK(hipMalloc((void**)&ok, sizeof(int32_t) * (size_t)(world + 1)));

  mi355q_exec_options opts{};
  opts.stream = s;
  opts.out_buffer = table;
  // size-independent checks of this rank's share of the final table
  auto check_table = [&](const mi355q_result* mine, const mi355q_exec_report& rep) {
    std::vector<int64_t> h((size_t)(E * rq));
    MQCK(mi355q_result_copy_to_host(mine, h.data(), E * row_bytes));
    int64_t groups = 0, counted = 0;
    bool owned = true;
    for (int64_t e = 0; e < E; ++e) {
      if (h[e * rq] == INT64_MAX) continue;
      ++groups;
      counted += h[e * rq + 1];
      const int64_t home = murmur3_u64((uint64_t)h[e * rq]) % (uint64_t)E;
      owned = owned && (world == 1 || (home >= my_lo && home < my_hi));
    }
    REQ(owned);
    group_counts[rank] = groups;
    group_counts[world + rank] = counted;
    std::printf("[rank %d] %lld rows in %zu fragments: step + merge %.3f ms (kernel %s, %d launches, %.3f ms on the stream), %lld groups in my range\n",
                rank, (long long)local, frag_rows.size(), step_ms[rank], rep.kernel_name, rep.n_launches, rep.total_ms, (long long)groups);
  };
  for (int iter = 0; iter < 3; ++iter) {   // iteration 0 warms the workspace up; the last one is timed
    if (sim) g_loop->bar.wait();
    else NCCK(ncclAllReduce(ok, ok, 1, ncclInt32, ncclMin, comm, s));   // a barrier on the stream
    HIPCK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    mi355q_result* res = nullptr;
    mi355q_pending* pend = nullptr;
    mi355q_result* fresh = nullptr;
    // --sim: the same sequence with the loopback transport; the calls that use the library's per-device workspace are
    // made in turns and completed (the N ranks share device 0)
    auto exchange_sim = [&]() {
      {
        std::lock_guard<std::mutex> turn(g_loop->device_turn);
        MQCK(mi355q_shard_pads(res, world, kPadRows, pads, ok, s));
        HIPCK(hipStreamSynchronize(s));
      }
      std::vector<const char*> snd(world);
      std::vector<size_t> nb(world);
      std::vector<char*> rcv(world);
      for (int r = 0; r < world; ++r) {
        snd[r] = (const char*)table + bound[r] * row_bytes;
        nb[r] = (size_t)((bound[r + 1] - bound[r]) * row_bytes);
        rcv[r] = (char*)recv + (int64_t)r * my_len * row_bytes;
      }
      LOOPCK(loop_all_to_all(*g_loop, rank, snd, nb, rcv, s));
      for (int r = 0; r < world; ++r) {
        snd[r] = (const char*)pads + (int64_t)r * kPadRows * row_bytes;
        nb[r] = (size_t)(kPadRows * row_bytes);
        rcv[r] = (char*)recv_pads + (int64_t)r * kPadRows * row_bytes;
      }
      LOOPCK(loop_all_to_all(*g_loop, rank, snd, nb, rcv, s));
      LOOPCK(loop_all_reduce(*g_loop, rank, ok, world, false, s));
      std::lock_guard<std::mutex> turn(g_loop->device_turn);
      MQCK(mi355q_result_create(&q, dev, fresh_buf, &fresh));
      std::vector<const void*> slices(world), padv(world);
      for (int r = 0; r < world; ++r) {
        slices[r] = (const char*)recv + (int64_t)r * my_len * row_bytes;
        padv[r] = (const char*)recv_pads + (int64_t)r * kPadRows * row_bytes;
      }
      int32_t e = mi355q_shard_merge_slices(fresh, slices.data(), padv.data(), world, kPadRows, my_lo, my_hi, s);
      if (e == MI355Q_ERR_UNSUPPORTED) {
        MQCK(mi355q_shard_merge_range(fresh, recv, (int64_t)world * my_len, my_lo, my_hi, s));
        MQCK(mi355q_shard_merge_range(fresh, recv_pads, (int64_t)world * kPadRows, my_lo, my_hi, s));
      } else {
        MQCK(e);
      }
      HIPCK(hipStreamSynchronize(s));
    };
    if (sim) {
      mi355q_exec_report rep{};
      {
        std::lock_guard<std::mutex> turn(g_loop->device_turn);
        MQCK(mi355q_execute_async(&plan, &in, &opts, &res, &pend));
        int32_t wc = mi355q_wait(pend, &rep);
        if (wc == MI355Q_STEP_RECOMPUTED) wc = MI355Q_OK;   // nothing was enqueued behind the step yet
        MQCK(wc);
        HIPCK(hipStreamSynchronize(s));
      }
      if (world > 1) exchange_sim();
      step_ms[rank] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (iter == 2) {
        std::vector<int32_t> h_ok(world, 1);
        if (world > 1) HIPCK(hipMemcpy(h_ok.data(), ok, sizeof(int32_t) * world, hipMemcpyDeviceToHost));
        for (int r = 0; r < world; ++r) REQ(h_ok[r] == 1);
        std::lock_guard<std::mutex> turn(g_loop->device_turn);
        check_table(world > 1 ? fresh : res, rep);
      }
      g_loop->bar.wait();   // nobody re-runs its step into `table` while a peer may still be reading a slice of it
      if (fresh) mi355q_result_free(fresh);
      mi355q_result_free(res);
      continue;
    }
    MQCK(mi355q_execute_async(&plan, &in, &opts, &res, &pend));
    auto exchange = [&]() {
      MQCK(mi355q_shard_pads(res, world, kPadRows, pads, ok, s));
      NCCK(ncclGroupStart());
      for (int r = 0; r < world; ++r) {
        NCCK(ncclSend((const char*)table + bound[r] * row_bytes, (size_t)((bound[r + 1] - bound[r]) * rq), ncclInt64, r, comm, s));
        NCCK(ncclRecv((char*)recv + (int64_t)r * my_len * row_bytes, (size_t)(my_len * rq), ncclInt64, r, comm, s));
      }
      NCCK(ncclGroupEnd());
      NCCK(ncclGroupStart());
      for (int r = 0; r < world; ++r) {
        NCCK(ncclSend((const char*)pads + (int64_t)r * kPadRows * row_bytes, (size_t)(kPadRows * rq), ncclInt64, r, comm, s));
        NCCK(ncclRecv((char*)recv_pads + (int64_t)r * kPadRows * row_bytes, (size_t)(kPadRows * rq), ncclInt64, r, comm, s));
      }
      NCCK(ncclGroupEnd());
      // ok[r] = 1 when pad r of MY table ends in an empty slot; every rank needs min over ranks and pads
      NCCK(ncclAllReduce(ok, ok, world, ncclInt32, ncclMin, comm, s));
      MQCK(mi355q_result_create(&q, dev, fresh_buf, &fresh));
      std::vector<const void*> slices(world), padv(world);
      for (int r = 0; r < world; ++r) {
        slices[r] = (const char*)recv + (int64_t)r * my_len * row_bytes;
        padv[r] = (const char*)recv_pads + (int64_t)r * kPadRows * row_bytes;
      }
      int32_t e = mi355q_shard_merge_slices(fresh, slices.data(), padv.data(), world, kPadRows, my_lo, my_hi, s);
      if (e == MI355Q_ERR_UNSUPPORTED) {
        MQCK(mi355q_shard_merge_range(fresh, recv, (int64_t)world * my_len, my_lo, my_hi, s));
        MQCK(mi355q_shard_merge_range(fresh, recv_pads, (int64_t)world * kPadRows, my_lo, my_hi, s));
      } else {
        MQCK(e);
      }
    };
    if (world > 1) exchange();
    mi355q_exec_report rep{};
    int32_t wc = mi355q_wait(pend, &rep);
    // MI355Q_STEP_RECOMPUTED: the step was re-run inside the wait (spill overflow), so the pads and slices enqueued
    // behind the first launches came from the abandoned table.  The exchange is a collective: every rank redoes it
    // when any rank has to (flag agreed on with one small all-reduce).
    int32_t* redo = ok + world;
    int32_t h_redo = wc == MI355Q_STEP_RECOMPUTED ? 1 : 0;
    if (wc == MI355Q_STEP_RECOMPUTED) wc = MI355Q_OK;
    MQCK(wc);
    if (world > 1) {
      HIPCK(hipMemcpyAsync(redo, &h_redo, sizeof(int32_t), hipMemcpyHostToDevice, s));
      NCCK(ncclAllReduce(redo, redo, 1, ncclInt32, ncclMax, comm, s));
      HIPCK(hipMemcpyAsync(&h_redo, redo, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      HIPCK(hipStreamSynchronize(s));
      if (h_redo) {
        mi355q_result_free(fresh);
        fresh = nullptr;
        exchange();
      }
    }
    HIPCK(hipStreamSynchronize(s));
    step_ms[rank] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (iter == 2) {
      std::vector<int32_t> h_ok(world, 1);
      if (world > 1) HIPCK(hipMemcpy(h_ok.data(), ok, sizeof(int32_t) * world, hipMemcpyDeviceToHost));
      for (int r = 0; r < world; ++r) REQ(h_ok[r] == 1);   // else: the general partition path (mi355q_shard_partition)
      check_table(world > 1 ? fresh : res, rep);
    }
    if (fresh) mi355q_result_free(fresh);
    mi355q_result_free(res);
  }
  // rows passing the filter on this rank, by the non-grouped scan kernel
  {
    const mi355q_plan cp = headline_plan(n_keys, false);
    mi355q_result* cr = nullptr;
    mi355q_exec_options co{};
    co.stream = s;
    std::unique_lock<std::mutex> turn;
    if (sim) turn = std::unique_lock<std::mutex>(g_loop->device_turn);
    MQCK(mi355q_execute(&cp, &in, &co, &cr, nullptr));
    int64_t c = 0;
    MQCK(mi355q_result_copy_to_host(cr, &c, 8));
    group_counts[2 * world + rank] = c;
    mi355q_result_free(cr);
  }
  if (comm) ncclCommDestroy(comm);
  (void)hipFree(table); (void)hipFree(fresh_buf); (void)hipFree(recv); (void)hipFree(pads); (void)hipFree(recv_pads); (void)hipFree(ok);
  (void)hipFree(d_key); (void)hipFree(d_val); (void)hipFree(d_fil);
  if (sim) g_loop->bar.wait();
  if (!sim || rank == 0) (void)mi355q_release_workspace(dev);
}

}  // namespace

int main(int argc, char** argv) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
    std::printf("multi_check: no GPU\n");
    return 77;
  }
  int arg0 = 1;
  int sim_ranks = 0;
  if (argc > 2 && std::strcmp(argv[1], "--sim") == 0) {
    sim_ranks = std::max(1, std::atoi(argv[2]));
    arg0 = 3;
  }
  const int world = sim_ranks ? sim_ranks : (argc > arg0 ? std::min(std::atoi(argv[arg0]), n_dev) : n_dev);
  if (!sim_ranks && argc > arg0) ++arg0;
  const int64_t total_rows = argc > arg0 ? (int64_t)std::atof(argv[arg0]) : 256000000;
  const int64_t n_keys = argc > arg0 + 1 ? (int64_t)std::atof(argv[arg0 + 1]) : 2000000;
  std::printf("multi_check: %d %s, %lld rows, %lld keys\n", world, sim_ranks ? "rank(s) simulated on device 0" : "device(s)",
              (long long)total_rows, (long long)n_keys);
  LoopComm loop;
  if (sim_ranks) {
    if (hipSetDevice(0) != hipSuccess) return 1;
    loop.world = world;
    loop.bar.n = world;
    loop.send_ptr.assign((size_t)world * world, nullptr);
    loop.send_bytes.assign((size_t)world * world, 0);
    loop.ready.resize(world);
    loop.done.resize(world);
    for (int r = 0; r < world; ++r)
      if (hipEventCreateWithFlags(&loop.ready[r], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&loop.done[r], hipEventDisableTiming) != hipSuccess)
        return 1;
    g_loop = &loop;
  }
  ncclUniqueId id{};
  if (!sim_ranks && ncclGetUniqueId(&id) != ncclSuccess) {
    std::printf("multi_check: ncclGetUniqueId failed\n");
    return 1;
  }
  std::vector<int64_t> counts((size_t)3 * world, 0);
  std::vector<double> ms((size_t)world, 0.0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r, world, total_rows, n_keys, id, counts.data(), ms.data());
  for (auto& t : th) t.join();
  int64_t groups = 0, counted = 0, passing = 0;
  double worst = 0.0;
  for (int r = 0; r < world; ++r) {
    groups += counts[r];
    counted += counts[world + r];
    passing += counts[2 * world + r];
    worst = std::max(worst, ms[r]);
  }
  std::printf("multi_check: %lld groups over all ranks (expected %lld), SUM(COUNT(*)) %lld == rows passing the filter %lld, "
              "slowest rank %.3f ms -> %.3g rows/s\n", (long long)groups, (long long)std::min(n_keys, total_rows), (long long)counted,
              (long long)passing, worst, total_rows / (worst * 1e-3));
  if (groups != std::min(n_keys, total_rows) && total_rows >= 50 * n_keys) failures.fetch_add(1);
  if (counted != passing) failures.fetch_add(1);
  std::printf(failures.load() ? "multi_check: %d FAILURE(S)\n" : "multi_check: ok\n", failures.load());
  return failures.load() ? 1 : 0;
}
