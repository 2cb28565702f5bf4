// libnbody_hip — several GPUs behind one context (nbody_create_multi, include/nbody_hip.h).
//
// The reference steps its world from ONE thread with ONE call, `world.update(STEP_SIZE, &mut counter)`
// (/root/reference src/main.rs:120), whose only parallel region is a map over targets with read-only sources
// (main.rs:406-416).  So the step shards by target, and the host keeps its single call: the handle returned by
// nbody_create_multi fronts G devices of one node, every nbody_update_* on it runs on all of them, and the one
// real exchange of a step — the new positions (direct sum) or the walked slice's rows (Barnes-Hut) — happens inside
// the library as an all-gather over xGMI.
//
//   * One process, one worker thread per device (each owns that device's nbody_ctx: a full replica of the rows, 20 B
//     per body — 336 MB at N = 16.7 M, nothing against 288 GB), so phases that wait on their device (tree builds)
//     overlap across devices and RCCL sees one caller per rank.
//   * Direct sum.  Device d owns the target blocks {c*G + d : c < C} of `block` bodies each.  Block c*G + d is computed
//     straight into its place in the NEXT position array, and chunk c = blocks [c*G, (c+1)*G) is then a contiguous
//     region: one in-place ncclAllGather per chunk on a communication stream of its own, ordered after that chunk's
//     kernels by an event, while the compute stream goes on with chunk c+1.  Only the last chunk's gather is exposed.
//     (Velocities never move during direct steps; they are gathered lazily when a call needs whole rows.)
//     Why chunks and not "local sources first" (SURVEY §8e): the step's preparation — hazard scan and near/far split,
//     nearfar.hip — needs ALL positions of the step, so no pair of step k+1 can start before the gather of step k has
//     landed; a local-sources-first pass would have to run with the clamp (+10 % on 1/G of the pairs), which at
//     N = 1 M, G = 8 costs more (0.33 ms) than the 8 MB gather it hides (0.1-0.3 ms).  Chunks cost nothing once a
//     chunk still fills the chip (>= 262 144 targets), so C = 1 at N = 1 M / 8 GPUs and C = 8 at N = 16.7 M / 8 GPUs.
//   * Barnes-Hut.  "Replicas of the tree, shards of the targets": every device builds the same tree (deterministic,
//     on the device), walks and integrates the d-th slice of the tree-ordered targets, and the slices' {row, position,
//     velocity} records travel in ONE packed all-gather.  Bit-identical to the single-device step.
//   * Exchange.  RCCL (ncclCommInitAll + ncclAllGather; librccl is opened at run time, so a host without it still
//     loads this library) — or direct peer copies (hipMemcpyPeerAsync of each block to every peer, the one-shot
//     all-gather of a fully connected xGMI node), which also works with one physical device listed several times:
//     that is how the sharding logic is rehearsed on a one-GPU box (tests/test_gpu_multi.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the pointers of `Rccl`

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "ctx.h"
#include "multi_sync.hpp"

namespace nbody {

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  bool load(std::string* err) {
    if (handle) return true;
    // a copy the process already holds (PyTorch-ROCm bundles one) is reused; otherwise the system's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
    if (!handle)
      for (const char* n : names)
        if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!handle) {
      *err = std::string("librccl could not be opened (") + (dlerror() ? dlerror() : "?") + ")";
      return false;
    }
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    GetVersion = (decltype(GetVersion))dlsym(handle, "ncclGetVersion");
    CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
    if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) {
      *err = "librccl lacks ncclCommInitAll / ncclAllGather";
      handle = nullptr;
      return false;
    }
    return true;
  }
};

constexpr int kMaxChunks = 16;
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The calling thread's current device is its own business: calls that visit several devices put it back.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct Multi {
  int G = 0;
  std::vector<int> dev;
  std::vector<nbody_ctx*> sub;
  int exchange = NBODY_EXCHANGE_RCCL;
  int chunks_wanted = 0;  // 0: by size
  Rccl rccl;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> comm_stream;
  std::vector<std::vector<hipEvent_t>> ev_chunk;  // [device][chunk]: that chunk's kernels are done
  std::vector<hipEvent_t> ev_done;                // [device]: this device's communication stream has drained
  // NBODY_TRACE: timed events of a direct call's LAST step, [device][chunk] — the chunk's kernels have ended (compute stream), its
  // gather begins / has ended (communication stream): says whether the exchange hides behind the next chunk's kernels
  std::vector<std::vector<hipEvent_t>> tr_kend, tr_gbeg, tr_gend;
  Pool pool;
  Barrier barrier;
  // layout of the current upload
  int64_t n = 0;
  bool f64 = false;
  int chunks = 1;
  int64_t block = 0;
  int64_t slice = 0;         // tree steps: rows per device, ceil(n / G)
  bool vel_sharded = false;  // direct steps left each device with the velocities of its own blocks only
  std::vector<void*> posbuf[2];  // [parity][device]: the position arrays of a direct call (read parity, write parity ^ 1)
  std::vector<char*> xbuf;       // [device]: G packed sections {rows u32 | pos | vel} of a tree step's exchange
  size_t xbuf_bytes = 0, xsec = 0, xoff_pos = 0, xoff_vel = 0;

  ~Multi() {
    pool.shutdown();
    for (int d = 0; d < (int)sub.size(); ++d) {
      if (!sub[(size_t)d]) continue;
      (void)hipSetDevice(dev[(size_t)d]);
      if ((size_t)d < comm_stream.size() && comm_stream[(size_t)d]) (void)hipStreamSynchronize(comm_stream[(size_t)d]);
      if ((size_t)d < comm.size() && comm[(size_t)d] && rccl.CommDestroy) (void)rccl.CommDestroy(comm[(size_t)d]);
      if ((size_t)d < xbuf.size() && xbuf[(size_t)d]) (void)hipFree(xbuf[(size_t)d]);
      if ((size_t)d < ev_chunk.size())
        for (auto e : ev_chunk[(size_t)d]) (void)hipEventDestroy(e);
      if ((size_t)d < ev_done.size() && ev_done[(size_t)d]) (void)hipEventDestroy(ev_done[(size_t)d]);
      for (auto* set : {&tr_kend, &tr_gbeg, &tr_gend})
        if ((size_t)d < set->size())
          for (auto e : (*set)[(size_t)d]) (void)hipEventDestroy(e);
      if ((size_t)d < comm_stream.size() && comm_stream[(size_t)d]) (void)hipStreamDestroy(comm_stream[(size_t)d]);
      ctx_destroy_single(sub[(size_t)d]);
    }
  }

  int64_t block_begin(int c, int d) const { return ((int64_t)c * G + d) * block; }
  int64_t block_count(int c, int d) const { return std::clamp<int64_t>(n - block_begin(c, d), 0, block); }
  int64_t local_total(int d) const {
    int64_t t = 0;
    for (int c = 0; c < chunks; ++c) t += block_count(c, d);
    return t;
  }
  int64_t slice_begin(int d) const { return std::min<int64_t>((int64_t)d * slice, n); }
  int64_t slice_count(int d) const { return std::clamp<int64_t>(n - (int64_t)d * slice, 0, slice); }

  int hip_fail(int d, hipError_t e, const char* what) {
    return ctx_fail(sub[(size_t)d], NBODY_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  }
#define MHIP(d, call)                                      \
  do {                                                     \
    hipError_t e__ = (call);                               \
    if (e__ != hipSuccess) return hip_fail((d), e__, #call); \
  } while (0)

  // Device d's part of an all-gather whose rank-r piece is `bytes` bytes at `off_r(r)` of the same-shaped buffer
  // `base[r]` on every device; pieces are equally spaced (`stride`), `mine` bytes of device d's piece are valid.
  // Enqueued on device d's communication stream.
  int gather_piece(int d, const std::vector<char*>& base, size_t region_off, size_t stride, size_t mine) {
    char* my_region = base[(size_t)d] + region_off;
    if (exchange == NBODY_EXCHANGE_RCCL) {
      ncclResult_t r = rccl.AllGather(my_region + (size_t)d * stride, my_region, stride, ncclInt8, comm[(size_t)d], comm_stream[(size_t)d]);
      if (r != ncclSuccess) return ctx_fail(sub[(size_t)d], NBODY_ERR_HIP, std::string("ncclAllGather: ") + rccl.GetErrorString(r));
      return NBODY_OK;
    }
    if (mine == 0) return NBODY_OK;
    for (int p = 0; p < G; ++p) {
      if (p == d) continue;
      char* dst = base[(size_t)p] + region_off + (size_t)d * stride;
      const char* src = my_region + (size_t)d * stride;
      if (dev[(size_t)p] == dev[(size_t)d]) MHIP(d, hipMemcpyAsync(dst, src, mine, hipMemcpyDeviceToDevice, comm_stream[(size_t)d]));
      else MHIP(d, hipMemcpyPeerAsync(dst, dev[(size_t)p], src, dev[(size_t)d], mine, comm_stream[(size_t)d]));
    }
    return NBODY_OK;
  }
  // After device d enqueued its last piece: `stream` (on device d) continues once everything it is owed has landed.
  // Every rank passes through the same barriers whatever happens to it (a rank that left early would strand the others),
  // and all of them return the same verdict.
  int gather_finish(int d, hipStream_t stream) {
    hipError_t e = hipEventRecord(ev_done[(size_t)d], comm_stream[(size_t)d]);
    if (exchange == NBODY_EXCHANGE_RCCL) {  // the collective on my stream completes when my receive buffer is whole
      if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_done[(size_t)d], 0);
      return e == hipSuccess ? NBODY_OK : hip_fail(d, e, "gather_finish");
    }
    bool all = barrier.arrive(e == hipSuccess);  // every device has recorded its event
    for (int p = 0; p < G && all && e == hipSuccess; ++p) e = hipStreamWaitEvent(stream, ev_done[(size_t)p], 0);
    all = barrier.arrive(e == hipSuccess) && all;  // nobody re-records an event a peer has yet to wait on
    if (e != hipSuccess) return hip_fail(d, e, "gather_finish");
    return all ? NBODY_OK : NBODY_ERR_HIP;
  }
};

namespace {

// The failure of a call as the front context reports it: every rank's own message (a rank that merely stood down
// because another one failed has none).
int front_fail(nbody_ctx* front, int who, int rc) {
  Multi* M = front->multi;
  std::string msg;
  for (int d = 0; d < M->G; ++d) {
    const std::string& e = M->sub[(size_t)d]->err;
    if (e.empty()) continue;
    if (!msg.empty()) msg += "; ";
    msg += "device " + std::to_string(M->dev[(size_t)d]) + " (rank " + std::to_string(d) + "): " + e;
  }
  if (msg.empty()) msg = "rank " + std::to_string(who) + " failed";
  front->err = msg;
  return rc;
}
void clear_errors(Multi& M) {
  for (nbody_ctx* S : M.sub) S->err.clear();
}

int choose_chunks(const Multi& M, int64_t n) {
  int c = M.chunks_wanted > 0 ? M.chunks_wanted : env_int("NBODY_MULTI_CHUNKS", 0);
  if (c <= 0) {
    if (M.G == 1) return 1;
    const int64_t n_local = (n + M.G - 1) / M.G;
    c = (int)std::clamp<int64_t>(n_local / 262144, 1, 8);  // a chunk should still fill the chip (see the head of this file)
  }
  return std::clamp(c, 1, kMaxChunks);
}

// ---- direct steps: device d's share of the call
int direct_worker(Multi& M, int d, float delta, int n_steps) {
  nbody_ctx* S = M.sub[(size_t)d];
  hipError_t e0 = hipSetDevice(S->device);
  if (e0 != hipSuccess) return M.hip_fail(d, e0, "hipSetDevice");
  State<float>& s = S->sf;
  auto& st = s.set[s.cur];
  const int64_t n = s.n;
  const int G = M.G, C = M.chunks;
  const size_t rowb = sizeof(float2);
  int rc = ctx_ensure_workspace(S, ctx_direct_ws_bytes(n, std::min<int64_t>(M.block, n)));
  if (rc == NBODY_OK) rc = ctx_ensure_mass_classes(S);
  bool ok = rc == NBODY_OK;
  const int64_t total = M.local_total(d);
  int64_t tmax = 0;  // this device's largest block
  for (int c = 0; c < C; ++c) tmax = std::max(tmax, M.block_count(c, d));
  std::vector<char*> bases((size_t)G);
  // one rank under RCCL still goes through its (no-op) all-gather, so that the RCCL path runs on a one-GPU box too
  const bool exch = G > 1 || M.exchange == NBODY_EXCHANGE_RCCL;
  // NBODY_TRACE: time the last step's chunks against their gathers (events with timing, made on first use)
  const bool trace = exch && C > 1 && env_int("NBODY_TRACE", 0) != 0;
  if (trace && M.tr_kend[(size_t)d].size() < (size_t)C) {
    for (auto* set : {&M.tr_kend, &M.tr_gbeg, &M.tr_gend})
      while ((*set)[(size_t)d].size() < (size_t)C) {
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) != hipSuccess) return M.hip_fail(d, hipGetLastError(), "hipEventCreate");
        (*set)[(size_t)d].push_back(ev);
      }
  }
  int par = 0;
  for (int step = 0; step < n_steps; ++step, par ^= 1) {
    const bool timed = trace && step == n_steps - 1;
    float2* cur = (float2*)M.posbuf[par][(size_t)d];
    float2* nxt = (float2*)M.posbuf[par ^ 1][(size_t)d];
    for (int p = 0; p < G; ++p) bases[(size_t)p] = (char*)M.posbuf[par ^ 1][(size_t)p];
    if (ok) {
      rc = ctx_direct_prep(S, S->stream, n, cur, st.mass, direct_mass_hint(s), total, tmax, S->params.clamp, S->params.arith, S->workspace,
                           S->workspace_bytes);
      ok = rc == NBODY_OK;
    }
    for (int c = 0; c < C; ++c) {
      const int64_t tb = M.block_begin(c, d), nt = M.block_count(c, d);
      if (ok && nt > 0) {
        rc = ctx_direct_run(S, S->stream, n, cur, st.mass, direct_mass_hint(s), tb, nt, st.vel + tb, nxt + tb, nullptr, delta, S->params.clamp,
                            S->params.arith, total, tmax, S->workspace, S->workspace_bytes, S->timer);
        ok = rc == NBODY_OK;
      }
      if (!exch) continue;
      if (ok) {  // the chunk's gather starts when its kernels are done; the compute stream goes on with the next chunk
        hipError_t e = hipEventRecord(M.ev_chunk[(size_t)d][(size_t)c], S->stream);
        if (e == hipSuccess && timed) e = hipEventRecord(M.tr_kend[(size_t)d][(size_t)c], S->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(M.comm_stream[(size_t)d], M.ev_chunk[(size_t)d][(size_t)c], 0);
        if (e == hipSuccess && timed) e = hipEventRecord(M.tr_gbeg[(size_t)d][(size_t)c], M.comm_stream[(size_t)d]);
        if (e != hipSuccess) { rc = M.hip_fail(d, e, "chunk event"); ok = false; }
      }
      if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;  // some rank failed: nobody enters the exchange
      rc = M.gather_piece(d, bases, (size_t)c * G * M.block * rowb, (size_t)M.block * rowb, (size_t)nt * rowb);
      ok = rc == NBODY_OK;
      if (ok && timed && hipEventRecord(M.tr_gend[(size_t)d][(size_t)c], M.comm_stream[(size_t)d]) != hipSuccess) (void)hipGetLastError();
    }
    if (exch) {
      if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;
      rc = M.gather_finish(d, S->stream);  // the next step's preparation needs every position
      ok = rc == NBODY_OK;                 // a failure here travels to the next barrier: all ranks leave together
    } else if (!ok) {
      return rc;
    }
  }
  if (!ok) return rc;
  if (par) std::swap(st.pos, s.pos_next);
  hipError_t e = hipStreamSynchronize(S->stream);
  if (e != hipSuccess) return M.hip_fail(d, e, "hipStreamSynchronize");
  if (trace && n_steps > 0) {
    // chunk c's gather against chunk c + 1's kernels: it should BEGIN before they end (the streams overlap at all) and, where the
    // exchange is shorter than a chunk's kernels, END before they do (only the last chunk's gather is exposed)
    for (int c = 0; c + 1 < C; ++c) {
      if (M.block_count(c + 1, d) <= 0) continue;
      float lead_ms = 0.f, slack_ms = 0.f, took_ms = 0.f;
      if (hipEventElapsedTime(&lead_ms, M.tr_gbeg[(size_t)d][(size_t)c], M.tr_kend[(size_t)d][(size_t)c + 1]) != hipSuccess ||
          hipEventElapsedTime(&slack_ms, M.tr_gend[(size_t)d][(size_t)c], M.tr_kend[(size_t)d][(size_t)c + 1]) != hipSuccess ||
          hipEventElapsedTime(&took_ms, M.tr_gbeg[(size_t)d][(size_t)c], M.tr_gend[(size_t)d][(size_t)c]) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      std::fprintf(stderr, "[nbody] multi: rank %d (device %d) chunk %d: gather began %.1f us before chunk %d's kernels ended, took %.1f us, "
                           "ended %.1f us %s them\n", d, M.dev[(size_t)d], c, 1e3 * lead_ms, c + 1, 1e3 * took_ms, 1e3 * std::fabs(slack_ms),
                   slack_ms >= 0.f ? "before" : "after");
    }
  }
  s.tree_valid = false;
  S->steps_done += (uint64_t)n_steps;
  return NBODY_OK;
}

// The velocities of every device's own blocks to all devices (same block layout as the positions).
int vel_sync_worker(Multi& M, int d) {
  nbody_ctx* S = M.sub[(size_t)d];
  hipError_t e0 = hipSetDevice(S->device);
  if (e0 != hipSuccess) return M.hip_fail(d, e0, "hipSetDevice");
  const size_t rowb = M.f64 ? sizeof(double2) : sizeof(float2);
  std::vector<char*> bases((size_t)M.G);
  for (int p = 0; p < M.G; ++p) {
    nbody_ctx* P = M.sub[(size_t)p];
    bases[(size_t)p] = M.f64 ? (char*)P->sd.set[P->sd.cur].vel : (char*)P->sf.set[P->sf.cur].vel;
  }
  hipError_t e = hipEventRecord(M.ev_chunk[(size_t)d][0], S->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(M.comm_stream[(size_t)d], M.ev_chunk[(size_t)d][0], 0);
  bool ok = e == hipSuccess;
  int rc = ok ? NBODY_OK : M.hip_fail(d, e, "event");
  for (int c = 0; c < M.chunks; ++c) {
    if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;
    rc = M.gather_piece(d, bases, (size_t)c * M.G * M.block * rowb, (size_t)M.block * rowb, (size_t)M.block_count(c, d) * rowb);
    ok = rc == NBODY_OK;
  }
  if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;
  rc = M.gather_finish(d, S->stream);
  if (rc) return rc;
  e = hipStreamSynchronize(S->stream);
  return e == hipSuccess ? NBODY_OK : M.hip_fail(d, e, "hipStreamSynchronize");
}

// ---- tree steps: device d builds the whole tree, walks and integrates its slice, then the slices are exchanged
int tree_worker(Multi& M, int d, int kind, double delta, int n_steps, nbody_counting* cnt0, double* exchange_s) {
  nbody_ctx* S = M.sub[(size_t)d];
  hipError_t e0 = hipSetDevice(S->device);
  if (e0 != hipSuccess) return M.hip_fail(d, e0, "hipSetDevice");
  const int G = M.G;
  const int64_t begin = M.slice_begin(d), count = M.slice_count(d);
  int rc = NBODY_OK;
  bool ok = true;
  for (int step = 0; step < n_steps; ++step) {
    if (ok) {  // (a failure of the previous step's exchange travels to this step's first barrier: all ranks leave together)
      rc = ctx_update_tree_shard(S, M.f64, kind, delta, begin, count, d == 0 ? cnt0 : nullptr);
      ok = rc == NBODY_OK;
    }
    if (G == 1 && M.exchange != NBODY_EXCHANGE_RCCL) {
      if (!ok) return rc;
      continue;
    }
    const double t0 = now_s();
    char* mine = M.xbuf[(size_t)d] + (size_t)d * M.xsec;
    if (ok && count > 0) {  // (waits for the step: the records are complete when it returns)
      rc = ctx_export_slice(S, begin, count, mine, mine + M.xoff_pos, mine + M.xoff_vel);
      ok = rc == NBODY_OK;
    }
    if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;
    rc = M.gather_piece(d, M.xbuf, 0, M.xsec, count > 0 ? M.xsec : 0);
    ok = rc == NBODY_OK;
    if (!M.barrier.arrive(ok)) return ok ? NBODY_ERR_HIP : rc;
    rc = M.gather_finish(d, S->stream);
    ok = rc == NBODY_OK;
    for (int r = 0; r < G && ok; ++r) {
      const int64_t cr = M.slice_count(r);
      if (r == d || cr == 0) continue;
      const char* sec = M.xbuf[(size_t)d] + (size_t)r * M.xsec;
      rc = ctx_import_rows(S, cr, sec, sec + M.xoff_pos, sec + M.xoff_vel);
      ok = rc == NBODY_OK;
    }
    if (d == 0 && exchange_s) *exchange_s += now_s() - t0;
  }
  return ok ? NBODY_OK : rc;
}

template <class T> int replicate_rows(Multi& M) {
  DeviceGuard guard;
  nbody_ctx* P = M.sub[0];
  auto state = [](nbody_ctx* c) -> State<T>& {
    if constexpr (sizeof(T) == 8) return c->sd; else return c->sf;
  };
  State<T>& s0 = state(P);
  using T2 = typename State<T>::T2;
  hipError_t e = hipSetDevice(P->device);
  if (e != hipSuccess) return M.hip_fail(0, e, "hipSetDevice");
  const size_t n = (size_t)s0.n;
  auto& a = s0.set[s0.cur];
  for (int d = 1; d < M.G; ++d) {
    nbody_ctx* S = M.sub[(size_t)d];
    State<T>& s = state(S);
    s.cur = s0.cur;
    auto& b = s.set[s.cur];
    if (n) {
      const int dd = S->device, d0 = P->device;
      auto cp = [&](void* dst, const void* src, size_t bytes) {
        return dd == d0 ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, P->stream) : hipMemcpyPeerAsync(dst, dd, src, d0, bytes, P->stream);
      };
      e = cp(b.pos, a.pos, n * sizeof(T2));
      if (e == hipSuccess) e = cp(b.vel, a.vel, n * sizeof(T2));
      if (e == hipSuccess) e = cp(b.weight, a.weight, n * 4);
      if (e == hipSuccess) e = cp(b.ids, a.ids, n * 4);
      if (e == hipSuccess) e = cp(b.mass, a.mass, n * sizeof(T));
      if (e != hipSuccess) return M.hip_fail(0, e, "replicate rows");
    }
    s.h_weight_stale = true;  // the host mirror of the weights is in another row order now
    ++s.row_epoch;
    s.tree_valid = false;
    s.wt_hist_n = -1;
  }
  e = hipStreamSynchronize(P->stream);
  return e == hipSuccess ? NBODY_OK : M.hip_fail(0, e, "hipStreamSynchronize");
}

}  // namespace

// ================================================================================================ ctx.h interface
void multi_destroy(nbody_ctx* front) {
  if (!front) return;
  delete front->multi;
  front->multi = nullptr;
  delete front;
}

nbody_ctx* multi_peek(const nbody_ctx* front) { return front->multi->sub[0]; }

int multi_set_params(nbody_ctx* front) {
  for (nbody_ctx* S : front->multi->sub) S->params = front->params;
  return NBODY_OK;
}

int multi_upload(nbody_ctx* front, bool f64, int64_t n, const void* pos, const void* vel, const uint32_t* w) {
  Multi& M = *front->multi;
  if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!pos || !vel))) return ctx_fail(front, NBODY_ERR_INVALID, "upload: bad arguments");
  const int C = choose_chunks(M, n);
  int64_t block = (n + (int64_t)M.G * C - 1) / ((int64_t)M.G * C);
  block = (block + 63) & ~(int64_t)63;  // whole waves of targets
  if (block < 64) block = 64;
  const int64_t cap = (int64_t)M.G * C * block;
  if (cap > 0x7fffffffLL) return ctx_fail(front, NBODY_ERR_INVALID, "upload: too many bodies for this many devices");
  int who = 0;
  clear_errors(M);
  int rc = M.pool.run([&](int d) {
    nbody_ctx* S = M.sub[(size_t)d];
    S->row_capacity = cap;
    S->params = front->params;
    return ctx_upload(S, f64, n, pos, vel, w);
  }, &who);
  if (rc) return front_fail(front, who, rc);
  M.n = n;
  M.f64 = f64;
  M.chunks = C;
  M.block = block;
  M.slice = (n + M.G - 1) / M.G;
  M.vel_sharded = false;
  front->has_f32 = !f64;
  front->has_f64 = f64;
  // tree exchange: G sections of {rows u32 | positions | velocities} for ceil(n / G) rows
  const size_t rows = (size_t)std::max<int64_t>(M.slice, 1), e2 = f64 ? sizeof(double2) : sizeof(float2);
  M.xoff_pos = align256(rows * 4);
  M.xoff_vel = M.xoff_pos + align256(rows * e2);
  M.xsec = M.xoff_vel + align256(rows * e2);
  const size_t need = M.xsec * (size_t)M.G;
  DeviceGuard guard;
  if (need > M.xbuf_bytes) {
    for (int d = 0; d < M.G; ++d) {
      hipError_t e = hipSetDevice(M.dev[(size_t)d]);
      if (e == hipSuccess && M.xbuf[(size_t)d]) e = hipFree(M.xbuf[(size_t)d]);
      M.xbuf[(size_t)d] = nullptr;
      if (e == hipSuccess) e = hipMalloc((void**)&M.xbuf[(size_t)d], need);
      if (e != hipSuccess) {
        M.xbuf_bytes = 0;
        return front_fail(front, d, M.hip_fail(d, e, "hipMalloc (exchange buffer)"));
      }
    }
    M.xbuf_bytes = need;
  }
  return NBODY_OK;
}

int multi_primary(nbody_ctx* front, nbody_ctx** out) {
  Multi& M = *front->multi;
  *out = M.sub[0];
  if (M.vel_sharded && M.n > 0) {
    int who = 0;
    clear_errors(M);
    int rc = M.pool.run([&](int d) { return vel_sync_worker(M, d); }, &who);
    if (rc) return front_fail(front, who, rc);
  }
  M.vel_sharded = false;
  return NBODY_OK;
}

int multi_replicate(nbody_ctx* front) {
  Multi& M = *front->multi;
  if (M.G == 1) return NBODY_OK;
  int rc = M.f64 ? replicate_rows<double>(M) : replicate_rows<float>(M);
  return rc ? front_fail(front, 0, rc) : NBODY_OK;
}

int multi_update_direct(nbody_ctx* front, float delta, int n_steps, nbody_counting* counter) {
  Multi& M = *front->multi;
  if (!front->has_f32) return ctx_fail(front, NBODY_ERR_INVALID, "update_direct_f32: no f32 particles uploaded");
  if (n_steps < 0) return ctx_fail(front, NBODY_ERR_INVALID, "update_direct_f32: n_steps < 0");
  if (n_steps == 0 || M.n == 0) return NBODY_OK;
  const double t0 = now_s();
  for (int d = 0; d < M.G; ++d) {
    State<float>& s = M.sub[(size_t)d]->sf;
    M.posbuf[0][(size_t)d] = s.set[s.cur].pos;
    M.posbuf[1][(size_t)d] = s.pos_next;
  }
  int who = 0;
  clear_errors(M);
  int rc = M.pool.run([&](int d) { return direct_worker(M, d, delta, n_steps); }, &who);
  if (rc) return front_fail(front, who, rc);
  if (M.G > 1) M.vel_sharded = true;
  // force and integrate are one fused kernel: the whole call is booked under sum_gravity (as for one device)
  const double dt = now_s() - t0;
  front->counting.sum_gravity += dt;
  if (counter) counter->sum_gravity += dt;
  front->steps_done += (uint64_t)n_steps;
  return NBODY_OK;
}

int multi_update_tree(nbody_ctx* front, bool f64, int kind, double delta, int n_steps, nbody_counting* counter) {
  Multi& M = *front->multi;
  if (f64 ? !front->has_f64 : !front->has_f32) return ctx_fail(front, NBODY_ERR_INVALID, "update_tree: no particles of this precision uploaded");
  if (n_steps < 0) return ctx_fail(front, NBODY_ERR_INVALID, "update_tree: n_steps < 0");
  nbody_ctx* p = nullptr;
  int rc = multi_primary(front, &p);  // whole rows on every device before the trees are built
  if (rc) return rc;
  if (M.G == 1 && lab_int("NBODY_MULTI_FORCE_EXCHANGE", 0) == 0) {
    // one device: nothing to shard or exchange — the single-device step driver, steps ahead of the host and all (the
    // tests set NBODY_MULTI_FORCE_EXCHANGE=1 to push a lone rank through the sliced step and its one-rank all-gather)
    nbody_counting c1{};
    rc = ctx_update_tree(p, f64, kind, delta, n_steps, &c1);
    if (rc) { front->err = p->err; return rc; }
    front->counting.build_bvh += c1.build_bvh;
    front->counting.sum_gravity += c1.sum_gravity;
    front->counting.post_calculations += c1.post_calculations;
    if (counter) {
      counter->build_bvh += c1.build_bvh;
      counter->sum_gravity += c1.sum_gravity;
      counter->post_calculations += c1.post_calculations;
    }
    front->steps_done += (uint64_t)n_steps;
    return NBODY_OK;
  }
  nbody_counting c0{};
  double exchange_s = 0.0;
  int who = 0;
  clear_errors(M);
  rc = M.pool.run([&](int d) { return tree_worker(M, d, kind, delta, n_steps, &c0, &exchange_s); }, &who);
  if (rc) return front_fail(front, who, rc);
  // the phases as the first device saw them; the exchange of the slices belongs to what follows the force map
  c0.post_calculations += exchange_s;
  front->counting.build_bvh += c0.build_bvh;
  front->counting.sum_gravity += c0.sum_gravity;
  front->counting.post_calculations += c0.post_calculations;
  if (counter) {
    counter->build_bvh += c0.build_bvh;
    counter->sum_gravity += c0.sum_gravity;
    counter->post_calculations += c0.post_calculations;
  }
  front->steps_done += (uint64_t)n_steps;
  return NBODY_OK;
}

}  // namespace nbody

// ================================================================================================ C ABI
using namespace nbody;
#define NB_API extern "C" __attribute__((visibility("default")))

NB_API int nbody_create_multi_ex(nbody_ctx** out, int n_devices, const int* device_ids, int exchange, int chunks) {
  if (!out) return ctx_fail(nullptr, NBODY_ERR_INVALID, "nbody_create_multi: out is NULL");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return ctx_fail(nullptr, NBODY_ERR_INVALID, "nbody_create_multi: n_devices must be 1..64");
  if (exchange != NBODY_EXCHANGE_RCCL && exchange != NBODY_EXCHANGE_PEER)
    return ctx_fail(nullptr, NBODY_ERR_INVALID, "nbody_create_multi: unknown exchange");
  if (chunks < 0 || chunks > kMaxChunks) return ctx_fail(nullptr, NBODY_ERR_INVALID, "nbody_create_multi: chunks must be 0 (by size) .. 16");
  std::vector<int> ids((size_t)n_devices);
  for (int d = 0; d < n_devices; ++d) ids[(size_t)d] = device_ids ? device_ids[d] : d;
  bool distinct = true;
  for (int a = 0; a < n_devices; ++a)
    for (int b = a + 1; b < n_devices; ++b) distinct = distinct && ids[(size_t)a] != ids[(size_t)b];
  if (exchange == NBODY_EXCHANGE_RCCL && !distinct)
    return ctx_fail(nullptr, NBODY_ERR_INVALID,
                    "nbody_create_multi: RCCL needs one rank per physical device; a device listed twice (a rehearsal on one GPU) "
                    "takes NBODY_EXCHANGE_PEER");
  nbody_ctx* front = new (std::nothrow) nbody_ctx();
  Multi* M = new (std::nothrow) Multi();
  if (!front || !M) {
    delete front;
    delete M;
    return ctx_fail(nullptr, NBODY_ERR_NOMEM, "nbody_create_multi: out of host memory");
  }
  front->multi = M;
  DeviceGuard guard;
  nbody_default_params(&front->params);
  M->G = n_devices;
  M->dev = ids;
  M->exchange = exchange;
  M->chunks_wanted = chunks;
  M->sub.assign((size_t)n_devices, nullptr);
  M->comm.assign((size_t)n_devices, nullptr);
  M->comm_stream.assign((size_t)n_devices, nullptr);
  M->tr_kend.assign((size_t)n_devices, {});
  M->tr_gbeg.assign((size_t)n_devices, {});
  M->tr_gend.assign((size_t)n_devices, {});
  M->ev_chunk.assign((size_t)n_devices, {});
  M->ev_done.assign((size_t)n_devices, nullptr);
  M->xbuf.assign((size_t)n_devices, nullptr);
  M->posbuf[0].assign((size_t)n_devices, nullptr);
  M->posbuf[1].assign((size_t)n_devices, nullptr);
  M->barrier.n = n_devices;
  M->pool.barrier = &M->barrier;
  auto bail = [&](int rc) {
    multi_destroy(front);
    return rc;
  };
  for (int d = 0; d < n_devices; ++d) {
    int rc = ctx_create_single(&M->sub[(size_t)d], ids[(size_t)d]);  // (leaves its message in the thread's create error)
    if (rc) return bail(rc);
    hipError_t e = hipSetDevice(ids[(size_t)d]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&M->comm_stream[(size_t)d], hipStreamNonBlocking);
    for (int c = 0; c < kMaxChunks && e == hipSuccess; ++c) {
      hipEvent_t ev = nullptr;
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess) M->ev_chunk[(size_t)d].push_back(ev);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&M->ev_done[(size_t)d], hipEventDisableTiming);
    if (e != hipSuccess) return bail(ctx_fail(nullptr, NBODY_ERR_HIP, std::string("nbody_create_multi: ") + hipGetErrorString(e)));
  }
  // peer access between the physical devices (the peer copies, and RCCL's own P2P transport)
  for (int a = 0; a < n_devices; ++a)
    for (int b = 0; b < n_devices; ++b) {
      if (ids[(size_t)a] == ids[(size_t)b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, ids[(size_t)a], ids[(size_t)b]) != hipSuccess || !can) continue;
      (void)hipSetDevice(ids[(size_t)a]);
      hipError_t e = hipDeviceEnablePeerAccess(ids[(size_t)b], 0);
      if (e != hipSuccess) (void)hipGetLastError();  // already enabled, or left to staged copies
    }
  if (exchange == NBODY_EXCHANGE_RCCL) {
    std::string why;
    if (!M->rccl.load(&why)) return bail(ctx_fail(nullptr, NBODY_ERR_NO_DEVICE, "nbody_create_multi: " + why));
    ncclResult_t r = M->rccl.CommInitAll(M->comm.data(), n_devices, ids.data());
    if (r != ncclSuccess)
      return bail(ctx_fail(nullptr, NBODY_ERR_HIP, std::string("nbody_create_multi: ncclCommInitAll: ") + M->rccl.GetErrorString(r)));
  }
  M->pool.start(n_devices);
  *out = front;
  return NBODY_OK;
}

NB_API int nbody_create_multi(nbody_ctx** out, int n_devices, const int* device_ids) {
  const char* x = getenv("NBODY_MULTI_EXCHANGE");
  const int exchange = (x && (!strcmp(x, "peer") || !strcmp(x, "1"))) ? NBODY_EXCHANGE_PEER : NBODY_EXCHANGE_RCCL;
  return nbody_create_multi_ex(out, n_devices, device_ids, exchange, 0);
}

NB_API int nbody_multi_info(const nbody_ctx* ctx, int* n_devices, int* exchange, int* chunks, int64_t* block) {
  if (!ctx) return NBODY_ERR_INVALID;
  if (!ctx->multi) {
    if (n_devices) *n_devices = 1;
    if (exchange) *exchange = -1;
    if (chunks) *chunks = 1;
    if (block) *block = ctx->has_f32 ? ctx->sf.n : (ctx->has_f64 ? ctx->sd.n : 0);
    return NBODY_OK;
  }
  const Multi& M = *ctx->multi;
  if (n_devices) *n_devices = M.G;
  if (exchange) *exchange = M.exchange;
  if (chunks) *chunks = M.chunks;
  if (block) *block = M.block;
  return NBODY_OK;
}

NB_API int nbody_multi_comm_count(const nbody_ctx* ctx, int* n_ranks) {
  if (!ctx || !n_ranks) return NBODY_ERR_INVALID;
  *n_ranks = 0;
  if (!ctx->multi) return NBODY_OK;
  const Multi& M = *ctx->multi;
  if (M.exchange != NBODY_EXCHANGE_RCCL || M.comm.empty() || !M.comm[0] || !M.rccl.CommCount) return NBODY_OK;
  int count = 0;
  ncclResult_t r = M.rccl.CommCount(M.comm[0], &count);
  if (r != ncclSuccess) return NBODY_ERR_HIP;
  *n_ranks = count;
  return NBODY_OK;
}
