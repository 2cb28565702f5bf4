// libnbody_hip — context and extern "C" surface (include/nbody_hip.h).
//
// Host side of the drop-in for World::update (/root/reference src/main.rs:388-425).  The context owns the
// device image of `World.particles` (main.rs:37-39) as SoA arrays; every entry point is one of the three
// phases of update (build / force / integrate) or a copy in/out.  There is no CPU fallback anywhere in this
// file: the only host computation is the tree build, which the reference also does on the host and in
// sequence (bvh_tree.rs:56-96); forces and integration always run on the GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "common.h"
#include "direct_kernels.h"
#include "bvh_build.h"
#include "bvh_build64.h"
#include "delta_codec.h"
#include "delta_decoder.hpp"
#include "delta_snapshot.h"
#include "exact_sum.h"
#include "exact_sum64.h"
#include "quad_build.h"
#include "render.h"
#include "tree_build.hpp"
#include "tree_kernels.h"
#include "walk_split.h"

using namespace nbody;

// ------------------------------------------------------------------------------------------------ timer
hipError_t nbody_timer::begin(hipStream_t s, Pair* out) {
  if (pool.size() >= 256) {
    hipError_t e = drain();
    if (e != hipSuccess) return e;
  }
  Pair p{};
  if (!free_.empty()) {
    p = free_.back();
    free_.pop_back();
  } else {
    hipError_t e = hipEventCreate(&p.a);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&p.b);
    if (e != hipSuccess) return e;
  }
  *out = p;
  return hipEventRecord(p.a, s);
}
hipError_t nbody_timer::end(hipStream_t s, const Pair& p) {
  hipError_t e = hipEventRecord(p.b, s);
  pool.push_back(p);
  return e;
}
hipError_t nbody_timer::drain() {
  for (auto& p : pool) {
    hipError_t e = hipEventSynchronize(p.b);
    if (e != hipSuccess) return e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p.a, p.b);
    if (e != hipSuccess) return e;
    total_ms += ms;
    launches++;
    free_.push_back(p);
  }
  pool.clear();
  return hipSuccess;
}
nbody_timer::~nbody_timer() {
  for (auto& p : pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto& p : free_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
}

// ------------------------------------------------------------------------------------------------ context
#include "ctx.h"

namespace {

thread_local std::string g_create_error;

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct nbody_host_tree {
  bool is_f64 = false;
  TreeHost<float> tf;
  TreeHost<double> td;
};

namespace {

int fail(nbody_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}
int fail_hip(nbody_ctx* c, hipError_t e, const char* what) {
  return fail(c, NBODY_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, call)                                   \
  do {                                                    \
    hipError_t e__ = (call);                              \
    if (e__ != hipSuccess) return fail_hip(c, e__, #call); \
  } while (0)

template <class P> void free_dev(P*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}
template <class T> void free_state(State<T>& s) {
  for (auto& st : s.set) { free_dev(st.pos); free_dev(st.vel); free_dev(st.weight); free_dev(st.ids); free_dev(st.mass); }
  free_dev(s.pos_next); free_dev(s.acc); free_dev(s.geom0); free_dev(s.geom1); free_dev(s.link); free_dev(s.order_dev);
  free_dev(s.node_depth); free_dev(s.node_mass); free_dev(s.node_size); free_dev(s.qb_scratch); free_dev(s.bb_scratch); free_dev(s.ws_scratch); free_dev(s.ws_terms); free_dev(s.wt_hist);
  s.node_aux_cap = 0; s.qb_scratch_bytes = 0; s.bb_scratch_bytes = 0; s.h_weight_stale = false;
  s.ws_scratch_bytes = 0; s.ws_capacity = 0; s.ws_backoff = 0; s.quad_depth_hint = 0; s.wt_hist_n = -1; s.bvh_levels_hint = 0; s.bvh_levels_stable = 0; s.ahead_total_due = false;
  s.tree_host_stale = false; s.n_nodes = 0;
  s.node_cap = 0; s.n = 0; s.tree_valid = false; s.tree.clear();
  s.h_pos.clear(); s.h_weight.clear();
  free_dev(s.classes.rank); free_dev(s.classes.pad_slots); free_dev(s.classes.tile_mass);
  s.classes = typename State<T>::MassClasses{};
}

template <class T> State<T>& state_of(nbody_ctx* c);
template <> State<float>& state_of<float>(nbody_ctx* c) { return c->sf; }
template <> State<double>& state_of<double>(nbody_ctx* c) { return c->sd; }
template <class T> bool has_state(const nbody_ctx* c);
template <> bool has_state<float>(const nbody_ctx* c) { return c->has_f32; }
template <> bool has_state<double>(const nbody_ctx* c) { return c->has_f64; }

// -------------------------------------------------------------------------------------------- direct config

// NBODY_WALK_SPLIT (big-leaf BVH walk): 0 the fused walk only, 1 one pass through LDS when it pays (default), 3 one pass whenever
// possible; the laboratory build also knows 4 / 2, the three-pass design of round 1 (when it pays / whenever possible).
int walk_split_mode() {
  const int mode = env_int("NBODY_WALK_SPLIT", 1);
  return (!kLabBuild && (mode == 2 || mode == 4)) ? 1 : mode;
}

// How the direct kernel covers (n_tgt x n_src): enough waves to fill 256 CUs x 4 SIMDs x 8 waves.
DirectConfig choose_direct_config(int64_t n_src, int64_t n_tgt, bool uniform = true) {
  DirectConfig c;
  c.use_asm = lab_int("NBODY_DIRECT_ASM", 3);
  // near/far split: 0.05 ms (65 536 bodies) to 0.16 ms (1 M) of preparation per step against 10 % of the pair work: it pays
  // from 65 536 x 65 536 pairs on (profiles/r03_nearfar_hash_grid.txt; the sort-based split of rounds 1-2 broke even at
  // twice that).  NBODY_DIRECT_NEARFAR: 0 never, 1 by size (default), 2 always.
  const int nf = env_int("NBODY_DIRECT_NEARFAR", 1);
  c.nearfar = nf == 2 || (nf == 1 && (double)n_src * (double)n_tgt >= 4294967296.0);
  // measured at N = 1M (profiles/r01_direct_mass_variants.txt): 1 target/thread with the hand-ordered block wins for
  // equal masses (44.0 %) and for per-body masses (39.8 % vs 36.1 % for 2 targets/thread)
  (void)uniform;
  c.tpt = lab_int("NBODY_DIRECT_TPT", 1);
  if (c.tpt != 1 && c.tpt != 2) c.tpt = 1;
  // 256 CUs x 32 wave slots hold 8192 waves; several rounds of waves balance the tail, so the sources are split
  // over blockIdx.y until there are ~16 rounds (measured, profiles/r01_direct_gsplit_sweep.txt: 131072 targets
  // 40.5 % -> 43.8 %, 65536 x 65536 33.5 % -> 43.1 %, 1M x 1M 43.8 % -> 44.1 %)
  const int64_t want_waves = 131072;
  int64_t waves = 4 * ((n_tgt + 64 * c.tpt - 1) / (64 * c.tpt));
  int64_t g = waves > 0 ? (want_waves + waves - 1) / waves : 1;
  if (g < 1) g = 1;
  if (g > 16) g = 16;
  // ... and until a split's sources (8 B each) are at most half an XCD's 4 MB L2, where 16 splits can do that: the work-groups of
  // one split run together (blockIdx.x varies fastest), so the range they stream stays L2-resident instead of being re-read
  // from HBM by waves that have drifted apart (N = 1 M: 2 -> 4 splits, FETCH_SIZE 524 -> 82 MB per launch, the same 173.1 ms)
  const int64_t g_l2 = (n_src + 262143) / 262144;
  if (g_l2 <= 16 && g < g_l2) g = g_l2;
  g = lab_int("NBODY_DIRECT_GSPLIT", (int)g);
  if (g < 1) g = 1;
  if (g > 64) g = 64;
  while (g > 1 && n_src / g < 2048) g /= 2;  // a split should still hold a couple of tiles
  c.gsplit = (int)g;
  return c;
}

constexpr size_t kFlagBytes = 256;

// The partial-sum area holds gsplit(n) x n entries for every block size n <= n_tgt (a shard's last block may be
// shorter than the others and then takes a LARGER gsplit: 262080 rows -> 9 where 262144 -> 8).  gsplit(n) =
// ceil(want_waves / (4 ceil(n/64))) <= 16, so gsplit(n) n < 2097152 + n and <= 16 n; direct_run clamps what an
// environment override could still push past it.
size_t direct_partial_bytes(int64_t n_src, int64_t n_tgt) {
  size_t partial = 0;
  for (bool uni : {false, true}) {
    DirectConfig c = choose_direct_config(n_src, n_tgt, uni);
    size_t p = (size_t)c.gsplit * (size_t)n_tgt * sizeof(float2);
    if (p > partial) partial = p;
  }
  size_t any_block = (size_t)std::min<int64_t>(16 * n_tgt, 2097152 + n_tgt) * sizeof(float2);
  if (any_block > partial) partial = any_block;
  const int64_t g_l2 = (n_src + 262143) / 262144;  // (the L2 rule of choose_direct_config holds for every block size)
  if (g_l2 <= 16 && (size_t)(g_l2 * n_tgt) * sizeof(float2) > partial) partial = (size_t)(g_l2 * n_tgt) * sizeof(float2);
  return (partial + 255) & ~(size_t)255;
}
size_t direct_ws_bytes(int64_t n_src, int64_t n_tgt) {
  return kFlagBytes + direct_partial_bytes(n_src, n_tgt) + nearfar_layout(n_src).total;
}

// Mass classes of the context's f32 rows in their current order (ctx.h, State::MassClasses): built on the host from the
// weights (static between uploads; only the row order moves, with the tree builds), cached until the rows are permuted.
// Classes are taken in ascending weight, bodies inside a class in ascending row: everything downstream stays a function
// of the inputs alone (bitwise reproducible).  Not used for equal masses, for "one mass but for a few bodies" (the near
// list carries those) or for more than 32 distinct masses (the per-body kernel runs then).
int ensure_mass_classes(nbody_ctx* c) {
  State<float>& s = c->sf;
  auto& mc = s.classes;
  if (mc.epoch == s.row_epoch) return NBODY_OK;
  // A captured direct step (DirectGraph) has the class arrays' addresses baked into its kernel arguments: they are about to be
  // freed and rebuilt for the new row order, so the graph goes with them (an even number of tree steps permutes the
  // rows and leaves every pointer of the graph's key where it was).
  c->direct_graph.reset();
  mc.epoch = s.row_epoch;
  mc.usable = false;
  const int64_t n = s.n;
  if (lab_int("NBODY_DIRECT_NO_CLASSES", 0) != 0 || s.uniform_mass > 0.f || s.sparse_base > 0.f || n < 32768) return NBODY_OK;
  if (s.h_weight_stale) {
    s.h_weight.resize((size_t)n);
    HIPCHK(c, hipMemcpyAsync(s.h_weight.data(), s.set[s.cur].weight, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    s.h_weight_stale = false;
  }
  uint32_t vals[kMaxMassClasses];
  int64_t counts[kMaxMassClasses];
  int k = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t w = s.h_weight[(size_t)i];
    int j = 0;
    while (j < k && vals[j] != w) ++j;
    if (j == k) {
      if (k == (int)kMaxMassClasses) return NBODY_OK;  // too many distinct masses
      vals[k] = w;
      counts[k] = 0;
      ++k;
    }
    ++counts[j];
  }
  if (k < 2) return NBODY_OK;
  int order[kMaxMassClasses];
  for (int j = 0; j < k; ++j) order[j] = j;
  std::sort(order, order + k, [&](int a, int b) { return vals[a] < vals[b]; });
  int64_t start[kMaxMassClasses], fill[kMaxMassClasses];
  int64_t slots = 0;
  std::vector<float> tile_mass;
  std::vector<uint32_t> pads;
  for (int r = 0; r < k; ++r) {
    const int j = order[r];
    start[j] = slots;
    fill[j] = 0;
    const int64_t padded = (counts[j] + kDirectTile - 1) / kDirectTile * kDirectTile;
    for (int64_t t = 0; t < padded / kDirectTile; ++t) tile_mass.push_back((float)vals[j]);  // `weight as f32`, main.rs:360
    for (int64_t q = slots + counts[j]; q < slots + padded; ++q) pads.push_back((uint32_t)q);
    slots += padded;
  }
  std::vector<uint32_t> rank((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t w = s.h_weight[(size_t)i];
    int j = 0;
    while (vals[j] != w) ++j;
    rank[(size_t)i] = (uint32_t)(start[j] + fill[j]++);
  }
  free_dev(mc.rank); free_dev(mc.pad_slots); free_dev(mc.tile_mass);
  HIPCHK(c, hipMalloc((void**)&mc.rank, (size_t)n * 4));
  HIPCHK(c, hipMalloc((void**)&mc.pad_slots, (pads.size() + 1) * 4));
  HIPCHK(c, hipMalloc((void**)&mc.tile_mass, (tile_mass.size() + 1) * 4));
  HIPCHK(c, hipMemcpyAsync(mc.rank, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if (!pads.empty()) HIPCHK(c, hipMemcpyAsync(mc.pad_slots, pads.data(), pads.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(mc.tile_mass, tile_mass.data(), tile_mass.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  mc.n_classes = k;
  mc.n_slots = slots;
  mc.n_pad_slots = (int)pads.size();
  mc.usable = true;
  if (env_int("NBODY_TRACE", 0) != 0)
    std::fprintf(stderr, "[nbody] direct: %d mass classes, %lld bodies in %lld slots\n", k, (long long)n, (long long)slots);
  return NBODY_OK;
}
// the classes of `c` if they describe exactly these sources in their current row order
const State<float>::MassClasses* classes_for(const nbody_ctx* c, int64_t n_src, const void* mass_all) {
  if (!c || !c->has_f32) return nullptr;
  const State<float>& s = c->sf;
  const auto& mc = s.classes;
  if (!mc.usable || mc.epoch != s.row_epoch || n_src != s.n || mass_all != (const void*)s.set[s.cur].mass) return nullptr;
  return &mc;
}

// A direct step = one preparation over ALL positions (hazard scan, near/far split, the decision word) followed by one
// or more runs, each over a block of targets.  `n_tgt_total` (all targets this device computes in the step) decides
// whether the near/far split pays; `n_tgt_max` (the largest block of one run) sizes the partial-sum area, so that
// preparation and runs agree on the workspace layout.
struct DirectPlan {
  int arith = 0;
  bool uni = false;
  float sparse_base = 0.f;  // > 0: all masses equal this but a few bodies', which travel with the near list
  const State<float>::MassClasses* classes = nullptr;  // masses in a few classes: the far copy in class order, equal-mass arithmetic per tile
  bool nearfar = false;
  bool couples = false;  // the far copy in couples {xA, xB, yA, yB}, padded to whole 16-source iterations (the packed kernels)
  bool stream_m = false; // free per-body masses through the streamed main pass: the far copy carries 1 / mass in slot order
  int use_hazard = 0;
  size_t partial_bytes = 0;
};
int direct_plan(nbody_ctx* c, int64_t n_src, const void* mass_all, float uniform_mass, int64_t n_tgt_total, int64_t n_tgt_max, float clamp,
                int arith, const void* ws, size_t ws_bytes, DirectPlan* out) {
  if (n_src < 0 || n_tgt_total < 0 || n_tgt_max < 0 || n_tgt_max > n_tgt_total || n_tgt_total > n_src || n_src > 0x7fffffffLL)
    return fail(c, NBODY_ERR_INVALID, "direct_step: bad target/source counts");
  if (arith < NBODY_ARITH_AUTO || arith > NBODY_ARITH_EXACT) return fail(c, NBODY_ERR_INVALID, "direct_step: bad arith");
  if (!ws || ws_bytes < direct_ws_bytes(n_src, n_tgt_max)) return fail(c, NBODY_ERR_INVALID, "direct_step: workspace too small");
  // FAST's zero-distance bias needs clamp >= 2^-19 (HISTORY.md §4.1); smaller clamps always take EXACT.
  if (arith != NBODY_ARITH_EXACT && !(clamp >= 1.9073486328125e-06f)) arith = NBODY_ARITH_EXACT;
  DirectPlan p;
  p.arith = arith;
  p.uni = uniform_mass > 0.f && lab_int("NBODY_DIRECT_NO_UNIFORM", 0) == 0;
  {
    const DirectConfig c0 = choose_direct_config(n_src, n_tgt_total, p.uni);
    p.nearfar = c0.nearfar;
    p.couples = c0.nearfar && c0.use_asm >= 2 && c0.tpt == 1;
  }
  // uniform_mass < 0: every mass is -uniform_mass except a sparse set; the split hands those to direct_finish, so the
  // main pass runs at the equal-mass rate.  Without the split (small problems) the per-body-mass kernel is used.
  if (uniform_mass < 0.f && p.nearfar && lab_int("NBODY_DIRECT_NO_UNIFORM", 0) == 0 && lab_int("NBODY_DIRECT_NO_SPARSE", 0) == 0)
    p.sparse_base = -uniform_mass;
  if (!p.uni && p.sparse_base == 0.f && p.nearfar) p.classes = classes_for(c, n_src, mass_all);
  p.stream_m = !p.uni && p.sparse_base == 0.f && !p.classes && p.couples && lab_int("NBODY_DIRECT_ASM", 3) >= 3;
  p.use_hazard = arith == NBODY_ARITH_AUTO;
  p.partial_bytes = direct_partial_bytes(n_src, n_tgt_max);
  *out = p;
  return NBODY_OK;
}

// Decides, on the stream, which kernels of this step do the work (flags[kFlagState]).
int direct_prep(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
                int64_t n_tgt_total, int64_t n_tgt_max, float clamp, int arith, void* ws, size_t ws_bytes) {
  DirectPlan p;
  int rc = direct_plan(c, n_src, mass_all, uniform_mass, n_tgt_total, n_tgt_max, clamp, arith, ws, ws_bytes, &p);
  if (rc) return rc;
  if (n_tgt_total == 0 || p.arith == NBODY_ARITH_EXACT) return NBODY_OK;
  if (!pos_all || !mass_all) return fail(c, NBODY_ERR_INVALID, "direct_step: null pos_all/mass_all");
  int* flags = (int*)ws;
  HIPCHK(c, hipMemsetAsync(flags, 0, kFlagBytes, stream));
  if (p.use_hazard && !(p.nearfar && n_src > 0))  // (with the split on, nf_insert checks the positions as it reads them)
    HIPCHK(c, launch_hazard_scan(stream, (const float*)pos_all, 2 * n_src, flags));
  if (p.nearfar) {
    char* nf_scratch = (char*)ws + kFlagBytes + p.partial_bytes;
    NearFarLayout L = nearfar_layout(n_src);
    const float2* pos_far = nullptr;
    const uint32_t* near_list = nullptr;
    const float* minv_far = nullptr;
    HIPCHK(c, launch_nearfar(stream, (const float2*)pos_all, (const float*)mass_all, p.sparse_base, (int)n_src, clamp, p.use_hazard, flags,
                             nf_scratch, L, &pos_far, &near_list, p.classes ? p.classes->rank : nullptr,
                             p.classes ? p.classes->pad_slots : nullptr, p.classes ? p.classes->n_pad_slots : 0, p.couples,
                             p.stream_m ? &minv_far : nullptr));
  } else {
    HIPCHK(c, launch_decide_simple(stream, p.use_hazard, flags));
  }
  return NBODY_OK;
}

// Force + integration for the targets [tgt_begin, tgt_begin + n_tgt) under the decision direct_prep left in `ws`.
int direct_run(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
               int64_t tgt_begin, int64_t n_tgt, void* vel, void* pos_out, void* acc_out, float delta, float clamp, int arith,
               int64_t n_tgt_total, int64_t n_tgt_max, void* ws, size_t ws_bytes, nbody_timer* timer) {
  if (n_tgt < 0 || tgt_begin < 0 || tgt_begin + n_tgt > n_src || n_tgt > n_tgt_max)
    return fail(c, NBODY_ERR_INVALID, "direct_step: bad target/source range");
  DirectPlan p;
  int rc = direct_plan(c, n_src, mass_all, uniform_mass, n_tgt_total, n_tgt_max, clamp, arith, ws, ws_bytes, &p);
  if (rc) return rc;
  if (n_tgt == 0) return NBODY_OK;
  if (!pos_all || !mass_all) return fail(c, NBODY_ERR_INVALID, "direct_step: null pos_all/mass_all");
  if ((vel == nullptr) != (pos_out == nullptr))
    return fail(c, NBODY_ERR_INVALID, "direct_step: vel and pos_out must both be given or both be NULL");
  if (!vel && !acc_out) return fail(c, NBODY_ERR_INVALID, "direct_step: nothing to compute");

  DirectConfig cfg = choose_direct_config(n_src, n_tgt, p.uni);
  cfg.nearfar = p.nearfar;
  while (cfg.gsplit > 1 && (size_t)cfg.gsplit * (size_t)n_tgt * sizeof(float2) > p.partial_bytes) --cfg.gsplit;
  if ((size_t)cfg.gsplit * (size_t)n_tgt * sizeof(float2) > p.partial_bytes)
    return fail(c, NBODY_ERR_INVALID, "direct_step: the partial sums of this block do not fit the workspace's layout");
  int* flags = (int*)ws;
  DirectArgs a{};
  a.pos_all = (const float2*)pos_all;
  a.src_pos = a.pos_all;
  a.mass_all = (const float*)mass_all;
  a.n_src = (int)n_src;
  a.tgt_begin = (int)tgt_begin;
  a.n_tgt = (int)n_tgt;
  a.vel = (float2*)vel;
  a.pos_out = (float2*)pos_out;
  a.acc_out = (float2*)acc_out;
  a.partial = (float2*)((char*)ws + kFlagBytes);
  a.delta = delta;
  a.clamp = clamp;
  a.uniform_mass = p.uni ? uniform_mass : 0.f;
  a.flags = flags;
  a.run_state = -1;

  if (p.arith == NBODY_ARITH_EXACT) {
    TimerScope ts(timer, stream);
    HIPCHK(c, launch_direct_exact(stream, a));
    return NBODY_OK;
  }
  const float2* pos_far = nullptr;
  const uint32_t* near_list = nullptr;
  const float* minv_far = nullptr;
  if (cfg.nearfar) {
    char* nf_scratch = (char*)ws + kFlagBytes + p.partial_bytes;
    NearFarLayout L = nearfar_layout(n_src);
    pos_far = (const float2*)(nf_scratch + L.pos_far);
    near_list = (const uint32_t*)(nf_scratch + L.near_list);
    if (p.stream_m) minv_far = (const float*)(nf_scratch + L.minv_far);
  }
  {
    TimerScope ts(timer, stream);
    if (cfg.nearfar) {  // state 0: main pass over the far sources without the clamp, near sources added by finish
      DirectArgs a0 = a;
      a0.src_pos = pos_far;
      a0.src_couples = p.couples ? 1 : 0;
      a0.src_minv = minv_far;
      if (p.couples) a0.n_src = (int)far_padded(n_src);
      a0.near_list = near_list;
      a0.to_partial = 1;
      a0.run_state = 0;
      if (p.sparse_base > 0.f) a0.uniform_mass = p.sparse_base;  // the odd masses sit in the near list (state 1 reads them all)
      if (p.classes) {  // the far copy is in class order, padded: the equal-mass instantiation, a tile's mass in its closing FMA
        a0.n_src = (int)(p.couples ? far_padded(p.classes->n_slots) : p.classes->n_slots);
        a0.uniform_mass = 1.0f;
        a0.tile_mass = p.classes->tile_mass;
      }
      HIPCHK(c, launch_direct_fast(stream, a0, cfg, true));
    }
    DirectArgs a1 = a;  // state 1: one clamped pass over every source
    a1.to_partial = cfg.gsplit > 1;
    a1.run_state = 1;
    HIPCHK(c, launch_direct_fast(stream, a1, cfg, false));
  }
  if (cfg.nearfar) {
    DirectArgs a0 = a;
    a0.near_list = near_list;
    a0.run_state = 0;
    HIPCHK(c, launch_direct_finish(stream, a0, cfg.gsplit, true));
  }
  if (cfg.gsplit > 1) {
    DirectArgs a1 = a;
    a1.run_state = 1;
    HIPCHK(c, launch_direct_finish(stream, a1, cfg.gsplit, false));
  }
  if (p.use_hazard) {  // state 2
    DirectArgs a2 = a;
    a2.run_state = 2;
    HIPCHK(c, launch_direct_exact(stream, a2));
  }
  return NBODY_OK;
}

int direct_step_dev(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all,
                    float uniform_mass, int64_t tgt_begin, int64_t n_tgt, void* vel, void* pos_out, void* acc_out,
                    float delta, float clamp, int arith, void* ws, size_t ws_bytes, nbody_timer* timer) {
  if (n_src < 0 || n_tgt < 0 || tgt_begin < 0 || tgt_begin + n_tgt > n_src || n_src > 0x7fffffffLL)
    return fail(c, NBODY_ERR_INVALID, "direct_step: bad target/source range");
  if (n_tgt == 0) return NBODY_OK;
  if (!pos_all || !mass_all) return fail(c, NBODY_ERR_INVALID, "direct_step: null pos_all/mass_all");
  if ((vel == nullptr) != (pos_out == nullptr))
    return fail(c, NBODY_ERR_INVALID, "direct_step: vel and pos_out must both be given or both be NULL");
  if (!vel && !acc_out) return fail(c, NBODY_ERR_INVALID, "direct_step: nothing to compute");
  int rc = direct_prep(c, stream, n_src, pos_all, mass_all, uniform_mass, n_tgt, n_tgt, clamp, arith, ws, ws_bytes);
  if (rc) return rc;
  return direct_run(c, stream, n_src, pos_all, mass_all, uniform_mass, tgt_begin, n_tgt, vel, pos_out, acc_out, delta, clamp, arith,
                    n_tgt, n_tgt, ws, ws_bytes, timer);
}

int ensure_workspace(nbody_ctx* c, size_t bytes) {
  if (c->workspace_bytes >= bytes) return NBODY_OK;
  free_dev(c->workspace);
  c->workspace_bytes = 0;
  HIPCHK(c, hipMalloc(&c->workspace, bytes));
  c->workspace_bytes = bytes;
  return NBODY_OK;
}

// -------------------------------------------------------------------------------------------- upload / download
template <class T> int upload(nbody_ctx* c, int64_t n, const T* pos, const T* vel, const uint32_t* w) {
  if (!c) return NBODY_ERR_INVALID;
  if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!pos || !vel))) return fail(c, NBODY_ERR_INVALID, "upload: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  c->direct_graph.reset();
  free_state(c->sf);
  free_state(c->sd);
  c->has_f32 = c->has_f64 = false;
  c->dl_key_next = true;  // new bodies: the next delta snapshot starts a sequence
  State<T>& s = state_of<T>(c);
  using T2 = typename State<T>::T2;
  s.n = n;
  // tree_walk_wave reads leaf particles 8 at a time: padded; a multi-device context asks for whole blocks (row_capacity)
  const size_t nn = (size_t)std::max<int64_t>(n > 0 ? n : 1, c->row_capacity) + 16;
  for (auto& st : s.set) {
    HIPCHK(c, hipMalloc((void**)&st.pos, nn * sizeof(T2)));
    HIPCHK(c, hipMalloc((void**)&st.vel, nn * sizeof(T2)));
    HIPCHK(c, hipMalloc((void**)&st.weight, nn * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&st.ids, nn * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&st.mass, nn * sizeof(T)));
  }
  HIPCHK(c, hipMalloc((void**)&s.pos_next, nn * sizeof(T2)));
  HIPCHK(c, hipMalloc((void**)&s.acc, nn * sizeof(T2)));
  HIPCHK(c, hipMalloc((void**)&s.order_dev, nn * sizeof(uint32_t)));
  s.cur = 0;
  s.h_weight.resize((size_t)n);
  std::vector<uint32_t> ids((size_t)n);
  std::vector<T> mass((size_t)n);
  bool uniform = n > 0;
  for (int64_t i = 0; i < n; ++i) {
    s.h_weight[(size_t)i] = w ? w[i] : 1u;
    uniform = uniform && s.h_weight[(size_t)i] == s.h_weight[0];
    ids[(size_t)i] = (uint32_t)i;
    mass[(size_t)i] = (T)s.h_weight[(size_t)i];  // `weight as f32`
  }
  if (n > 0) {
    auto& st = s.set[0];
    HIPCHK(c, hipMemcpyAsync(st.pos, pos, (size_t)n * sizeof(T2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(st.vel, vel, (size_t)n * sizeof(T2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(st.weight, s.h_weight.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(st.ids, ids.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(st.mass, mass.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  s.h_weight_stale = false;
  ++s.row_epoch;
  s.uniform_mass = (uniform && s.h_weight[0] > 0) ? (float)s.h_weight[0] : 0.f;
  s.sparse_base = 0.f;
  if (!uniform && n > 0) {  // one mass but for a few bodies?  (majority vote, then a count)
    uint32_t cand = 0;
    int64_t votes = 0, odd = 0;
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t wv = s.h_weight[(size_t)i];
      if (votes == 0) { cand = wv; votes = 1; }
      else votes += (wv == cand) ? 1 : -1;
    }
    for (int64_t i = 0; i < n; ++i) odd += s.h_weight[(size_t)i] != cand;
    if (cand > 0 && odd <= n / 256) s.sparse_base = (float)cand;
  }
  s.tree_valid = false;
  if (sizeof(T) == 4) c->has_f32 = true; else c->has_f64 = true;
  return NBODY_OK;
}

template <class T> int download(nbody_ctx* c, T* pos, T* vel, uint32_t* w, uint32_t* ids) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "download: no particles of this precision uploaded");
  State<T>& s = state_of<T>(c);
  using T2 = typename State<T>::T2;
  HIPCHK(c, hipSetDevice(c->device));
  auto& st = s.set[s.cur];
  const size_t n = (size_t)s.n;
  if (n) {
    if (pos) HIPCHK(c, hipMemcpyAsync(pos, st.pos, n * sizeof(T2), hipMemcpyDeviceToHost, c->stream));
    if (vel) HIPCHK(c, hipMemcpyAsync(vel, st.vel, n * sizeof(T2), hipMemcpyDeviceToHost, c->stream));
    if (w) HIPCHK(c, hipMemcpyAsync(w, st.weight, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (ids) HIPCHK(c, hipMemcpyAsync(ids, st.ids, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return NBODY_OK;
}

// -------------------------------------------------------------------------------------------- tree phases
template <class T> int ensure_node_buffers(nbody_ctx* c, State<T>& s, size_t m) {
  using G4 = typename TreeHost<T>::G4;
  using L4 = typename TreeHost<T>::L4;
  if (m > s.node_cap) {
    free_dev(s.geom0); free_dev(s.geom1); free_dev(s.link);
    s.node_cap = 0;
    size_t cap = m + m / 4 + 64;
    HIPCHK(c, hipMalloc(&s.geom0, cap * sizeof(G4)));
    HIPCHK(c, hipMalloc(&s.geom1, cap * sizeof(G4)));
    HIPCHK(c, hipMalloc(&s.link, cap * sizeof(L4)));
    s.node_cap = cap;
  }
  return NBODY_OK;
}

template <class T> int upload_tree(nbody_ctx* c, State<T>& s) {
  const size_t m = s.tree.size();
  using G4 = typename TreeHost<T>::G4;
  using L4 = typename TreeHost<T>::L4;
  int rc0 = ensure_node_buffers<T>(c, s, m);
  if (rc0) return rc0;
  HIPCHK(c, hipMemcpyAsync(s.geom0, s.tree.geom0.data(), m * sizeof(G4), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(s.geom1, s.tree.geom1.data(), m * sizeof(G4), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(s.link, s.tree.link.data(), m * sizeof(L4), hipMemcpyHostToDevice, c->stream));
  if (s.n) HIPCHK(c, hipMemcpyAsync(s.order_dev, s.tree.order.data(), (size_t)s.n * 4, hipMemcpyHostToDevice, c->stream));
  return NBODY_OK;
}

template <class T> int ensure_node_aux(nbody_ctx* c, State<T>& s, size_t m) {
  if (m > s.node_aux_cap) {
    free_dev(s.node_depth); free_dev(s.node_mass); free_dev(s.node_size);
    s.node_aux_cap = 0;
    size_t cap = m + m / 4 + 64;
    HIPCHK(c, hipMalloc((void**)&s.node_depth, cap * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&s.node_mass, cap * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&s.node_size, cap * sizeof(typename State<T>::T2)));
    s.node_aux_cap = cap;
  }
  return NBODY_OK;
}

// BVH built on the device (bvh_build.hip, f32 only).  Returns NBODY_OK, an error, or 1 when the device build declines.
// The same for f64 rows (bvh_build64.hip): every level enqueued blind, one question at the end.
int bvh_build_device64(nbody_ctx* c, State<double>& s) {
  const int n = (int)s.n;
  const int leaf = c->params.leaf_size;
  const Bvh64Layout L = bvh64_layout(n, leaf);
  if (s.bb_scratch_bytes < L.total) {
    free_dev(s.bb_scratch);
    s.bb_scratch_bytes = 0;
    HIPCHK(c, hipMalloc((void**)&s.bb_scratch, L.total));
    s.bb_scratch_bytes = L.total;
  }
  int rc = ensure_node_buffers<double>(c, s, (size_t)L.node_cap);
  if (rc) return rc;
  rc = ensure_node_aux<double>(c, s, (size_t)L.node_cap);
  if (rc) return rc;
  auto& in = s.set[s.cur];
  auto& out = s.set[1 - s.cur];
  HIPCHK(c, bvh64_begin(c->stream, in.pos, n, s.bb_scratch, L));
  // the levels a balanced tree has, plus a margin (the mean split is not the median: real trees run a few levels deeper);
  // a tree that is deeper still goes on four levels at a time
  int lv_end = std::min(bvh64_first_levels(n, leaf) + 4, kB64Levels);
  lv_end = std::max(1, std::min(lab_int("NBODY_BVH_BLIND_LEVELS", lv_end), kB64Levels));  // tests force the long way
  int hostf[kB64FlagWords + kB64Levels + 2];
  auto finish_and_ask = [&]() -> int {
    HIPCHK(c, bvh64_finish(c->stream, in.weight, n, lv_end, s.bb_scratch, L, s.order_dev, s.geom0, s.geom1, s.link, s.node_depth, s.node_mass,
                           s.node_size));
    HIPCHK(c, hipMemcpyAsync(hostf, s.bb_scratch + L.flags, kB64FlagWords * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hostf + kB64FlagWords, s.bb_scratch + L.opencount, (kB64Levels + 2) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NBODY_OK;
  };
  HIPCHK(c, bvh64_levels(c->stream, n, leaf, 0, lv_end, s.bb_scratch, L));
  rc = finish_and_ask();
  if (rc) return rc;
  while (hostf[kB64Fallback] == 0 && hostf[kB64FlagWords + lv_end] != 0) {  // open nodes were left behind
    if (lv_end >= kB64Levels) return 1;  // deeper than the device follows (coincident points): the host builder reports it
    const int lv = lv_end;
    lv_end = std::min(lv_end + 4, kB64Levels);
    HIPCHK(c, bvh64_levels(c->stream, n, leaf, lv, lv_end, s.bb_scratch, L));
    rc = finish_and_ask();
    if (rc) return rc;
  }
  if (env_int("NBODY_TRACE", 0) != 0)
    std::fprintf(stderr, "[nbody] device bvh build (f64): %d nodes, depth %d, %d levels enqueued, %d scan restarts, %d prepared runs used, fallback %d\n",
                 hostf[kB64NodeCount], hostf[kB64MaxDepth], lv_end, hostf[kB64Stops], hostf[kB64RunsUsed], hostf[kB64Fallback]);
  const int m = hostf[kB64NodeCount];
  if (hostf[kB64Fallback] != 0 || m <= 0 || m > L.node_cap) return 1;
  GatherArgs<double> g{};  // rows into tree order, as the in-place partition leaves `self.particles` (bvh_tree.rs:73-77)
  g.perm = s.order_dev;
  g.n = n;
  g.pos_in = in.pos; g.pos_out = out.pos;
  g.weight_in = in.weight;
  g.mass_out = out.mass;
  g.vel_in = in.vel; g.vel_out = out.vel;
  g.weight_out = out.weight;
  g.ids_in = in.ids; g.ids_out = out.ids;
  HIPCHK(c, launch_gather<double>(c->stream, g));
  s.cur = 1 - s.cur;
    ++s.row_epoch;
  s.h_weight_stale = true;
  s.n_nodes = m;
  s.tree_kind = NBODY_TREE_BVH;
  s.tree_max_depth = hostf[kB64MaxDepth];
  s.tree_host_stale = true;
  s.tree_valid = true;
  c->bvh_stops = hostf[kB64Stops];
  return NBODY_OK;
}

template <class T> int bvh_build_device(nbody_ctx* c, State<T>& s) {
  if constexpr (!std::is_same<T, float>::value) {
    return env_int("NBODY_TREE_BUILD_HOST", 0) != 0 ? 1 : bvh_build_device64(c, s);
  } else {
    const int n = (int)s.n;
    const int leaf = c->params.leaf_size;
    BvhBuildLayout L = bvh_build_layout(n, leaf);
    if (s.bb_scratch_bytes < L.total) {
      free_dev(s.bb_scratch);
      s.bb_scratch_bytes = 0;
      HIPCHK(c, hipMalloc((void**)&s.bb_scratch, L.total));
      s.bb_scratch_bytes = L.total;
    }
    auto& in = s.set[s.cur];
    auto& out = s.set[1 - s.cur];
    s.bb_flags_clean = false;
    HIPCHK(c, bvh_build_begin(c->stream, in.pos, n, s.bb_scratch, L));
    int rc = ensure_node_buffers<T>(c, s, (size_t)L.node_cap);
    if (rc) return rc;
    rc = ensure_node_aux<T>(c, s, (size_t)L.node_cap);
    if (rc) return rc;
    // Everything is enqueued blind — the long-node levels a balanced tree has (plus two), the subtrees, the numbering,
    // the row gather — and checked once at the end: asking in between costs a round trip per question, an empty level
    // a few microseconds.  A lopsided tree still has long nodes then: levels two at a time (asking after each pair)
    // until none is left, then the tail once more for the subtrees that were not there the first time.
    const int first_levels = bvh_build_first_levels(n);
    int lv_end = first_levels > 0 ? first_levels + 2 : 0;
    if (lv_end > 0) lv_end = std::max(1, lab_int("NBODY_BVH_BLIND_LEVELS", lv_end));  // tests force the lopsided path
    if (lv_end > kBvhKeyDepth + 1) lv_end = kBvhKeyDepth + 1;
    int hostf[kBvhFlagWords + kBvhLevels];
    auto tail = [&](int sub_start) -> int {
      GatherArgs<T> g{};  // rows into tree order, as the in-place partition leaves `self.particles` (bvh_tree.rs:73-77)
      g.perm = bvh_build_order(s.bb_scratch, L);  // read where the build left it, and copied out on the way
      g.perm_copy = s.order_dev;
      g.n = n;
      g.pos_in = in.pos; g.pos_out = out.pos;
      g.weight_in = in.weight;
      g.mass_out = out.mass;
      g.vel_in = in.vel; g.vel_out = out.vel;
      g.weight_out = out.weight;
      g.ids_in = in.ids; g.ids_out = out.ids;
      HIPCHK(c, bvh_build_finish(c->stream, in.weight, n, leaf, sub_start, s.bb_scratch, L, nullptr, s.geom0, s.geom1, s.link,
                                 s.node_depth, s.node_mass, s.node_size, &g));  // (numbering and row gather in one launch)
      return NBODY_OK;
    };
    auto ask = [&]() -> int {
      HIPCHK(c, hipMemcpyAsync(hostf, s.bb_scratch + L.flags, kBvhFlagWords * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(hostf + kBvhFlagWords, s.bb_scratch + L.bigcount, kBvhLevels * sizeof(int), hipMemcpyDeviceToHost,
                               c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      return NBODY_OK;
    };
    if (lv_end > 0) HIPCHK(c, bvh_build_levels(c->stream, n, leaf, 0, lv_end, s.bb_scratch, L));
    rc = tail(0);
    if (rc) return rc;
    rc = ask();
    if (rc) return rc;
    if (hostf[kBvhFallback] != 0) return 1;
    if (lv_end > 0 && hostf[kBvhFlagWords + lv_end] != 0) {  // long nodes were left behind
      const int sub_start = hostf[kBvhSubCount];
      while (hostf[kBvhFlagWords + lv_end] != 0) {
        if (lv_end >= kBvhKeyDepth + 1) return 1;
        const int lv = lv_end;
        lv_end = lv_end + 2 > kBvhKeyDepth + 1 ? kBvhKeyDepth + 1 : lv_end + 2;
        HIPCHK(c, bvh_build_levels(c->stream, n, leaf, lv, lv_end, s.bb_scratch, L));
        rc = ask();
        if (rc) return rc;
        if (hostf[kBvhFallback] != 0) return 1;
      }
      rc = tail(sub_start);
      if (rc) return rc;
      rc = ask();
      if (rc) return rc;
      if (hostf[kBvhFallback] != 0) return 1;
    }
    if (env_int("NBODY_TRACE", 0) != 0)
      std::fprintf(stderr, "[nbody] device bvh build: %d nodes, depth %d, %d subtrees, %d long-node levels, %d scan restarts, %d prepared chunk runs used\n",
                   hostf[kBvhNodes], hostf[kBvhMaxDepth], hostf[kBvhSubCount], lv_end, hostf[kBvhStops], hostf[kBvhRunsUsed]);
#ifdef NB_BVH_TIMING
    std::fprintf(stderr, "[nbody] bvh_subtrees, slowest group per phase (10 ns ticks): load %d, level 1 %d, level 2 %d, level 3 %d, other levels %d, leaves %d, upward %d, store %d\n",
                 hostf[kBvhDebug], hostf[kBvhDebug + 1], hostf[kBvhDebug + 2], hostf[kBvhDebug + 3], hostf[kBvhDebug + 4], hostf[kBvhDebug + 5],
                 hostf[kBvhDebug + 6], hostf[kBvhDebug + 7]);
#endif
    const int m = hostf[kBvhNodes];
    if (m <= 0 || m > L.node_cap || hostf[kBvhNodeCount] > L.node_cap || hostf[kBvhBadIndex] != 0) return 1;
    {  // how many long-node levels this tree had: what a step enqueued ahead of the host enqueues blind next time
      int used = 0;
      while (used < kBvhLevels - 1 && hostf[kBvhFlagWords + used] != 0) ++used;
      s.bvh_levels_hint = used;
      s.bvh_levels_stable = 0;
    }
    s.cur = 1 - s.cur;
    ++s.row_epoch;
    s.h_weight_stale = true;
    s.n_nodes = m;
    s.tree_kind = NBODY_TREE_BVH;
    s.tree_max_depth = hostf[kBvhMaxDepth];
    s.tree_host_stale = true;
    s.tree_valid = true;
    c->bvh_stops = hostf[kBvhStops];
    return NBODY_OK;
  }
}

// Quad tree built on the device (quad_build.hip).  Returns NBODY_OK, an error, or 1 when the device build declines.
template <class T> int quad_build_device(nbody_ctx* c, State<T>& s) {
  const int n = (int)s.n;
  QuadBuildLayout L = quad_build_layout(n);
  if (s.qb_scratch_bytes < L.total) {
    free_dev(s.qb_scratch);
    s.qb_scratch_bytes = 0;
    HIPCHK(c, hipMalloc((void**)&s.qb_scratch, L.total));
    s.qb_scratch_bytes = L.total;
  }
  auto& in = s.set[s.cur];
  auto& out = s.set[1 - s.cur];
  const T rx = (T)c->params.quad_root_x, ry = (T)c->params.quad_root_y, rh = (T)c->params.quad_root_h;
  // the sorts only look at as many levels as the tree is expected to have: the last quad tree's depth plus three (all 31
  // the first time, and again whenever that turns out to be too few)
  int sort_levels = s.quad_depth_hint > 0 ? s.quad_depth_hint + 3 : 31;
  int flags[3] = {0, 0, 0};
  for (;;) {
    HIPCHK(c, quad_build_phase_a<T>(c->stream, in.pos, n, rx, ry, rh, s.qb_scratch, L, s.order_dev, sort_levels));
    HIPCHK(c, hipMemcpyAsync(flags, s.qb_scratch + L.flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    {
      // The leaves' own copies of their points, in tree order: enqueued before the host asks for the node count, so the
      // device gathers while the host waits, and the leaf statistics of phase B read rows that lie side by side.
      GatherArgs<T> g{};
      g.perm = s.order_dev;
      g.n = n;
      g.pos_in = in.pos; g.pos_out = out.pos;
      g.weight_in = in.weight;
      g.weight_out = out.weight;
      g.mass_out = out.mass;
      HIPCHK(c, launch_gather<T>(c->stream, g));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (env_int("NBODY_TRACE", 0) != 0)
      std::fprintf(stderr, "[nbody] device quad build: sorted by %d levels, flags %d, %d nodes, depth %d\n", sort_levels, flags[0], flags[1], flags[2]);
    if ((flags[0] & 2) != 0 && sort_levels < 31) { sort_levels = 31; continue; }
    break;
  }
  if ((flags[0] & 1) != 0 || flags[1] <= 0) { s.quad_depth_hint = 0; return 1; }
  s.quad_depth_hint = flags[2];
  const int m = flags[1];
  int rc = ensure_node_buffers<T>(c, s, (size_t)m);
  if (rc) return rc;
  rc = ensure_node_aux<T>(c, s, (size_t)m);
  if (rc) return rc;
  HIPCHK(c, quad_build_phase_b<T>(c->stream, out.pos, out.weight, n, rx, ry, rh, s.qb_scratch, L, nullptr, m, flags[2],
                                  s.geom0, s.geom1, s.link, s.node_depth, s.node_mass));
  s.n_nodes = m;
  s.tree_kind = NBODY_TREE_QUAD;
  s.tree_max_depth = flags[2];
  s.tree_host_stale = true;
  s.tree_valid = true;
  return NBODY_OK;
}

// Host image of a device-built tree, for the export API.
template <class T> int download_tree(nbody_ctx* c, State<T>& s) {
  if (!s.tree_host_stale) return NBODY_OK;
  using G4 = typename TreeHost<T>::G4;
  using L4 = typename TreeHost<T>::L4;
  const size_t m = (size_t)s.n_nodes;
  auto& t = s.tree;
  t.clear();
  t.kind = s.tree_kind;
  t.max_depth = s.tree_max_depth;
  t.geom0.resize(m); t.geom1.resize(m); t.link.resize(m); t.mass_u32.resize(m); t.order.resize((size_t)s.n);
  HIPCHK(c, hipMemcpyAsync(t.geom0.data(), s.geom0, m * sizeof(G4), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(t.geom1.data(), s.geom1, m * sizeof(G4), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(t.link.data(), s.link, m * sizeof(L4), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(t.mass_u32.data(), s.node_mass, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (s.n) HIPCHK(c, hipMemcpyAsync(t.order.data(), s.order_dev, (size_t)s.n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  t.size_x.resize(m); t.size_y.resize(m);
  if (s.tree_kind == NBODY_TREE_BVH) {  // boundary.size as the build computed it (max - min)
    std::vector<typename State<T>::T2> sz(m);
    HIPCHK(c, hipMemcpy(sz.data(), s.node_size, m * sizeof(typename State<T>::T2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; ++i) { t.size_x[i] = (T)sz[i].x; t.size_y[i] = (T)sz[i].y; }
    s.tree_host_stale = false;
    return NBODY_OK;
  }
  for (size_t i = 0; i < m; ++i) {
    // height is not stored on the device; hi - lo would round.  Recover it exactly from the parent chain:
    // a child's height is its parent's height / 2 (quad_tree.rs:172), the root's is the parameter.
    t.size_x[i] = t.size_y[i] = T(0);
  }
  {
    std::vector<int> depth(m);
    HIPCHK(c, hipMemcpy(depth.data(), s.node_depth, m * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<T> hd((size_t)t.max_depth + 2);
    hd[0] = (T)c->params.quad_root_h;
    for (size_t d = 1; d < hd.size(); ++d) hd[d] = hd[d - 1] / (T)2.0;
    for (size_t i = 0; i < m; ++i) t.size_x[i] = t.size_y[i] = hd[(size_t)depth[i]];
  }
  s.tree_host_stale = false;
  return NBODY_OK;
}

// ---- phase timing by events (see PhaseEvents in ctx.h)
// `prev`: the step enqueued just before this one, with nothing in between — its end event doubles as this step's start
// (every recorded event is a marker in the queue, ~6 us of idle stream: two back to back would be the largest gap of a step).
// ---- ... and by the kernels' own clock for the steps enqueued ahead (ctx.h, stamp_*): no event records between the phases
constexpr int kStampSlots = 256;
int close_open_stamp(nbody_ctx* c) {  // the last stamped step's end, when no stamped step follows it directly
  if (c->stamp_open < 0) return NBODY_OK;
  HIPCHK(c, launch_stamp(c->stream, c->stamp_dev + 4 * (size_t)c->stamp_open + 3));
  c->stamp_open = -1;
  return NBODY_OK;
}
int phase_begin(nbody_ctx* c, PhaseEvents* out, const PhaseEvents* prev = nullptr) {
  if (int rc = close_open_stamp(c)) return rc;
  PhaseEvents p;
  for (int k = prev ? 1 : 0; k < 4; ++k) {
    if (!c->ph_free.empty()) {
      p.e[k] = c->ph_free.back();
      c->ph_free.pop_back();
    } else {
      HIPCHK(c, hipEventCreate(&p.e[k]));
    }
  }
  if (prev) {
    p.e[0] = prev->e[3];
    p.borrowed = true;
  } else {
    HIPCHK(c, hipEventRecord(p.e[0], c->stream));
  }
  *out = p;
  return NBODY_OK;
}
int phase_mark(nbody_ctx* c, const PhaseEvents& p, int k) {
  HIPCHK(c, hipEventRecord(p.e[k], c->stream));
  if (k == 3) c->ph_pending.push_back(p);
  return NBODY_OK;
}
// Reads every recorded step's phases into the context's (and the caller's) Counting.  Waits for them.
int phase_drain(nbody_ctx* c) {
  if (int rc = close_open_stamp(c)) return rc;
  if (!c->stamp_pending.empty()) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> h((size_t)kStampSlots * 4);
    HIPCHK(c, hipMemcpy(h.data(), c->stamp_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int slot : c->stamp_pending) {
      const unsigned long long* t = h.data() + 4 * (size_t)slot;
      const double sec[3] = {1e-8 * (double)(long long)(t[1] - t[0]), 1e-8 * (double)(long long)(t[2] - t[1]), 1e-8 * (double)(long long)(t[3] - t[2])};  // 100 MHz ticks
      c->counting.build_bvh += sec[0];
      c->counting.sum_gravity += sec[1];
      c->counting.post_calculations += sec[2];
      if (c->ph_counter) {
        c->ph_counter->build_bvh += sec[0];
        c->ph_counter->sum_gravity += sec[1];
        c->ph_counter->post_calculations += sec[2];
      }
    }
    c->stamp_pending.clear();
  }
  for (auto& p : c->ph_pending) {
    HIPCHK(c, hipEventSynchronize(p.e[3]));
    float ms[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) HIPCHK(c, hipEventElapsedTime(&ms[k], p.e[k], p.e[k + 1]));
    c->counting.build_bvh += 1e-3 * ms[0];
    c->counting.sum_gravity += 1e-3 * ms[1];
    c->counting.post_calculations += 1e-3 * ms[2];
    if (c->ph_counter) {
      c->ph_counter->build_bvh += 1e-3 * ms[0];
      c->ph_counter->sum_gravity += 1e-3 * ms[1];
      c->ph_counter->post_calculations += 1e-3 * ms[2];
    }
  }
  for (auto& p : c->ph_pending)
    for (int k = p.borrowed ? 1 : 0; k < 4; ++k) c->ph_free.push_back(p.e[k]);
  c->ph_pending.clear();
  return NBODY_OK;
}

// The host's mirror of the weights, in the current row order.
template <class T> int refresh_host_weights(nbody_ctx* c, State<T>& s) {
  if (!s.h_weight_stale) return NBODY_OK;
  const int64_t n = s.n;
  s.h_weight.resize((size_t)n);
  if (n) HIPCHK(c, hipMemcpyAsync(s.h_weight.data(), s.set[s.cur].weight, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  s.h_weight_stale = false;
  return NBODY_OK;
}

// A linearised tree in s.tree (the host builders', or a caller's: walk_tree) becomes the tree the walks use: its records go
// to the device and the rows into its order, as the in-place partition leaves `self.particles` (bvh_tree.rs:73-77); for the
// quad tree the leaf-ordered copies.  s.h_weight must be current (refresh_host_weights).
template <class T> int install_host_tree(nbody_ctx* c, State<T>& s, int kind) {
  const int64_t n = s.n;
  s.n_nodes = (int)s.tree.size();
  s.tree_kind = s.tree.kind;
  s.tree_max_depth = s.tree.max_depth;
  int rc = upload_tree(c, s);
  if (rc) return rc;
  auto& in = s.set[s.cur];
  auto& out = s.set[1 - s.cur];
  GatherArgs<T> g{};
  g.perm = s.order_dev;
  g.n = n;
  g.pos_in = in.pos; g.pos_out = out.pos;
  g.weight_in = in.weight;
  g.mass_out = out.mass;
  if (kind == NBODY_TREE_BVH) {
    g.vel_in = in.vel; g.vel_out = out.vel;
    g.weight_out = out.weight;
    g.ids_in = in.ids; g.ids_out = out.ids;
    HIPCHK(c, launch_gather<T>(c->stream, g));
    s.cur = 1 - s.cur;
    ++s.row_epoch;
    // host mirror of the row order
    s.h_tmp.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) s.h_tmp[(size_t)i] = s.h_weight[s.tree.order[(size_t)i]];
    s.h_weight.swap(s.h_tmp);
  } else {
    HIPCHK(c, launch_gather<T>(c->stream, g));
  }
  s.tree_valid = true;
  return NBODY_OK;
}


// Phase 1 of update (main.rs:398-401): snapshot + build + upward pass.  After it, for the BVH, set[cur] holds the
// permuted particles and set[1-cur].pos the pre-build snapshot (`cloned`); for the quad tree set[1-cur].pos/.mass
// hold the leaf-ordered copies the leaves own.
template <class T> int tree_build_phase(nbody_ctx* c, State<T>& s, int kind) {
  using T2 = typename State<T>::T2;
  if (kind != NBODY_TREE_BVH && kind != NBODY_TREE_QUAD) return fail(c, NBODY_ERR_INVALID, "unknown tree kind");
  const int64_t n = s.n;
  s.tree_valid = false;
  s.tree_host_stale = false;
  c->last_build_device = true;
  c->bvh_stops = 0;
  if (kind == NBODY_TREE_QUAD && n > 0 && env_int("NBODY_TREE_BUILD_HOST", 0) == 0) {
    int rc = quad_build_device<T>(c, s);
    if (rc != 1) return rc;  // 1 = the device build declined (too deep for its key / sizes): host builder below
  }
  if (kind == NBODY_TREE_BVH && n > 0 && c->params.leaf_size >= 1 && env_int("NBODY_TREE_BUILD_HOST", 0) == 0) {
    int rc = bvh_build_device<T>(c, s);
    if (rc != 1) return rc;
  }
  c->last_build_device = false;
  int rcw = refresh_host_weights<T>(c, s);  // the host builder reads the weights in the current row order
  if (rcw) return rcw;
  const bool trace = env_int("NBODY_TRACE", 0) != 0;
  double tt0 = now_s();
  s.h_pos.resize((size_t)(2 * n));
  if (n) {
    HIPCHK(c, hipMemcpyAsync(s.h_pos.data(), s.set[s.cur].pos, (size_t)n * sizeof(T2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  double tt1 = now_s();
  if (kind == NBODY_TREE_BVH) {
    if (c->params.leaf_size < 1) return fail(c, NBODY_ERR_INVALID, "leaf_size must be >= 1");
    build_bvh<T>(s.h_pos.data(), s.h_weight.data(), n, c->params.leaf_size, s.tree);
  } else {
    build_quad<T>(s.h_pos.data(), s.h_weight.data(), n, (T)c->params.quad_root_x, (T)c->params.quad_root_y,
                  (T)c->params.quad_root_h, s.tree);
  }
  double tt2 = now_s();
  if (trace) std::fprintf(stderr, "[nbody] host tree build: D2H %.3f ms, build %.3f ms (%zu nodes)\n", 1e3 * (tt1 - tt0), 1e3 * (tt2 - tt1), s.tree.size());
  if (s.tree.overflow)
    return fail(c, NBODY_ERR_DEGENERATE, "tree build exceeded the depth cap (more coincident points than a leaf holds)");
  return install_host_tree<T>(c, s, kind);
}

// Phase 2 (main.rs:406-416).  tgt_pos == nullptr: the particles themselves.
template <class T>
int tree_walk_phase(nbody_ctx* c, State<T>& s, int kind, const void* tgt_pos, int64_t n_tgt, void* acc,
                    int64_t slice_begin = 0, int64_t slice_count = -1) {
  using T2w = typename State<T>::T2;
  WalkArgs<T> w{};
  w.geom0 = s.geom0; w.geom1 = s.geom1; w.link = s.link;
  w.n_nodes = s.n_nodes;
  w.big_leaves = kind == NBODY_TREE_BVH && c->params.leaf_size >= 16;
  w.fast = c->params.arith == NBODY_ARITH_FAST;  // AUTO and EXACT walk with the reference's operations
  w.theta = (T)c->params.theta;
  w.clamp = (T)c->params.clamp;
  w.acc = acc;
  w.stats = c->want_stats ? c->stats_dev : nullptr;
  if (w.stats) HIPCHK(c, hipMemsetAsync(c->stats_dev, 0, 3 * sizeof(unsigned long long), c->stream));
  if (kind == NBODY_TREE_BVH) {
    w.leaf_pos = s.set[s.cur].pos;
    w.leaf_mass = s.set[s.cur].mass;
    if (tgt_pos) { w.tgt_pos = tgt_pos; w.n_tgt = n_tgt; }
    else {
      // AS_WRITTEN: accelerations are computed for the snapshot's rows (main.rs:406-412 iterate `cloned`)
      const T2w* base = (c->params.order == NBODY_ORDER_AS_WRITTEN) ? s.set[1 - s.cur].pos : s.set[s.cur].pos;
      if (slice_count >= 0) {  // a contiguous block of rows (tree order = row order after the build's permutation)
        w.tgt_pos = base + slice_begin;
        w.acc = (T2w*)acc + slice_begin;
        w.n_tgt = slice_count;
      } else {
        w.tgt_pos = base;
        w.n_tgt = s.n;
      }
    }
  } else {
    w.leaf_pos = s.set[1 - s.cur].pos;
    w.leaf_mass = s.set[1 - s.cur].mass;
    if (tgt_pos) { w.tgt_pos = tgt_pos; w.n_tgt = n_tgt; }
    else if (slice_count >= 0) { w.tgt_pos = s.set[s.cur].pos; w.n_tgt = slice_count; w.tgt_index = s.order_dev + slice_begin; }
    else { w.tgt_pos = s.set[s.cur].pos; w.n_tgt = s.n; w.tgt_index = s.order_dev; }
  }
  bool done = false;
  {
    // Big leaves: a leaf's terms are evaluated lane = particle (walk_split.hip): in one pass with the terms handed over
    // through LDS (walk_tile), or in three passes through a term array.  NBODY_WALK_SPLIT: 0 never (fused walk), 1 one pass
    // when it pays (default), 3 one pass whenever possible; laboratory build only: 4 / 2 three passes when it pays / whenever
    // possible (the round-1 design the one-pass walk replaced; the product treats them as 1).
    const int mode = walk_split_mode();
    const bool tile_mode = mode == 3 || (mode == 1 && w.n_tgt >= 4096);
    const bool eligible = w.big_leaves && !w.stats && w.n_tgt > 0 && w.n_nodes > 0 && lab_int("NBODY_WALK_PER_THREAD", 0) == 0;
    if (eligible && tile_mode && (mode == 3 || s.ws_backoff == 0)) {  // one pass, terms through LDS (walk_tile)
      const WalkSplitLayout L = walk_split_layout(w.n_tgt);
      if (s.ws_scratch_bytes < L.total) {
        free_dev(s.ws_scratch);
        s.ws_scratch_bytes = 0;
        s.wt_hist_n = -1;
        HIPCHK(c, hipMalloc((void**)&s.ws_scratch, L.total));
        s.ws_scratch_bytes = L.total;
        HIPCHK(c, hipMemsetAsync(s.ws_scratch + L.scan_state, 0, L.scan_state_bytes, c->stream));  // walk_scan_est_tail keeps its books there
      }
      const bool self = tgt_pos == nullptr;
      if (self && !s.wt_hist) {
        HIPCHK(c, hipMalloc((void**)&s.wt_hist, (size_t)(s.n > 0 ? s.n : 1) * 4));
        HIPCHK(c, hipMemsetAsync(s.wt_hist, 0, (size_t)(s.n > 0 ? s.n : 1) * 4, c->stream));  // a shard's slice never writes the other ids
        s.wt_hist_n = -1;
      }
      // the targets' particle ids (the snapshot's rows under AS_WRITTEN, the permuted rows otherwise)
      const uint32_t* tgt_ids = !self ? nullptr
                                : ((c->params.order == NBODY_ORDER_AS_WRITTEN) ? s.set[1 - s.cur].ids : s.set[s.cur].ids) + (slice_count >= 0 ? slice_begin : 0);
      const bool hist = self && s.wt_hist_n == w.n_tgt && s.wt_hist_begin == slice_begin && lab_int("NBODY_WALK_TILE_COUNT", 0) == 0;
      int shift = 0;
      while (hist && (s.wt_total >> shift) >= (1ull << 31)) ++shift;
      if (hist && lab_int("NBODY_WALK_TILE_POISON", 0) != 0)  // test hook: a history whose scan wraps must be noticed
        HIPCHK(c, hipMemsetAsync(s.wt_hist, 0xFF, (size_t)s.n * 4, c->stream));
      int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      unsigned long long total = 0;
      for (int estimate = hist ? 1 : 0; !done; estimate = 2) {
        {
          TimerScope ts(c->timer, c->stream);
          HIPCHK(c, launch_tree_walk_tile(c->stream, w, s.ws_scratch, L, tgt_ids, self ? s.wt_hist : nullptr, estimate, shift));
        }
        HIPCHK(c, hipMemcpyAsync(info, s.ws_scratch + L.info, sizeof(info), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(&total, &info[6], 8);
        if (env_int("NBODY_TRACE", 0) != 0)
          std::fprintf(stderr, "[nbody] tile walk: %llu terms, estimate %s (shift %d, total %d), %d per wave, overflow %d\n", total,
                       estimate == 1 ? "from the last walk" : (estimate == 0 ? "counted" : "none"), shift, info[0], info[3], info[1]);
        // an estimate whose scan does not fit (a counted one past 2^32 terms; counts of older walks under another theta
        // in a shard's new slice): walk without one (the counts it leaves behind are scaled next time)
        done = info[1] == 0;
        if (!done && estimate == 2) return fail(c, NBODY_ERR_HIP, "tile walk: overflow flag without an estimate");
      }
      s.wt_hist_n = self ? w.n_tgt : -1;
      s.wt_hist_begin = slice_begin;
      s.wt_total = total;
      // a walk in which the average target takes a sixteenth of all particles (small theta on the needle boxes) is nearly
      // the direct sum: every lane wants every leaf and the fused walk's lane = target is the cheaper arrangement; look
      // again in 64 walks
      if (mode != 3 && (double)total > (double)w.n_tgt * (double)s.n / 16.0) s.ws_backoff = 64;
#ifdef NBODY_LAB
    } else if (std::is_same<T, float>::value && eligible && (mode == 2 || (mode == 4 && w.n_tgt >= 4096 && s.ws_backoff == 0))) {
      s.wt_hist_n = -1;
      const int64_t hard_cap = ((int64_t)1 << 31) - 65536;  // terms (16 GB; the offsets are 32 bits wide); past that the fused walk
      const WalkSplitLayout L = walk_split_layout(w.n_tgt);
      if (s.ws_scratch_bytes < L.total) {
        free_dev(s.ws_scratch);
        s.ws_scratch_bytes = 0;
        HIPCHK(c, hipMalloc((void**)&s.ws_scratch, L.total));
        s.ws_scratch_bytes = L.total;
        HIPCHK(c, hipMemsetAsync(s.ws_scratch + L.scan_state, 0, L.scan_state_bytes, c->stream));
      }
      for (int attempt = 0; attempt < 2 && !done; ++attempt) {
        int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        {
          TimerScope ts(c->timer, c->stream);
          if constexpr (std::is_same<T, float>::value)
            HIPCHK(c, launch_tree_walk_split(c->stream, w, s.ws_scratch, L, s.ws_terms, s.ws_capacity));
        }
        HIPCHK(c, hipMemcpyAsync(info, s.ws_scratch + L.info, sizeof(info), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (env_int("NBODY_TRACE", 0) != 0)
          std::fprintf(stderr, "[nbody] split walk: %d terms, overflow %d, %d terms per term-pass wave; longest such wave %d us (leaf %d us, node %d us; timing builds)\n",
                       info[0], info[1], info[3], info[5] >> 20, (info[5] >> 10) & 1023, info[5] & 1023);
        if (info[1] == 0) {
          done = true;
        } else if (info[2] != 0 || info[0] > hard_cap) {
          s.ws_backoff = 64;  // too many terms for this tree: fused walk for a while
          break;
        } else {  // the term array was too small (or absent): half as much again, once
          free_dev(s.ws_terms);
          s.ws_capacity = 0;
          int64_t want = (int64_t)info[0] + info[0] / 2 + 4096;
          if (want > hard_cap) want = hard_cap;
          if (hipMalloc(&s.ws_terms, (size_t)want * sizeof(float2)) != hipSuccess) {  // no room: the fused walk needs none
            (void)hipGetLastError();
            s.ws_terms = nullptr;
            s.ws_backoff = 64;
            break;
          }
          s.ws_capacity = want;
        }
      }
#endif
    } else if (s.ws_backoff > 0 && eligible) {
      --s.ws_backoff;
    }
  }
  if (!done) {
    TimerScope ts(c->timer, c->stream);
    HIPCHK(c, launch_tree_walk<T>(c->stream, w, lab_int("NBODY_WALK_PER_THREAD", 0) == 0));
  }
  if (w.stats) {
    HIPCHK(c, hipMemcpyAsync(c->last_stats, c->stats_dev, sizeof(c->last_stats), hipMemcpyDeviceToHost, c->stream));
  }
  return NBODY_OK;
}

// A whole f32 BVH step enqueued AHEAD of the host's knowledge of it.  The plain sequence asks the device three
// questions per step (is the build complete?  did the walk's estimate wrap?  how many terms were there?) and the old
// step driver added a wait at every phase boundary: five round trips on a 1.2 ms step.  Here the stream gets, in one go,
//     build (blind: the levels the last tree had, plus one) -> verdict of the build, ON THE DEVICE -> row gather ->
//     the walk's preparation (estimate scan, wrap check, budget) -> a 0.5 KB copy of {verdict, flags, level counters,
//     walk info} to pinned memory + an event -> the walk kernel, reading the node count from device memory and
//     returning at once if the verdict or the preparation said no
// and the host waits for that EVENT only — it fires when the long kernel starts, so the integration and the whole next
// step's build are enqueued while the walk runs and the stream never drains between steps.  Everything the host
// decides on is known before the walk; the rows a step starts from stay intact until it has decided (the gather writes
// the other set, the integration is enqueued after the decision), so a step whose speculation fails is simply done
// again by the plain sequence.  The walk's exact term count (next estimate's scale) is read one step late.
// Returns NBODY_OK (step done), 1 (not applicable / speculation failed: take the plain sequence), or an error.
constexpr int kSpecWords = 2 + 128 + 8 + 8;  // verdict | flags + level counters (512 B) | info before the walk | info after it

template <class T> int step_ahead_collect(nbody_ctx* c, State<T>& s) {  // the previous ahead-step's term count, if one is due
  if (!s.ahead_total_due) return NBODY_OK;
  s.ahead_total_due = false;
  const int* post = c->spec_host + 2 + 128 + 8;
  unsigned long long total = 0;
  std::memcpy(&total, &post[6], 8);
  s.wt_total = total;
  if ((double)total > (double)s.n * (double)s.n / 16.0) s.ws_backoff = 64;  // nearly the direct sum: the fused walk for a while
  return NBODY_OK;
}

template <class T> int bvh_step_ahead(nbody_ctx* c, State<T>& s, T delta, PhaseEvents* chain) {
  if constexpr (!std::is_same<T, float>::value) {
    return 1;
  } else {
    const int n = (int)s.n;
    const int leaf = c->params.leaf_size;
    const int mode = walk_split_mode();
    if (n < 4096 || leaf < 16 || mode != 1 || c->want_stats || s.ws_backoff != 0) return 1;
    if (!s.wt_hist || s.wt_hist_n != n || s.wt_hist_begin != 0) return 1;  // no walk of these targets to estimate from yet
    if (env_int("NBODY_STEP_AHEAD", 1) == 0 || env_int("NBODY_TREE_BUILD_HOST", 0) != 0 || lab_int("NBODY_WALK_PER_THREAD", 0) != 0 ||
        lab_int("NBODY_WALK_TILE_COUNT", 0) != 0)
      return 1;
    const BvhBuildLayout L = bvh_build_layout(n, leaf);
    const WalkSplitLayout WL = walk_split_layout(n);
    if (s.bb_scratch_bytes < L.total || s.ws_scratch_bytes < WL.total || s.node_cap < (size_t)L.node_cap || s.node_aux_cap < (size_t)L.node_cap)
      return 1;  // the plain sequence sizes the buffers the first time
    static_assert(kBvhFlagWords + kBvhLevels <= 128, "flags and level counters travel as one 512-byte block");
    if (L.bigcount - L.flags + kBvhLevels * sizeof(int) > 128 * sizeof(int)) return 1;
    if (!c->spec_dev) {
      HIPCHK(c, hipMalloc((void**)&c->spec_dev, (2 + kSpecWords) * sizeof(int)));  // the verdict, then the record packed for the host
      HIPCHK(c, hipHostMalloc((void**)&c->spec_host, kSpecWords * sizeof(int), hipHostMallocMapped));
      HIPCHK(c, hipHostGetDevicePointer((void**)&c->spec_host_dev, c->spec_host, 0));
      HIPCHK(c, hipEventCreateWithFlags(&c->spec_event, hipEventDisableTiming));
    }
    // Phase timing: by the step's own kernels (the 100 MHz wall clock written at the three boundaries: no event records, each of
    // which leaves ~6 us of idle stream) when the walk's preparation is the one fused kernel; by events otherwise.
    const bool fused_scan = n <= std::min<int64_t>(kWalkFusedScanMaxTargets, lab_int("NBODY_WALK_FUSED_SCAN_MAX", (int)kWalkFusedScanMaxTargets));
    const bool stamps = fused_scan && lab_int("NBODY_PHASE_STAMPS", 1) != 0;
    PhaseEvents ph;
    int rc = NBODY_OK;
    unsigned long long* stamp = nullptr;  // this step's slot
    unsigned long long* stamp_prev_end = nullptr;
    int slot = -1;
    if (stamps) {
      if (!c->stamp_dev) {
        HIPCHK(c, hipMalloc((void**)&c->stamp_dev, (size_t)kStampSlots * 4 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->stamp_dev, 0, (size_t)kStampSlots * 4 * sizeof(unsigned long long), c->stream));
      }
      slot = c->stamp_next;
      c->stamp_next = (c->stamp_next + 1) % kStampSlots;
      stamp = c->stamp_dev + 4 * (size_t)slot;
      if (c->stamp_open >= 0) stamp_prev_end = c->stamp_dev + 4 * (size_t)c->stamp_open + 3;  // bvh_init closes the step before
      c->stamp_open = -1;
    } else {
      rc = phase_begin(c, &ph, chain->e[3] ? chain : nullptr);
    }
    *chain = PhaseEvents{};
    if (rc) return rc;
    auto& in = s.set[s.cur];
    auto& out = s.set[1 - s.cur];
    // ---- build: as many long-node levels as the last tree had, plus one (a balanced tree's, plus two, the first time)
    const int first_levels = bvh_build_first_levels(n);
    int lv_end = first_levels > 0 ? first_levels + 2 : 0;
    // (a lopsided tree has more than a balanced one + 2).  The spare level is four launches that find nothing to do (19 us of a
    // 1.1 ms step): once the count has stood for eight builds it is dropped — the verdict still checks that no long node is
    // left (bigcount[lv_end] == 0), and a tree that grows a level then costs ONE repeated step and brings the spare back.
    if (lv_end > 0 && s.bvh_levels_hint > 0)
      lv_end = s.bvh_levels_hint + ((s.bvh_levels_stable >= 8 && lab_int("NBODY_BVH_SPARE_LEVEL", 0) == 0) ? 0 : 1);
    if (lv_end > 0) lv_end = std::max(1, lab_int("NBODY_BVH_BLIND_LEVELS", lv_end));  // tests: too few levels, the verdict fails
    if (lv_end > kBvhKeyDepth + 1) lv_end = kBvhKeyDepth + 1;
    const bool flags_clean = s.bb_flags_clean;
    s.bb_flags_clean = false;
    HIPCHK(c, bvh_build_begin(c->stream, in.pos, n, s.bb_scratch, L, flags_clean, stamp, stamp_prev_end));
    if (lv_end > 0) HIPCHK(c, bvh_build_levels(c->stream, n, leaf, 0, lv_end, s.bb_scratch, L));
    int* walk_info = (int*)(s.ws_scratch + WL.info);
    GatherArgs<T> g{};
    g.perm = bvh_build_order(s.bb_scratch, L);  // read where the build left it, and copied out on the way
    g.perm_copy = s.order_dev;
    g.zero8 = walk_info;  // the estimate check's counters (launch_tree_walk_tile_prep below)
    g.n = n;
    g.pos_in = in.pos; g.pos_out = out.pos;
    g.weight_in = in.weight;
    g.mass_out = out.mass;
    g.vel_in = in.vel; g.vel_out = out.vel;
    g.weight_out = out.weight;
    g.ids_in = in.ids; g.ids_out = out.ids;
    // (the numbering's launch gathers the rows too: bvh_emit_gather)
    HIPCHK(c, bvh_build_finish(c->stream, in.weight, n, leaf, 0, s.bb_scratch, L, nullptr, s.geom0, s.geom1, s.link, s.node_depth,
                               s.node_mass, s.node_size, &g));
    if (!stamps) rc = phase_mark(c, ph, 1);
    if (rc) return rc;
    // ---- walk (rows as after the build: `out` is the permuted set, `in` the snapshot)
    WalkArgs<T> w{};
    w.geom0 = s.geom0; w.geom1 = s.geom1; w.link = s.link;
    w.n_nodes = 0;
    w.n_nodes_dev = c->spec_dev;
    w.big_leaves = 1;
    w.fast = c->params.arith == NBODY_ARITH_FAST;
    w.theta = (T)c->params.theta;
    w.clamp = (T)c->params.clamp;
    w.acc = s.acc;
    w.leaf_pos = out.pos;
    w.leaf_mass = out.mass;
    const bool as_written = c->params.order == NBODY_ORDER_AS_WRITTEN;
    w.tgt_pos = as_written ? in.pos : out.pos;
    w.n_tgt = n;
    const uint32_t* tgt_ids = as_written ? in.ids : out.ids;
    int shift = 0;
    while ((s.wt_total >> shift) >= (1ull << 31)) ++shift;
    if (lab_int("NBODY_WALK_TILE_POISON", 0) != 0)  // test hook: a history whose scan wraps must be noticed
      HIPCHK(c, hipMemsetAsync(s.wt_hist, 0xFF, (size_t)s.n * 4, c->stream));
    int64_t waves = 0;
    // The estimate check's last work-group concludes on the build (the verdict the walk kernel reads), packs verdict, build
    // flags and walk info for one copy to the host and clears the build's counters for the next step.
    TileTail tail;
    tail.flags = (const int*)(s.bb_scratch + L.flags);
    tail.flag_words = 128;
    tail.bigcount = (const int*)(s.bb_scratch + L.bigcount);
    tail.level_end = lv_end;
    tail.node_cap = L.node_cap;
    tail.verdict = c->spec_dev;
    tail.pack = c->spec_host_dev;  // (straight into the host's pinned record: no copy on the stream)
    tail.clear = (int*)(s.bb_scratch + L.flags);
    tail.clear_words = (int)((L.zero_end - L.flags) / sizeof(int));
    tail.info_zeroed = true;
    // (one kernel instead of three: 17 us against 32 at 151 405 targets, 31 against 67 at a million; NBODY_WALK_FUSED_SCAN_MAX=0: the three)
    tail.fused_scan = fused_scan;
    tail.stamp = stamps ? stamp + 1 : nullptr;
    HIPCHK(c, launch_tree_walk_tile_prep<T>(c->stream, w, s.ws_scratch, WL, tgt_ids, s.wt_hist, 1, shift, &waves, &tail));
    s.bb_flags_clean = true;
    int* h = c->spec_host;
    HIPCHK(c, hipEventRecord(c->spec_event, c->stream));  // (the tail kernel has written h[0 .. 2 + 128 + 8) by then)
    {
      TimerScope ts(c->timer, c->stream);
      HIPCHK(c, launch_tree_walk_tile_main<T>(c->stream, w, s.ws_scratch, WL, tgt_ids, s.wt_hist, waves));
    }
    if (!stamps) rc = phase_mark(c, ph, 2);
    if (rc) return rc;
    // ---- the step's one wait: for the event in front of the walk kernel
    HIPCHK(c, hipEventSynchronize(c->spec_event));
    rc = step_ahead_collect<T>(c, s);  // (the previous step's copies are older than this event)
    if (rc) return rc;
    const int* flags = h + 2;
    const int* bigcount = flags + (L.bigcount - L.flags) / sizeof(int);
    const int* info = h + 2 + 128;
    if (env_int("NBODY_TRACE", 0) != 0) {
      std::fprintf(stderr, "[nbody] step ahead: build verdict %d (%d nodes, depth %d, fallback %d, %d blind levels)\n", h[1], flags[kBvhNodes],
                   flags[kBvhMaxDepth], flags[kBvhFallback], lv_end);
      std::fprintf(stderr, "[nbody] tile walk (step ahead): estimate from the last walk (shift %d, total %d), %d per wave, overflow %d\n", shift,
                   info[0], info[3], info[1]);
    }
    if (h[1] == 0) {  // the build needs more levels or the host builder: nothing was integrated, `in` is intact
      s.wt_hist_n = -1;  // (the walk returned at once and left zeros in the history)
      s.bvh_levels_hint = 0;
      s.bvh_levels_stable = 0;
      if (!stamps) (void)phase_mark(c, ph, 3);  // (a stamped slot is simply not booked: the plain sequence that follows books the step)
      return 1;
    }
    // the build stands: what bvh_build_device records
    int used = 0;
    while (used < kBvhLevels - 1 && bigcount[used] != 0) ++used;
    s.bvh_levels_stable = used == s.bvh_levels_hint ? s.bvh_levels_stable + 1 : 0;
    s.bvh_levels_hint = used;
    s.cur = 1 - s.cur;
    ++s.row_epoch;
    s.h_weight_stale = true;
    s.n_nodes = flags[kBvhNodes];
    s.tree_kind = NBODY_TREE_BVH;
    s.tree_max_depth = flags[kBvhMaxDepth];
    s.tree_host_stale = true;
    s.tree_valid = true;
    c->bvh_stops = flags[kBvhStops];
    c->last_build_device = true;
    if (info[1] != 0) {  // the estimate's scan wrapped (the walk kernel returned at once): walk again the plain way
      rc = tree_walk_phase<T>(c, s, NBODY_TREE_BVH, nullptr, 0, s.acc);
      if (rc) return rc;
    } else {
      s.ahead_total_due = true;
    }
    Gate carry;  // the walk's info after the walk (its exact term count), read one step late: the integration's first threads take it along
    carry.carry_src = walk_info;
    carry.carry_dst = c->spec_host_dev + 2 + 128 + 8;
    carry.carry_words = 8;
    carry.stamp = stamps ? stamp + 2 : nullptr;
    HIPCHK(c, launch_integrate<T>(c->stream, s.set[s.cur].pos, s.set[s.cur].vel, s.acc, s.n, delta, carry));
    if (stamps) {
      c->stamp_open = slot;  // its end: the next stamped step's first kernel, or close_open_stamp
      c->stamp_pending.push_back(slot);
      return NBODY_OK;
    }
    rc = phase_mark(c, ph, 3);
    if (!rc) *chain = ph;  // the next step starts where this one ends
    return rc;
  }
}

// `async`: return once everything is enqueued (for a step ahead: once the host's one decision per step is made) instead of
// waiting for the last step; nbody_wait (or any call that reads the rows) completes it.
template <class T> int update_tree(nbody_ctx* c, int kind, T delta, int n_steps, nbody_counting* counter, bool async = false) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "update_tree: no particles of this precision uploaded");
  if (n_steps < 0) return fail(c, NBODY_ERR_INVALID, "update_tree: n_steps < 0");
  HIPCHK(c, hipSetDevice(c->device));
  State<T>& s = state_of<T>(c);
  c->ph_counter = counter;
  auto done = [&](int rc) {
    int rc2 = phase_drain(c);  // (waits for the last step: the call is synchronous)
    if (!rc2 && s.ahead_total_due) {
      hipError_t e = hipStreamSynchronize(c->stream);
      rc2 = e == hipSuccess ? step_ahead_collect<T>(c, s) : fail_hip(c, e, "hipStreamSynchronize");
    }
    c->ph_counter = nullptr;
    return rc ? rc : rc2;
  };
  PhaseEvents chain;  // the step before, when the next one follows it directly on the stream
  for (int step = 0; step < n_steps; ++step) {
    if (c->ph_pending.size() >= 64 || c->stamp_pending.size() >= 64) {
      int rc = phase_drain(c);
      if (rc) return done(rc);
      chain = PhaseEvents{};
    }
    if (kind == NBODY_TREE_BVH) {
      int rc = bvh_step_ahead<T>(c, s, delta, &chain);
      if (rc < 0) return done(rc);
      if (rc == NBODY_OK) {
        ++c->steps_done;
        continue;
      }
    }
    // the plain sequence: no host wait between the phases but those a phase needs for itself
    chain = PhaseEvents{};
    PhaseEvents ph;
    int rc = phase_begin(c, &ph);
    if (rc) return done(rc);
    rc = tree_build_phase<T>(c, s, kind);
    if (rc) return done(rc);
    rc = phase_mark(c, ph, 1);
    if (rc) return done(rc);
    rc = tree_walk_phase<T>(c, s, kind, nullptr, 0, s.acc);
    if (rc) return done(rc);
    rc = phase_mark(c, ph, 2);
    if (rc) return done(rc);
    hipError_t e = launch_integrate<T>(c->stream, s.set[s.cur].pos, s.set[s.cur].vel, s.acc, s.n, delta);
    if (e != hipSuccess) return done(fail_hip(c, e, "launch_integrate"));
    rc = phase_mark(c, ph, 3);
    if (rc) return done(rc);
    ++c->steps_done;
  }
  if (async) {  // the phase events and the last walk's term count are collected by nbody_wait or the next synchronous call
    // (a stamped step's end is written now: whatever the caller does before its next step is not this step's integration)
    if (int rc = close_open_stamp(c)) return rc;
    c->ph_counter = nullptr;
    return NBODY_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return done(NBODY_OK);
}

// One tree step of a rank that owns the slice [begin, begin+count) of the tree-ordered targets: the tree is built over
// ALL particles (every rank holds them all and builds the same tree), the walk and the integration touch only the slice.
template <class T> int update_tree_shard(nbody_ctx* c, int kind, T delta, int64_t begin, int64_t count, nbody_counting* counter) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "update_tree_shard: no particles of this precision uploaded");
  State<T>& s = state_of<T>(c);
  if (begin < 0 || count < 0 || begin + count > s.n) return fail(c, NBODY_ERR_INVALID, "update_tree_shard: slice out of range");
  HIPCHK(c, hipSetDevice(c->device));
  c->ph_counter = counter;
  auto done = [&](int rc) {
    int rc2 = phase_drain(c);
    c->ph_counter = nullptr;
    return rc ? rc : rc2;
  };
  PhaseEvents ph;
  int rc = phase_begin(c, &ph);
  if (rc) return done(rc);
  rc = tree_build_phase<T>(c, s, kind);
  if (rc) return done(rc);
  rc = phase_mark(c, ph, 1);
  if (rc) return done(rc);
  rc = tree_walk_phase<T>(c, s, kind, nullptr, 0, s.acc, begin, count);
  if (rc) return done(rc);
  rc = phase_mark(c, ph, 2);
  if (rc) return done(rc);
  const uint32_t* rows = kind == NBODY_TREE_QUAD ? s.order_dev + begin : nullptr;
  hipError_t e = launch_integrate_rows<T>(c->stream, s.set[s.cur].pos, s.set[s.cur].vel, s.acc, rows, begin, count, delta);
  if (e != hipSuccess) return done(fail_hip(c, e, "launch_integrate_rows"));
  rc = phase_mark(c, ph, 3);
  if (rc) return done(rc);
  s.shard_kind = kind;
  ++c->steps_done;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return done(NBODY_OK);
}
template <class T>
int export_slice(nbody_ctx* c, int64_t begin, int64_t count, void* rows_dev, void* pos_dev, void* vel_dev) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "export_slice: no particles of this precision uploaded");
  State<T>& s = state_of<T>(c);
  if (!s.tree_valid) return fail(c, NBODY_ERR_INVALID, "export_slice: no tree step yet");
  if (begin < 0 || count < 0 || begin + count > s.n || !rows_dev || !pos_dev || !vel_dev)
    return fail(c, NBODY_ERR_INVALID, "export_slice: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t* rows = s.shard_kind == NBODY_TREE_QUAD ? s.order_dev + begin : nullptr;
  HIPCHK(c, launch_export_rows<T>(c->stream, s.set[s.cur].pos, s.set[s.cur].vel, rows, begin, count, (uint32_t*)rows_dev, pos_dev, vel_dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NBODY_OK;
}
template <class T> int import_rows_api(nbody_ctx* c, int64_t n_rows, const void* rows_dev, const void* pos_dev, const void* vel_dev) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "import_rows: no particles of this precision uploaded");
  State<T>& s = state_of<T>(c);
  if (n_rows < 0 || (n_rows > 0 && (!rows_dev || !pos_dev || !vel_dev))) return fail(c, NBODY_ERR_INVALID, "import_rows: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_import_rows<T>(c->stream, s.set[s.cur].pos, s.set[s.cur].vel, (const uint32_t*)rows_dev, n_rows, s.n, pos_dev, vel_dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  s.tree_valid = false;
  return NBODY_OK;
}

template <class T> int accel_built_tree(nbody_ctx* c, State<T>& s, int kind, int64_t n_targets, const T* target_xy, T* acc_xy);
template <class T> int accel_tree(nbody_ctx* c, int kind, int64_t n_targets, const T* target_xy, T* acc_xy) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "accel_tree: no particles of this precision uploaded");
  if (!acc_xy) return fail(c, NBODY_ERR_INVALID, "accel_tree: acc_xy is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  State<T>& s = state_of<T>(c);
  int rc = tree_build_phase<T>(c, s, kind);
  if (rc) return rc;
  return accel_built_tree<T>(c, s, kind, n_targets, target_xy, acc_xy);
}
// ... the walk alone, over the tree that is installed (the library's build or a caller's tree)
template <class T> int accel_built_tree(nbody_ctx* c, State<T>& s, int kind, int64_t n_targets, const T* target_xy, T* acc_xy) {
  using T2 = typename State<T>::T2;
  int rc = NBODY_OK;
  if (!target_xy) {
    // particles themselves, post-build row order, regardless of params.order
    const void* tp = s.set[s.cur].pos;
    int saved = c->params.order;
    c->params.order = NBODY_ORDER_CONSISTENT;
    rc = tree_walk_phase<T>(c, s, kind, kind == NBODY_TREE_BVH ? tp : nullptr, s.n, s.acc);
    c->params.order = saved;
    if (rc) return rc;
    if (s.n) HIPCHK(c, hipMemcpyAsync(acc_xy, s.acc, (size_t)s.n * sizeof(T2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NBODY_OK;
  }
  if (n_targets < 0) return fail(c, NBODY_ERR_INVALID, "accel_tree: n_targets < 0");
  if (n_targets == 0) return NBODY_OK;
  T2 *tp = nullptr, *ta = nullptr;
  HIPCHK(c, hipMalloc((void**)&tp, (size_t)n_targets * sizeof(T2)));
  hipError_t e = hipMalloc((void**)&ta, (size_t)n_targets * sizeof(T2));
  if (e != hipSuccess) { (void)hipFree(tp); return fail_hip(c, e, "hipMalloc"); }
  e = hipMemcpyAsync(tp, target_xy, (size_t)n_targets * sizeof(T2), hipMemcpyHostToDevice, c->stream);
  rc = (e == hipSuccess) ? tree_walk_phase<T>(c, s, kind, tp, n_targets, ta) : fail_hip(c, e, "hipMemcpyAsync");
  if (!rc) {
    e = hipMemcpyAsync(acc_xy, ta, (size_t)n_targets * sizeof(T2), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail_hip(c, e, "download acc");
  }
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(tp);
  (void)hipFree(ta);
  return rc;
}

// ---- a caller's tree (the force map alone, main.rs:406-416, for a host that keeps bvh_tree.rs:56-158) --------
// What the walks rely on and a foreign tree has to prove before it reaches the device: every skip link points forward (the
// walk's node index only ever grows: it ends), subtrees nest, an inner node's children are i + 1, skip[i + 1], ... and its range is
// their ranges one after the other (the walks take the range of an inner node whose children are two leaves in one step),
// leaves are single nodes, every range lies inside the particles, `order` is a permutation.  BVH: two children (BVHTree::Root,
// bvh_tree.rs:28); quad: one to four (quad_tree.rs:47-50).  The geometry and the masses are only ever operands.
bool tree_shape_ok(int kind, int64_t m, const int32_t* is_leaf, const int64_t* first, const int64_t* count, const int64_t* skip,
                   int64_t n, const uint32_t* order, int* max_depth, std::string& why) {
  auto bad = [&](int64_t i, const char* what) {
    why = "node " + std::to_string(i) + ": " + what;
    return false;
  };
  if (kind != NBODY_TREE_BVH && kind != NBODY_TREE_QUAD) { why = "unknown tree kind"; return false; }
  if (m < 1 || m > (int64_t)INT32_MAX - 1) { why = "n_nodes out of range"; return false; }
  if (n < 0 || n > (int64_t)INT32_MAX - 64) { why = "particle count out of range"; return false; }
  if (!is_leaf || !first || !count || !skip || (n > 0 && !order)) { why = "a tree array is NULL"; return false; }
  const int max_kids = kind == NBODY_TREE_BVH ? 2 : 4, min_kids = kind == NBODY_TREE_BVH ? 2 : 1;
  struct Open { int64_t id, end, cursor; int kids; };
  std::vector<Open> open;
  int deepest = 0;
  auto close = [&](const Open& o) {
    if (o.cursor != first[o.id] + count[o.id]) return bad(o.id, "its range is not its children's ranges one after the other");
    if (o.kids < min_kids || o.kids > max_kids) return bad(o.id, kind == NBODY_TREE_BVH ? "a BVH root has two children" : "a quad root has one to four children");
    return true;
  };
  for (int64_t i = 0; i < m; ++i) {
    while (!open.empty() && open.back().end == i) {
      if (!close(open.back())) return false;
      open.pop_back();
    }
    if (i > 0 && open.empty()) return bad(i, "lies outside the root's subtree (skip[0] must be n_nodes)");
    if (skip[i] <= i || skip[i] > m) return bad(i, "skip does not point forward inside the tree");
    if (first[i] < 0 || count[i] < 0 || first[i] > n || count[i] > n - first[i]) return bad(i, "range outside the particles");
    if (!open.empty()) {
      Open& parent = open.back();
      if (skip[i] > parent.end) return bad(i, "subtree reaches past its parent's");
      if (first[i] != parent.cursor) return bad(i, "range does not follow its sibling's");
      parent.cursor += count[i];
      ++parent.kids;
    }
    if (is_leaf[i]) {
      if (skip[i] != i + 1) return bad(i, "a leaf with nodes below it");
    } else {
      if (skip[i] == i + 1) return bad(i, "a root without children");
      open.push_back({i, skip[i], first[i], 0});
      if ((int)open.size() > deepest) deepest = (int)open.size();
    }
  }
  while (!open.empty()) {
    if (open.back().end != m) return bad(open.back().id, "subtree ends past the last node");
    if (!close(open.back())) return false;
    open.pop_back();
  }
  if (skip[0] != m) return bad(0, "skip[0] must be n_nodes");
  if (first[0] != 0 || count[0] != n) return bad(0, "the root's range must be every particle");
  std::vector<bool> seen((size_t)n, false);
  for (int64_t k = 0; k < n; ++k) {
    if (order[k] >= (uint64_t)n || seen[order[k]]) { why = "order is not a permutation of the rows"; return false; }
    seen[order[k]] = true;
  }
  if (max_depth) *max_depth = deepest;
  return true;
}

template <class T>
int walk_tree(nbody_ctx* c, int kind, int64_t m, const T* geom, const uint32_t* mass, const int32_t* is_leaf, const int64_t* first,
              const int64_t* count, const int64_t* skip, const uint32_t* order, int64_t n_targets, const T* target_xy, T* acc_xy) {
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "walk_tree: no particles of this precision uploaded");
  if (!geom || !mass) return fail(c, NBODY_ERR_INVALID, "walk_tree: geom or mass is NULL");
  if (!acc_xy) return fail(c, NBODY_ERR_INVALID, "walk_tree: acc_xy is NULL");
  if (target_xy && n_targets < 0) return fail(c, NBODY_ERR_INVALID, "walk_tree: n_targets < 0");
  State<T>& s = state_of<T>(c);
  std::string why;
  int depth = 0;
  if (!tree_shape_ok(kind, m, is_leaf, first, count, skip, s.n, order, &depth, why)) return fail(c, NBODY_ERR_INVALID, "walk_tree: " + why);
  HIPCHK(c, hipSetDevice(c->device));
  int rc = refresh_host_weights<T>(c, s);
  if (rc) return rc;
  s.tree_valid = false;
  s.tree_host_stale = false;
  c->last_build_device = false;
  c->bvh_stops = 0;
  TreeHost<T>& t = s.tree;
  t.clear();
  t.kind = kind;
  t.max_depth = depth;
  t.geom0.resize((size_t)m); t.geom1.resize((size_t)m); t.link.resize((size_t)m);
  t.size_x.resize((size_t)m); t.size_y.resize((size_t)m); t.mass_u32.resize((size_t)m);
  for (int64_t i = 0; i < m; ++i) {
    const size_t k = (size_t)i;
    if (kind == NBODY_TREE_BVH) {
      const T* g = geom + 6 * k;
      const T w = g[2], h = g[3];
      const T tx = sse_max(w, h), ty = sse_max(h, w);  // size.max(size.yx()), main.rs:371 (as build_bvh and bvh_emit)
      t.geom0[k] = {g[0], g[1], g[0] + w, g[1] + h};
      t.geom1[k] = {g[4], g[5], (T)mass[k], tx * ty};
      t.size_x[k] = w; t.size_y[k] = h;
    } else {
      const T* g = geom + 5 * k;
      const T h = g[2];
      t.geom0[k] = {g[0], g[1], g[0] + h, g[1] + h};
      t.geom1[k] = {g[3], g[4], (T)mass[k], h * h};
      t.size_x[k] = h; t.size_y[k] = h;
    }
    t.mass_u32[k] = mass[k];
    t.link[k] = {(int32_t)skip[k], (int32_t)first[k], (int32_t)count[k], is_leaf[k] ? 1 : 0};
  }
  t.order.assign(order, order + s.n);
  rc = install_host_tree<T>(c, s, kind);
  if (rc) return rc;
  return accel_built_tree<T>(c, s, kind, n_targets, target_xy, acc_xy);
}

template <class T>
void tree_export_host(const TreeHost<T>& t, T* geom, uint32_t* mass, int32_t* is_leaf, int64_t* first, int64_t* count,
                      int64_t* skip, uint32_t* order) {
  const size_t m = t.size();
  for (size_t i = 0; i < m; ++i) {
    if (geom) {
      if (t.kind == NBODY_TREE_BVH) {
        T* g = geom + 6 * i;
        g[0] = t.geom0[i].a; g[1] = t.geom0[i].b; g[2] = t.size_x[i]; g[3] = t.size_y[i];
        g[4] = t.geom1[i].a; g[5] = t.geom1[i].b;
      } else {
        T* g = geom + 5 * i;
        g[0] = t.geom0[i].a; g[1] = t.geom0[i].b; g[2] = t.size_x[i]; g[3] = t.geom1[i].a; g[4] = t.geom1[i].b;
      }
    }
    if (mass) mass[i] = t.mass_u32[i];
    if (is_leaf) is_leaf[i] = t.link[i].is_leaf;
    if (first) first[i] = t.link[i].first;
    if (count) count[i] = t.link[i].count;
    if (skip) skip[i] = t.link[i].skip;
  }
  if (order && !t.order.empty()) std::memcpy(order, t.order.data(), t.order.size() * sizeof(uint32_t));
}

template <class T>
int tree_export(const nbody_ctx* cc, T* geom, uint32_t* mass, int32_t* is_leaf, int64_t* first, int64_t* count,
                int64_t* skip, uint32_t* order) {
  nbody_ctx* c = const_cast<nbody_ctx*>(cc);
  if (!c) return NBODY_ERR_INVALID;
  if (!has_state<T>(c)) return fail(c, NBODY_ERR_INVALID, "tree_export: no particles of this precision uploaded");
  State<T>& s = state_of<T>(c);
  if (!s.tree_valid) return fail(c, NBODY_ERR_INVALID, "tree_export: no tree built yet");
  int rc = download_tree<T>(c, s);
  if (rc) return rc;
  tree_export_host<T>(s.tree, geom, mass, is_leaf, first, count, skip, order);
  return NBODY_OK;
}

}  // namespace

// ================================================================================================ C ABI
#define NB_API extern "C" __attribute__((visibility("default")))

NB_API int nbody_abi_version(void) { return NBODY_ABI_VERSION; }

NB_API int nbody_default_params(nbody_params* p) {
  if (!p) return NBODY_ERR_INVALID;
  p->theta = 50.0f;          // main.rs:35
  p->clamp = 0.001f;         // main.rs:247-248
  p->leaf_size = 64;         // bvh_tree.rs:37
  p->order = NBODY_ORDER_AS_WRITTEN;
  p->arith = NBODY_ARITH_AUTO;
  p->quad_root_x = 0.0f;
  p->quad_root_y = 0.0f;
  p->quad_root_h = 100000.0f;  // HEIGHT, main.rs:31
  return NBODY_OK;
}

NB_API int nbody_create(nbody_ctx** out, int device_id) { return nbody::ctx_create_single(out, device_id); }

int nbody::ctx_create_single(nbody_ctx** out, int device_id) {
  if (!out) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, NBODY_ERR_NO_DEVICE,
                std::string("nbody_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                    "); this library has no CPU path");
  if (device_id < 0 || device_id >= count) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: device_id out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceProperties");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, NBODY_ERR_NO_DEVICE, std::string("nbody_create: device is ") + prop.gcnArchName +
                                                  ", kernels are built for gfx950 (MI355X) only");
  nbody_ctx* c = new (std::nothrow) nbody_ctx();
  if (!c) return fail(nullptr, NBODY_ERR_NOMEM, "nbody_create: out of host memory");
  c->device = device_id;
  nbody_default_params(&c->params);
  e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->snap_event, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&c->stats_dev, 3 * sizeof(unsigned long long));
  if (e != hipSuccess) {
    int rc = fail_hip(nullptr, e, "nbody_create");
    delete c;
    return rc;
  }
  *out = c;
  return NBODY_OK;
}

static void free_snapshot(nbody_ctx* c);
static void free_delta(nbody_ctx* c);
NB_API void nbody_destroy(nbody_ctx* c) {
  if (!c) return;
  if (c->multi) { nbody::multi_destroy(c); return; }
  nbody::ctx_destroy_single(c);
}
void nbody::ctx_destroy_single(nbody_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->direct_graph.reset();
  free_state(c->sf);
  free_state(c->sd);
  free_dev(c->workspace);
  free_dev(c->stats_dev);
  free_dev(c->frame_work);
  free_dev(c->frame_rgba);
  free_dev(c->spec_dev);
  free_dev(c->stamp_dev);
  if (c->spec_host) (void)hipHostFree(c->spec_host);
  if (c->spec_event) (void)hipEventDestroy(c->spec_event);
  for (auto& p : c->ph_pending)
    for (int k = p.borrowed ? 1 : 0; k < 4; ++k)
      if (p.e[k]) (void)hipEventDestroy(p.e[k]);
  for (auto e : c->ph_free) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  free_snapshot(c);
  free_delta(c);
  if (c->snap_event) (void)hipEventDestroy(c->snap_event);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

NB_API const char* nbody_last_error(const nbody_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

NB_API int nbody_set_params(nbody_ctx* c, const nbody_params* p) {
  if (!c || !p) return NBODY_ERR_INVALID;
  if (p->leaf_size < 1) return fail(c, NBODY_ERR_INVALID, "set_params: leaf_size < 1");
  if (p->order != NBODY_ORDER_AS_WRITTEN && p->order != NBODY_ORDER_CONSISTENT) return fail(c, NBODY_ERR_INVALID, "set_params: bad order");
  if (p->arith < NBODY_ARITH_AUTO || p->arith > NBODY_ARITH_EXACT) return fail(c, NBODY_ERR_INVALID, "set_params: bad arith");
  c->params = *p;
  if (c->multi) return nbody::multi_set_params(c);
  return NBODY_OK;
}
NB_API int nbody_get_params(const nbody_ctx* c, nbody_params* out) {
  if (!c || !out) return NBODY_ERR_INVALID;
  *out = c->params;
  return NBODY_OK;
}

// A handle made by nbody_create_multi fronts several devices.  Calls that only read or that act on "the current rows"
// are served by the first device once the replicas agree (multi_primary brings them up to date); `mutates` marks the
// calls after which the other replicas must be refreshed from it (multi_replicate).
#define NB_VIA_PRIMARY(c, mutates, expr)                    \
  do {                                                      \
    if ((c) && (c)->multi) {                                \
      nbody_ctx* front__ = (c);                             \
      nbody_ctx* p = nullptr;                               \
      int rc__ = nbody::multi_primary(front__, &p);         \
      if (rc__) return rc__;                                \
      rc__ = (expr);                                        \
      if (rc__) { front__->err = p->err; return rc__; }     \
      return (mutates) ? nbody::multi_replicate(front__) : NBODY_OK; \
    }                                                       \
  } while (0)

NB_API int nbody_upload_f32(nbody_ctx* c, int64_t n, const float* pos, const float* vel, const uint32_t* w) {
  if (c && c->multi) return nbody::multi_upload(c, false, n, pos, vel, w);
  return upload<float>(c, n, pos, vel, w);
}
NB_API int nbody_upload_f64(nbody_ctx* c, int64_t n, const double* pos, const double* vel, const uint32_t* w) {
  if (c && c->multi) return nbody::multi_upload(c, true, n, pos, vel, w);
  return upload<double>(c, n, pos, vel, w);
}
NB_API int nbody_download_f32(nbody_ctx* c, float* pos, float* vel, uint32_t* w, uint32_t* ids) {
  NB_VIA_PRIMARY(c, false, download<float>(p, pos, vel, w, ids));
  return download<float>(c, pos, vel, w, ids);
}
NB_API int nbody_download_f64(nbody_ctx* c, double* pos, double* vel, uint32_t* w, uint32_t* ids) {
  NB_VIA_PRIMARY(c, false, download<double>(p, pos, vel, w, ids));
  return download<double>(c, pos, vel, w, ids);
}
// ---- snapshot hand-off (main.rs:136-139) --------------------------------------------------------------------------
static void free_snapshot(nbody_ctx* c) {
  free_dev(c->snap_pos); free_dev(c->snap_vel); free_dev(c->snap_w); free_dev(c->snap_ids);
  if (c->snap_hpos) (void)hipHostFree(c->snap_hpos);
  if (c->snap_hvel) (void)hipHostFree(c->snap_hvel);
  if (c->snap_hw) (void)hipHostFree(c->snap_hw);
  if (c->snap_hids) (void)hipHostFree(c->snap_hids);
  c->snap_hpos = c->snap_hvel = nullptr;
  c->snap_hw = c->snap_hids = nullptr;
  c->snap_bytes2 = 0;
  c->snap_n = 0;
  c->snap_pending = false;
}
template <class T> int snapshot_begin(nbody_ctx* c, State<T>& s) {
  using T2 = typename State<T>::T2;
  const size_t n = (size_t)s.n, b2 = n * sizeof(T2);
  if (c->snap_n != s.n || c->snap_bytes2 != b2) {
    free_snapshot(c);
    if (n) {
      HIPCHK(c, hipMalloc(&c->snap_pos, b2));
      HIPCHK(c, hipMalloc(&c->snap_vel, b2));
      HIPCHK(c, hipMalloc((void**)&c->snap_w, n * 4));
      HIPCHK(c, hipMalloc((void**)&c->snap_ids, n * 4));
      HIPCHK(c, hipHostMalloc(&c->snap_hpos, b2, hipHostMallocDefault));
      HIPCHK(c, hipHostMalloc(&c->snap_hvel, b2, hipHostMallocDefault));
      HIPCHK(c, hipHostMalloc((void**)&c->snap_hw, n * 4, hipHostMallocDefault));
      HIPCHK(c, hipHostMalloc((void**)&c->snap_hids, n * 4, hipHostMallocDefault));
    }
    c->snap_n = s.n;
    c->snap_bytes2 = b2;
  }
  c->snap_f64 = sizeof(T) == 8;
  auto& st = s.set[s.cur];
  if (n) {
    // the rows are copied aside on the stream the steps run on (ordered after the last step, microseconds), so that
    // later steps may overwrite them; the slow leg to the host runs on its own stream, alongside those steps
    HIPCHK(c, hipMemcpyAsync(c->snap_pos, st.pos, b2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_vel, st.vel, b2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_w, st.weight, n * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_ids, st.ids, n * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->snap_event, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->snap_event, 0));
    HIPCHK(c, hipMemcpyAsync(c->snap_hpos, c->snap_pos, b2, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_hvel, c->snap_vel, b2, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_hw, c->snap_w, n * 4, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipMemcpyAsync(c->snap_hids, c->snap_ids, n * 4, hipMemcpyDeviceToHost, c->copy_stream));
  }
  c->snap_step = c->steps_done;
  c->snap_pending = true;
  return NBODY_OK;
}
NB_API int nbody_snapshot_begin(nbody_ctx* c) {
  if (!c) return NBODY_ERR_INVALID;
  NB_VIA_PRIMARY(c, false, nbody_snapshot_begin(p));
  if (!c->has_f32 && !c->has_f64) return fail(c, NBODY_ERR_INVALID, "snapshot_begin: no particles uploaded");
  if (c->snap_pending) return fail(c, NBODY_ERR_INVALID, "snapshot_begin: a snapshot is still pending (take it with nbody_snapshot_end)");
  HIPCHK(c, hipSetDevice(c->device));
  return c->has_f32 ? snapshot_begin<float>(c, c->sf) : snapshot_begin<double>(c, c->sd);
}
NB_API int nbody_snapshot_pending(const nbody_ctx* c) {
  if (c && c->multi) return nbody_snapshot_pending(nbody::multi_peek(c));
  return c && c->snap_pending ? 1 : 0;
}
static int snapshot_end(nbody_ctx* c, bool f64, void* pos, void* vel, uint32_t* w, uint32_t* ids, uint64_t* step) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) {  // the pending snapshot lives on the first device; taking it does not need the replicas to agree
    nbody_ctx* p = nbody::multi_peek(c);
    int rc = snapshot_end(p, f64, pos, vel, w, ids, step);
    if (rc) c->err = p->err;
    return rc;
  }
  if (!c->snap_pending) return fail(c, NBODY_ERR_INVALID, "snapshot_end: no snapshot pending");
  if (c->snap_f64 != f64) return fail(c, NBODY_ERR_INVALID, "snapshot_end: the pending snapshot has the other precision");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  const size_t n = (size_t)c->snap_n;
  if (n) {
    if (pos) std::memcpy(pos, c->snap_hpos, c->snap_bytes2);
    if (vel) std::memcpy(vel, c->snap_hvel, c->snap_bytes2);
    if (w) std::memcpy(w, c->snap_hw, n * 4);
    if (ids) std::memcpy(ids, c->snap_hids, n * 4);
  }
  if (step) *step = c->snap_step;
  c->snap_pending = false;
  return NBODY_OK;
}
NB_API int nbody_snapshot_end_f32(nbody_ctx* c, float* pos, float* vel, uint32_t* w, uint32_t* ids, uint64_t* step) {
  return snapshot_end(c, false, pos, vel, w, ids, step);
}
NB_API int nbody_snapshot_end_f64(nbody_ctx* c, double* pos, double* vel, uint32_t* w, uint32_t* ids, uint64_t* step) {
  return snapshot_end(c, true, pos, vel, w, ids, step);
}

// ---- delta snapshots (the commented experiment of main.rs:107-134; format: delta_codec.h) -------------------------
static void free_delta(nbody_ctx* c) {
  for (auto& k : c->dl_keys) free_dev(k);
  free_dev(c->dl_widths); free_dev(c->dl_words); free_dev(c->dl_offsets); free_dev(c->dl_scan); free_dev(c->dl_payload);
  free_dev(c->dl_total);
  if (c->dl_host) (void)hipHostFree(c->dl_host);
  if (c->dl_htotal) (void)hipHostFree(c->dl_htotal);
  c->dl_host = nullptr;
  c->dl_htotal = nullptr;
  c->dl_n = -1;
  c->dl_bits = 0;
  c->dl_key_next = true;
  c->dl_pending = false;
}
template <class T> int delta_begin(nbody_ctx* c, State<T>& s) {
  const int bits = (int)sizeof(T) * 8;
  const int64_t n = s.n;
  const size_t nblk = delta_blocks(n), npad = nblk * 64, kb = 2 * npad * sizeof(T), wb = delta_width_bytes(n);
  if (c->dl_n != n || c->dl_bits != bits) {
    free_delta(c);
    for (auto& k : c->dl_keys) HIPCHK(c, hipMalloc(&k, kb ? kb : 8));
    HIPCHK(c, hipMalloc((void**)&c->dl_widths, wb ? wb : 8));
    HIPCHK(c, hipMalloc((void**)&c->dl_words, 2 * nblk * 4 + 8));
    HIPCHK(c, hipMalloc((void**)&c->dl_offsets, 2 * nblk * 4 + 8));
    c->dl_scan_bytes = delta_scan_temp_bytes(n);
    HIPCHK(c, hipMalloc(&c->dl_scan, c->dl_scan_bytes ? c->dl_scan_bytes : 8));
    HIPCHK(c, hipMalloc((void**)&c->dl_payload, 2 * nblk * (size_t)bits * 8 + 8));
    HIPCHK(c, hipMalloc((void**)&c->dl_total, 8));
    HIPCHK(c, hipHostMalloc((void**)&c->dl_host, delta_bound(n, bits), hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&c->dl_htotal, 8, hipHostMallocDefault));
    if (wb) HIPCHK(c, hipMemsetAsync(c->dl_widths, 0, wb, c->stream));  // the padding bytes stay zero
    c->dl_n = n;
    c->dl_bits = bits;
    c->dl_key_next = true;
  }
  const bool key = c->dl_key_next;
  if (key && kb)
    for (auto& k : c->dl_keys) HIPCHK(c, hipMemsetAsync(k, 0, kb, c->stream));
  void* cur = c->dl_keys[c->dl_cur];
  const void* prev = c->dl_keys[(c->dl_cur + 2) % 3];
  const void* prev2 = c->dl_keys[(c->dl_cur + 1) % 3];
  auto& st = s.set[s.cur];
  HIPCHK(c, launch_delta_encode<T>(c->stream, n, st.pos, st.ids, cur, prev, prev2, c->dl_widths, c->dl_words, c->dl_offsets,
                                   c->dl_scan, c->dl_scan_bytes, c->dl_payload, c->dl_total));
  HIPCHK(c, hipEventRecord(c->snap_event, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->snap_event, 0));
  // the size of the stream is known on the device only: fetch it, then start the transfer proper (which later steps overlap)
  HIPCHK(c, hipMemcpyAsync(c->dl_htotal, c->dl_total, 8, hipMemcpyDeviceToHost, c->copy_stream));
  HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  const uint64_t total = *c->dl_htotal;
  if (total > 2 * nblk * (uint64_t)bits) return fail(c, NBODY_ERR_HIP, "delta_begin: the encoder reported an impossible size");
  uint8_t* h = c->dl_host;
  std::memset(h, 0, kDeltaHeader);
  h[0] = 'N'; h[1] = 'B'; h[2] = 'D'; h[3] = '1';
  h[4] = (uint8_t)bits;
  h[5] = key ? 1 : 0;
  const uint64_t n64 = (uint64_t)n, step = c->steps_done;
  std::memcpy(h + 8, &n64, 8);
  std::memcpy(h + 16, &step, 8);
  std::memcpy(h + 24, &total, 8);
  if (wb) HIPCHK(c, hipMemcpyAsync(h + kDeltaHeader, c->dl_widths, wb, hipMemcpyDeviceToHost, c->copy_stream));
  if (total) HIPCHK(c, hipMemcpyAsync(h + kDeltaHeader + wb, c->dl_payload, total * 8, hipMemcpyDeviceToHost, c->copy_stream));
  c->dl_stream_bytes = kDeltaHeader + wb + (size_t)total * 8;
  c->dl_cur = (c->dl_cur + 1) % 3;  // the oldest keys are overwritten next time
  c->dl_key_next = false;
  c->dl_step = step;
  c->dl_pending = true;
  return NBODY_OK;
}
NB_API int nbody_delta_begin(nbody_ctx* c) {
  if (!c) return NBODY_ERR_INVALID;
  NB_VIA_PRIMARY(c, false, nbody_delta_begin(p));
  if (!c->has_f32 && !c->has_f64) return fail(c, NBODY_ERR_INVALID, "delta_begin: no particles uploaded");
  if (c->dl_pending) return fail(c, NBODY_ERR_INVALID, "delta_begin: a stream is still pending (take it with nbody_delta_end)");
  HIPCHK(c, hipSetDevice(c->device));
  return c->has_f32 ? delta_begin<float>(c, c->sf) : delta_begin<double>(c, c->sd);
}
NB_API int nbody_delta_pending(const nbody_ctx* c) {
  if (c && c->multi) return nbody_delta_pending(nbody::multi_peek(c));
  return c && c->dl_pending ? 1 : 0;
}
NB_API int nbody_delta_end(nbody_ctx* c, uint8_t* out, size_t cap, size_t* bytes_out, uint64_t* step_out) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) {
    nbody_ctx* p = nbody::multi_peek(c);
    int rc = nbody_delta_end(p, out, cap, bytes_out, step_out);
    if (rc) c->err = p->err;
    return rc;
  }
  if (!c->dl_pending) return fail(c, NBODY_ERR_INVALID, "delta_end: no stream pending");
  if (bytes_out) *bytes_out = c->dl_stream_bytes;
  if (step_out) *step_out = c->dl_step;
  if (!out || cap < c->dl_stream_bytes) return fail(c, NBODY_ERR_INVALID, "delta_end: the output buffer is smaller than the stream");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  std::memcpy(out, c->dl_host, c->dl_stream_bytes);
  c->dl_pending = false;
  return NBODY_OK;
}
NB_API int nbody_delta_reset(nbody_ctx* c) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) {
    nbody_ctx* p = nbody::multi_peek(c);
    int rc = nbody_delta_reset(p);
    if (rc) c->err = p->err;
    return rc;
  }
  if (c->dl_pending) return fail(c, NBODY_ERR_INVALID, "delta_reset: a stream is still pending");
  c->dl_key_next = true;
  return NBODY_OK;
}
NB_API size_t nbody_delta_bound(int64_t n, int is_f64) { return n < 0 ? 0 : delta_bound(n, is_f64 ? 64 : 32); }

// The receiving side: plain host code (delta_decoder.hpp; the consumer of a snapshot is a host thread, main.rs:147-150).
struct nbody_delta_decoder {
  DeltaDecoder d;
};
NB_API nbody_delta_decoder* nbody_delta_decoder_create(void) { return new (std::nothrow) nbody_delta_decoder(); }
NB_API void nbody_delta_decoder_destroy(nbody_delta_decoder* d) { delete d; }
NB_API const char* nbody_delta_decoder_error(const nbody_delta_decoder* d) { return d ? d->d.err.c_str() : "null decoder"; }
NB_API int64_t nbody_delta_decoder_count(const nbody_delta_decoder* d) { return d ? d->d.n : -1; }
NB_API int nbody_delta_decoder_is_f64(const nbody_delta_decoder* d) { return d && d->d.bits == 64 ? 1 : 0; }
NB_API uint64_t nbody_delta_decoder_step(const nbody_delta_decoder* d) { return d ? d->d.step : 0; }
NB_API int nbody_delta_decoder_apply(nbody_delta_decoder* d, const uint8_t* stream, size_t bytes) {
  if (!d) return NBODY_ERR_INVALID;
  try {  // nothing may unwind through the C ABI (the decoder itself already turns a failed allocation into a refusal)
    return d->d.apply(stream, bytes) ? NBODY_OK : NBODY_ERR_INVALID;
  } catch (...) {
    return NBODY_ERR_NOMEM;
  }
}
NB_API int nbody_delta_decoder_set_max_bodies(nbody_delta_decoder* d, int64_t max_bodies) {
  if (!d || max_bodies < 0) return NBODY_ERR_INVALID;
  d->d.max_bodies = (uint64_t)max_bodies;
  return NBODY_OK;
}
NB_API int nbody_delta_decoder_positions_f32(const nbody_delta_decoder* d, float* pos) {
  return d && d->d.positions<float, uint32_t>(pos) ? NBODY_OK : NBODY_ERR_INVALID;
}
NB_API int nbody_delta_decoder_positions_f64(const nbody_delta_decoder* d, double* pos) {
  return d && d->d.positions<double, uint64_t>(pos) ? NBODY_OK : NBODY_ERR_INVALID;
}

template <class T> int render_rows(nbody_ctx* c, State<T>& s, uint32_t height, uint32_t render_px, uint8_t* rgba_out) {
  auto& st = s.set[s.cur];
  HIPCHK(c, launch_render<T>(c->stream, s.n, st.pos, st.vel, st.weight, height, render_px, c->frame_work, c->frame_rgba));
  HIPCHK(c, hipMemcpyAsync(rgba_out, c->frame_rgba, (size_t)render_px * render_px * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NBODY_OK;
}
NB_API int nbody_render_rgba(nbody_ctx* c, uint32_t height, uint32_t render_px, uint8_t* rgba_out) {
  if (!c) return NBODY_ERR_INVALID;
  NB_VIA_PRIMARY(c, false, nbody_render_rgba(p, height, render_px, rgba_out));
  if (!rgba_out) return fail(c, NBODY_ERR_INVALID, "render: null output");
  if (!c->has_f32 && !c->has_f64) return fail(c, NBODY_ERR_INVALID, "render: no particles uploaded");
  // main.rs:51-52 divide by HEIGHT / RENDER_HEIGHT: a cell of 0 world units or a last cell past the frame is an
  // out-of-range index upstream (a panic): refuse instead
  if (render_px == 0 || render_px > 16384 || height == 0 || height % render_px != 0 || height > (1u << 24))
    return fail(c, NBODY_ERR_INVALID, "render: render_px must divide height (both > 0, height <= 2^24, render_px <= 16384)");
  const int64_t n = c->has_f32 ? c->sf.n : c->sd.n;
  if (n > (1 << 24)) return fail(c, NBODY_ERR_INVALID, "render: more than 2^24 rows");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->frame_px != render_px) {
    free_dev(c->frame_work); free_dev(c->frame_rgba);
    c->frame_px = 0;
    HIPCHK(c, hipMalloc((void**)&c->frame_work, sizeof(uint32_t) * 2 * (size_t)render_px * render_px));
    HIPCHK(c, hipMalloc((void**)&c->frame_rgba, (size_t)render_px * render_px * 4));
    c->frame_px = render_px;
  }
  return c->has_f32 ? render_rows<float>(c, c->sf, height, render_px, rgba_out) : render_rows<double>(c, c->sd, height, render_px, rgba_out);
}
NB_API int nbody_render_rgba_dev(void* stream, int64_t n, int is_f64, const void* pos_xy, const void* vel_xy, const void* weight_u32,
                                 uint32_t height, uint32_t render_px, void* work_u32, void* rgba_dev) {
  if (n < 0 || n > (1 << 24) || render_px == 0 || height == 0 || height % render_px != 0 || height > (1u << 24) || !work_u32 || !rgba_dev ||
      (n > 0 && (!pos_xy || !vel_xy || !weight_u32)))
    return NBODY_ERR_INVALID;
  hipError_t e = is_f64 ? launch_render<double>((hipStream_t)stream, n, pos_xy, vel_xy, (const uint32_t*)weight_u32, height, render_px,
                                                (uint32_t*)work_u32, (uint8_t*)rgba_dev)
                        : launch_render<float>((hipStream_t)stream, n, pos_xy, vel_xy, (const uint32_t*)weight_u32, height, render_px,
                                               (uint32_t*)work_u32, (uint8_t*)rgba_dev);
  return e == hipSuccess ? NBODY_OK : NBODY_ERR_HIP;
}

NB_API int64_t nbody_num_particles(const nbody_ctx* c) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_num_particles(nbody::multi_peek(c));
  return c->has_f32 ? c->sf.n : (c->has_f64 ? c->sd.n : 0);
}

NB_API int nbody_update_direct_f32(nbody_ctx* c, float delta, int n_steps, nbody_counting* counter) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return nbody::multi_update_direct(c, delta, n_steps, counter);
  if (!c->has_f32) return fail(c, NBODY_ERR_INVALID, "update_direct_f32: no f32 particles uploaded");
  if (n_steps < 0) return fail(c, NBODY_ERR_INVALID, "update_direct_f32: n_steps < 0");
  HIPCHK(c, hipSetDevice(c->device));
  State<float>& s = c->sf;
  int rc = ensure_workspace(c, direct_ws_bytes(s.n, s.n));
  if (rc) return rc;
  const double t_begin = now_s();
  int step = 0;
  // ---- graph replay of step pairs (no timer attached: event records do not belong in a captured graph)
  const bool want_graph = n_steps >= 4 && s.n > 0 && s.n <= (1 << 17) && !c->timer && env_int("NBODY_DIRECT_GRAPH", 1) != 0;
  rc = ensure_mass_classes(c);  // (host work and copies: before any capture)
  if (rc) return rc;
  if (want_graph) {
    auto& st = s.set[s.cur];
    DirectGraph& g = c->direct_graph;
    const std::string sig = direct_env_signature();
    const auto& mc = s.classes;
    const bool stale = !g.exec || g.n != s.n || g.pos_a != st.pos || g.pos_b != s.pos_next || g.vel != st.vel ||
                       g.mass != st.mass || g.ws != c->workspace || g.delta != delta || g.clamp != c->params.clamp ||
                       g.uniform != direct_mass_hint(s) || g.arith != c->params.arith || g.env != sig || g.row_epoch != s.row_epoch ||
                       g.cls_usable != mc.usable || g.cls_rank != (const void*)mc.rank || g.cls_tile_mass != (const void*)mc.tile_mass;
    if (stale) {
      g.reset();
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      int rc1 = NBODY_OK, rc2 = NBODY_OK;
      if (e == hipSuccess) {
        rc1 = direct_step_dev(c, c->stream, s.n, st.pos, st.mass, direct_mass_hint(s), 0, s.n, st.vel, s.pos_next, nullptr, delta,
                              c->params.clamp, c->params.arith, c->workspace, c->workspace_bytes, nullptr);
        if (!rc1)
          rc2 = direct_step_dev(c, c->stream, s.n, s.pos_next, st.mass, direct_mass_hint(s), 0, s.n, st.vel, st.pos, nullptr, delta,
                                c->params.clamp, c->params.arith, c->workspace, c->workspace_bytes, nullptr);
        e = hipStreamEndCapture(c->stream, &graph);
      }
      if (e == hipSuccess && !rc1 && !rc2 && graph) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess || rc1 || rc2 || !g.exec) {
        g.reset();  // capture is an optimisation: fall through to eager steps
        (void)hipGetLastError();
      } else {
        g.n = s.n; g.pos_a = st.pos; g.pos_b = s.pos_next; g.vel = st.vel; g.mass = st.mass; g.ws = c->workspace;
        g.delta = delta; g.clamp = c->params.clamp; g.uniform = direct_mass_hint(s); g.arith = c->params.arith; g.env = sig;
        g.row_epoch = s.row_epoch; g.cls_usable = mc.usable; g.cls_rank = mc.rank; g.cls_tile_mass = mc.tile_mass;
      }
    }
    if (g.exec) {
      for (; step + 2 <= n_steps; step += 2) HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // an even number of steps: positions are back in st.pos
    }
  }
  for (; step < n_steps; ++step) {
    auto& st = s.set[s.cur];
    rc = direct_step_dev(c, c->stream, s.n, st.pos, st.mass, direct_mass_hint(s), 0, s.n, st.vel, s.pos_next, nullptr, delta,
                         c->params.clamp, c->params.arith, c->workspace, c->workspace_bytes, c->timer);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::swap(st.pos, s.pos_next);
  }
  // force and integrate are one fused kernel: the whole call is booked under sum_gravity
  const double dt = now_s() - t_begin;
  c->counting.sum_gravity += dt;
  if (counter) counter->sum_gravity += dt;
  s.tree_valid = false;
  c->steps_done += (uint64_t)n_steps;
  return NBODY_OK;
}

NB_API int nbody_accel_direct_f32(nbody_ctx* c, float* acc_xy) {
  if (!c) return NBODY_ERR_INVALID;
  NB_VIA_PRIMARY(c, false, nbody_accel_direct_f32(p, acc_xy));
  if (!c->has_f32) return fail(c, NBODY_ERR_INVALID, "accel_direct_f32: no f32 particles uploaded");
  if (!acc_xy) return fail(c, NBODY_ERR_INVALID, "accel_direct_f32: acc_xy is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  State<float>& s = c->sf;
  int rc = ensure_workspace(c, direct_ws_bytes(s.n, s.n));
  if (rc) return rc;
  rc = ensure_mass_classes(c);
  if (rc) return rc;
  auto& st = s.set[s.cur];
  rc = direct_step_dev(c, c->stream, s.n, st.pos, st.mass, direct_mass_hint(s), 0, s.n, nullptr, nullptr, s.acc, 0.f, c->params.clamp,
                       c->params.arith, c->workspace, c->workspace_bytes, c->timer);
  if (rc) return rc;
  if (s.n) HIPCHK(c, hipMemcpyAsync(acc_xy, s.acc, (size_t)s.n * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NBODY_OK;
}

NB_API int nbody_update_tree_f32(nbody_ctx* c, int kind, float delta, int n_steps, nbody_counting* counter) {
  if (c && c->multi) return nbody::multi_update_tree(c, false, kind, (double)delta, n_steps, counter);
  return update_tree<float>(c, kind, delta, n_steps, counter);
}
NB_API int nbody_update_tree_f64(nbody_ctx* c, int kind, double delta, int n_steps, nbody_counting* counter) {
  if (c && c->multi) return nbody::multi_update_tree(c, true, kind, delta, n_steps, counter);
  return update_tree<double>(c, kind, delta, n_steps, counter);
}
// Asynchronous form of nbody_update_tree_f32 and its completion.
NB_API int nbody_update_tree_async_f32(nbody_ctx* c, int kind, float delta, int n_steps) {
  if (c && c->multi) return nbody::multi_update_tree(c, false, kind, (double)delta, n_steps, nullptr);  // (synchronous there)
  return update_tree<float>(c, kind, delta, n_steps, nullptr, true);
}
NB_API int nbody_wait(nbody_ctx* c) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return NBODY_OK;  // every call on a multi-device context is synchronous
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = phase_drain(c);
  if (!rc && c->has_f32) rc = step_ahead_collect<float>(c, c->sf);
  return rc;
}
static int not_on_multi(nbody_ctx* c, const char* what) {
  return fail(c, NBODY_ERR_INVALID, std::string(what) + ": a context made by nbody_create_multi shards its steps itself");
}
NB_API int nbody_update_tree_shard_f32(nbody_ctx* c, int kind, float delta, int64_t begin, int64_t count, nbody_counting* counter) {
  if (c && c->multi) return not_on_multi(c, "update_tree_shard");
  return update_tree_shard<float>(c, kind, delta, begin, count, counter);
}
NB_API int nbody_update_tree_shard_f64(nbody_ctx* c, int kind, double delta, int64_t begin, int64_t count, nbody_counting* counter) {
  if (c && c->multi) return not_on_multi(c, "update_tree_shard");
  return update_tree_shard<double>(c, kind, delta, begin, count, counter);
}
NB_API int nbody_export_slice_dev(nbody_ctx* c, int64_t begin, int64_t count, void* rows_u32, void* pos_xy, void* vel_xy) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return not_on_multi(c, "export_slice");
  return c->has_f64 ? export_slice<double>(c, begin, count, rows_u32, pos_xy, vel_xy)
                    : export_slice<float>(c, begin, count, rows_u32, pos_xy, vel_xy);
}
NB_API int nbody_import_rows_dev(nbody_ctx* c, int64_t n_rows, const void* rows_u32, const void* pos_xy, const void* vel_xy) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return not_on_multi(c, "import_rows");
  return c->has_f64 ? import_rows_api<double>(c, n_rows, rows_u32, pos_xy, vel_xy)
                    : import_rows_api<float>(c, n_rows, rows_u32, pos_xy, vel_xy);
}
// (a BVH build permutes the rows of the device that ran it: the other replicas are refreshed afterwards)
NB_API int nbody_accel_tree_f32(nbody_ctx* c, int kind, int64_t n_targets, const float* target_xy, float* acc_xy) {
  NB_VIA_PRIMARY(c, true, accel_tree<float>(p, kind, n_targets, target_xy, acc_xy));
  return accel_tree<float>(c, kind, n_targets, target_xy, acc_xy);
}
NB_API int nbody_accel_tree_f64(nbody_ctx* c, int kind, int64_t n_targets, const double* target_xy, double* acc_xy) {
  NB_VIA_PRIMARY(c, true, accel_tree<double>(p, kind, n_targets, target_xy, acc_xy));
  return accel_tree<double>(c, kind, n_targets, target_xy, acc_xy);
}
NB_API int nbody_walk_tree_f32(nbody_ctx* c, int kind, int64_t n_nodes, const float* geom, const uint32_t* mass, const int32_t* is_leaf,
                               const int64_t* leaf_first, const int64_t* leaf_count, const int64_t* skip, const uint32_t* order,
                               int64_t n_targets, const float* target_xy, float* acc_xy) {
  NB_VIA_PRIMARY(c, true, walk_tree<float>(p, kind, n_nodes, geom, mass, is_leaf, leaf_first, leaf_count, skip, order, n_targets, target_xy, acc_xy));
  return walk_tree<float>(c, kind, n_nodes, geom, mass, is_leaf, leaf_first, leaf_count, skip, order, n_targets, target_xy, acc_xy);
}
NB_API int nbody_walk_tree_f64(nbody_ctx* c, int kind, int64_t n_nodes, const double* geom, const uint32_t* mass, const int32_t* is_leaf,
                               const int64_t* leaf_first, const int64_t* leaf_count, const int64_t* skip, const uint32_t* order,
                               int64_t n_targets, const double* target_xy, double* acc_xy) {
  NB_VIA_PRIMARY(c, true, walk_tree<double>(p, kind, n_nodes, geom, mass, is_leaf, leaf_first, leaf_count, skip, order, n_targets, target_xy, acc_xy));
  return walk_tree<double>(c, kind, n_nodes, geom, mass, is_leaf, leaf_first, leaf_count, skip, order, n_targets, target_xy, acc_xy);
}
NB_API int nbody_tree_validate(int kind, int64_t n_nodes, const int32_t* is_leaf, const int64_t* leaf_first, const int64_t* leaf_count,
                               const int64_t* skip, int64_t n_particles, const uint32_t* order, char* reason, size_t reason_cap) {
  std::string why;
  const bool ok = tree_shape_ok(kind, n_nodes, is_leaf, leaf_first, leaf_count, skip, n_particles, order, nullptr, why);
  if (reason && reason_cap) {
    const size_t k = why.size() < reason_cap - 1 ? why.size() : reason_cap - 1;
    std::memcpy(reason, why.data(), k);
    reason[k] = 0;
  }
  return ok ? NBODY_OK : NBODY_ERR_INVALID;
}

NB_API int nbody_tree_info(const nbody_ctx* c, nbody_tree_view* out) {
  if (!c || !out) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_tree_info(nbody::multi_peek(c), out);
  nbody_ctx* mc = const_cast<nbody_ctx*>(c);
  if (c->has_f32 && c->sf.tree_valid) {
    out->n_nodes = c->sf.n_nodes; out->kind = c->sf.tree_kind; out->max_depth = c->sf.tree_max_depth;
    return NBODY_OK;
  }
  if (c->has_f64 && c->sd.tree_valid) {
    out->n_nodes = c->sd.n_nodes; out->kind = c->sd.tree_kind; out->max_depth = c->sd.tree_max_depth;
    return NBODY_OK;
  }
  return fail(mc, NBODY_ERR_INVALID, "tree_info: no tree built yet");
}
NB_API int nbody_tree_export_f32(const nbody_ctx* c, float* geom, uint32_t* mass, int32_t* is_leaf, int64_t* first,
                                 int64_t* count, int64_t* skip, uint32_t* order) {
  if (c && c->multi) c = nbody::multi_peek(c);
  return tree_export<float>(c, geom, mass, is_leaf, first, count, skip, order);
}
NB_API int nbody_tree_export_f64(const nbody_ctx* c, double* geom, uint32_t* mass, int32_t* is_leaf, int64_t* first,
                                 int64_t* count, int64_t* skip, uint32_t* order) {
  if (c && c->multi) c = nbody::multi_peek(c);
  return tree_export<double>(c, geom, mass, is_leaf, first, count, skip, order);
}
template <class T>
static int host_tree_build(int kind, int64_t n, const T* pos, const uint32_t* w, const nbody_params* p, nbody_host_tree** out) {
  if (!out) return fail(nullptr, NBODY_ERR_INVALID, "host_tree_build: out is NULL");
  *out = nullptr;
  if (n < 0 || n > 0x7fffffffLL || (n > 0 && !pos)) return fail(nullptr, NBODY_ERR_INVALID, "host_tree_build: bad arguments");
  nbody_params dflt;
  nbody_default_params(&dflt);
  if (!p) p = &dflt;
  if (p->leaf_size < 1) return fail(nullptr, NBODY_ERR_INVALID, "host_tree_build: leaf_size < 1");
  auto* h = new (std::nothrow) nbody_host_tree();
  if (!h) return fail(nullptr, NBODY_ERR_NOMEM, "host_tree_build: out of memory");
  TreeHost<T>* t;
  if constexpr (sizeof(T) == 8) { h->is_f64 = true; t = &h->td; } else { t = &h->tf; }
  if (kind == NBODY_TREE_BVH) build_bvh<T>(pos, w, n, p->leaf_size, *t);
  else if (kind == NBODY_TREE_QUAD) build_quad<T>(pos, w, n, (T)p->quad_root_x, (T)p->quad_root_y, (T)p->quad_root_h, *t);
  else { delete h; return fail(nullptr, NBODY_ERR_INVALID, "host_tree_build: unknown tree kind"); }
  *out = h;
  return t->overflow ? NBODY_ERR_DEGENERATE : NBODY_OK;
}

NB_API int nbody_host_tree_build_f32(int kind, int64_t n, const float* pos_xy, const uint32_t* weight,
                                     const nbody_params* p, nbody_host_tree** out) {
  return host_tree_build<float>(kind, n, pos_xy, weight, p, out);
}
NB_API int nbody_host_tree_build_f64(int kind, int64_t n, const double* pos_xy, const uint32_t* weight,
                                     const nbody_params* p, nbody_host_tree** out) {
  return host_tree_build<double>(kind, n, pos_xy, weight, p, out);
}
NB_API void nbody_host_tree_free(nbody_host_tree* t) { delete t; }
NB_API int nbody_host_tree_info(const nbody_host_tree* t, nbody_tree_view* out) {
  if (!t || !out) return NBODY_ERR_INVALID;
  if (t->is_f64) { out->n_nodes = (int64_t)t->td.size(); out->kind = t->td.kind; out->max_depth = t->td.max_depth; }
  else { out->n_nodes = (int64_t)t->tf.size(); out->kind = t->tf.kind; out->max_depth = t->tf.max_depth; }
  return NBODY_OK;
}
NB_API int nbody_host_tree_export_f32(const nbody_host_tree* t, float* geom, uint32_t* mass, int32_t* is_leaf,
                                      int64_t* first, int64_t* count, int64_t* skip, uint32_t* order) {
  if (!t || t->is_f64) return fail(nullptr, NBODY_ERR_INVALID, "host_tree_export_f32: not an f32 tree");
  tree_export_host<float>(t->tf, geom, mass, is_leaf, first, count, skip, order);
  return NBODY_OK;
}
NB_API int nbody_host_tree_export_f64(const nbody_host_tree* t, double* geom, uint32_t* mass, int32_t* is_leaf,
                                      int64_t* first, int64_t* count, int64_t* skip, uint32_t* order) {
  if (!t || !t->is_f64) return fail(nullptr, NBODY_ERR_INVALID, "host_tree_export_f64: not an f64 tree");
  tree_export_host<double>(t->td, geom, mass, is_leaf, first, count, skip, order);
  return NBODY_OK;
}

NB_API int nbody_tree_walk_stats(nbody_ctx* c, int enable, uint64_t* node_visits, uint64_t* accepted, uint64_t* leaf_pairs) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return not_on_multi(c, "tree_walk_stats (each device walks a slice)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (node_visits) *node_visits = c->last_stats[0];
  if (accepted) *accepted = c->last_stats[1];
  if (leaf_pairs) *leaf_pairs = c->last_stats[2];
  c->want_stats = enable != 0;
  return NBODY_OK;
}

NB_API int nbody_get_counting(const nbody_ctx* c, nbody_counting* out) {
  if (!c || !out) return NBODY_ERR_INVALID;
  *out = c->counting;
  return NBODY_OK;
}

NB_API size_t nbody_direct_workspace_bytes(int64_t n_sources, int64_t n_targets) {
  if (n_sources < 0 || n_targets < 0) return 0;
  return direct_ws_bytes(n_sources, n_targets);
}

NB_API int nbody_direct_step_dev(void* stream, int64_t n_sources, const void* pos_all, const void* mass_all,
                                 float uniform_mass, int64_t target_begin, int64_t n_targets, void* vel, void* pos_out, void* acc_out,
                                 float delta, float clamp, int arith, void* workspace, size_t workspace_bytes,
                                 nbody_timer* timer) {
  return direct_step_dev(nullptr, (hipStream_t)stream, n_sources, pos_all, mass_all, uniform_mass, target_begin, n_targets, vel,
                         pos_out, acc_out, delta, clamp, arith, workspace, workspace_bytes, timer);
}

NB_API int nbody_direct_workspace_peek(void* stream, const void* workspace, int32_t out[4]) {
  if (!workspace || !out) return fail(nullptr, NBODY_ERR_INVALID, "workspace_peek: bad arguments");
  hipError_t e = hipMemcpyAsync(out, workspace, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return e == hipSuccess ? NBODY_OK : fail_hip(nullptr, e, "workspace_peek");
}

NB_API int nbody_weights_to_mass_dev(void* stream, int64_t n, const void* weight_u32, void* mass_f32) {
  if (n < 0 || (n > 0 && (!weight_u32 || !mass_f32))) return fail(nullptr, NBODY_ERR_INVALID, "weights_to_mass: bad arguments");
  hipError_t e = launch_weights_to_mass((hipStream_t)stream, (const uint32_t*)weight_u32, (float*)mass_f32, n);
  return e == hipSuccess ? NBODY_OK : fail_hip(nullptr, e, "weights_to_mass");
}

NB_API int nbody_selftest_exact_sum(const float* x, int64_t n, int tile, int seq_run, float* out_sum, int64_t* out_restarts) {
  if ((!x && n > 0) || n < 0 || tile < 1 || seq_run < 1 || !out_sum) return NBODY_ERR_INVALID;
  *out_sum = xsum::emulate_fold(x, n, tile, seq_run, out_restarts);
  return NBODY_OK;
}
NB_API int nbody_selftest_div_pair(int device, const float* nx, const float* ny, const float* den, int64_t n, float* qx, float* qy) {
  if (n < 0 || (n > 0 && (!nx || !ny || !den || !qx || !qy))) return NBODY_ERR_INVALID;
  if (n == 0) return NBODY_OK;
  if (hipSetDevice(device) != hipSuccess) return NBODY_ERR_NO_DEVICE;
  float* d = nullptr;
  const size_t b = (size_t)n * sizeof(float);
  if (hipMalloc((void**)&d, 5 * b) != hipSuccess) return NBODY_ERR_HIP;
  hipError_t e = hipMemcpy(d, nx, b, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + n, ny, b, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 2 * n, den, b, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_div_pair_selftest(nullptr, d, d + n, d + 2 * n, n, d + 3 * n, d + 4 * n);
  if (e == hipSuccess) e = hipMemcpy(qx, d + 3 * n, b, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(qy, d + 4 * n, b, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? NBODY_OK : NBODY_ERR_HIP;
}
NB_API int nbody_selftest_exact_sum_chunked(const float* x, int64_t n, int chunk, float* out_sum, int64_t* out_runs_used) {
  if ((!x && n > 0) || n < 0 || chunk < 1 || !out_sum) return NBODY_ERR_INVALID;
  // segments of 8 addends per thread, as bvh_chunk_runs cuts a chunk
  *out_sum = xsum::emulate_fold_chunked2(x, n, chunk, chunk >= 8 ? 8 : 1, out_runs_used);
  return NBODY_OK;
}
NB_API int nbody_selftest_exact_sum_f64(const double* x, int64_t n, int tile, int seq_run, double* out_sum, int64_t* out_restarts) {
  if ((!x && n > 0) || n < 0 || tile < 1 || seq_run < 1 || !out_sum) return NBODY_ERR_INVALID;
  *out_sum = xsum64::emulate_fold(x, n, tile, seq_run, out_restarts);
  return NBODY_OK;
}
NB_API int nbody_selftest_exact_sum_f64_segmented(const double* x, int64_t n, int seg, double* out_sum, int64_t* out_runs_used) {
  if ((!x && n > 0) || n < 0 || seg < 1 || !out_sum) return NBODY_ERR_INVALID;
  *out_sum = xsum64::emulate_fold_segmented(x, n, seg, out_runs_used);
  return NBODY_OK;
}
NB_API int nbody_bvh_build_restarts(const nbody_ctx* ctx) {
  if (ctx && ctx->multi) ctx = nbody::multi_peek(ctx);
  return ctx ? ctx->bvh_stops : 0;
}
NB_API int nbody_last_build_on_device(const nbody_ctx* ctx) {
  if (ctx && ctx->multi) ctx = nbody::multi_peek(ctx);
  return ctx && ctx->last_build_device ? 1 : 0;
}

NB_API int nbody_timer_create(nbody_timer** out) {
  if (!out) return NBODY_ERR_INVALID;
  *out = new (std::nothrow) nbody_timer();
  return *out ? NBODY_OK : NBODY_ERR_NOMEM;
}
NB_API void nbody_timer_destroy(nbody_timer* t) { delete t; }
NB_API int nbody_timer_read(nbody_timer* t, int reset, double* avg_ms, int64_t* launches) {
  if (!t) return NBODY_ERR_INVALID;
  hipError_t e = t->drain();
  if (e != hipSuccess) return fail_hip(nullptr, e, "timer drain");
  if (avg_ms) *avg_ms = t->launches ? t->total_ms / (double)t->launches : 0.0;
  if (launches) *launches = t->launches;
  if (reset) { t->total_ms = 0.0; t->launches = 0; }
  return NBODY_OK;
}
NB_API int nbody_set_timer(nbody_ctx* c, nbody_timer* t) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) nbody::multi_peek(c)->timer = t;  // the first device's kernels are the ones timed
  c->timer = t;
  return NBODY_OK;
}

NB_API int nbody_direct_prep_dev(void* stream, int64_t n_sources, const void* pos_all, const void* mass_all, float uniform_mass,
                                 int64_t n_targets_total, int64_t n_targets_max, float clamp, int arith, void* workspace,
                                 size_t workspace_bytes) {
  return direct_prep(nullptr, (hipStream_t)stream, n_sources, pos_all, mass_all, uniform_mass, n_targets_total, n_targets_max, clamp,
                     arith, workspace, workspace_bytes);
}
NB_API int nbody_direct_run_dev(void* stream, int64_t n_sources, const void* pos_all, const void* mass_all, float uniform_mass,
                                int64_t target_begin, int64_t n_targets, void* vel, void* pos_out, void* acc_out, float delta,
                                float clamp, int arith, int64_t n_targets_total, int64_t n_targets_max, void* workspace,
                                size_t workspace_bytes, nbody_timer* timer) {
  return direct_run(nullptr, (hipStream_t)stream, n_sources, pos_all, mass_all, uniform_mass, target_begin, n_targets, vel, pos_out,
                    acc_out, delta, clamp, arith, n_targets_total, n_targets_max, workspace, workspace_bytes, timer);
}
NB_API void* nbody_get_stream(const nbody_ctx* c) {
  if (c && c->multi) c = nbody::multi_peek(c);
  return c ? (void*)c->stream : nullptr;
}

// ---- what multi.hip needs of this translation unit (ctx.h)
namespace nbody {
int ctx_fail(nbody_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }
int ctx_upload(nbody_ctx* c, bool f64, int64_t n, const void* pos, const void* vel, const uint32_t* w) {
  return f64 ? upload<double>(c, n, (const double*)pos, (const double*)vel, w) : upload<float>(c, n, (const float*)pos, (const float*)vel, w);
}
int ctx_update_tree(nbody_ctx* c, bool f64, int kind, double delta, int n_steps, nbody_counting* counter) {
  return f64 ? update_tree<double>(c, kind, delta, n_steps, counter) : update_tree<float>(c, kind, (float)delta, n_steps, counter);
}
int ctx_update_tree_shard(nbody_ctx* c, bool f64, int kind, double delta, int64_t begin, int64_t count, nbody_counting* counter) {
  return f64 ? update_tree_shard<double>(c, kind, delta, begin, count, counter)
             : update_tree_shard<float>(c, kind, (float)delta, begin, count, counter);
}
int ctx_export_slice(nbody_ctx* c, int64_t begin, int64_t count, void* rows, void* pos, void* vel) {
  return c->has_f64 ? export_slice<double>(c, begin, count, rows, pos, vel) : export_slice<float>(c, begin, count, rows, pos, vel);
}
int ctx_import_rows(nbody_ctx* c, int64_t n_rows, const void* rows, const void* pos, const void* vel) {
  return c->has_f64 ? import_rows_api<double>(c, n_rows, rows, pos, vel) : import_rows_api<float>(c, n_rows, rows, pos, vel);
}
size_t ctx_direct_ws_bytes(int64_t n_src, int64_t n_tgt) { return direct_ws_bytes(n_src, n_tgt); }
int ctx_ensure_workspace(nbody_ctx* c, size_t bytes) { return ensure_workspace(c, bytes); }
int ctx_ensure_mass_classes(nbody_ctx* c) { return c && c->has_f32 ? ensure_mass_classes(c) : NBODY_OK; }
int ctx_direct_prep(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
                    int64_t n_tgt_total, int64_t n_tgt_max, float clamp, int arith, void* ws, size_t ws_bytes) {
  return direct_prep(c, stream, n_src, pos_all, mass_all, uniform_mass, n_tgt_total, n_tgt_max, clamp, arith, ws, ws_bytes);
}
int ctx_direct_run(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
                   int64_t tgt_begin, int64_t n_tgt, void* vel, void* pos_out, void* acc_out, float delta, float clamp, int arith,
                   int64_t n_tgt_total, int64_t n_tgt_max, void* ws, size_t ws_bytes, nbody_timer* timer) {
  return direct_run(c, stream, n_src, pos_all, mass_all, uniform_mass, tgt_begin, n_tgt, vel, pos_out, acc_out, delta, clamp, arith,
                    n_tgt_total, n_tgt_max, ws, ws_bytes, timer);
}
}  // namespace nbody
