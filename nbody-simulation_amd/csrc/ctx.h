// Internal: the context of libnbody_hip and the functions its translation units share (capi.hip: one device;
// multi.hip: several devices behind the same handle).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "common.h"
#include "env.h"
#include "tree_build.hpp"

namespace nbody {

struct Multi;  // multi.hip: the devices of a context made by nbody_create_multi

template <class T> struct State {
  using T2 = typename Vec2T<T>::type;
  struct Set {
    T2* pos = nullptr;
    T2* vel = nullptr;
    uint32_t* weight = nullptr;
    uint32_t* ids = nullptr;
    T* mass = nullptr;
  };
  int64_t n = 0;
  Set set[2];
  int cur = 0;
  T2* pos_next = nullptr;  // direct step output, swapped with set[cur].pos
  float uniform_mass = 0.f;  // > 0 when every weight is the same value (checked on upload)
  float sparse_base = 0.f;   // > 0 when every weight is this value but for at most n/256 bodies (the reference's scene)
  // Mass classes (direct step, masses that differ freely but take at most 32 values): the far copy of the positions is
  // written in class order, every class padded to whole 1024-source tiles, so the main pass runs the equal-mass arithmetic
  // tile by tile.  Built lazily for the current row order (row_epoch moves whenever a build permutes the rows).
  struct MassClasses {
    uint64_t epoch = ~0ull;        // row_epoch these arrays were built for
    bool usable = false;           // 2 .. 32 distinct masses, none of the cheaper cases
    int n_classes = 0;
    int64_t n_slots = 0;           // bodies + padding
    int n_pad_slots = 0;
    uint32_t* rank = nullptr;      // [n] device: body -> slot
    uint32_t* pad_slots = nullptr; // [n_pad_slots] device
    float* tile_mass = nullptr;    // [n_slots / 1024] device
  } classes;
  uint64_t row_epoch = 0;
  T2* acc = nullptr;
  // tree
  void* geom0 = nullptr;
  void* geom1 = nullptr;
  void* link = nullptr;
  size_t node_cap = 0;
  uint32_t* order_dev = nullptr;
  TreeHost<T> tree;          // host image of the last build (filled lazily after a device build)
  bool tree_valid = false;
  bool tree_host_stale = false;  // the last build ran on the device and has not been downloaded
  int n_nodes = 0, tree_kind = 0, tree_max_depth = 0;
  int shard_kind = 0;            // tree kind of the last sharded step (decides how a slice maps to rows)
  int* node_depth = nullptr;     // device build: depth of every node
  uint32_t* node_mass = nullptr; // device build: u32 mass of every node
  T2* node_size = nullptr;       // device BVH build: boundary.size of every node
  size_t node_aux_cap = 0;
  char* qb_scratch = nullptr;
  size_t qb_scratch_bytes = 0;
  char* bb_scratch = nullptr;    // device BVH build
  size_t bb_scratch_bytes = 0;
  bool h_weight_stale = false;   // a device BVH build permuted the rows without touching h_weight (row_epoch moves with it)
  int quad_depth_hint = 0;       // depth of the last device-built quad tree (how many levels the next build sorts by)
  int bvh_levels_hint = 0;       // long-node levels the last device-built BVH had (how many the next step enqueues blind)
  int bvh_levels_stable = 0;     // consecutive step-ahead builds that had exactly that many: after eight the spare blind level is dropped
  bool bb_flags_clean = false;   // the BVH build's flags and level counters are zero (the step enqueued ahead left them so)
  bool ahead_total_due = false;  // the last ahead-step's exact term count has not been read back yet
  // split walk (walk_split.hip): counts/offsets scratch and the term array
  char* ws_scratch = nullptr;
  size_t ws_scratch_bytes = 0;
  void* ws_terms = nullptr;
  int64_t ws_capacity = 0;       // terms
  int ws_backoff = 0;            // steps for which the split walk is not tried (the last one needed too much memory)
  // one-pass walk (walk_tile): the counts in ws_scratch are those of the last walk over the context's own particles
  int64_t wt_hist_n = -1, wt_hist_begin = 0;
  unsigned long long wt_total = 0;  // terms of that walk
  uint32_t* wt_hist = nullptr;      // [n] per particle id: its terms in that walk
  std::vector<T> h_pos;
  std::vector<uint32_t> h_weight;  // current row order
  std::vector<uint32_t> h_tmp;
};


// The `uniform_mass` argument of the direct step for this state: > 0 all equal, < 0 all equal to its magnitude but a few, 0 neither.
template <class T> inline float direct_mass_hint(const State<T>& s) { return s.uniform_mass > 0.f ? s.uniform_mass : -s.sparse_base; }

}  // namespace nbody

// Two consecutive direct steps (A -> B -> A position buffers) captured once as a hipGraph and replayed: a small-N step
// is ~15 launches (hazard scan, near/far split, gated kernels), i.e. launch-bound (N = 1024: 78 us per eager step
// against 13 us of kernels).  The graph is rebuilt when anything it baked in changes.
namespace nbody {
struct DirectGraph {
  hipGraphExec_t exec = nullptr;
  int64_t n = -1;
  const void *pos_a = nullptr, *pos_b = nullptr, *vel = nullptr, *mass = nullptr, *ws = nullptr;
  float delta = 0.f, clamp = 0.f, uniform = 0.f;
  int arith = -1;
  // what the step derives from the row order: the mass classes' arrays (their addresses are kernel arguments of the capture)
  uint64_t row_epoch = ~0ull;
  bool cls_usable = false;
  const void *cls_rank = nullptr, *cls_tile_mass = nullptr;
  std::string env;  // the NBODY_DIRECT_* switches read at capture time
  void reset() {
    if (exec) (void)hipGraphExecDestroy(exec);
    exec = nullptr;
    n = -1;
  }
};
inline std::string direct_env_signature() {
  std::string sig;
  const char* nf = getenv("NBODY_DIRECT_NEARFAR");
  sig += nf ? nf : "-";
  sig += ';';
  for (const char* k : {"NBODY_DIRECT_ASM", "NBODY_DIRECT_TPT", "NBODY_DIRECT_GSPLIT", "NBODY_DIRECT_NO_UNIFORM", "NBODY_DIRECT_NO_CLASSES",
                        "NBODY_DIRECT_NO_SPARSE"}) {  // (laboratory build only)
    const char* v = lab_str(k);
    sig += v ? v : "-";
    sig += ';';
  }
  return sig;
}

}  // namespace nbody

namespace nbody {
// The three phases of a tree step (Counting, main.rs:74-79) timed by four events on the step's stream: the stream idles
// while the host works inside a phase (a host-side build), so device-timeline intervals cover host time too, and no
// phase boundary needs a host synchronisation.
struct PhaseEvents {
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
  bool borrowed = false;  // e[0] is the previous step's e[3]
};
}  // namespace nbody

struct nbody_ctx {
  nbody::Multi* multi = nullptr;   // non-null: this handle fronts several devices (multi.hip) and owns no device state itself
  int64_t row_capacity = 0;        // rows the particle arrays are allocated for (>= n; a multi context pads to its block layout)
  nbody::DirectGraph direct_graph;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  nbody_params params{};
  nbody_counting counting{};
  nbody_timer* timer = nullptr;
  bool has_f32 = false, has_f64 = false;
  nbody::State<float> sf;
  nbody::State<double> sd;
  void* workspace = nullptr;
  int bvh_stops = 0;  // exact-sum restarts of the last device BVH build (diagnostic)
  bool last_build_device = false;
  size_t workspace_bytes = 0;
  unsigned long long* stats_dev = nullptr;
  uint32_t* frame_work = nullptr;  // render: per-pixel counters
  hipStream_t copy_stream = nullptr;  // snapshot transfers, concurrent with the steps on `stream`
  hipEvent_t snap_event = nullptr;
  bool snap_pending = false;
  uint64_t steps_done = 0, snap_step = 0;
  // snapshot staging: device copy of the rows, pinned host image
  void *snap_pos = nullptr, *snap_vel = nullptr, *snap_hpos = nullptr, *snap_hvel = nullptr;
  uint32_t *snap_w = nullptr, *snap_ids = nullptr, *snap_hw = nullptr, *snap_hids = nullptr;
  size_t snap_bytes2 = 0;  // bytes of one position array the staging holds
  int64_t snap_n = 0;
  bool snap_f64 = false;
  // delta snapshots (delta_snapshot.hip): three key arrays in rotation (this / previous / the one before), the pieces
  // of the stream on the device, the assembled stream in pinned memory
  void* dl_keys[3] = {nullptr, nullptr, nullptr};
  int dl_cur = 0;
  uint8_t* dl_widths = nullptr;
  uint32_t *dl_words = nullptr, *dl_offsets = nullptr;
  void* dl_scan = nullptr;
  size_t dl_scan_bytes = 0;
  uint64_t *dl_payload = nullptr, *dl_total = nullptr, *dl_htotal = nullptr;
  uint8_t* dl_host = nullptr;
  int64_t dl_n = -1;
  int dl_bits = 0;
  bool dl_key_next = true, dl_pending = false;
  size_t dl_stream_bytes = 0;
  uint64_t dl_step = 0;
  uint8_t* frame_rgba = nullptr;
  uint32_t frame_px = 0;
  unsigned long long last_stats[3] = {0, 0, 0};
  bool want_stats = false;
  // Phase clock of the steps enqueued ahead (capi.hip, bvh_step_ahead): instead of three event records per step (each ~6 us of idle
  // stream) the step's own kernels write the 100 MHz wall clock at the phase boundaries: slot = {build begins, walk begins,
  // integration begins, step ends}; read back when the phases are drained.
  unsigned long long* stamp_dev = nullptr;               // [kStampSlots][4]
  std::vector<int> stamp_pending;                        // slots of finished steps not yet read
  int stamp_next = 0;
  int stamp_open = -1;                                   // slot whose end is still to be written (by the next step's first kernel, or a stamp kernel)
  std::vector<nbody::PhaseEvents> ph_pending;           // recorded phase events
  std::vector<hipEvent_t> ph_free;                       // reusable events
  nbody_counting* ph_counter = nullptr;                // the caller's counter of the call in progress
  // a BVH step enqueued whole, ahead of the host's knowledge of its build (capi.hip, bvh_step_ahead)
  int* spec_dev = nullptr;    // [2] verdict of the build: node count or 0, ok
  int* spec_host_dev = nullptr;  // spec_host as the device addresses it (kernels write the record there themselves)
  int* spec_host = nullptr;   // pinned: verdict [2] | build flags + level counters [128] | walk info before [8] and after [8] the walk
  hipEvent_t spec_event = nullptr;
};


namespace nbody {
// ---- capi.hip
int ctx_fail(nbody_ctx* c, int code, const std::string& msg);  // c == NULL: the thread's create error
int ctx_create_single(nbody_ctx** out, int device_id);
void ctx_destroy_single(nbody_ctx* c);
int ctx_upload(nbody_ctx* c, bool f64, int64_t n, const void* pos, const void* vel, const uint32_t* w);
int ctx_update_tree(nbody_ctx* c, bool f64, int kind, double delta, int n_steps, nbody_counting* counter);  // the single-device call
int ctx_update_tree_shard(nbody_ctx* c, bool f64, int kind, double delta, int64_t begin, int64_t count, nbody_counting* counter);
int ctx_export_slice(nbody_ctx* c, int64_t begin, int64_t count, void* rows, void* pos, void* vel);
int ctx_import_rows(nbody_ctx* c, int64_t n_rows, const void* rows, const void* pos, const void* vel);
size_t ctx_direct_ws_bytes(int64_t n_src, int64_t n_tgt);
int ctx_ensure_workspace(nbody_ctx* c, size_t bytes);
int ctx_ensure_mass_classes(nbody_ctx* c);  // State::MassClasses of the f32 rows, for the direct steps that follow
int ctx_direct_prep(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
                    int64_t n_tgt_total, int64_t n_tgt_max, float clamp, int arith, void* ws, size_t ws_bytes);
int ctx_direct_run(nbody_ctx* c, hipStream_t stream, int64_t n_src, const void* pos_all, const void* mass_all, float uniform_mass,
                   int64_t tgt_begin, int64_t n_tgt, void* vel, void* pos_out, void* acc_out, float delta, float clamp, int arith,
                   int64_t n_tgt_total, int64_t n_tgt_max, void* ws, size_t ws_bytes, nbody_timer* timer);
// ---- multi.hip (`front` is the handle nbody_create_multi returned)
void multi_destroy(nbody_ctx* front);
int multi_set_params(nbody_ctx* front);
int multi_upload(nbody_ctx* front, bool f64, int64_t n, const void* pos, const void* vel, const uint32_t* w);
nbody_ctx* multi_peek(const nbody_ctx* front);            // the first device's context, as it is
int multi_primary(nbody_ctx* front, nbody_ctx** out);     // ... after bringing every replica up to date
int multi_replicate(nbody_ctx* front);                    // the first device's rows to every other replica
int multi_update_direct(nbody_ctx* front, float delta, int n_steps, nbody_counting* counter);
int multi_update_tree(nbody_ctx* front, bool f64, int kind, double delta, int n_steps, nbody_counting* counter);
}  // namespace nbody
