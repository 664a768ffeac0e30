// liblynxhip: C-ABI entry points (include/lynx_hip.h) over the gfx950 kernels in
// lynx_device.hpp.  One lynx_ctx per GPU and per process: one HIP stream, a caching
// device allocator (so that `track()` never pays hipMalloc on the hot path), the
// moment-reduction scratch and, optionally, an RCCL communicator.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "lynx_device.hpp"
#include "lynx_grad.hpp"
#include "lynx_units.hpp"
#include "lynx_grad_units.hpp"

using namespace lynx;

// Launch-plan switches.  The planner picks a form of the computation by shape (wave tiles or per-particle accesses,
// particles per lane, which stream builds and reduces, ...); every knob forces one of the forms it would otherwise pick
// for another shape, for tests (test_every_kernel_variant_gives_the_default_answer) and the A/B scripts under
// scripts/gpu/.  They are read from the environment ONCE, when the context is created (lynx_ctx_reload_knobs reads them
// again); a `track` call reads none.  -1 = not set: the planner decides.
struct Knobs {
  int xpose = -1;             // LYNX_XPOSE             wave tiles through LDS (1) / per-particle accesses (0)
  int unroll = -1;            // LYNX_UNROLL            particles per lane: 1 | 2 | 4
  int mom = -1;               // LYNX_MOM               moment accumulation mode of the float32 kernels: 2 | 3
  int min_tiles_per_wg = -1;  // LYNX_MIN_TILES_PER_WG
  int async_build = -1;       // LYNX_ASYNC_BUILD       build on the second stream (1) / in line (0)
  int side_reduce = -1;       // LYNX_SIDE_REDUCE       moment reduction on the side stream (1) / in line (0)
  int build_host_wait = -1;   // LYNX_BUILD_HOST_WAIT   the host (1) / the main stream (0) waits for an asynchronous build
  int gather_overlap = -1;    // LYNX_GATHER_OVERLAP    RCCL gather on the side stream (1) / in line (0)
  int lanes_build_min_batch = 256;  // LYNX_LANES_BUILD_MIN_BATCH  lanes = samples build from this batch on
  int piece = 8;              // LYNX_PIECE             elements per piece of the lanes build (tests: 1 and 3 make short lattices grow a pair tree)
  int pair_levels_fused = 1;  // LYNX_PAIR_LEVELS_FUSED narrow pair trees in one launch (wide ones always take one per level)
  int fuse_max_chunks = 0;    // LYNX_FUSE_MAX_CHUNKS   fused build prologue for samples of <= n workgroups (0: never)
  int merge_steps = 1;        // LYNX_MERGE_STEPS       [run, cavity] pairs as one unit
  int reduce_wide = 0;        // LYNX_REDUCE_WIDE       1: one 1024-thread workgroup per sample for beams of few samples with a few hundred records each
  int reduce_ticket = 1;      // LYNX_REDUCE_TICKET     samples with more records than one workgroup walks: both levels in one launch (0: two launches)
  int one_round = -1;         // LYNX_ONE_ROUND         at most this many workgroups per CU in a launch (0: no cap; default: 3 for single-map launches of a few rounds)
  int track_units = 1;        // LYNX_TRACK_UNITS       structured step loop (2: insist)
  int unit_pairs = 1;         // LYNX_UNIT_PAIRS        lattices of merged [run, cavity] pairs of class U: the kernel written for that form (0: the general one, 2: insist)
  int bwd_units = 1;          // LYNX_BWD_UNITS         structured reverse pass (2, for tests: every odd sample is left to the dense kernel all the same)
  int bwd_merge = 1;          // LYNX_BWD_MERGE         merged pairs in the reverse pass
  int bwd_pairs = 1;          // LYNX_BWD_PAIRS         two particles per lane in the float32 reverse pass
  int build_in_tail = 1;      // LYNX_BUILD_IN_TAIL     start the next build in the tail of the streaming kernel
  int small_inline = -1;      // LYNX_SMALL_INLINE      short calls: build, stream and reduce back to back on ONE stream, no events
  int bwd_reuse_table = 1;    // LYNX_BWD_REUSE_TABLE   the reverse pass reads the step table the forward call just built (0: builds its own)
  int host_visible_records = 1;  // LYNX_HOST_VISIBLE_RECORDS  moment records of a few samples in host memory the GPU writes through (0: device memory)
  int alternate_order = 1;    // LYNX_ALTERNATE_ORDER   long calls walk the batch forwards and backwards in turn (0: always forwards, 2: every call backwards)
  int inline_pool = 1;        // LYNX_INLINE_POOL       small lattices: the parameters by value in the kernel arguments (0: always from memory)
};

static void load_knobs(Knobs* k) {
  *k = Knobs();
  const struct { const char* name; int* value; } table[] = {
      {"LYNX_XPOSE", &k->xpose}, {"LYNX_UNROLL", &k->unroll}, {"LYNX_MOM", &k->mom},
      {"LYNX_MIN_TILES_PER_WG", &k->min_tiles_per_wg}, {"LYNX_ASYNC_BUILD", &k->async_build},
      {"LYNX_SIDE_REDUCE", &k->side_reduce}, {"LYNX_BUILD_HOST_WAIT", &k->build_host_wait},
      {"LYNX_GATHER_OVERLAP", &k->gather_overlap}, {"LYNX_LANES_BUILD_MIN_BATCH", &k->lanes_build_min_batch},
      {"LYNX_PIECE", &k->piece}, {"LYNX_PAIR_LEVELS_FUSED", &k->pair_levels_fused},
      {"LYNX_FUSE_MAX_CHUNKS", &k->fuse_max_chunks}, {"LYNX_MERGE_STEPS", &k->merge_steps},
      {"LYNX_REDUCE_WIDE", &k->reduce_wide}, {"LYNX_REDUCE_TICKET", &k->reduce_ticket}, {"LYNX_TRACK_UNITS", &k->track_units}, {"LYNX_ONE_ROUND", &k->one_round}, {"LYNX_UNIT_PAIRS", &k->unit_pairs}, {"LYNX_BWD_UNITS", &k->bwd_units},
      {"LYNX_BWD_MERGE", &k->bwd_merge}, {"LYNX_BWD_PAIRS", &k->bwd_pairs}, {"LYNX_BUILD_IN_TAIL", &k->build_in_tail},
      {"LYNX_SMALL_INLINE", &k->small_inline}, {"LYNX_INLINE_POOL", &k->inline_pool}, {"LYNX_ALTERNATE_ORDER", &k->alternate_order}, {"LYNX_HOST_VISIBLE_RECORDS", &k->host_visible_records}, {"LYNX_BWD_REUSE_TABLE", &k->bwd_reuse_table}};
  for (const auto& t : table) {
    const char* v = getenv(t.name);
    if (v && *v) *t.value = atoi(v);
  }
}
static inline int knob(int value, int dflt) { return value >= 0 ? value : dflt; }

struct lynx_ctx {
  Knobs knobs;
  int device = 0;
  hipStream_t stream = nullptr;
  // Second stream for k_build.  The map build of a `track` call depends on the lattice and the
  // incoming energy only, so it runs underneath the previous call's streaming kernel:
  //   s_build:  [wait: table slot free] k_build(n) -> table[n % kTableSlots]   record ev_built
  //   host   :  waits for ev_built (LYNX_BUILD_HOST_WAIT=0: the main stream does, with a barrier packet)
  //   stream :  k_track_direct(n)                                              ev_streamed rides on its dispatch
  // With three table slots the build of call n+1 may start as soon as the streaming kernel of call n-2 has finished,
  // i.e. it runs underneath kernel n-1 and the host, which waits for it, stays two calls ahead of the GPU.
  // Every build is followed at once by the main stream's wait on it, so anything enqueued on the
  // main stream later (copies, other kernels, lynx_sync) is ordered after every build so far.  The
  // build stream in turn waits for the main stream when (a) it reuses a table slot (ev_streamed) or
  // (b) the main stream may have written what the build reads (`main_dirty`, `main_wrote`).
  hipStream_t s_build = nullptr;
  static constexpr int kTableSlots = 3;
  hipEvent_t ev_built[kTableSlots] = {}, ev_streamed_own[kTableSlots] = {}, ev_mark = nullptr;
  hipEvent_t ev_streamed[kTableSlots] = {};  // the slot's current "streamed" event: its own, or a profiled launch's
  bool streamed_valid[kTableSlots] = {};
  unsigned seq = 0;
  bool main_dirty = false;            // unsynchronised device writes on the main stream (any buffer)
  // Nothing enqueued on the main stream since the host last waited for it.  Kept by the library itself (set by
  // sync_main, cleared by every entry point that enqueues): asking the runtime -- hipStreamQuery -- puts a marker
  // packet into the queue whose release costs the next kernel ~5 us behind a kernel that left dirty lines in L2.
  bool main_idle = true;
  bool plain_events = false;  // events with HIP's default (system-scope) fence: jobs of more than one rank
  unsigned long_calls = 0;    // streaming launches of 256 MB and more so far (TrackArgs::reversed)
  // The step table (and unit records) the latest forward call built, for a reverse pass that follows it directly on the
  // same lattice, incoming energy and merge form: it reads them instead of building its own (BASELINE config 5: 80 us
  // of 2.27 ms).  Good while no later call has taken a table slot (`seq`), the lattice has not been written to
  // (`version`) and no entry point that writes device memory on the caller's behalf has run since.
  struct FwdTable {
    const lynx_lattice* lat = nullptr;
    uint64_t version = 0;
    const void* energy = nullptr;
    int slot = -1;
    int merged = 0;
    bool units = false;
    size_t energy_bytes = 0;
    unsigned seq = 0;
    bool valid = false;
    // device memory [dst, dst + bytes) is about to be written on the caller's behalf
    void written(const void* dst, size_t bytes) {
      const char *a = (const char*)dst, *e = (const char*)energy;
      if (valid && a < e + energy_bytes && a + bytes > e) valid = false;
    }
  } fwd_table;
  const void* main_wrote = nullptr;   // energy buffer the last streaming kernel published on the main stream
  std::mutex mu;                      // allocator maps: finalizers may run on other threads
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // host-mapped status words the kernels can raise a flag in (k_cavity_flags: energy <= 0 at a cavity); the host
  // looks at them whenever it has waited for the GPU anyway
  int32_t* h_status = nullptr;
  int32_t* d_status = nullptr;
  // "streaming kernel k has reached its tail" (TrackArgs.tail_*): the word the kernels write, and how many have been
  // launched with it
  unsigned int* d_tail_flag = nullptr;
  uint32_t tail_seq = 0;
  int can_wait_value = 0;
  std::string err;
  hipDeviceProp_t prop;
  // caching allocator: size class -> free blocks; live pointer -> size class
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> live;
  std::vector<std::pair<const char*, size_t>> host_ranges;  // the host-visible blocks there are (live or cached)
  // internal scratch (grown on demand, stream-ordered reuse)
  void* scratch_level = nullptr;  // second level of the moment reduction (long beams)
  size_t scratch_level_bytes = 0;
  unsigned int* scratch_tickets = nullptr;  // ... and its per-sample tickets (k_reduce_moments_ticket): zero between launches
  size_t scratch_tickets_count = 0;
  void* scratch_obs = nullptr;  // per-workgroup sums of x, y at the observers [B][chunks][2 * LYNX_MAX_OBSERVERS]
  size_t scratch_obs_bytes = 0;
  void* scratch_erun = nullptr;      // k_cavity_flags: every sample's energy on its way through the cavities
  size_t scratch_erun_bytes = 0;
  void* scratch_esteps = nullptr;    // k_cavity_flags_spec -> lanes build: every sample's energy behind 0, 1, ... step cavities [k][Bp]
  size_t scratch_esteps_bytes = 0;
  int32_t* d_spec_valid = nullptr;   // ... and whether they hold (k_cavity_flags)
  void* scratch_products = nullptr;  // lanes = samples build: piece / pair products [slot][49][Bp] float64
  size_t scratch_products_bytes = 0;
  void* scratch_coefs = nullptr;     // ... and cavity coefficients [S][8][Bp]
  size_t scratch_coefs_bytes = 0;
  static constexpr int kTableBwd = kTableSlots, kTablePb = kTableSlots + 1;
  void* scratch_steps[kTableSlots + 2] = {};  // the ring of step tables, the reverse pass's own, the ParameterBeam lanes path's
  size_t scratch_steps_bytes[kTableSlots + 2] = {};
  void* scratch_units_bwd[2] = {nullptr, nullptr};  // ... and of the reverse pass's own table
  size_t scratch_units_bwd_bytes[2] = {0, 0};
  void* scratch_units[2 * kTableSlots] = {};  // compact unit records of multi-step float32 programs (lynx_units.hpp) and their class-D extras, per table slot
  size_t scratch_units_bytes[2 * kTableSlots] = {};
  void* scratch_grad[3] = {nullptr, nullptr, nullptr};  // backward: partials, T_bar, build scratch
  size_t scratch_grad_bytes[3] = {0, 0, 0};
  ncclComm_t comm = nullptr;
  int comm_ranks = 0;
  // The SIDE stream: what follows a streaming kernel without anything on this GPU waiting for it -- the reduction of
  // the workgroups' moment records and, with a communicator, the RCCL gather of the result -- runs here, underneath
  // the next call's streaming kernel, so that the main stream carries streaming kernels back to back.
  //   * a side operation starts behind the stop event that rides on its streaming kernel's dispatch (or a marker);
  //   * nothing on the main stream waits for it by itself: entry points that read moment records on the device
  //     join first (`join_side`), the host sees results through lynx_buf_d2h / lynx_sync, which wait for this stream;
  //   * the blocks it touches are kept out of the allocator until its "done" event has fired (`side_ops`);
  //   * the workgroups' records go through a ring of kPartialRing buffers; the host checks (and, rarely, waits)
  //     that a buffer's last reduction is done before a streaming kernel is given it again.
  hipStream_t s_side = nullptr;
  hipEvent_t ev_side_in = nullptr;    // side -> main (join_side)
  hipEvent_t ev_main_mark = nullptr;  // main -> side (a side operation whose input the main stream produces)
  struct SideOp {
    hipEvent_t done;
    const void* a;  // blocks the operation reads or writes (null: none)
    void* b;
    bool a_freed, b_freed;  // lynx_buf_free came while it was in flight: released when it is done
    bool owned;             // `done` goes back to side_events when the operation retires (else it belongs to a ring slot)
  };
  std::vector<SideOp> side_ops;         // in flight, oldest first (one stream: they finish in order)
  std::vector<hipEvent_t> side_events;  // idle "done" events
  bool side_busy = false;               // s_side may still be writing a block: readers wait for it
  bool level_on_side = false;           // which stream the users of scratch_level ran on last
  const void* side_wrote = nullptr;     // moment block the last side reduction writes (a gather of it needs no marker)
  static constexpr int kPartialRing = 4;
  struct PartialSlot {
    void* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t track_done = nullptr;  // rides on the streaming kernel's dispatch
    hipEvent_t reduced = nullptr;     // recorded on s_side behind the reduction
    bool pending = false;             // `reduced` has been recorded and not yet seen complete
  } partial_ring[kPartialRing];
  unsigned partial_seq = 0;
  // per-launch profiling of k_track (lynx_profile_begin / _end)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<double> prof_ms;  // every launch of the last finished profile, in launch order
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_gather_events;  // ... and every RCCL gather (on the stream it ran on)
  std::vector<double> prof_gather_ms;
  hipEvent_t last_stream_stop = nullptr;
};

struct lynx_lattice {
  lynx_ctx* ctx = nullptr;
  int dtype = LYNX_F32;
  int64_t batch = 0;
  int32_t n_elems = 0, n_steps = 0;
  int32_t n_observers = 0;  // steps with LYNX_STEP_FLAG_OBSERVE (kept current by lynx_lattice_set_flags)
  bool has_cavity = false;  // any cavity element: its whole-batch predicates are evaluated on the device
  int32_t n_cavities = 0;
  int32_t* d_cav_words = nullptr;  // one word per cavity: the predicates OR-ed over the batch (k_cavity_flags_spec); kept zero between calls
  int2* d_cavs = nullptr;          // the cavities in lattice order: (element, its step if it is one of its own else -1)
  int32_t* d_cav_before = nullptr;  // [S + 1]: step cavities in front of step s (StepEnergies, lynx_device.hpp)
  int32_t n_step_cavities = 0;
  int64_t pool_count = 0;
  std::vector<lynx_elem> h_elems;
  std::vector<lynx_step> h_steps;
  lynx_elem* d_elems = nullptr;
  lynx_step* d_steps = nullptr;
  int32_t* d_elem_step = nullptr;
  void* d_pool = nullptr;
  // lanes = samples build (large batches): pieces, the levels of the pairwise tree over them, and
  // where each step's product ends up -- planned once from the step structure
  int plan_piece_len = 0;
  int32_t n_pieces = 0, n_slots = 0;
  std::vector<std::pair<int32_t, int32_t>> levels;  // (first task, number of tasks) per level
  BuildPiece* d_pieces = nullptr;
  PairTask* d_tasks = nullptr;
  int32_t* d_step_slot = nullptr;
  // multi-step float32 programs as units (lynx_units.hpp), one plan per value of `merged` (the forward and the reverse
  // pass may disagree about it, and a training loop that alternates between them must not re-plan every call):
  // whether it has been made since the flags last changed, whether the program fits it, and what k_emit_steps is told
  // about every step
  UnitPlan units[2];
  bool units_made[2] = {false, false};
  bool units_ok[2] = {false, false};
  int32_t* d_step_unit[2] = {nullptr, nullptr};
  // the reverse pass's (element, parameter) tasks, kind by kind (lynx_grad.hpp: bwd_kind_params), made on first use
  unsigned short* d_bwd_tasks = nullptr;
  int32_t n_bwd_tasks = 0;
  // Small lattices without cavities: the parameter pool travels by value in the arguments of the one-workgroup-per-
  // sample kernels (InlinePool, lynx_device.hpp).  A parameter write is a memcpy on the host; d_pool is brought up to
  // date when a kernel that reads it is next launched (sync_pool).
  uint64_t version = 0;  // parameter and flag writes so far
  bool prog_pool = false;
  bool pool_stale = false;
  InlinePool<kInlinePoolLarge> h_pool;
};

static thread_local std::string g_err;

static int fail(lynx_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail((ctx), LYNX_ERR_HIP,                                                      \
                  std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +   \
                      std::to_string(__LINE__) + ")");                                      \
  } while (0)

#define NCCL_TRY(ctx, expr)                                                                 \
  do {                                                                                      \
    ncclResult_t _e = (expr);                                                               \
    if (_e != ncclSuccess)                                                                  \
      return fail((ctx), LYNX_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(_e)); \
  } while (0)

// The context's device as this thread's current one.  hipGetDevice reads a thread-local of the runtime; hipSetDevice,
// which every entry point used to call, takes the runtime's lock -- once per thread is enough.
static hipError_t use_device(lynx_ctx* ctx) {
  int current = -1;
  if (hipGetDevice(&current) == hipSuccess && current == ctx->device) return hipSuccess;
  return hipSetDevice(ctx->device);
}

// Events between the context's own streams, and the time stamps around a profiled launch, are recorded WITHOUT the
// system-scope fence HIP puts behind an event by default: that fence writes back and invalidates for the benefit of
// the host and of other devices, which nothing that waits on these events needs -- a kernel's own end-of-kernel release
// is what makes its results visible to the other queues of this device, host read-backs go through a copy and a stream
// synchronisation of their own, and RCCL fences what it sends.  Same box, alternating (scripts/gpu/r4/evscope.sh):
// BASELINE config 3 49.7-50.7 -> 47.0 us/step, the 128-sample shard of config 4 139.8-140.8 -> 135.8-136.2, config 3
// at 8 M particles 170.8-174.8 -> 168.5; hipEventReleaseToDevice instead: no change.
// With MORE THAN ONE RANK the events keep HIP's default: there a recorded event may also stand between RCCL's traffic from
// other devices and whoever reads it next, and none of that has run on hardware the builder had (lynx_ctx::plain_events).
static unsigned sync_event_flags(const lynx_ctx* ctx) {
  return hipEventDisableTiming | (ctx->plain_events ? 0u : (unsigned)hipEventDisableSystemFence);
}
static unsigned timing_event_flags(const lynx_ctx* ctx) {
  return hipEventDefault | (ctx->plain_events ? 0u : (unsigned)hipEventDisableSystemFence);
}

static hipError_t sync_main(lynx_ctx* ctx) {
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) ctx->main_idle = true;
  return e;
}

static size_t dtype_size(int dtype) { return dtype == LYNX_F64 ? 8 : 4; }

static size_t size_class(size_t bytes) {
  if (bytes == 0) bytes = 1;
  const size_t g = bytes >= (1u << 20) ? (1u << 20) : 256;
  return (bytes + g - 1) / g * g;
}

// Small result blocks the host reads right after the call (a sample's moment record: 288 bytes) can live in HOST memory
// the GPU writes through: reading them back then is a wait for the stream and a memcpy, not a copy command with a
// round trip of its own (BASELINE config 2 with sigma_x read after every call: NOTES.md).  Same allocator, another
// size-class name space (kHostVisible set in the class); `is_host_visible` tells lynx_buf_d2h and friends.
constexpr size_t kHostVisible = (size_t)1 << 62;
constexpr size_t kHostVisibleMax = 4096;

// caller holds ctx->mu
static void free_block(lynx_ctx* ctx, void* p, size_t size_class_key) {
  if (size_class_key & kHostVisible) {
    for (size_t i = 0; i < ctx->host_ranges.size(); ++i)
      if (ctx->host_ranges[i].first == (const char*)p) {
        ctx->host_ranges.erase(ctx->host_ranges.begin() + (long)i);
        break;
      }
    (void)hipHostFree(p);
  } else {
    (void)hipFree(p);
  }
}

static bool is_host_visible(lynx_ctx* ctx, const void* p) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  for (const auto& r : ctx->host_ranges)
    if ((const char*)p >= r.first && (const char*)p < r.first + r.second) return true;
  return false;
}

static int ctx_alloc_host_visible(lynx_ctx* ctx, size_t bytes, void** out) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  const size_t sc = size_class(bytes) | kHostVisible;
  auto it = ctx->free_blocks.find(sc);
  if (it != ctx->free_blocks.end()) {
    *out = it->second;
    ctx->free_blocks.erase(it);
    ctx->live[*out] = sc;
    return LYNX_OK;
  }
  void *p = nullptr, *d = nullptr;
  hipError_t e = hipHostMalloc(&p, sc & ~kHostVisible, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&d, p, 0);
  if (e != hipSuccess || d != p) {  // (one address for both sides is what the rest of the library assumes)
    if (p) (void)hipHostFree(p);
    (void)hipGetLastError();
    return fail(ctx, LYNX_ERR_NOMEM, std::string("hipHostMalloc (host-visible block): ") + hipGetErrorString(e));
  }
  ctx->live[p] = sc;
  ctx->host_ranges.emplace_back((const char*)p, sc & ~kHostVisible);
  *out = p;
  return LYNX_OK;
}

static int ctx_alloc(lynx_ctx* ctx, size_t bytes, void** out) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  const size_t sc = size_class(bytes);
  auto it = ctx->free_blocks.find(sc);
  if (it != ctx->free_blocks.end()) {
    *out = it->second;
    ctx->free_blocks.erase(it);
    ctx->live[*out] = sc;
    return LYNX_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, sc);
  if (e != hipSuccess) {
    // give cached blocks back and retry once
    for (auto& kv : ctx->free_blocks) free_block(ctx, kv.second, kv.first);
    ctx->free_blocks.clear();
    e = hipMalloc(&p, sc);
    if (e != hipSuccess)
      return fail(ctx, LYNX_ERR_NOMEM, "hipMalloc(" + std::to_string(sc) + "): " + hipGetErrorString(e));
  }
  ctx->live[p] = sc;
  *out = p;
  return LYNX_OK;
}

// caller holds ctx->mu
static void release_block(lynx_ctx* ctx, void* p) {
  auto it = ctx->live.find(p);
  if (it == ctx->live.end()) return;
  ctx->free_blocks.emplace(it->second, p);
  ctx->live.erase(it);
}

// Side operations whose "done" event has fired leave the in-flight list (oldest first: s_side runs them in order);
// blocks that were freed meanwhile go back to the allocator now.  `wait`: block on every one of them.  Caller holds ctx->mu.
static void retire_side_ops(lynx_ctx* ctx, bool wait) {
  // a freed block goes back only with the LAST operation in flight that touches it (a reduction and the gather of
  // its result name the same block): an earlier one hands its mark on
  auto hand_on_or_release = [ctx](const void* p) {
    for (size_t k = 1; k < ctx->side_ops.size(); ++k) {
      lynx_ctx::SideOp& later = ctx->side_ops[k];
      if (later.a == p) { later.a_freed = true; return; }
      if (later.b == p) { later.b_freed = true; return; }
    }
    release_block(ctx, const_cast<void*>(p));
  };
  while (!ctx->side_ops.empty()) {
    lynx_ctx::SideOp& g = ctx->side_ops.front();
    if (wait) (void)hipEventSynchronize(g.done);
    else if (hipEventQuery(g.done) != hipSuccess) break;
    if (g.a_freed) hand_on_or_release(g.a);
    if (g.b_freed && g.b != g.a) hand_on_or_release(g.b);
    if (g.owned) ctx->side_events.push_back(g.done);
    ctx->side_ops.erase(ctx->side_ops.begin());
  }
}

static int ctx_free(lynx_ctx* ctx, void* p) {
  if (!p) return LYNX_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (ctx->live.find(p) == ctx->live.end()) return fail(ctx, LYNX_ERR_INVALID, "lynx_buf_free: unknown pointer");
  if (!ctx->side_ops.empty()) {
    retire_side_ops(ctx, false);
    // a block an operation on the side stream still reads or writes: stream order on the main stream does not
    // protect it, so it stays out of the allocator until that operation is done
    bool held = false;
    for (auto& g : ctx->side_ops) {
      if (g.a == p) { g.a_freed = true; held = true; }
      if (g.b == p) { g.b_freed = true; held = true; }
    }
    if (held) return LYNX_OK;
  }
  release_block(ctx, p);
  return LYNX_OK;
}

static int ensure_scratch(lynx_ctx* ctx, void** buf, size_t* have, size_t need) {
  if (*have >= need) return LYNX_OK;
  if (*buf) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_build));
    HIP_TRY(ctx, sync_main(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_side));
    HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  const size_t sz = size_class(need);
  HIP_TRY(ctx, hipMalloc(buf, sz));
  *have = sz;
  return LYNX_OK;
}

// per-sample tickets of k_reduce_moments_ticket, zeroed when they are (re)allocated; the kernel leaves them zero
static int ensure_tickets(lynx_ctx* ctx, size_t count) {
  if (ctx->scratch_tickets_count >= count) return LYNX_OK;
  void* buf = ctx->scratch_tickets;
  size_t have = ctx->scratch_tickets_count * sizeof(unsigned int);
  const int rc = ensure_scratch(ctx, &buf, &have, std::max<size_t>(count, 1024) * sizeof(unsigned int));
  if (rc) return rc;
  ctx->scratch_tickets = (unsigned int*)buf;
  ctx->scratch_tickets_count = have / sizeof(unsigned int);
  // (hipMemset runs on the null stream and need not have finished when it returns; the context's streams do not wait for
  // that stream -- a reduction launched right behind this took tickets that the fill then wiped: one moment record of
  // zeros in the first call of a process, once in about thirty processes)
  HIP_TRY(ctx, hipMemsetAsync(buf, 0, have, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));
  return LYNX_OK;
}

// The API functions below get C linkage from their declarations in include/lynx_hip.h.

const char* lynx_version(void) { return "lynxhip 0.1.0 (gfx950)"; }

int lynx_device_count(int* count) {
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, LYNX_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return LYNX_OK;
}

// the context's own inter-stream events (again, with other flags, when a communicator of more than one rank arrives)
static int make_sync_events(lynx_ctx* ctx) {
  const auto make = [&](hipEvent_t* e) -> hipError_t {
    if (*e) (void)hipEventDestroy(*e);
    *e = nullptr;
    return hipEventCreateWithFlags(e, sync_event_flags(ctx));
  };
  HIP_TRY(ctx, make(&ctx->ev_side_in));
  HIP_TRY(ctx, make(&ctx->ev_main_mark));
  for (auto& slot : ctx->partial_ring) {
    HIP_TRY(ctx, make(&slot.track_done));
    HIP_TRY(ctx, make(&slot.reduced));
    slot.pending = false;
  }
  for (int i = 0; i < lynx_ctx::kTableSlots; ++i) {
    HIP_TRY(ctx, make(&ctx->ev_built[i]));
    HIP_TRY(ctx, make(&ctx->ev_streamed_own[i]));
    ctx->streamed_valid[i] = false;
  }
  HIP_TRY(ctx, make(&ctx->ev_mark));
  for (hipEvent_t e : ctx->side_events) (void)hipEventDestroy(e);  // idle "done" events: made again on demand
  ctx->side_events.clear();
  return LYNX_OK;
}

int lynx_ctx_create(int device, lynx_ctx** out) {
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, LYNX_ERR_HIP,
                std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
  if (device < 0 || device >= n) return fail(nullptr, LYNX_ERR_INVALID, "device ordinal out of range");
  lynx_ctx* ctx = new lynx_ctx();
  load_knobs(&ctx->knobs);
  ctx->device = device;
  HIP_TRY(nullptr, hipSetDevice(device));
  HIP_TRY(nullptr, hipGetDeviceProperties(&ctx->prop, device));
  HIP_TRY(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  {
    // the build is short and the next streaming kernel waits for it: let its workgroups go first
    int prio_low = 0, prio_high = 0;
    HIP_TRY(nullptr, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    HIP_TRY(nullptr, hipStreamCreateWithPriority(&ctx->s_build, hipStreamNonBlocking, prio_high));
    HIP_TRY(nullptr, hipStreamCreateWithPriority(&ctx->s_side, hipStreamNonBlocking, prio_high));
  }
  {
    const char* world = getenv("WORLD_SIZE");  // (the launcher's; lynx_comm_init looks again)
    const char* plain = getenv("LYNX_PLAIN_EVENTS");  // (one rank's share of a job rehearsed in a process of its own)
    ctx->plain_events = (world && atoi(world) > 1) || (plain && atoi(plain) != 0);
    const int rc = make_sync_events(ctx);
    if (rc) return rc;
  }
  HIP_TRY(nullptr, hipEventCreate(&ctx->ev_start));
  HIP_TRY(nullptr, hipEventCreate(&ctx->ev_stop));
  HIP_TRY(nullptr, hipHostMalloc((void**)&ctx->h_status, 2 * sizeof(int32_t), hipHostMallocMapped));
  ctx->h_status[0] = ctx->h_status[1] = 0;
  HIP_TRY(nullptr, hipHostGetDevicePointer((void**)&ctx->d_status, ctx->h_status, 0));
  HIP_TRY(nullptr, hipMalloc((void**)&ctx->d_spec_valid, sizeof(int32_t)));
  HIP_TRY(nullptr, hipMemset(ctx->d_spec_valid, 0, sizeof(int32_t)));
  // the word hipStreamWaitValue32 polls: signal memory (what HIP documents for it); without it the build simply
  // starts at the head of the streaming kernel instead of in its tail
  if (hipDeviceGetAttribute(&ctx->can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) ctx->can_wait_value = 0;
  if (ctx->can_wait_value &&
      hipExtMallocWithFlags((void**)&ctx->d_tail_flag, 8, hipMallocSignalMemory) != hipSuccess) {
    (void)hipGetLastError();
    ctx->d_tail_flag = nullptr;
    ctx->can_wait_value = 0;
  }
  if (ctx->d_tail_flag) HIP_TRY(nullptr, hipMemset(ctx->d_tail_flag, 0, 8));
  HIP_TRY(nullptr, hipDeviceSynchronize());  // (hipMemset runs on the null stream and need not have finished when it returns)
  *out = ctx;
  return LYNX_OK;
}

int lynx_ctx_destroy(lynx_ctx* ctx) {
  if (!ctx) return LYNX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->s_build);
  (void)sync_main(ctx);
  (void)hipStreamSynchronize(ctx->s_side);
  if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    retire_side_ops(ctx, true);
  }
  for (auto* list : {&ctx->prof_events, &ctx->prof_gather_events})
    for (auto& pr : *list) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  (void)hipEventDestroy(ctx->ev_side_in);
  (void)hipEventDestroy(ctx->ev_main_mark);
  for (hipEvent_t e : ctx->side_events) (void)hipEventDestroy(e);
  for (auto& slot : ctx->partial_ring) {
    (void)hipEventDestroy(slot.track_done);
    (void)hipEventDestroy(slot.reduced);
    if (slot.buf) (void)hipFree(slot.buf);
  }
  (void)hipStreamDestroy(ctx->s_side);
  for (auto& kv : ctx->free_blocks) free_block(ctx, kv.second, kv.first);
  for (auto& kv : ctx->live) free_block(ctx, kv.first, kv.second);
  if (ctx->scratch_level) (void)hipFree(ctx->scratch_level);
  if (ctx->scratch_tickets) (void)hipFree(ctx->scratch_tickets);
  if (ctx->scratch_obs) (void)hipFree(ctx->scratch_obs);
  if (ctx->scratch_erun) (void)hipFree(ctx->scratch_erun);
  if (ctx->scratch_esteps) (void)hipFree(ctx->scratch_esteps);
  if (ctx->d_spec_valid) (void)hipFree(ctx->d_spec_valid);
  if (ctx->scratch_products) (void)hipFree(ctx->scratch_products);
  if (ctx->scratch_coefs) (void)hipFree(ctx->scratch_coefs);
  for (int i = 0; i < lynx_ctx::kTableSlots + 2; ++i)
    if (ctx->scratch_steps[i]) (void)hipFree(ctx->scratch_steps[i]);
  for (int i = 0; i < 2 * lynx_ctx::kTableSlots; ++i)
    if (ctx->scratch_units[i]) (void)hipFree(ctx->scratch_units[i]);
  for (int i = 0; i < 2; ++i)
    if (ctx->scratch_units_bwd[i]) (void)hipFree(ctx->scratch_units_bwd[i]);
  for (int i = 0; i < 3; ++i)
    if (ctx->scratch_grad[i]) (void)hipFree(ctx->scratch_grad[i]);
  for (int i = 0; i < lynx_ctx::kTableSlots; ++i) {
    (void)hipEventDestroy(ctx->ev_built[i]);
    (void)hipEventDestroy(ctx->ev_streamed_own[i]);
  }
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  if (ctx->d_tail_flag) (void)hipFree(ctx->d_tail_flag);
  (void)hipEventDestroy(ctx->ev_mark);
  (void)hipEventDestroy(ctx->ev_start);
  (void)hipEventDestroy(ctx->ev_stop);
  (void)hipStreamDestroy(ctx->s_build);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return LYNX_OK;
}

const char* lynx_last_error(lynx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int lynx_device_info(lynx_ctx* ctx, lynx_device_info_t* out) {
  memset(out, 0, sizeof(*out));
  snprintf(out->arch, sizeof(out->arch), "%s", ctx->prop.gcnArchName);
  // some driver stacks report no marketing name: the architecture then stands in for it
  if (ctx->prop.name[0]) snprintf(out->name, sizeof(out->name), "%s", ctx->prop.name);
  else snprintf(out->name, sizeof(out->name), "AMD GPU (%s)", ctx->prop.gcnArchName);
  out->compute_units = ctx->prop.multiProcessorCount;
  out->lds_bytes_per_cu = (int32_t)ctx->prop.maxSharedMemoryPerMultiProcessor;
  out->hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
  return LYNX_OK;
}

// s_side is the one stream whose work nothing on the main stream waits for by itself.
// Host readers (lynx_buf_d2h, lynx_sync): wait until it has drained.
static int wait_for_side(lynx_ctx* ctx) {
  if (ctx->side_busy) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_side));
    ctx->side_busy = false;
    std::lock_guard<std::mutex> lock(ctx->mu);
    retire_side_ops(ctx, true);
    for (auto& slot : ctx->partial_ring) slot.pending = false;
  }
  return LYNX_OK;
}

// a fresh or recycled "done" event for a side operation
static int side_event(lynx_ctx* ctx, hipEvent_t* out) {
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    retire_side_ops(ctx, false);
    // bounded backlog: the host blocks on the oldest operation rather than run arbitrarily far ahead
    while (ctx->side_ops.size() >= 32) {
      (void)hipEventSynchronize(ctx->side_ops.front().done);
      retire_side_ops(ctx, false);
    }
    if (!ctx->side_events.empty()) {
      *out = ctx->side_events.back();
      ctx->side_events.pop_back();
    }
  }
  if (!*out) HIP_TRY(ctx, hipEventCreateWithFlags(out, sync_event_flags(ctx)));
  return LYNX_OK;
}

// `done` has just been recorded on s_side behind an operation that touches blocks a and b
static void side_op_issued(lynx_ctx* ctx, hipEvent_t done, const void* a, void* b, bool owned = true) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->side_ops.push_back(lynx_ctx::SideOp{done, a, b, false, false, owned});
  ctx->side_busy = true;
}

// What the kernels flagged since the last look; the caller has just waited for the GPU.
static int check_status(lynx_ctx* ctx) {
  if (ctx->h_status && __atomic_load_n(&ctx->h_status[0], __ATOMIC_ACQUIRE) != 0) {
    const int elem = __atomic_load_n(&ctx->h_status[1], __ATOMIC_ACQUIRE);
    __atomic_store_n(&ctx->h_status[0], 0, __ATOMIC_RELEASE);
    __atomic_store_n(&ctx->h_status[1], 0, __ATOMIC_RELEASE);
    return fail(ctx, LYNX_ERR_ENERGY,
                "Initial energy must be larger than 0 (a beam reached the cavity at element " + std::to_string(elem) +
                    " of its program with energy <= 0 or NaN; lynx/accelerator/cavity.py:260)");
  }
  return LYNX_OK;
}

// Device readers on the main stream (anything that may be handed a moment record or a gathered block): the main
// stream waits for what the side stream has been given so far.
static int join_side(lynx_ctx* ctx) {
  if (ctx->side_busy) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_side_in, ctx->s_side));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side_in, 0));
  }
  return LYNX_OK;
}

int lynx_sync(lynx_ctx* ctx) {
  HIP_TRY(ctx, use_device(ctx));
  HIP_TRY(ctx, sync_main(ctx));
  const int rc = wait_for_side(ctx);
  return rc ? rc : check_status(ctx);
}

int lynx_timer_start(lynx_ctx* ctx) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  return LYNX_OK;
}

int lynx_timer_stop(lynx_ctx* ctx, float* elapsed_ms) {
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
  ctx->main_idle = true;
  HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
  return LYNX_OK;
}

int lynx_profile_begin(lynx_ctx* ctx) {
  // events of an earlier, unfinished profile may still stand in for step-table slots' "streamed" events: nothing may
  // wait on them once they are destroyed
  if (!ctx->prof_events.empty()) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_build));
    HIP_TRY(ctx, sync_main(ctx));
    for (bool& v : ctx->streamed_valid) v = false;
    ctx->last_stream_stop = nullptr;
  }
  for (auto& pr : ctx->prof_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  ctx->prof_events.clear();
  for (auto& pr : ctx->prof_gather_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  ctx->prof_gather_events.clear();
  ctx->profiling = true;
  return LYNX_OK;
}

int lynx_profile_end(lynx_ctx* ctx, double* total_ms, int64_t* launches) {
  ctx->profiling = false;
  HIP_TRY(ctx, sync_main(ctx));
  // every build is followed by its streaming kernel on the main stream, so both streams are idle now;
  // the per-launch stop events stood in for the step-table slots' "streamed" events and go away here
  for (bool& v : ctx->streamed_valid) v = false;
  double total = 0.0;
  ctx->prof_ms.clear();
  for (auto& pr : ctx->prof_events) {
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
    ctx->prof_ms.push_back(ms);
    total += ms;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  *total_ms = total;
  *launches = (int64_t)ctx->prof_events.size();
  ctx->prof_events.clear();
  ctx->prof_gather_ms.clear();
  if (!ctx->prof_gather_events.empty()) {
    const int rc = wait_for_side(ctx);  // the gathers' stop events must have fired
    if (rc) return rc;
  }
  for (auto& pr : ctx->prof_gather_events) {
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
    ctx->prof_gather_ms.push_back(ms);
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  ctx->prof_gather_events.clear();
  return LYNX_OK;
}

int lynx_profile_launches(lynx_ctx* ctx, double* ms_out, int64_t capacity, int64_t* launches) {
  if (!ctx || !launches || (capacity > 0 && !ms_out)) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  *launches = (int64_t)ctx->prof_ms.size();
  for (int64_t i = 0; i < capacity && i < *launches; ++i) ms_out[i] = ctx->prof_ms[(size_t)i];
  return LYNX_OK;
}

int lynx_profile_gathers(lynx_ctx* ctx, double* ms_out, int64_t capacity, int64_t* gathers) {
  if (!ctx || !gathers || (capacity > 0 && !ms_out)) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  *gathers = (int64_t)ctx->prof_gather_ms.size();
  for (int64_t i = 0; i < capacity && i < *gathers; ++i) ms_out[i] = ctx->prof_gather_ms[(size_t)i];
  return LYNX_OK;
}

int lynx_ctx_reload_knobs(lynx_ctx* ctx) {
  if (!ctx) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  // plans made under the old settings (lanes-build pieces, unit plans) are keyed by what they depend on and are
  // re-made on demand; nothing in flight reads the knobs
  load_knobs(&ctx->knobs);
  return LYNX_OK;
}

int lynx_buf_alloc(lynx_ctx* ctx, size_t bytes, void** d_out) {
  HIP_TRY(ctx, use_device(ctx));
  return ctx_alloc(ctx, bytes, d_out);
}

// A block for a RESULT the host is going to read back (a ParameterBeam's outgoing mu and cov, a moment record): small
// ones come from host memory the GPU writes through, so that lynx_buf_d2h is a wait and a memcpy (ctx_alloc_host_visible);
// large ones, and all of them once the context has a communicator, are device memory like any other block.
int lynx_buf_alloc_result(lynx_ctx* ctx, size_t bytes, void** d_out) {
  HIP_TRY(ctx, use_device(ctx));
  if (bytes <= kHostVisibleMax && ctx->comm_ranks == 0 && ctx->knobs.host_visible_records) return ctx_alloc_host_visible(ctx, bytes, d_out);
  return ctx_alloc(ctx, bytes, d_out);
}

int lynx_buf_free(lynx_ctx* ctx, void* d_ptr) {
  if (ctx && d_ptr) ctx->fwd_table.written(d_ptr, 1);  // (a freed energy block may come back with other contents)
  return ctx_free(ctx, d_ptr);
}

int lynx_buf_h2d(lynx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (bytes == 0) return LYNX_OK;
  ctx->fwd_table.written(d_dst, bytes);
  if (is_host_visible(ctx, d_dst)) {  // host memory the GPU reads through: wait for whoever still uses it, then write
    const int rc = wait_for_side(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, sync_main(ctx));
    memcpy(d_dst, h_src, bytes);
    return LYNX_OK;
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));  // h_src is pageable and may be freed
  return LYNX_OK;
}

int lynx_buf_d2h(lynx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return LYNX_OK;
  {
    const int rc = wait_for_side(ctx);  // the block may be a moment record or a gathered block
    if (rc) return rc;
  }
  if (is_host_visible(ctx, d_src)) {  // the GPU wrote it through to host memory: no copy command, just the wait
    HIP_TRY(ctx, sync_main(ctx));
    memcpy(h_dst, d_src, bytes);
    return check_status(ctx);
  }
  HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));
  return check_status(ctx);  // what came back may be the result of a program that met a cavity with energy <= 0
}

int lynx_buf_d2d(lynx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (bytes == 0) return LYNX_OK;
  {
    const int rc = join_side(ctx);  // the source may be a moment record the side stream is still reducing
    if (rc) return rc;
  }
  ctx->fwd_table.written(d_dst, bytes);
  // (either side may be a host-visible block: the runtime works the direction out from the addresses)
  HIP_TRY(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDefault, ctx->stream));
  ctx->main_dirty = true;
  return LYNX_OK;
}

int lynx_buf_memset(lynx_ctx* ctx, void* d_dst, int value, size_t bytes) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (bytes == 0) return LYNX_OK;
  {
    const int rc = join_side(ctx);
    if (rc) return rc;
  }
  ctx->fwd_table.written(d_dst, bytes);
  HIP_TRY(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  ctx->main_dirty = true;
  return LYNX_OK;
}

int lynx_pool_trim(lynx_ctx* ctx) {
  HIP_TRY(ctx, hipStreamSynchronize(ctx->s_build));
  HIP_TRY(ctx, sync_main(ctx));
  {
    const int rc = wait_for_side(ctx);
    if (rc) return rc;
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
  for (auto& kv : ctx->free_blocks) free_block(ctx, kv.second, kv.first);
  ctx->free_blocks.clear();
  return LYNX_OK;
}

// ---- lattice ---------------------------------------------------------------------------

// observers must be single-identity-element run steps, at most LYNX_MAX_OBSERVERS of them
static int count_observers(lynx_ctx* ctx, lynx_lattice* lat) {
  int32_t n = 0;
  for (int32_t s = 0; s < lat->n_steps; ++s) {
    const lynx_step& st = lat->h_steps[s];
    if (!(st.flags & LYNX_STEP_FLAG_OBSERVE)) continue;
    if (st.kind != LYNX_STEP_RUN || st.last != st.first + 1 || lat->h_elems[st.first].kind != LYNX_KIND_IDENTITY)
      return fail(ctx, LYNX_ERR_INVALID, "an observer step must be a run of exactly one identity element");
    ++n;
  }
  if (n > LYNX_MAX_OBSERVERS) return fail(ctx, LYNX_ERR_INVALID, "too many observer steps in one program");
  lat->n_observers = n;
  return LYNX_OK;
}

static int params_of_kind(int kind) {
  switch (kind) {
    case LYNX_KIND_IDENTITY: return 0;
    case LYNX_KIND_DRIFT: return 1;
    case LYNX_KIND_QUADRUPOLE: return 5;
    case LYNX_KIND_DIPOLE: return 8;
    case LYNX_KIND_HCOR:
    case LYNX_KIND_VCOR: return 2;
    case LYNX_KIND_CAVITY: return 4;
    case LYNX_KIND_CUSTOM: return 49;
    case LYNX_KIND_BASE_RMATRIX: return 4;
    case LYNX_KIND_ROTATION: return 1;
    case LYNX_KIND_MISALIGNMENT: return 3;
    case LYNX_KIND_SOLENOID: return 4;
    case LYNX_KIND_UNDULATOR: return 1;
    default: return -1;
  }
}

// d_pool as current as the pool that travels in the kernel arguments: before any kernel that reads the parameters from memory
static int sync_pool(lynx_ctx* ctx, lynx_lattice* lat) {
  if (!lat->pool_stale) return LYNX_OK;
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_pool, lat->h_pool.q, (size_t)lat->pool_count * dtype_size(lat->dtype), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));
  lat->pool_stale = false;
  return LYNX_OK;
}

int lynx_lattice_create(lynx_ctx* ctx, int dtype, int64_t batch, int32_t n_elems,
                        const lynx_elem* elems, int32_t n_steps, const lynx_step* steps,
                        const void* pool, int64_t pool_count, lynx_lattice** out) {
  *out = nullptr;
  if (dtype != LYNX_F32 && dtype != LYNX_F64) return fail(ctx, LYNX_ERR_INVALID, "bad dtype");
  if (batch <= 0 || n_elems < 0 || n_steps < 0 || pool_count < 0)
    return fail(ctx, LYNX_ERR_INVALID, "bad lattice sizes");
  // Validate every index the kernels will dereference: a bad table must never reach the GPU.
  std::vector<int32_t> elem_step(n_elems, -1);
  for (int32_t e = 0; e < n_elems; ++e) {
    const int np = params_of_kind(elems[e].kind);
    if (np < 0) return fail(ctx, LYNX_ERR_INVALID, "element " + std::to_string(e) + ": unknown kind");
    const int64_t bs = elems[e].batch_stride;
    if (bs != 0 && bs < np) return fail(ctx, LYNX_ERR_INVALID, "element batch_stride smaller than its row");
    const int64_t lo = elems[e].param_offset;
    const int64_t hi = lo + (batch - 1) * bs + np;
    if (lo < 0 || hi > pool_count)
      return fail(ctx, LYNX_ERR_INVALID, "element " + std::to_string(e) + ": parameters outside the pool");
  }
  int32_t next = 0;
  for (int32_t s = 0; s < n_steps; ++s) {
    const lynx_step& st = steps[s];
    if (st.first != next || st.last <= st.first || st.last > n_elems)
      return fail(ctx, LYNX_ERR_INVALID, "steps must tile the element list in order");
    if (st.kind == LYNX_STEP_CAVITY) {
      if (st.last != st.first + 1 || elems[st.first].kind != LYNX_KIND_CAVITY)
        return fail(ctx, LYNX_ERR_INVALID, "cavity step must hold exactly one cavity element");
    } else if (st.kind != LYNX_STEP_RUN) {
      return fail(ctx, LYNX_ERR_INVALID, "unknown step kind");
    }
    for (int32_t e = st.first; e < st.last; ++e) elem_step[e] = s;
    next = st.last;
  }
  if (next != n_elems) return fail(ctx, LYNX_ERR_INVALID, "steps do not cover every element");

  HIP_TRY(ctx, use_device(ctx));
  lynx_lattice* lat = new lynx_lattice();
  lat->ctx = ctx;
  lat->dtype = dtype;
  lat->batch = batch;
  lat->n_elems = n_elems;
  lat->n_steps = n_steps;
  lat->pool_count = pool_count;
  lat->h_elems.assign(elems, elems + n_elems);
  lat->h_steps.assign(steps, steps + n_steps);
  for (int32_t e = 0; e < n_elems; ++e) lat->n_cavities += elems[e].kind == LYNX_KIND_CAVITY ? 1 : 0;
  lat->has_cavity = lat->n_cavities > 0;
  const size_t es = dtype_size(dtype);
  int rc;
  if ((rc = count_observers(ctx, lat))) {
    delete lat;
    return rc;
  }
  if ((rc = ctx_alloc(ctx, std::max<size_t>(1, n_elems) * sizeof(lynx_elem), (void**)&lat->d_elems)) ||
      (rc = ctx_alloc(ctx, std::max<size_t>(1, n_steps) * sizeof(lynx_step), (void**)&lat->d_steps)) ||
      (rc = ctx_alloc(ctx, std::max<size_t>(1, n_elems) * sizeof(int32_t), (void**)&lat->d_elem_step)) ||
      (rc = ctx_alloc(ctx, std::max<size_t>(1, pool_count) * es, &lat->d_pool))) {
    delete lat;
    return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_elems, elems, n_elems * sizeof(lynx_elem), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_steps, steps, n_steps * sizeof(lynx_step), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_elem_step, elem_step.data(), n_elems * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_pool, pool, pool_count * es, hipMemcpyHostToDevice, ctx->stream));
  if (lat->has_cavity) {
    if ((rc = ctx_alloc(ctx, (size_t)lat->n_cavities * sizeof(int32_t), (void**)&lat->d_cav_words))) {
      delete lat;
      return rc;
    }
    HIP_TRY(ctx, hipMemsetAsync(lat->d_cav_words, 0, (size_t)lat->n_cavities * sizeof(int32_t), ctx->stream));
    std::vector<int2> cavs;
    for (int32_t e = 0; e < n_elems; ++e)
      if (elems[e].kind == LYNX_KIND_CAVITY) cavs.push_back(make_int2(e, steps[elem_step[e]].kind == LYNX_STEP_CAVITY ? elem_step[e] : -1));
    if ((rc = ctx_alloc(ctx, cavs.size() * sizeof(int2), (void**)&lat->d_cavs))) {
      delete lat;
      return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(lat->d_cavs, cavs.data(), cavs.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    std::vector<int32_t> before((size_t)n_steps + 1, 0);
    for (int32_t s = 0; s < n_steps; ++s) before[s + 1] = before[s] + (steps[s].kind == LYNX_STEP_CAVITY ? 1 : 0);
    lat->n_step_cavities = before[n_steps];
    if ((rc = ctx_alloc(ctx, before.size() * sizeof(int32_t), (void**)&lat->d_cav_before))) {
      delete lat;
      return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(lat->d_cav_before, before.data(), before.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, sync_main(ctx));  // `cavs` and `before` go out of scope
  }
  HIP_TRY(ctx, sync_main(ctx));
  memset(&lat->h_pool, 0, sizeof(lat->h_pool));
  lat->prog_pool = !lat->has_cavity && n_steps > 0 && (size_t)pool_count * es <= sizeof(lat->h_pool);  // (a cavity's predicates are evaluated by kernels of their own)
  if (lat->prog_pool) memcpy(lat->h_pool.q, pool, (size_t)pool_count * es);
  *out = lat;
  return LYNX_OK;
}

int lynx_lattice_update_params(lynx_lattice* lat, int64_t offset, int64_t count, const void* host) {
  lynx_ctx* ctx = lat->ctx;
  if (offset < 0 || count < 0 || offset + count > lat->pool_count)
    return fail(ctx, LYNX_ERR_INVALID, "lynx_lattice_update_params: range outside the pool");
  const size_t es = dtype_size(lat->dtype);
  ++lat->version;
  if (lat->prog_pool) {  // the kernels of a small lattice read the parameters in their arguments: d_pool follows when it is needed
    memcpy(reinterpret_cast<unsigned char*>(lat->h_pool.q) + offset * es, host, count * es);
    lat->pool_stale = true;
    return LYNX_OK;
  }
  HIP_TRY(ctx, hipMemcpyAsync((char*)lat->d_pool + offset * es, host, count * es, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));
  return LYNX_OK;
}

int lynx_lattice_set_flags(lynx_lattice* lat, const int32_t* elem_flags, const int32_t* step_flags) {
  lynx_ctx* ctx = lat->ctx;
  for (int32_t e = 0; e < lat->n_elems; ++e) lat->h_elems[e].flags = elem_flags[e];
  for (int32_t s = 0; s < lat->n_steps; ++s) lat->h_steps[s].flags = step_flags[s];
  lat->units_made[0] = lat->units_made[1] = false;  // the proposed classes depend on the elements' flags
  ++lat->version;
  {
    const int rc = count_observers(ctx, lat);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_elems, lat->h_elems.data(), lat->n_elems * sizeof(lynx_elem), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(lat->d_steps, lat->h_steps.data(), lat->n_steps * sizeof(lynx_step), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, sync_main(ctx));
  return LYNX_OK;
}

int lynx_lattice_destroy(lynx_lattice* lat) {
  if (!lat) return LYNX_OK;
  lynx_ctx* ctx = lat->ctx;
  if (ctx->fwd_table.lat == lat) ctx->fwd_table.valid = false;  // (the next lattice may get this address)
  ctx_free(ctx, lat->d_elems);
  ctx_free(ctx, lat->d_steps);
  ctx_free(ctx, lat->d_elem_step);
  ctx_free(ctx, lat->d_pool);
  if (lat->d_cav_words) ctx_free(ctx, lat->d_cav_words);
  if (lat->d_cavs) ctx_free(ctx, lat->d_cavs);
  if (lat->d_cav_before) ctx_free(ctx, lat->d_cav_before);
  for (int32_t* p : lat->d_step_unit)
    if (p) ctx_free(ctx, p);
  if (lat->d_bwd_tasks) ctx_free(ctx, lat->d_bwd_tasks);
  if (lat->d_pieces) ctx_free(ctx, lat->d_pieces);
  if (lat->d_tasks) ctx_free(ctx, lat->d_tasks);
  if (lat->d_step_slot) ctx_free(ctx, lat->d_step_slot);
  delete lat;
  return LYNX_OK;
}

static LatticeDev dev_view(const lynx_lattice* lat) {
  LatticeDev d;
  d.elems = lat->d_elems;
  d.steps = lat->d_steps;
  d.elem_step = lat->d_elem_step;
  d.pool = lat->d_pool;
  d.batch = lat->batch;
  d.n_elems = lat->n_elems;
  d.n_steps = lat->n_steps;
  return d;
}

template <typename K>
static int allow_lds(lynx_ctx* ctx, K kernel, size_t bytes) {
  if (bytes > 160 * 1024)
    return fail(ctx, LYNX_ERR_INVALID,
                "program needs " + std::to_string(bytes) + " B of LDS (> 160 KiB): split the lattice");
  if (bytes > 48 * 1024)
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return LYNX_OK;
}

// ---- build + compose -------------------------------------------------------------------

// Launch shape of k_build: 256 threads; chunks of <= 64 elements (<= 32 for float32 lattices, whose staging area
// would otherwise halve the resident workgroups) when the batch fills the GPU, of <= 128 when it does not -- then the
// tree depth is what a call waits for.  (Round 2 gave such a build 1024 threads; measured alone on one sample's
// 128-element float64 FODO, scripts/gpu/experiments/build_phases.hip: 1024 threads 18.0 us, 512: 13.1, 256: 11.3,
// 128: 13.6 -- sixteen waves of eight busy lanes each fetch the builders' code sixteen times and take 8.4 us for the
// first quadrupole where four waves of 32 take 4.6.)
// The long chunk needs ~90-115 KB of LDS: such a workgroup cannot be placed on a CU before a streaming kernel's
// workgroups there have drained, so a build that runs underneath the previous call's streaming kernel (`underneath`)
// keeps the short one (128-sample shard of BASELINE config 4: 18 us alone either way, 117-150 us with the long chunk vs
// the short chunk's share of the GPU under the streaming kernel).
template <typename T>
static void build_shape(lynx_ctx* ctx, const lynx_lattice* lat, bool underneath, int* threads, int* chunk) {
  const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const bool deep = lat->batch * 2 <= cus && !underneath;
  *threads = 256;
  // (float32, at most one workgroup per CU: 64 again -- half the rounds; the 128-sample shard of BASELINE config 4
  // underneath its streaming kernel: 0.1573 -> 0.1543 ms/step, medians of four, same box)
  const int limit = deep ? 128 : ((sizeof(T) == 4 && lat->batch > cus) ? 32 : 64);
  *chunk = build_chunk(lat->n_elems, limit);
}

// Whole-batch cavity predicates for this call's energies, on `stream`, in front of whatever builds maps
// there (no-op for lattices without cavities).
template <typename T>
static int launch_cavity_flags(lynx_ctx* ctx, lynx_lattice* lat, hipStream_t stream, const void* d_energy_in,
                               bool for_lanes_build = false) {
  if (!lat->has_cavity || lat->n_steps == 0) return LYNX_OK;
  int rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_erun, &ctx->scratch_erun_bytes, (size_t)lat->batch * sizeof(T)))) return rc;
  // the lanes build reads every sample's energy in front of every step from what the first kernel computes on its way
  const int64_t Bp = (lat->batch + 63) / 64 * 64;
  T* e_steps = nullptr;
  if (for_lanes_build) {
    if ((rc = ensure_scratch(ctx, &ctx->scratch_esteps, &ctx->scratch_esteps_bytes, (size_t)(lat->n_step_cavities + 1) * Bp * sizeof(T)))) return rc;
    e_steps = (T*)ctx->scratch_esteps;
  }
  // the predicates for every cavity at once, assuming every batch gains energy (one lane per sample), then one small
  // workgroup that checks the assumption, publishes the bits and -- should it not hold -- walks the serial way
  hipLaunchKernelGGL(k_cavity_flags_spec<T>, dim3((unsigned)((lat->batch + 255) / 256)), dim3(256), 0, stream, dev_view(lat),
                     lat->d_cavs, lat->n_cavities, (const T*)d_energy_in, lat->d_cav_words, e_steps, Bp);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_cavity_flags<T>, dim3(1), dim3(256), 0, stream, dev_view(lat), lat->d_elems, lat->d_steps,
                     (const T*)d_energy_in, (T*)ctx->scratch_erun, ctx->d_status, lat->d_cavs, lat->n_cavities,
                     lat->d_cav_words, e_steps ? ctx->d_spec_valid : (int32_t*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

// Plan of the lanes = samples build for this lattice: pieces of <= L elements inside each step, then
// levels of pair products until every step is one slot.
static int plan_lanes_build(lynx_ctx* ctx, lynx_lattice* lat, int L) {
  if (lat->plan_piece_len == L && lat->d_pieces) return LYNX_OK;
  std::vector<BuildPiece> pieces;
  std::vector<std::vector<int32_t>> live(lat->n_steps);
  for (int32_t s = 0; s < lat->n_steps; ++s) {
    const lynx_step& st = lat->h_steps[s];
    for (int32_t e = st.first; e < st.last; e += L) {
      live[s].push_back((int32_t)pieces.size());
      pieces.push_back(BuildPiece{e, std::min<int32_t>(e + L, st.last), s, 0});
    }
  }
  int32_t next = (int32_t)pieces.size();
  std::vector<PairTask> tasks;
  std::vector<std::pair<int32_t, int32_t>> levels;
  for (;;) {
    const int32_t first = (int32_t)tasks.size();
    for (auto& slots : live) {
      if (slots.size() < 2) continue;
      std::vector<int32_t> merged;
      for (size_t k = 0; k + 1 < slots.size(); k += 2) {
        tasks.push_back(PairTask{slots[k], slots[k + 1], next, 0});
        merged.push_back(next++);
      }
      if (slots.size() & 1) merged.push_back(slots.back());
      slots.swap(merged);
    }
    if ((int32_t)tasks.size() == first) break;
    levels.emplace_back(first, (int32_t)tasks.size() - first);
  }
  std::vector<int32_t> step_slot(std::max<int32_t>(1, lat->n_steps), 0);
  for (int32_t s = 0; s < lat->n_steps; ++s) step_slot[s] = live[s][0];
  if (lat->d_pieces) ctx_free(ctx, lat->d_pieces);
  if (lat->d_tasks) ctx_free(ctx, lat->d_tasks);
  if (lat->d_step_slot) ctx_free(ctx, lat->d_step_slot);
  lat->d_pieces = nullptr; lat->d_tasks = nullptr; lat->d_step_slot = nullptr;
  int rc;
  if ((rc = ctx_alloc(ctx, std::max<size_t>(1, pieces.size()) * sizeof(BuildPiece), (void**)&lat->d_pieces)) ||
      (rc = ctx_alloc(ctx, std::max<size_t>(1, tasks.size()) * sizeof(PairTask), (void**)&lat->d_tasks)) ||
      (rc = ctx_alloc(ctx, step_slot.size() * sizeof(int32_t), (void**)&lat->d_step_slot)))
    return rc;
  // plain synchronous copies: planning happens once per lattice structure
  HIP_TRY(ctx, hipMemcpy(lat->d_pieces, pieces.data(), pieces.size() * sizeof(BuildPiece), hipMemcpyHostToDevice));
  if (!tasks.empty()) HIP_TRY(ctx, hipMemcpy(lat->d_tasks, tasks.data(), tasks.size() * sizeof(PairTask), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(lat->d_step_slot, step_slot.data(), step_slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  lat->plan_piece_len = L;
  lat->n_pieces = (int32_t)pieces.size();
  lat->n_slots = next;
  lat->levels.swap(levels);
  return LYNX_OK;
}

template <typename T>
static int launch_build_lanes(lynx_ctx* ctx, lynx_lattice* lat, hipStream_t stream, const void* d_energy_in,
                              void* d_steps_out, void* d_energy_out, int merge_pairs, float* d_units = nullptr,
                              float* d_extras = nullptr) {
  int rc;
  if ((rc = plan_lanes_build(ctx, lat, std::max(1, ctx->knobs.piece)))) return rc;
  const int64_t groups = (lat->batch + 63) / 64, Bp = groups * 64;
  if (lat->n_pieces > 65535 || lat->n_steps > 65535) return fail(ctx, LYNX_ERR_INVALID, "lattice too long for the lanes build");
  if ((rc = ensure_scratch(ctx, &ctx->scratch_products, &ctx->scratch_products_bytes, (size_t)lat->n_slots * 49 * Bp * sizeof(double))))
    return rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_coefs, &ctx->scratch_coefs_bytes, (size_t)std::max(1, lat->n_steps) * 8 * Bp * sizeof(T))))
    return rc;
  const LatticeDev lv = dev_view(lat);
  const StepEnergies<T> se{lat->has_cavity ? (const T*)ctx->scratch_esteps : (const T*)nullptr, lat->d_cav_before, ctx->d_spec_valid, Bp};
  if ((rc = allow_lds(ctx, k_build_pieces<T>, build_pieces_lds<T>()))) return rc;
  hipLaunchKernelGGL(k_build_pieces<T>, dim3((unsigned)groups, (unsigned)lat->n_pieces), dim3(64), build_pieces_lds<T>(), stream, lv,
                     lat->d_pieces, (const T*)d_energy_in, Bp, (double*)ctx->scratch_products, (T*)ctx->scratch_coefs, se);
  HIP_TRY(ctx, hipGetLastError());
  // narrow trees (BASELINE config 4: 8, 4, 2, 1 tasks per level) in one launch: underneath a streaming kernel every
  // launch of the chain costs ~20 us, the product itself 3-6 (config 4: 0.995 -> 0.985 ms/step, same box)
  bool narrow = !lat->levels.empty() && (int)lat->levels.size() <= kPairLevelsMax && ctx->knobs.pair_levels_fused != 0;
  for (const auto& level : lat->levels) narrow = narrow && level.second <= kPairLevelsMaxTasks;
  if (narrow) {
    PairLevels lv{};
    lv.n = (int32_t)lat->levels.size();
    for (int l = 0; l < lv.n; ++l) {
      lv.first[l] = lat->levels[l].first;
      lv.count[l] = lat->levels[l].second;
    }
    hipLaunchKernelGGL(k_pair_levels, dim3((unsigned)groups), dim3(256), 0, stream, lv, lat->d_tasks, lat->batch, Bp,
                       (double*)ctx->scratch_products);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    for (const auto& level : lat->levels) {
      hipLaunchKernelGGL(k_pair_products, dim3((unsigned)groups, (unsigned)level.second), dim3(64), 0, stream,
                         lat->d_tasks + level.first, lat->batch, Bp, (double*)ctx->scratch_products);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  hipLaunchKernelGGL(k_emit_steps<T>, dim3((unsigned)groups, (unsigned)lat->n_steps), dim3(64), emit_steps_lds<T>(), stream, lv,
                     lat->d_step_slot, (const T*)d_energy_in, Bp, (const double*)ctx->scratch_products,
                     (const T*)ctx->scratch_coefs, merge_pairs, (T*)d_steps_out, (T*)d_energy_out,
                     d_units ? lat->d_step_unit[merge_pairs ? 1 : 0] : (const int32_t*)nullptr,
                     lat->units[merge_pairs ? 1 : 0].n_units, d_units, d_extras, se);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

// `d_units` / `d_extras`: also pack the unit records of a multi-step float32 program (the lattice's unit plan for `merge_pairs` must have been made);
// the lanes build does it while it writes the table, the workgroup build with k_pack_units behind it
template <typename T>
static int launch_build(lynx_ctx* ctx, lynx_lattice* lat, hipStream_t stream, const void* d_energy_in,
                        void* d_steps_out, void* d_energy_out, int merge_pairs = 0, bool underneath = false,
                        float* d_units = nullptr, float* d_extras = nullptr) {
  // large batches: lanes = samples (an order of magnitude fewer wave-instructions); small ones: one
  // workgroup per sample, whose tree is shallower than a chain of launches
  const bool lanes = lat->n_steps > 0 && lat->batch >= ctx->knobs.lanes_build_min_batch;
  {
    const int rc = launch_cavity_flags<T>(ctx, lat, stream, d_energy_in, lanes);
    if (rc) return rc;
  }
  if (lanes) {
    const int rc = sync_pool(ctx, lat);
    return rc ? rc : launch_build_lanes<T>(ctx, lat, stream, d_energy_in, d_steps_out, d_energy_out, merge_pairs, d_units, d_extras);
  }
  int threads, chunk;
  build_shape<T>(ctx, lat, underneath, &threads, &chunk);
  const size_t lds = build_scratch_bytes(chunk, sizeof(T)) +
                     ((size_t)lat->n_steps * LYNX_STEP_STRIDE + lat->n_steps + 1) * sizeof(T);
  int rc;
  if (lat->prog_pool && ctx->knobs.inline_pool) {
    if ((size_t)lat->pool_count * sizeof(T) <= (size_t)kInlinePoolSmall) {
      if ((rc = allow_lds(ctx, k_build_inline<T, kInlinePoolSmall>, lds))) return rc;
      hipLaunchKernelGGL((k_build_inline<T, kInlinePoolSmall>), dim3((unsigned)lat->batch), dim3((unsigned)threads), lds, stream,
                         *reinterpret_cast<const InlinePool<kInlinePoolSmall>*>(&lat->h_pool), dev_view(lat),
                         (const T*)d_energy_in, (T*)d_steps_out, (T*)d_energy_out, chunk, merge_pairs);
    } else {
      if ((rc = allow_lds(ctx, k_build_inline<T, kInlinePoolLarge>, lds))) return rc;
      hipLaunchKernelGGL((k_build_inline<T, kInlinePoolLarge>), dim3((unsigned)lat->batch), dim3((unsigned)threads), lds, stream,
                         lat->h_pool, dev_view(lat), (const T*)d_energy_in, (T*)d_steps_out, (T*)d_energy_out, chunk, merge_pairs);
    }
  } else {
    if ((rc = sync_pool(ctx, lat)) || (rc = allow_lds(ctx, k_build<T>, lds))) return rc;
    hipLaunchKernelGGL(k_build<T>, dim3((unsigned)lat->batch), dim3((unsigned)threads), lds, stream, dev_view(lat),
                       (const T*)d_energy_in, (T*)d_steps_out, (T*)d_energy_out, chunk, merge_pairs);
  }
  HIP_TRY(ctx, hipGetLastError());
  if constexpr (sizeof(T) == 4) {
    if (d_units) {
      const UnitPlan& up = lat->units[merge_pairs ? 1 : 0];
      const int64_t n = lat->batch * up.n_units;
      hipLaunchKernelGGL(k_pack_units, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, up, lat->batch,
                         lat->n_steps, (const float*)d_steps_out, d_units, d_extras);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  return LYNX_OK;
}

int lynx_build_compose(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in, void* d_steps_out,
                       void* d_energy_out) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !lat || !d_energy_in || !d_steps_out) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (lat->batch > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "batch too large");
  HIP_TRY(ctx, use_device(ctx));
  ctx->main_dirty = true;
  return lat->dtype == LYNX_F64 ? launch_build<double>(ctx, lat, ctx->stream, d_energy_in, d_steps_out, d_energy_out)
                                : launch_build<float>(ctx, lat, ctx->stream, d_energy_in, d_steps_out, d_energy_out);
}

// ---- particle tracking -----------------------------------------------------------------

struct TrackPlan {
  hipEvent_t done = nullptr;  // recorded when the streaming kernel has finished (may be null)
  bool full_cov; // accumulate the whole covariance (21 products) instead of the property set (8)
  int unroll;    // particles per lane and iteration
  int mom_mode;  // 1 = float64 per particle, 2 = float32 partial sums per iteration, 3 = float32 lane sums
  bool xpose;    // wave tiles through LDS (full-width accesses) instead of per-particle accesses
  TrackArgs a;
  size_t lds;
  unsigned grid;
};

template <typename T>
static TrackPlan plan_track(lynx_ctx* ctx, const lynx_lattice* lat, int64_t B, int64_t N, int32_t S, bool fused,
                            bool moments, bool full_cov) {
  TrackPlan p;
  p.full_cov = full_cov;
  constexpr int P = 16 / (int)sizeof(T);  // particles per lane of a wave tile
  const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const Knobs& kn = ctx->knobs;
  const int64_t target = 512 * cus;  // workgroups per CU over the whole launch
  p.mom_mode = knob(kn.mom, sizeof(T) == 4 ? 2 : 1);
  if (sizeof(T) == 8 || p.mom_mode < 2 || p.mom_mode > 3) p.mom_mode = sizeof(T) == 4 ? 2 : 1;
  p.a.n_particles = N;
  p.a.fused_build = (fused && S > 0) ? 1 : 0;
  p.a.store = 0;
  p.a.build_chunk = 1;
  p.a.tail_flag = nullptr;
  p.a.tail_seq = 0;
  p.a.tail_wg = 0;
  p.a.stash_offset = 0;
  p.a.reversed = 0;
  // wave tiles: with non-temporal full-width stores +7 % on single-map float32 programs (BASELINE config 4:
  // 5.4 -> 5.8 TB/s) and +10 % (+7 % of that from the stores) on float64 (config 3 at 8 M particles); slower on
  // multi-step float32 programs, which want two particles per lane, not four
  // (a read-only pass -- lynx_moments, S = 0 -- has no stores to gain from and stays with per-particle loads)
  const bool single_map = S == 1 && lat && lat->h_steps[0].kind == LYNX_STEP_RUN &&
                          !(lat->h_steps[0].flags & LYNX_STEP_FLAG_OBSERVE);
  // (with the whole covariance in float32 lane sums the tile form runs out of registers: 1.23 vs 1.13 ms on C4)
  // (and short samples leave most wave tiles cut: 700 particles per sample 0.79 vs 0.72 ms per 20 M particles)
  p.xpose = !p.a.fused_build &&
            knob(kn.xpose, ((sizeof(T) == 8 || (single_map && !full_cov)) && N >= 8 * 64 * P) ? 1 : 0) != 0;
  int u = knob(kn.unroll, p.xpose ? P : (sizeof(T) == 4 ? (S > 1 ? 2 : 4) : 1));
  if (u != 1 && u != 2 && u != 4) u = 2;
  if (sizeof(T) == 8 && u > 2) u = 2;
  // small jobs: fewer particles per lane so that more workgroups exist
  while (u > 1 && B * ((N + 256 * u - 1) / (256 * u)) < 4 * cus) u >>= 1;
  if (u != P) p.xpose = false;
  p.unroll = u;
  const int64_t tile = 256 * (int64_t)u;
  const int64_t ntiles = (N + tile - 1) / tile;
  // A workgroup pays a fixed cost (table load, 29-value cross-lane reduction, partial
  // record), so it gets at least `min_tpw` tiles -- unless that would leave CUs idle.
  // Measured with the round-2 kernels (property-set moments, scripts/gpu/r2_c4sweep.sh, same box): single-map
  // programs like many short-lived workgroups -- 2 tiles each: C4 5.26 -> 5.49 TB/s, c3big 4.83 -> 5.09 --
  // while multi-step programs also fetch S step records per iteration and want long-lived ones (C5 with
  // 2 / 4 / 8 / 16 tiles: 1.25 / 1.14 / 1.10 / 1.09 ms).
  int64_t min_tpw = std::max(1, knob(kn.min_tiles_per_wg, S > 1 ? 16 : 2));
  while (min_tpw > 1 && B * ((ntiles + min_tpw - 1) / min_tpw) < 4 * cus) --min_tpw;
  int64_t chunks = std::max<int64_t>(1, std::min<int64_t>((ntiles + min_tpw - 1) / min_tpw, (target + B - 1) / B));
  // A launch of a FEW rounds of workgroups is better off as ONE round of longer-lived ones: a workgroup of the single-map
  // kernels takes ~10 us whatever it does (table load, first tile's latency, moment epilogue), three of them fit a CU
  // (162-164 registers), and BASELINE config 3 as worded -- one sample, 1 M particles, 1954 tiles -- ran 1954 one-tile
  // workgroups in 2.5 rounds: 28.9 us; 652 workgroups of three tiles: 23.7 (46 -> 39 us a step).  Launches of many rounds
  // keep their short-lived workgroups (capped the same way: c3big 155 -> 189 us, config 4 0.96 -> 1.2-1.3 ms).
  {
    const int64_t one_round = 3 * cus;
    const int cap = knob(kn.one_round, (S <= 1 && B * chunks > one_round && B * chunks <= 4 * one_round) ? 3 : 0);
    if (cap > 0) chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, cap * cus / B));
  }
  int64_t tpw = (ntiles + chunks - 1) / chunks;
  chunks = (ntiles + tpw - 1) / tpw;
  p.a.chunks = (int32_t)chunks;
  p.a.tiles_per_wg = (int32_t)tpw;
  // float32 lane sums for the whole workgroup are fine while a lane sees few particles
  if (p.mom_mode == 2 && kn.mom < 0 && tpw * u <= 32) p.mom_mode = 3;
  if (p.mom_mode == 3 && tpw * u > 64) p.mom_mode = 2;
  size_t scratch = 4 * kPartialStride * sizeof(double);
  if (moments)
    scratch = std::max<size_t>(scratch, (size_t)(full_cov ? kMomSlabScalars : kMomSlabScalarsCompact) * (p.mom_mode >= 2 ? 4 : 8));
  if (p.xpose) scratch = std::max<size_t>(scratch, (size_t)(kTrackThreads / 64) * kWaveTileBytes);
  if (p.a.fused_build) {
    p.a.build_chunk = build_chunk(lat->n_elems, sizeof(T) == 4 ? 32 : 64);
    scratch = std::max<size_t>(scratch, build_scratch_bytes(p.a.build_chunk, sizeof(T)));
  }
  scratch = (scratch + 15) / 16 * 16;
  p.a.lds_scratch_bytes = (int32_t)scratch;
  p.a.n_observers = lat ? lat->n_observers : 0;
  p.lds = (scratch + ((size_t)S * LYNX_STEP_STRIDE + S + 1) * sizeof(T) + 7) / 8 * 8 + (size_t)2 * p.a.n_observers * 256 * sizeof(double);
  p.grid = (unsigned)(B * chunks);
  return p;
}

template <typename T, int MOM, bool FULL, int UNROLL, bool FUSED, bool XPOSE>
static int launch_direct_inst(lynx_ctx* ctx, const TrackPlan& p, const LatticeDev& lv, const void* d_energy_in,
                              const void* d_p_in, void* d_p_out, void* d_energy_out, const void* d_steps,
                              double* d_partials, double* d_obs) {
  int rc = allow_lds(ctx, k_track_direct<T, MOM, FULL, UNROLL, FUSED, XPOSE>, p.lds);
  if (rc) return rc;
  // Events ride on the dispatch itself (hipExtLaunchKernelGGL: start / stop timestamps of this kernel)
  // instead of separate marker packets in front of and behind it: `done` is what the build stream waits
  // for before it reuses the step-table slot, and with profiling on (bench.py) the pair is also the
  // kernel's duration.
  hipEvent_t e0 = nullptr, e1 = p.done;
  if (ctx->profiling) {  // every profiled launch needs time stamps of its own
    HIP_TRY(ctx, hipEventCreateWithFlags(&e0, timing_event_flags(ctx)));
    HIP_TRY(ctx, hipEventCreateWithFlags(&e1, timing_event_flags(ctx)));
  }
  hipExtLaunchKernelGGL((k_track_direct<T, MOM, FULL, UNROLL, FUSED, XPOSE>), dim3(p.grid), dim3(kTrackThreads),
                        (std::uint32_t)p.lds, ctx->stream, e0, e1, 0u, lv, p.a, (const T*)d_energy_in, (const T*)d_p_in,
                        (T*)d_p_out, (T*)d_energy_out, (const T*)d_steps, d_partials, d_obs);
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->profiling) ctx->prof_events.emplace_back(e0, e1);
  ctx->last_stream_stop = e1;  // what "this streaming kernel has finished" is, for the build stream
  return LYNX_OK;
}

#define LYNX_LAUNCH_ARGS ctx, p, lv, d_energy_in, d_p_in, d_p_out, d_energy_out, d_steps, d_partials, d_obs
#define LYNX_LAUNCH_PARAMS                                                                                     \
  lynx_ctx *ctx, const TrackPlan &p, const LatticeDev &lv, const void *d_energy_in, const void *d_p_in,        \
      void *d_p_out, void *d_energy_out, const void *d_steps, double *d_partials, double *d_obs

template <typename T, int MOM, bool FULL, int U>
static int launch_direct_mu(LYNX_LAUNCH_PARAMS) {
  if (p.a.fused_build) {
    // the fused prologue is an option for jobs of a few workgroups: one particle per lane only
    if constexpr (U == 1) return launch_direct_inst<T, MOM, FULL, 1, true, false>(LYNX_LAUNCH_ARGS);
    else return fail(ctx, LYNX_ERR_INVALID, "fused build: one particle per lane");
  }
  if constexpr (U * 7 * sizeof(T) == 112) {
    if (p.xpose) return launch_direct_inst<T, MOM, FULL, U, false, true>(LYNX_LAUNCH_ARGS);
  }
  return launch_direct_inst<T, MOM, FULL, U, false, false>(LYNX_LAUNCH_ARGS);
}

template <typename T, int MOM, bool FULL>
static int launch_direct_m(LYNX_LAUNCH_PARAMS) {
  switch (p.unroll) {
    case 1: return launch_direct_mu<T, MOM, FULL, 1>(LYNX_LAUNCH_ARGS);
    case 2: return launch_direct_mu<T, MOM, FULL, 2>(LYNX_LAUNCH_ARGS);
    default:
      if constexpr (sizeof(T) == 4) return launch_direct_mu<T, MOM, FULL, 4>(LYNX_LAUNCH_ARGS);
      else return fail(ctx, LYNX_ERR_INVALID, "float64: at most 2 particles per lane");
  }
}

template <typename T>
static int launch_direct(LYNX_LAUNCH_PARAMS, bool moments) {
  const int mom = moments ? p.mom_mode : 0;
  if (mom == 0) return launch_direct_m<T, 0, false>(LYNX_LAUNCH_ARGS);
  if constexpr (sizeof(T) == 4) {
    if (mom == 3) return p.full_cov ? launch_direct_m<T, 3, true>(LYNX_LAUNCH_ARGS) : launch_direct_m<T, 3, false>(LYNX_LAUNCH_ARGS);
    return p.full_cov ? launch_direct_m<T, 2, true>(LYNX_LAUNCH_ARGS) : launch_direct_m<T, 2, false>(LYNX_LAUNCH_ARGS);
  } else {
    return p.full_cov ? launch_direct_m<T, 1, true>(LYNX_LAUNCH_ARGS) : launch_direct_m<T, 1, false>(LYNX_LAUNCH_ARGS);
  }
}
#undef LYNX_LAUNCH_ARGS
#undef LYNX_LAUNCH_PARAMS

// ---- multi-step float32 programs as units (lynx_units.hpp) ------------------------------------

// Class a unit's map is expected to have, from the kinds and whole-batch flags of its elements (the numbers are checked
// per sample by k_pack_units; a wrong guess costs speed, never correctness).
static int proposed_class(const lynx_lattice* lat, int32_t first, int32_t last) {
  int rank = 1;  // 1 = U, 2 = D, 3 = dense
  for (int32_t e = first; e < last; ++e) {
    const lynx_elem& el = lat->h_elems[e];
    switch (el.kind) {
      case LYNX_KIND_IDENTITY:
      case LYNX_KIND_DRIFT:
      case LYNX_KIND_HCOR:
      case LYNX_KIND_VCOR:
      case LYNX_KIND_CAVITY:
      case LYNX_KIND_UNDULATOR: break;
      case LYNX_KIND_QUADRUPOLE: rank = std::max(rank, (el.flags & LYNX_FLAG_TILT) ? 3 : 1); break;
      case LYNX_KIND_DIPOLE: rank = std::max(rank, 2); break;  // a tilted one fails the numeric check
      default: rank = 3; break;                                // solenoid, custom map, helper kinds
    }
  }
  return rank == 1 ? kClassU : rank == 2 ? kClassD : kClassDense;
}

// The program as units: a step, or (merged) a run together with the active cavity behind it.  False if it does not fit.
static bool plan_units(const lynx_lattice* lat, bool merged, UnitPlan* plan) {
  const int32_t S = lat->n_steps;
  plan->n_units = 0;
  for (int32_t s = 0; s < S; ++s) {
    const lynx_step& st = lat->h_steps[s];
    if (st.flags & LYNX_STEP_FLAG_OBSERVE) return false;
    const bool pair = merged && s + 1 < S && st.kind == LYNX_STEP_RUN && lat->h_steps[s + 1].kind == LYNX_STEP_CAVITY;
    const int32_t last = pair ? lat->h_steps[s + 1].last : st.last;
    if (pair) ++s;
    if (plan->n_units >= kMaxUnits || s > 255) return false;
    const int u = plan->n_units++;
    plan->slot[u] = (unsigned char)s;
    plan->pair[u] = pair ? 1 : 0;
    plan->cls[u] = (unsigned char)proposed_class(lat, st.first, last);
  }
  for (int u = plan->n_units; u < kMaxUnits; ++u) plan->slot[u] = plan->cls[u] = plan->pair[u] = 0;
  return plan->n_units > 0;
}

// The lattice's unit plan for `merged`, made once per lattice and flag change; with it the table k_emit_steps reads.
static int ensure_units_plan(lynx_ctx* ctx, lynx_lattice* lat, bool merged) {
  const int m = merged ? 1 : 0;
  if (lat->units_made[m]) return LYNX_OK;
  UnitPlan& up = lat->units[m];
  lat->units_ok[m] = plan_units(lat, merged, &up) && lat->batch * up.n_units * kUnitStride < 0x7fffffffLL;
  lat->units_made[m] = true;
  if (!lat->units_ok[m]) return LYNX_OK;
  std::vector<int32_t> code((size_t)lat->n_steps, -1);
  for (int u = 0; u < up.n_units; ++u) code[up.slot[u]] = step_unit_code(u, up.cls[u], up.pair[u]);
  int rc;
  if (!lat->d_step_unit[m] && (rc = ctx_alloc(ctx, code.size() * sizeof(int32_t), (void**)&lat->d_step_unit[m]))) return rc;
  // rare (new lattice, or its flags changed): nothing may still be reading the old table
  HIP_TRY(ctx, hipStreamSynchronize(ctx->s_build));
  HIP_TRY(ctx, sync_main(ctx));
  HIP_TRY(ctx, hipMemcpy(lat->d_step_unit[m], code.data(), code.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return LYNX_OK;
}

template <int MOM, bool FULL, int PAIRS, bool PAIR_FORM = false>
static int launch_units_inst(lynx_ctx* ctx, const TrackPlan& p, int32_t U, int32_t S, const void* d_p_in, void* d_p_out,
                             void* d_energy_out, const void* d_steps, const void* d_units, const void* d_extras,
                             double* d_partials) {
  const size_t lds = units_lds_bytes((size_t)p.a.lds_scratch_bytes, U);
  auto kernel = k_track_units<MOM, FULL, PAIRS>;
  if constexpr (PAIR_FORM) kernel = k_track_unit_pairs<MOM, FULL>;
  int rc = allow_lds(ctx, kernel, lds);
  if (rc) return rc;
  hipEvent_t e0 = nullptr, e1 = p.done;
  if (ctx->profiling) {  // every profiled launch needs time stamps of its own
    HIP_TRY(ctx, hipEventCreateWithFlags(&e0, timing_event_flags(ctx)));
    HIP_TRY(ctx, hipEventCreateWithFlags(&e1, timing_event_flags(ctx)));
  }
  hipExtLaunchKernelGGL(kernel, dim3(p.grid), dim3(kTrackThreads), (std::uint32_t)lds, ctx->stream, e0, e1,
                        0u, p.a, U, S, (const float*)d_p_in, (float*)d_p_out, (float*)d_energy_out, (const float*)d_steps,
                        (const float*)d_units, (const float*)d_extras, d_partials);
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->profiling) ctx->prof_events.emplace_back(e0, e1);
  ctx->last_stream_stop = e1;
  return LYNX_OK;
}

static int launch_units(lynx_ctx* ctx, const TrackPlan& p, int32_t U, int32_t S, const void* d_p_in, void* d_p_out,
                        void* d_energy_out, const void* d_steps, const void* d_units, const void* d_extras,
                        double* d_partials, bool moments, bool pair_form) {
#define LYNX_UNITS_ARGS ctx, p, U, S, d_p_in, d_p_out, d_energy_out, d_steps, d_units, d_extras, d_partials
  // every unit proposed as a merged [run, cavity] pair of class U: k_track_unit_pairs
  if (pair_form)
    return moments ? launch_units_inst<3, false, 1, true>(LYNX_UNITS_ARGS) : launch_units_inst<0, false, 1, true>(LYNX_UNITS_ARGS);
  if (p.unroll == 4) {  // two pairs per lane
    if (!moments) return launch_units_inst<0, false, 2>(LYNX_UNITS_ARGS);
    if (p.mom_mode == 3) return p.full_cov ? launch_units_inst<3, true, 2>(LYNX_UNITS_ARGS) : launch_units_inst<3, false, 2>(LYNX_UNITS_ARGS);
    return p.full_cov ? launch_units_inst<2, true, 2>(LYNX_UNITS_ARGS) : launch_units_inst<2, false, 2>(LYNX_UNITS_ARGS);
  }
  if (!moments) return launch_units_inst<0, false, 1>(LYNX_UNITS_ARGS);
  if (p.mom_mode == 3) return p.full_cov ? launch_units_inst<3, true, 1>(LYNX_UNITS_ARGS) : launch_units_inst<3, false, 1>(LYNX_UNITS_ARGS);
  return p.full_cov ? launch_units_inst<2, true, 1>(LYNX_UNITS_ARGS) : launch_units_inst<2, false, 1>(LYNX_UNITS_ARGS);
#undef LYNX_UNITS_ARGS
}

template <typename T>
static int track_particles_t(lynx_ctx* ctx, lynx_lattice* lat, const LatticeDev& lv, int64_t B, int64_t N,
                             const void* d_energy_in, const void* d_p_in, void* d_p_out, void* d_energy_out,
                             double* d_moments_out, int flags, double* d_observations = nullptr) {
  const int32_t S = lv.n_steps;
  const bool moments = (flags & (LYNX_TRACK_MOMENTS | LYNX_TRACK_COVARIANCE)) != 0;
  // Fused prologue vs separate build launch.  Default: separate launch (LYNX_FUSE_MAX_CHUNKS
  // = 0).  In the fused variant every workgroup of a sample rebuilds that sample's maps in
  // its prologue; that only pays with few, long-lived workgroups per sample, which stream
  // ~15 % slower than many small ones, and it forces the step table
  // through LDS instead of scalar loads.  Set LYNX_FUSE_MAX_CHUNKS=<n> to fuse whenever a
  // sample is covered by <= n workgroups.
  const bool full_cov = (flags & LYNX_TRACK_COVARIANCE) != 0;
  TrackPlan p = plan_track<T>(ctx, lat, B, N, S, true, moments, full_cov);
  bool fused = S > 0 && !(flags & LYNX_TRACK_TWO_KERNEL) && p.a.chunks <= ctx->knobs.fuse_max_chunks && p.unroll == 1;
  if (!fused) p = plan_track<T>(ctx, lat, B, N, S, false, moments, full_cov);
  p.a.store = d_p_out ? 1 : 0;
  const bool shared_in = (flags & LYNX_TRACK_SHARED_INPUT) != 0;
  p.a.in_stride = shared_in ? 0 : N * 7;
  p.a.merged_pairs = 0;
  const void* d_steps = nullptr;
  const void* d_units = nullptr;
  const void* d_extras = nullptr;
  int n_units = 0;
  bool use_units = false;
  bool tail = false;
  int rc;
  int slot = -1;
  bool async_build = false;
  bool short_call = false;
  if (S > 0 && !fused) {
    slot = (int)(ctx->seq++ % (unsigned)lynx_ctx::kTableSlots);
    const size_t need = (size_t)B * S * LYNX_STEP_STRIDE * sizeof(T);
    if ((rc = ensure_scratch(ctx, &ctx->scratch_steps[slot], &ctx->scratch_steps_bytes[slot], need))) return rc;
    // [run, cavity] pairs in merged form for the packed float32 step loop (one 7x7 application
    // per pair); LYNX_TRACK_SEQUENTIAL_STEPS keeps every step on its own
    bool has_pair = false;
    for (int32_t s = 1; s < S; ++s)
      has_pair |= lat->h_steps[s].kind == LYNX_STEP_CAVITY && lat->h_steps[s - 1].kind == LYNX_STEP_RUN &&
                  !(lat->h_steps[s - 1].flags & LYNX_STEP_FLAG_OBSERVE);
    p.a.merged_pairs = has_pair && sizeof(T) == 4 && p.unroll == 2 &&
                       !(flags & LYNX_TRACK_SEQUENTIAL_STEPS) && ctx->knobs.merge_steps;
    if (p.a.merged_pairs) {  // four floats of LDS per lane for pairs that take the rows form (kEntryStash)
      p.lds = (p.lds + 15) / 16 * 16;
      p.a.stash_offset = (int32_t)p.lds;
      p.lds += (size_t)kTrackThreads * 4 * sizeof(float);
    }
    // The second stream pays once the streaming kernel is long enough to hide a build under; below half a
    // million particles per call the extra event traffic costs more host time than the overlap returns
    // (BASELINE config 2: 31 -> 46 us per call with it).
    // SHORT calls in between (from half a million particles up to 128 MB of them: BASELINE config 3, 1 M particles,
    // 29 us of kernel) take what the queue says: kernels of one queue follow each other without a gap, a hop to
    // another queue costs 10-20 us of latency (kernel timeline, profiles/r04_*_timeline.txt).  If the main stream is
    // IDLE -- the caller waits for every result -- there is nothing to hide the build under and the hops are pure
    // loss: build, stream and reduce back to back on the main stream, no event at all.  If it is BUSY -- calls are
    // pipelined -- the build goes to the second stream, where it runs underneath the previous call's kernels, and only
    // the reduction stays in line (the side stream's hop is worth it for long kernels only).
    const bool half_million = B * N >= (int64_t)512 << 10;
    short_call = half_million && (size_t)B * N * 7 * sizeof(T) < ((size_t)128 << 20) && !lat->has_cavity;
    bool inline_all = false;
    if (short_call && ctx->knobs.async_build < 0)
      inline_all = knob(ctx->knobs.small_inline, ctx->main_idle ? 1 : 0) != 0;
    const bool async = knob(ctx->knobs.async_build, half_million && !inline_all ? 1 : 0) != 0;
    async_build = async;
    hipStream_t bs = async ? ctx->s_build : ctx->stream;
    if (async) {
      // the table slot was last read by the streaming kernel kTableSlots calls ago
      if (ctx->streamed_valid[slot]) HIP_TRY(ctx, hipStreamWaitEvent(bs, ctx->ev_streamed[slot], 0));
      // ... and the build starts in the TAIL of the streaming kernel before the one enqueued last (which has the GPU
      // to itself when it gets there), not at that kernel's head
      tail = ctx->can_wait_value && ctx->knobs.build_in_tail && !short_call;  // (a short kernel has no tail worth waiting for)
      if (tail && ctx->tail_seq >= 2)
        HIP_TRY(ctx, hipStreamWaitValue32(bs, ctx->d_tail_flag, ctx->tail_seq - 1, hipStreamWaitValueGte, 0xffffffffu));
      // what the build reads (energy, lattice pool) may have been written on the main stream
      if (ctx->main_dirty || (ctx->main_wrote && ctx->main_wrote == d_energy_in)) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_mark, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(bs, ctx->ev_mark, 0));
        ctx->main_dirty = false;
        ctx->main_wrote = nullptr;
      }
    }
    float* d_units_w = nullptr;
    float* d_extras_w = nullptr;
    if constexpr (sizeof(T) == 4) {
      // multi-step programs: walked as units with structured maps (lynx_units.hpp); LYNX_TRACK_UNITS=0 keeps the
      // dense step loop of k_track_direct
      // (LYNX_TRACK_UNITS=2: insist -- an error if this call cannot take the structured loop; for tests)
      const int want_units = ctx->knobs.track_units;
      // (k_track_units addresses a sample's particles with 32-bit byte offsets: samples below 4 GiB)
      if (S > 1 && (p.unroll == 2 || p.unroll == 4) && !p.xpose && p.a.n_observers == 0 && want_units &&
          (uint64_t)N * 28u + ((uint64_t)1 << 20) < ((uint64_t)1 << 32)) {
        if ((rc = ensure_units_plan(ctx, lat, p.a.merged_pairs != 0))) return rc;
        use_units = lat->units_ok[p.a.merged_pairs ? 1 : 0];
      }
      if (want_units == 2 && !use_units)
        return fail(ctx, LYNX_ERR_INVALID, "LYNX_TRACK_UNITS=2: this call does not take the structured step loop");
      if (use_units) {
        const int64_t n = B * lat->units[p.a.merged_pairs ? 1 : 0].n_units;
        const int xs = lynx_ctx::kTableSlots + slot;
        if ((rc = ensure_scratch(ctx, &ctx->scratch_units[slot], &ctx->scratch_units_bytes[slot], (size_t)n * kUnitStride * sizeof(float))) ||
            (rc = ensure_scratch(ctx, &ctx->scratch_units[xs], &ctx->scratch_units_bytes[xs], (size_t)n * kUnitExtraStride * sizeof(float))))
          return rc;
        d_units = d_units_w = (float*)ctx->scratch_units[slot];
        d_extras = d_extras_w = (float*)ctx->scratch_units[xs];
        n_units = lat->units[p.a.merged_pairs ? 1 : 0].n_units;
      }
    }
    if ((rc = launch_build<T>(ctx, lat, bs, d_energy_in, ctx->scratch_steps[slot], nullptr, p.a.merged_pairs, async, d_units_w, d_extras_w)))
      return rc;
    if (async) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_built[slot], bs));
      // The HOST waits for the build instead of the main stream: no barrier packet with a foreign signal in front of
      // the streaming kernel (measured on the 128-sample shard of BASELINE config 4: step - kernel 18.8 -> 8.5-10 us,
      // c3big 17.9 -> 8-14, config 4 itself 20 -> 14).  The build of this call started when the streaming kernel of
      // call n - kTableSlots + 1 began, so by now the GPU still has about two calls queued.  Not for lattices with
      // cavities: their build (k_cavity_flags + the lanes kernels) takes most of a streaming kernel's time next to a
      // VALU-bound kernel, and the host's wake-up would sit on the critical path (BASELINE config 5: 0.932 -> 0.947
      // ms/step with it).  LYNX_BUILD_HOST_WAIT=0 / 1 forces the main-stream / host wait.
      // Nor for short streaming kernels (BASELINE config 3, 1 M particles: 29 us of kernel, 22 us of build -- the
      // host would be the pacemaker: 48 -> 75 us/step): from 128 MB of particles per call.
      const bool long_kernel = (size_t)B * N * 7 * sizeof(T) >= ((size_t)128 << 20);
      if (knob(ctx->knobs.build_host_wait, (lat->has_cavity || !long_kernel) ? 0 : 1)) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_built[slot]));
      else HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_built[slot], 0));
    }
    d_steps = ctx->scratch_steps[slot];
    ctx->fwd_table.lat = lat;
    ctx->fwd_table.version = lat->version;
    ctx->fwd_table.energy = d_energy_in;
    ctx->fwd_table.energy_bytes = (size_t)B * sizeof(T);
    ctx->fwd_table.slot = slot;
    ctx->fwd_table.merged = p.a.merged_pairs;
    ctx->fwd_table.units = use_units;
    ctx->fwd_table.seq = ctx->seq;
    ctx->fwd_table.valid = d_energy_out != d_energy_in;  // (a call that overwrites its own incoming energy leaves nothing to come back to)
  }
  if (fused && lat) {
    if ((rc = sync_pool(ctx, lat))) return rc;  // (the prologue reads the parameters from memory)
    if ((rc = launch_cavity_flags<T>(ctx, lat, ctx->stream, d_energy_in))) return rc;
    ctx->main_dirty = true;  // it rewrote the lattice's flags on the main stream: a later build on s_build waits
  }
  if ((int64_t)B * p.a.chunks > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "grid too large");
  // Workgroup records: a ring of buffers, so that the reduction of call n (side stream) can still read its records
  // while the streaming kernels of calls n+1.. write theirs.  The host makes sure the slot's previous reduction is done.
  double* d_partials = nullptr;
  lynx_ctx::PartialSlot* ring = nullptr;
  // The reduction leaves the main stream once the streaming kernel is long enough to hide it under (same threshold
  // as the build's second stream: below it the extra event traffic costs more host time than it returns).
  const bool side = moments && knob(ctx->knobs.side_reduce, (B * N >= (int64_t)512 << 10 && !short_call) ? 1 : 0) != 0;
  if (moments) {
    ring = &ctx->partial_ring[ctx->partial_seq++ % lynx_ctx::kPartialRing];
    if (ring->pending) {
      if (hipEventQuery(ring->reduced) != hipSuccess) HIP_TRY(ctx, hipEventSynchronize(ring->reduced));
      ring->pending = false;
    }
    const size_t need = (size_t)B * p.a.chunks * kPartialStride * sizeof(double);
    if ((rc = ensure_scratch(ctx, &ring->buf, &ring->bytes, need))) return rc;
    d_partials = (double*)ring->buf;
  }
  double* d_obs = nullptr;
  if (p.a.n_observers) {
    if (!d_observations) return fail(ctx, LYNX_ERR_INVALID, "the program has observer steps: d_observations required");
    const size_t need = (size_t)B * p.a.chunks * 2 * LYNX_MAX_OBSERVERS * sizeof(double);
    if ((rc = ensure_scratch(ctx, &ctx->scratch_obs, &ctx->scratch_obs_bytes, need))) return rc;
    d_obs = (double*)ctx->scratch_obs;
  }
  // the time stamp on the dispatch is what a later build on the second stream waits for before it reuses the
  // table slot; a call whose build ran on the main stream leaves it out (it costs host time: BASELINE config 2 is
  // bound by the host's enqueue rate) and marks the main stream dirty instead, which makes the next asynchronous
  // build wait for everything enqueued here
  if (tail) {  // this kernel announces its tail: one of the workgroups of its last round of dispatches
    const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    p.a.tail_flag = ctx->d_tail_flag;
    p.a.tail_seq = ++ctx->tail_seq;
    p.a.tail_wg = (int32_t)std::max<int64_t>(0, (int64_t)p.grid - 4 * cus);
  }
  // Every other LONG call walks the batch from its end: what the previous pass left in the 256 MB Infinity Cache -- the
  // end of its incoming beam -- is then what this one reads first, if it is the same beam again (the optimisation loop:
  // one incoming beam, new settings every step).  BASELINE config 3 at 8 M particles (448 MB in): 171 -> 161.5 us/step;
  // config 4 (2.87 GB in): 980.6 -> 972.  Same workgroups, same records, another order of dispatch.
  if (!use_units && ctx->knobs.alternate_order &&
      (ctx->knobs.alternate_order == 2 || (size_t)B * N * 7 * sizeof(T) >= ((size_t)256 << 20)))
    p.a.reversed = ctx->knobs.alternate_order == 2 ? 1 : (int32_t)(ctx->long_calls++ & 1u);
  p.done = (slot >= 0 && async_build) ? ctx->ev_streamed_own[slot] : nullptr;
  if (side && !p.done) p.done = ring->track_done;  // the side stream's reduction starts behind it
  if (use_units) {
    const UnitPlan& up = lat->units[p.a.merged_pairs ? 1 : 0];
    bool pair_form = ctx->knobs.unit_pairs != 0;
    for (int u = 0; u < up.n_units && pair_form; ++u) pair_form = up.cls[u] == kClassU && up.pair[u];
    pair_form = pair_form && p.unroll != 4 && (!moments || (p.mom_mode == 3 && !p.full_cov));  // (the forms that fit 88 registers)
    if (ctx->knobs.unit_pairs == 2 && !pair_form)
      rc = fail(ctx, LYNX_ERR_INVALID, "LYNX_UNIT_PAIRS=2: this call does not take the kernel for lattices of [run, cavity] pairs");
    else
      rc = launch_units(ctx, p, n_units, S, d_p_in, d_p_out, d_energy_out, d_steps, d_units, d_extras, d_partials, moments, pair_form);
  } else {
    rc = launch_direct<T>(ctx, p, lv, d_energy_in, d_p_in, d_p_out, d_energy_out, d_steps, d_partials, d_obs, moments);
  }
  if (rc) {
    if (tail) --ctx->tail_seq;  // nothing will announce this number: a later build must not wait for it
    return rc;
  }
  if (p.a.n_observers) {
    hipLaunchKernelGGL(k_reduce_observers, dim3((unsigned)B), dim3(64), 0, ctx->stream, d_obs, p.a.chunks,
                       p.a.n_observers, (double)N, d_observations);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (slot >= 0) {
    ctx->ev_streamed[slot] = ctx->last_stream_stop;
    ctx->streamed_valid[slot] = async_build && ctx->last_stream_stop != nullptr;
    if (!async_build) ctx->main_dirty = true;
  }
  if (d_energy_out) ctx->main_wrote = d_energy_out;  // a later build that reads it must wait for this kernel
  if (moments) {
    // records of the workgroups -> record of the sample.  Many samples: one 256-thread workgroup per sample
    // walks up to 70 rows, more rows go through a level of <= 64 groups first (rows_per_group grows with the
    // beam).  Few samples with a few hundred rows each (BASELINE configs 2 and 3): one 1024-thread workgroup
    // per sample stages 448 rows per pass -- one launch instead of two (LYNX_REDUCE_WIDE=0: the level form).
    hipStream_t rs = ctx->stream;
    if (side) {
      rs = ctx->s_side;
      HIP_TRY(ctx, hipStreamWaitEvent(rs, ctx->last_stream_stop, 0));
    } else {
      // in line: the side stream may still be reducing into a block the allocator has not seen freed; the main
      // stream's own order covers everything else
      ctx->side_wrote = nullptr;
    }
    int rows = p.a.chunks;
    const double* level_in = d_partials;
    const bool wide = B <= 4 && rows > kReduceStage && rows <= 3 * kReduceStageWide && ctx->knobs.reduce_wide == 1;
    if (wide) {
      constexpr size_t lds = reduce_lds_bytes<1024, kReduceStageWide>();
      if ((rc = allow_lds(ctx, k_reduce_moments<true, 1024, kReduceStageWide>, lds))) return rc;
      hipLaunchKernelGGL((k_reduce_moments<true, 1024, kReduceStageWide>), dim3((unsigned)B), dim3(1024), lds, rs,
                         level_in, rows, rows, 1, d_moments_out);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      constexpr size_t lds = reduce_lds_bytes<256, kReduceStage>();
      if (rows > kReduceStage) {
        const int rpg = std::max((rows + 63) / 64, B <= 4 ? 32 : 1);  // (few samples: no point in groups of a handful of rows)
        const int groups = (rows + rpg - 1) / rpg;
        const size_t need = (size_t)B * groups * kPartialStride * sizeof(double);
        // one buffer: its users follow each other on one stream (side or main), and a change of stream joins first
        if ((rc = ensure_scratch(ctx, &ctx->scratch_level, &ctx->scratch_level_bytes, need))) return rc;
        if (ctx->level_on_side != side) {
          if (side) {  // earlier in-line levels ran on the main stream: the side stream waits for them
            HIP_TRY(ctx, hipEventRecord(ctx->ev_main_mark, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_side, ctx->ev_main_mark, 0));
          } else if ((rc = join_side(ctx))) {
            return rc;
          }
          ctx->level_on_side = side;
        }
        if (ctx->knobs.reduce_ticket) {
          // both levels in one launch: the workgroup that draws a sample's last ticket adds its group records up
          if ((rc = ensure_tickets(ctx, (size_t)B))) return rc;
          hipLaunchKernelGGL((k_reduce_moments_ticket<256, kReduceStage>), dim3((unsigned)(B * groups)), dim3(256), lds, rs,
                             level_in, rows, rpg, groups, (double*)ctx->scratch_level, ctx->scratch_tickets, d_moments_out);
          HIP_TRY(ctx, hipGetLastError());
          rows = 0;
        } else {
          hipLaunchKernelGGL((k_reduce_moments<false, 256, kReduceStage>), dim3((unsigned)(B * groups)), dim3(256), lds, rs,
                             level_in, rows, rpg, groups, (double*)ctx->scratch_level);
          HIP_TRY(ctx, hipGetLastError());
          level_in = (const double*)ctx->scratch_level;
          rows = groups;
        }
      }
      if (rows > 0) {
        hipLaunchKernelGGL((k_reduce_moments<true, 256, kReduceStage>), dim3((unsigned)B), dim3(256), lds, rs, level_in,
                           rows, rows, 1, d_moments_out);
        HIP_TRY(ctx, hipGetLastError());
      }
    }
    if (side) {
      // one event: the ring slot's.  The host has seen its previous recording complete before this call took the
      // slot, so whatever side operation still carries it has finished and retires on the next look.
      {
        std::lock_guard<std::mutex> lock(ctx->mu);
        retire_side_ops(ctx, false);
      }
      HIP_TRY(ctx, hipEventRecord(ring->reduced, rs));
      ring->pending = true;
      side_op_issued(ctx, ring->reduced, d_moments_out, nullptr, false);
      ctx->side_wrote = d_moments_out;
    }
  }
  return LYNX_OK;
}

int lynx_track_particles(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles, const void* d_energy_in,
                         const void* d_p_in, void* d_p_out, void* d_energy_out, double* d_moments_out,
                         int flags, double* d_observations) {
  if (!ctx || !lat || !d_p_in) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (n_particles <= 0) return fail(ctx, LYNX_ERR_INVALID, "n_particles must be > 0");
  if (lat->n_steps > 0 && !d_energy_in) return fail(ctx, LYNX_ERR_INVALID, "energy_in required");
  if ((flags & (LYNX_TRACK_MOMENTS | LYNX_TRACK_COVARIANCE)) && !d_moments_out)
    return fail(ctx, LYNX_ERR_INVALID, "LYNX_TRACK_MOMENTS / LYNX_TRACK_COVARIANCE need d_moments_out");
  if ((flags & LYNX_TRACK_SHARED_INPUT) && d_p_in == d_p_out)
    return fail(ctx, LYNX_ERR_INVALID, "a shared incoming beam cannot be tracked in place");
  HIP_TRY(ctx, use_device(ctx));
  LatticeDev lv = dev_view(lat);
  const int rc = lat->dtype == LYNX_F64
                     ? track_particles_t<double>(ctx, lat, lv, lat->batch, n_particles, d_energy_in, d_p_in, d_p_out,
                                                 d_energy_out, d_moments_out, flags, d_observations)
                     : track_particles_t<float>(ctx, lat, lv, lat->batch, n_particles, d_energy_in, d_p_in, d_p_out,
                                                d_energy_out, d_moments_out, flags, d_observations);
  ctx->main_idle = false;  // (whatever allocation inside may have waited for the stream: work follows it now)
  return rc;
}

int lynx_track_particles_new(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles, const void* d_energy_in,
                             const void* d_p_in, int flags, int32_t want_energy_out, void** d_out /* [4] */) {
  if (!ctx || !lat || !d_out) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  for (int k = 0; k < 4; ++k) d_out[k] = nullptr;
  if (n_particles <= 0) return fail(ctx, LYNX_ERR_INVALID, "n_particles must be > 0");
  const size_t es = dtype_size(lat->dtype);
  const bool moments = (flags & (LYNX_TRACK_MOMENTS | LYNX_TRACK_COVARIANCE)) != 0;
  int rc = ctx_alloc(ctx, (size_t)lat->batch * (size_t)n_particles * 7 * es, &d_out[0]);
  if (!rc && want_energy_out) rc = ctx_alloc(ctx, (size_t)lat->batch * es, &d_out[1]);
  if (!rc && moments) {
    // a few samples' records, one process: where the host reads them without a copy command (many samples, or records
    // that RCCL gathers: device memory)
    const size_t bytes = (size_t)lat->batch * LYNX_MOMENT_STRIDE * sizeof(double);
    if (bytes <= kHostVisibleMax && ctx->comm_ranks == 0 && ctx->knobs.host_visible_records) rc = ctx_alloc_host_visible(ctx, bytes, &d_out[2]);
    else rc = ctx_alloc(ctx, bytes, &d_out[2]);
  }
  if (!rc && lat->n_observers > 0) rc = ctx_alloc(ctx, (size_t)lat->batch * lat->n_observers * 2 * sizeof(double), &d_out[3]);
  if (!rc)
    rc = lynx_track_particles(ctx, lat, n_particles, d_energy_in, d_p_in, d_out[0], d_out[1], (double*)d_out[2], flags,
                              (double*)d_out[3]);
  if (rc) {  // nothing was enqueued on these blocks (every check of lynx_track_particles comes before its first launch ...
    const std::string why = ctx->err;
    (void)sync_main(ctx);  // ... but a failure halfway through a launch sequence may have: drain first)
    for (int k = 0; k < 4; ++k) {
      if (d_out[k]) (void)ctx_free(ctx, d_out[k]);
      d_out[k] = nullptr;
    }
    ctx->err = why;
  }
  return rc;
}

// ---- reverse pass -----------------------------------------------------------------------

// k_build_bwd's task list for this lattice: (element, parameter | energy) pairs kind by kind, every kind padded to whole
// waves of 64 (lynx_grad.hpp).  The kinds of a lattice never change: made once.
static int ensure_bwd_tasks(lynx_ctx* ctx, lynx_lattice* lat) {
  if (lat->d_bwd_tasks) return LYNX_OK;
  std::vector<unsigned short> tasks;
  if (lat->n_elems >= 4096) return fail(ctx, LYNX_ERR_INVALID, "reverse pass: at most 4095 elements per program");
  for (int k = 1; k <= LYNX_KIND_UNDULATOR; ++k) {
    const int np = bwd_kind_params(k);
    if (np == 0) continue;
    const size_t first = tasks.size();
    for (int32_t e = 0; e < lat->n_elems; ++e)
      if (lat->h_elems[e].kind == k)
        for (int q = 0; q <= np; ++q) tasks.push_back((unsigned short)(e * 16 + (q == np ? kGradParams : q)));
    if (tasks.size() > first)
      while (tasks.size() & 63) tasks.push_back(0xffffu);  // a wave never mixes kinds
  }
  if (tasks.empty()) tasks.assign(64, 0xffffu);
  int rc = ctx_alloc(ctx, tasks.size() * sizeof(unsigned short), (void**)&lat->d_bwd_tasks);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(lat->d_bwd_tasks, tasks.data(), tasks.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  lat->n_bwd_tasks = (int32_t)tasks.size();
  return LYNX_OK;
}

constexpr int kBwdWgsPerCu = 24;                     // workgroups per CU of the reverse streaming kernels
constexpr size_t kBwdMapsLdsBytes = (size_t)40 << 10;  // k_build_bwd keeps maps and prefix products in LDS up to this

template <typename T, typename Z = T>
static int track_backward_t(lynx_ctx* ctx, lynx_lattice* lat, int64_t N, const void* d_energy_in, const void* d_p_in,
                            const double* d_moments_fwd, const double* d_grad_moments, void* d_grad_params,
                            void* d_grad_energy_in, void* d_grad_p_in, const double* d_grad_observations) {
  const int64_t B = lat->batch;
  const int32_t S = lat->n_steps, E = lat->n_elems;
  int rc;
  // (k_build_bwd reads the parameters: from its arguments if the lattice's pool travels there, from memory otherwise)
  const bool pool_in_args = lat->prog_pool && ctx->knobs.inline_pool != 0;
  if (!pool_in_args && (rc = sync_pool(ctx, lat))) return rc;
  // forward step tables; packed float32 pairs walk [run, cavity] pairs in the merged form of the forward
  // kernel (LYNX_BWD_MERGE=0: every step on its own), k_build_bwd takes the cotangents apart again
  constexpr int W = LaneOf<Z>::W;
  BwdArgs a;
  int merged = 0;
  if (W == 2 && ctx->knobs.bwd_merge)
    for (int32_t s = 1; s < S; ++s)
      merged |= lat->h_steps[s].kind == LYNX_STEP_CAVITY && lat->h_steps[s - 1].kind == LYNX_STEP_RUN &&
                !(lat->h_steps[s - 1].flags & LYNX_STEP_FLAG_OBSERVE);
  a.n_units = 0;
  a.n_observers = 0;
  a.out_chunks = 0;
  a.n_work = 0;
  a.leave_odd = ctx->knobs.bwd_units == 2 ? 1 : 0;
  for (int32_t s = 0; s < S; ++s) {
    const bool pair = merged && s + 1 < S && lat->h_steps[s].kind == LYNX_STEP_RUN &&
                      lat->h_steps[s + 1].kind == LYNX_STEP_CAVITY && !(lat->h_steps[s].flags & LYNX_STEP_FLAG_OBSERVE);
    if (pair) ++s;
    if (a.n_units >= kBwdGroup * kBwdMaxGroups || s > 255)
      return fail(ctx, LYNX_ERR_INVALID,
                  "lynx_track_particles_backward: " + std::to_string(S) + " steps; this version parks at most " +
                      std::to_string(kBwdGroup * kBwdMaxGroups) + " (merge skippable elements or split the lattice)");
    a.unit_observer[a.n_units] = 0;
    if (lat->h_steps[s].flags & LYNX_STEP_FLAG_OBSERVE) a.unit_observer[a.n_units] = (unsigned char)(++a.n_observers);
    a.unit_slot[a.n_units++] = (unsigned char)s;
  }
  for (int u = a.n_units; u < kBwdGroup * kBwdMaxGroups; ++u) a.unit_slot[u] = a.unit_observer[u] = 0;
  const size_t steps_bytes = (size_t)B * S * LYNX_STEP_STRIDE * sizeof(T);
  if ((rc = ensure_scratch(ctx, &ctx->scratch_steps[lynx_ctx::kTableBwd], &ctx->scratch_steps_bytes[lynx_ctx::kTableBwd], steps_bytes))) return rc;
  // float32 packed pairs: samples whose units all have class U take the structured reverse kernel (lynx_grad_units.hpp)
  float* d_units = nullptr;
  float* d_extras = nullptr;
  bool all_proposed_u = false;  // the plan proposes class U for every unit: only a sample with a non-finite map is left to k_track_bwd
  if constexpr (W == 2) {
    if (ctx->knobs.bwd_units) {
      if ((rc = ensure_units_plan(ctx, lat, merged != 0))) return rc;
      const UnitPlan& up = lat->units[merged ? 1 : 0];
      // (the structured kernel parks the (s, delta) entering every unit in registers: up to kBwdUnitsMax units)
      bool same = lat->units_ok[merged ? 1 : 0] && up.n_units == a.n_units && a.n_units <= kBwdUnitsMax;
      for (int u = 0; same && u < a.n_units; ++u) same = up.slot[u] == a.unit_slot[u];
      if (same) {
        const int64_t n = B * a.n_units;
        if ((rc = ensure_scratch(ctx, &ctx->scratch_units_bwd[0], &ctx->scratch_units_bwd_bytes[0], (size_t)n * kUnitStride * sizeof(float))) ||
            (rc = ensure_scratch(ctx, &ctx->scratch_units_bwd[1], &ctx->scratch_units_bwd_bytes[1], (size_t)n * kUnitExtraStride * sizeof(float))))
          return rc;
        d_units = (float*)ctx->scratch_units_bwd[0];
        d_extras = (float*)ctx->scratch_units_bwd[1];
        all_proposed_u = true;
        for (int u = 0; u < up.n_units; ++u) all_proposed_u = all_proposed_u && up.cls[u] == kClassU;
      }
    }
  }
  // the forward call this reverse pass belongs to has built exactly this table (and these unit records) a moment ago:
  // read them where they are -- and make the slot's next build wait for this reverse pass too
  const lynx_ctx::FwdTable& ft = ctx->fwd_table;
  const bool reuse = ctx->knobs.bwd_reuse_table && ft.valid && ft.lat == lat && ft.version == lat->version &&
                     ft.energy == d_energy_in && ft.seq == ctx->seq && ft.merged == merged &&
                     ctx->scratch_steps_bytes[ft.slot] >= steps_bytes;
  const void* d_table = ctx->scratch_steps[lynx_ctx::kTableBwd];
  if (reuse) {
    d_table = ctx->scratch_steps[ft.slot];
    if (d_units && ft.units) {
      d_units = (float*)ctx->scratch_units[ft.slot];
      d_extras = (float*)ctx->scratch_units[lynx_ctx::kTableSlots + ft.slot];
    } else if (d_units) {  // the forward call walked the table itself (a single-map program): the records from its rows
      if constexpr (sizeof(T) == 4) {
        const UnitPlan& up = lat->units[merged ? 1 : 0];
        const int64_t n = B * up.n_units;
        hipLaunchKernelGGL(k_pack_units, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, up, B, S,
                           (const float*)d_table, d_units, d_extras);
        HIP_TRY(ctx, hipGetLastError());
      }
    }
  } else if ((rc = launch_build<T>(ctx, lat, ctx->stream, d_energy_in, ctx->scratch_steps[lynx_ctx::kTableBwd], nullptr, merged, false,
                                   d_units, d_extras))) {
    return rc;
  }
  ctx->main_dirty = true;

  // Z: what a lane carries -- one particle, or (float32) two as a packed pair
  using Geo = ExGeom<T, W>;
  const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const int64_t ntiles = (N + kTrackThreads * W - 1) / (kTrackThreads * W);
  int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(ntiles, ((int64_t)kBwdWgsPerCu * cus + B - 1) / B));
  const int64_t tpw = (ntiles + chunks - 1) / chunks;
  chunks = (ntiles + tpw - 1) / tpw;
  a.n_particles = N;
  a.chunks = (int32_t)chunks;
  a.tiles_per_wg = (int32_t)tpw;
  const size_t lds = ((size_t)4 * ExRows<W>::value * Geo::kPitch + (size_t)4 * S * 64) * sizeof(T);
  if (lds > 160 * 1024) return fail(ctx, LYNX_ERR_INVALID, "lynx_track_particles_backward: LDS budget exceeded");
  if ((rc = allow_lds(ctx, k_track_bwd<T, Z>, lds))) return rc;
  if ((int64_t)B * chunks > 0x7fffffffLL || (int64_t)B * S > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "grid too large");
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[0], &ctx->scratch_grad_bytes[0],
                           (size_t)B * chunks * S * kGradStride * sizeof(T))))
    return rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[1], &ctx->scratch_grad_bytes[1], (size_t)B * S * kGradStride * sizeof(T))))
    return rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[2], &ctx->scratch_grad_bytes[2],
                           (size_t)B * (2 * E + S + 1) * 49 * sizeof(T))))
    return rc;
  LatticeDev lv = dev_view(lat);
  if constexpr (W == 2) {
    if (d_units) {
      const size_t lds_u = bwd_units_lds_bytes(S);
      if (a.n_units <= 8) {
        if ((rc = allow_lds(ctx, k_track_bwd_units<8>, lds_u))) return rc;
        hipLaunchKernelGGL(k_track_bwd_units<8>, dim3((unsigned)(B * chunks)), dim3(kTrackThreads), lds_u, ctx->stream, a, S,
                           (const float*)d_p_in, (const float*)d_units, (const float*)d_extras, d_moments_fwd, d_grad_moments,
                           (float*)ctx->scratch_grad[0], (float*)d_grad_p_in);
      } else {
        if ((rc = allow_lds(ctx, k_track_bwd_units<kBwdUnitsMax>, lds_u))) return rc;
        hipLaunchKernelGGL(k_track_bwd_units<kBwdUnitsMax>, dim3((unsigned)(B * chunks)), dim3(kTrackThreads), lds_u, ctx->stream,
                           a, S, (const float*)d_p_in, (const float*)d_units, (const float*)d_extras, d_moments_fwd,
                           d_grad_moments, (float*)ctx->scratch_grad[0], (float*)d_grad_p_in);
      }
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  // Next to k_track_bwd_units this kernel is there for the samples that one left: workgroups of the others return at once --
  // 40 us for BASELINE config 5's 40 960 of them, which take 60 KB of LDS each to be placed.  When the plan proposes class
  // U for EVERY unit, only a sample with a non-finite map can be left: ONE workgroup per sample then (it walks all of
  // the sample's tiles, writes row 0 of the sample's partial sums and clears the rest).
  BwdArgs a_dense = a;
  a_dense.n_work = (int32_t)(B * chunks);
  int64_t dense_grid = B * chunks;
  if (d_units && all_proposed_u) {
    if (chunks > 1) {
      a_dense.out_chunks = (int32_t)chunks;
      a_dense.chunks = 1;
      a_dense.tiles_per_wg = (int32_t)ntiles;
    }
    a_dense.n_work = (int32_t)B;
    dense_grid = std::min<int64_t>(B, 2 * cus);  // (and as many workgroups as are resident at once: they look at B / grid samples each)
  }
  hipLaunchKernelGGL((k_track_bwd<T, Z>), dim3((unsigned)dense_grid), dim3(kTrackThreads), lds, ctx->stream, lv, a_dense,
                     (const T*)d_p_in, (const T*)d_table, d_moments_fwd, d_grad_moments,
                     (T*)ctx->scratch_grad[0], (T*)d_grad_p_in, (const float*)d_units, kUnitStride, kUnitClassShift, (int)kClassU,
                     d_grad_observations);
  HIP_TRY(ctx, hipGetLastError());
  if (chunks >= 16) {
    hipLaunchKernelGGL(k_reduce_tbar<T>, dim3((unsigned)(B * S)), dim3(256), 0, ctx->stream, (const T*)ctx->scratch_grad[0],
                       (int)chunks, (int)S, (T*)ctx->scratch_grad[1]);
  } else {  // four rows (sample, step) per wave
    const int64_t rows = B * S;
    hipLaunchKernelGGL((k_reduce_tbar_rows<T, 4>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, ctx->stream,
                       (const T*)ctx->scratch_grad[0], (int)chunks, (int)S, rows, (T*)ctx->scratch_grad[1]);
  }
  HIP_TRY(ctx, hipGetLastError());
  if constexpr (W == 2) {
    if (d_units) {  // class-U samples: the units' transverse blocks from the sample's S_x, S_y (lynx_grad_units.hpp)
      hipLaunchKernelGGL(k_finish_tbar_units, dim3((unsigned)B), dim3(64), 0, ctx->stream, a, S, (const float*)d_units,
                         (float*)ctx->scratch_grad[1]);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  if (reuse) {  // the forward call's table slot has been read up to here: its next build waits for this, too
    HIP_TRY(ctx, hipEventRecord(ctx->ev_streamed_own[ft.slot], ctx->stream));
    ctx->ev_streamed[ft.slot] = ctx->ev_streamed_own[ft.slot];
    ctx->streamed_valid[ft.slot] = true;
  }
  size_t lds2 = build_bwd_lds_fixed<T>(S, E);
  // maps + prefix products, then the kind-sorted task list and the elements' kinds (unsigned short each)
  const size_t maps_bytes = (size_t)(2 * E + S + 1) * 49 * sizeof(T);
  if ((rc = ensure_bwd_tasks(ctx, lat))) return rc;
  const int maps_in_lds = lds2 + maps_bytes <= kBwdMapsLdsBytes;
  if (maps_in_lds) lds2 += maps_bytes;
  HIP_TRY(ctx, hipMemsetAsync(d_grad_params, 0, (size_t)B * E * kGradParams * sizeof(T), ctx->stream));
  if (pool_in_args && (size_t)lat->pool_count * sizeof(T) <= (size_t)kInlinePoolSmall) {
    if ((rc = allow_lds(ctx, k_build_bwd_inline<T, kInlinePoolSmall>, lds2))) return rc;
    hipLaunchKernelGGL((k_build_bwd_inline<T, kInlinePoolSmall>), dim3((unsigned)B), dim3(256), lds2, ctx->stream,
                       *reinterpret_cast<const InlinePool<kInlinePoolSmall>*>(&lat->h_pool), lv, (const T*)d_energy_in,
                       (T*)ctx->scratch_grad[1], (T*)ctx->scratch_grad[2], (T*)d_grad_params, (T*)d_grad_energy_in, merged,
                       maps_in_lds, lat->d_bwd_tasks, lat->n_bwd_tasks);
  } else if (pool_in_args) {
    if ((rc = allow_lds(ctx, k_build_bwd_inline<T, kInlinePoolLarge>, lds2))) return rc;
    hipLaunchKernelGGL((k_build_bwd_inline<T, kInlinePoolLarge>), dim3((unsigned)B), dim3(256), lds2, ctx->stream, lat->h_pool, lv,
                       (const T*)d_energy_in, (T*)ctx->scratch_grad[1], (T*)ctx->scratch_grad[2], (T*)d_grad_params,
                       (T*)d_grad_energy_in, merged, maps_in_lds, lat->d_bwd_tasks, lat->n_bwd_tasks);
  } else {
    if ((rc = allow_lds(ctx, k_build_bwd<T>, lds2))) return rc;
    hipLaunchKernelGGL(k_build_bwd<T>, dim3((unsigned)B), dim3(256), lds2, ctx->stream, lv, (const T*)d_energy_in,
                       (T*)ctx->scratch_grad[1], (T*)ctx->scratch_grad[2], (T*)d_grad_params, (T*)d_grad_energy_in, merged,
                       maps_in_lds, lat->d_bwd_tasks, lat->n_bwd_tasks);
  }
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_track_particles_backward(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles, const void* d_energy_in,
                                  const void* d_p_in, const double* d_moments_fwd, const double* d_grad_moments,
                                  void* d_grad_params, void* d_grad_energy_in, void* d_grad_p_in,
                                  const double* d_grad_observations) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !lat || !d_energy_in || !d_p_in || !d_moments_fwd || !d_grad_moments || !d_grad_params || !d_grad_energy_in)
    return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (n_particles <= 0 || lat->n_steps <= 0) return fail(ctx, LYNX_ERR_INVALID, "empty program or beam");
  if (d_grad_observations && lat->n_observers == 0)
    return fail(ctx, LYNX_ERR_INVALID, "d_grad_observations given, but the program has no observer step");
  HIP_TRY(ctx, use_device(ctx));
  {
    const int rc = join_side(ctx);  // d_moments_fwd is what a reduction on the side stream writes
    if (rc) return rc;
  }
  return lat->dtype == LYNX_F64
             ? track_backward_t<double>(ctx, lat, n_particles, d_energy_in, d_p_in, d_moments_fwd, d_grad_moments,
                                        d_grad_params, d_grad_energy_in, d_grad_p_in, d_grad_observations)
         : ctx->knobs.bwd_pairs
             ? track_backward_t<float, lynx_f32x2>(ctx, lat, n_particles, d_energy_in, d_p_in, d_moments_fwd,
                                                   d_grad_moments, d_grad_params, d_grad_energy_in, d_grad_p_in,
                                                   d_grad_observations)
             : track_backward_t<float>(ctx, lat, n_particles, d_energy_in, d_p_in, d_moments_fwd, d_grad_moments,
                                       d_grad_params, d_grad_energy_in, d_grad_p_in, d_grad_observations);
}

template <typename T>
static int moments_backward_t(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in, const void* d_mu_in,
                              const void* d_cov_in, const void* d_mu_bar, const void* d_cov_bar, void* d_grad_params,
                              void* d_grad_energy_in, void* d_grad_mu_in, void* d_grad_cov_in) {
  const int64_t B = lat->batch;
  const int32_t S = lat->n_steps, E = lat->n_elems;
  int rc;
  if ((rc = sync_pool(ctx, lat))) return rc;  // (k_moments_bwd reads the parameters from memory)
  if ((rc = ensure_scratch(ctx, &ctx->scratch_steps[lynx_ctx::kTableBwd], &ctx->scratch_steps_bytes[lynx_ctx::kTableBwd], (size_t)B * S * LYNX_STEP_STRIDE * sizeof(T))))
    return rc;
  if ((rc = launch_build<T>(ctx, lat, ctx->stream, d_energy_in, ctx->scratch_steps[lynx_ctx::kTableBwd], nullptr))) return rc;
  ctx->main_dirty = true;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[0], &ctx->scratch_grad_bytes[0], (size_t)B * (S + 1) * 56 * sizeof(T))))
    return rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[1], &ctx->scratch_grad_bytes[1], (size_t)B * S * kGradStride * sizeof(T))))
    return rc;
  if ((rc = ensure_scratch(ctx, &ctx->scratch_grad[2], &ctx->scratch_grad_bytes[2],
                           (size_t)B * (2 * E + S + 1) * 49 * sizeof(T))))
    return rc;
  LatticeDev lv = dev_view(lat);
  hipLaunchKernelGGL(k_moments_bwd<T>, dim3((unsigned)B), dim3(64), 0, ctx->stream, lv, (const T*)ctx->scratch_steps[lynx_ctx::kTableBwd],
                     (const T*)d_mu_in, (const T*)d_cov_in, (const T*)d_mu_bar, (const T*)d_cov_bar,
                     (T*)ctx->scratch_grad[0], (T*)ctx->scratch_grad[1], (T*)d_grad_mu_in, (T*)d_grad_cov_in);
  HIP_TRY(ctx, hipGetLastError());
  size_t lds2 = build_bwd_lds_fixed<T>(S, E);
  // maps + prefix products, then the kind-sorted task list and the elements' kinds (unsigned short each)
  const size_t maps_bytes = (size_t)(2 * E + S + 1) * 49 * sizeof(T);
  if ((rc = ensure_bwd_tasks(ctx, lat))) return rc;
  const int maps_in_lds = lds2 + maps_bytes <= kBwdMapsLdsBytes;
  if (maps_in_lds) lds2 += maps_bytes;
  if ((rc = allow_lds(ctx, k_build_bwd<T>, lds2))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(d_grad_params, 0, (size_t)B * E * kGradParams * sizeof(T), ctx->stream));
  hipLaunchKernelGGL(k_build_bwd<T>, dim3((unsigned)B), dim3(256), lds2, ctx->stream, lv, (const T*)d_energy_in,
                     (T*)ctx->scratch_grad[1], (T*)ctx->scratch_grad[2], (T*)d_grad_params, (T*)d_grad_energy_in, 0,
                     maps_in_lds, lat->d_bwd_tasks, lat->n_bwd_tasks);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_track_moments_backward(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in, const void* d_mu_in,
                                const void* d_cov_in, const void* d_mu_bar, const void* d_cov_bar,
                                void* d_grad_params, void* d_grad_energy_in, void* d_grad_mu_in,
                                void* d_grad_cov_in) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !lat || !d_energy_in || !d_mu_in || !d_cov_in || !d_mu_bar || !d_cov_bar || !d_grad_params ||
      !d_grad_energy_in || !d_grad_mu_in || !d_grad_cov_in)
    return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (lat->n_steps <= 0) return fail(ctx, LYNX_ERR_INVALID, "empty program");
  if (lat->batch > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "batch too large");
  HIP_TRY(ctx, use_device(ctx));
  return lat->dtype == LYNX_F64
             ? moments_backward_t<double>(ctx, lat, d_energy_in, d_mu_in, d_cov_in, d_mu_bar, d_cov_bar, d_grad_params,
                                          d_grad_energy_in, d_grad_mu_in, d_grad_cov_in)
             : moments_backward_t<float>(ctx, lat, d_energy_in, d_mu_in, d_cov_in, d_mu_bar, d_cov_bar, d_grad_params,
                                         d_grad_energy_in, d_grad_mu_in, d_grad_cov_in);
}

int lynx_moments(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                 double* d_moments_out, int32_t covariance) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_p || !d_moments_out) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (batch <= 0 || n_particles <= 0) return fail(ctx, LYNX_ERR_INVALID, "bad shape");
  HIP_TRY(ctx, use_device(ctx));
  LatticeDev lv;
  memset(&lv, 0, sizeof(lv));
  lv.batch = batch;
  return dtype == LYNX_F64
             ? track_particles_t<double>(ctx, nullptr, lv, batch, n_particles, nullptr, d_p, nullptr, nullptr,
                                         d_moments_out, covariance ? LYNX_TRACK_COVARIANCE : LYNX_TRACK_MOMENTS)
             : track_particles_t<float>(ctx, nullptr, lv, batch, n_particles, nullptr, d_p, nullptr, nullptr,
                                        d_moments_out, covariance ? LYNX_TRACK_COVARIANCE : LYNX_TRACK_MOMENTS);
}

// ---- ParameterBeam ---------------------------------------------------------------------

template <typename T>
static int launch_track_moments(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in, const void* d_mu_in,
                                const void* d_cov_in, void* d_mu_out, void* d_cov_out, void* d_energy_out) {
  int rc;
  if constexpr (sizeof(T) == 4)  // (float64: 98 VGPRs of covariance per lane -- the compiler spills; workgroup form below)
  if (lat->n_steps > 0 && lat->batch >= ctx->knobs.lanes_build_min_batch) {
    // large batches: lanes = samples all the way (step table from the lanes build, then one lane per sample)
    const size_t need = (size_t)lat->batch * lat->n_steps * LYNX_STEP_STRIDE * sizeof(T);
    if ((rc = ensure_scratch(ctx, &ctx->scratch_steps[lynx_ctx::kTablePb], &ctx->scratch_steps_bytes[lynx_ctx::kTablePb], need))) return rc;
    if ((rc = launch_build<T>(ctx, lat, ctx->stream, d_energy_in, ctx->scratch_steps[lynx_ctx::kTablePb], nullptr, 0))) return rc;
    hipLaunchKernelGGL(k_apply_moments_lanes<T>, dim3((unsigned)((lat->batch + 63) / 64)), dim3(64), apply_moments_lds<T>(),
                       ctx->stream, dev_view(lat), (const T*)ctx->scratch_steps[lynx_ctx::kTablePb], (const T*)d_mu_in, (const T*)d_cov_in,
                       (T*)d_mu_out, (T*)d_cov_out, (T*)d_energy_out);
    HIP_TRY(ctx, hipGetLastError());
    return LYNX_OK;
  }
  // one workgroup per sample: one wave for big batches, four otherwise -- with chunks of up to 128 elements when the
  // batch leaves most of the GPU idle and the depth of the tree is what the call waits for (build_shape)
  const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  const bool deep = lat->batch * 2 <= cus;
  const unsigned threads = lat->batch <= 4096 ? 256u : 64u;
  const int chunk = build_chunk(lat->n_elems, deep ? 128 : 64);
  const size_t lds = build_scratch_bytes(chunk, sizeof(T)) +
                     ((size_t)lat->n_steps * LYNX_STEP_STRIDE + lat->n_steps + 1 + 8 + 49 + 49 + 16) * sizeof(T);
  if (lat->prog_pool && ctx->knobs.inline_pool) {  // (no cavities: no flags to evaluate)
    if ((size_t)lat->pool_count * sizeof(T) <= (size_t)kInlinePoolSmall) {
      if ((rc = allow_lds(ctx, k_track_moments_inline<T, kInlinePoolSmall>, lds))) return rc;
      hipLaunchKernelGGL((k_track_moments_inline<T, kInlinePoolSmall>), dim3((unsigned)lat->batch), dim3(threads), lds, ctx->stream,
                         *reinterpret_cast<const InlinePool<kInlinePoolSmall>*>(&lat->h_pool), dev_view(lat), (const T*)d_energy_in,
                         (const T*)d_mu_in, (const T*)d_cov_in, (T*)d_mu_out, (T*)d_cov_out, (T*)d_energy_out, chunk);
    } else {
      if ((rc = allow_lds(ctx, k_track_moments_inline<T, kInlinePoolLarge>, lds))) return rc;
      hipLaunchKernelGGL((k_track_moments_inline<T, kInlinePoolLarge>), dim3((unsigned)lat->batch), dim3(threads), lds, ctx->stream,
                         lat->h_pool, dev_view(lat), (const T*)d_energy_in, (const T*)d_mu_in, (const T*)d_cov_in, (T*)d_mu_out,
                         (T*)d_cov_out, (T*)d_energy_out, chunk);
    }
    HIP_TRY(ctx, hipGetLastError());
    return LYNX_OK;
  }
  if ((rc = sync_pool(ctx, lat)) || (rc = allow_lds(ctx, k_track_moments<T>, lds))) return rc;
  if ((rc = launch_cavity_flags<T>(ctx, lat, ctx->stream, d_energy_in))) return rc;
  hipLaunchKernelGGL(k_track_moments<T>, dim3((unsigned)lat->batch), dim3(threads), lds, ctx->stream, dev_view(lat),
                     (const T*)d_energy_in, (const T*)d_mu_in, (const T*)d_cov_in, (T*)d_mu_out, (T*)d_cov_out,
                     (T*)d_energy_out, chunk);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_track_moments(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in, const void* d_mu_in,
                       const void* d_cov_in, void* d_mu_out, void* d_cov_out, void* d_energy_out) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (ctx && d_energy_out) ctx->fwd_table.written(d_energy_out, 1);
  if (!ctx || !lat || !d_energy_in || !d_mu_in || !d_cov_in || !d_mu_out || !d_cov_out)
    return fail(ctx, LYNX_ERR_INVALID, "null argument");
  if (lat->batch > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "batch too large");
  HIP_TRY(ctx, use_device(ctx));
  ctx->main_dirty = true;
  return lat->dtype == LYNX_F64
             ? launch_track_moments<double>(ctx, lat, d_energy_in, d_mu_in, d_cov_in, d_mu_out, d_cov_out, d_energy_out)
             : launch_track_moments<float>(ctx, lat, d_energy_in, d_mu_in, d_cov_in, d_mu_out, d_cov_out, d_energy_out);
}

// ---- screen read-out -------------------------------------------------------------------------

int lynx_histogram2d(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                     const void* d_xedges, const void* d_yedges, int32_t nx, int32_t ny, int32_t* d_image) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_p || !d_xedges || !d_yedges || !d_image || batch <= 0 || n_particles <= 0 || nx <= 0 || ny <= 0)
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const size_t es = dtype_size(dtype);
  const size_t lds = ((size_t)nx + ny + 2) * es;
  if (lds > 64 * 1024) return fail(ctx, LYNX_ERR_INVALID, "screen resolution too large for the edge table");
  HIP_TRY(ctx, hipMemsetAsync(d_image, 0, (size_t)batch * nx * ny * sizeof(int32_t), ctx->stream));
  const int64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  int64_t chunks = std::max<int64_t>(1, std::min<int64_t>((n_particles + 1023) / 1024, (8 * cus + batch - 1) / batch));
  if (batch * chunks > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "grid too large");
  if (dtype == LYNX_F64)
    hipLaunchKernelGGL(k_histogram2d<double>, dim3((unsigned)(batch * chunks)), dim3(256), lds, ctx->stream,
                       (const double*)d_p, n_particles, (int)chunks, (const double*)d_xedges, (const double*)d_yedges,
                       nx, ny, d_image);
  else
    hipLaunchKernelGGL(k_histogram2d<float>, dim3((unsigned)(batch * chunks)), dim3(256), lds, ctx->stream,
                       (const float*)d_p, n_particles, (int)chunks, (const float*)d_xedges, (const float*)d_yedges, nx,
                       ny, d_image);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_diag_phase_trig(lynx_ctx* ctx, int64_t n, const float* d_x, int32_t packed, float* d_sin, float* d_cos) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_x || !d_sin || !d_cos || n <= 0) return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const int64_t threads = (n + 1) / 2;
  hipLaunchKernelGGL(k_diag_phase_trig, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_x, n,
                     (int)packed, d_sin, d_cos);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

// ---- aperture ----------------------------------------------------------------------------

int lynx_aperture_mask(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                       const void* d_x_max, const void* d_y_max, int32_t param_stride, int32_t elliptical,
                       unsigned char* d_mask, int32_t* d_counts, int64_t* d_offsets, int64_t* d_totals) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_p || !d_x_max || !d_y_max || !d_mask || !d_counts || !d_offsets || !d_totals || batch <= 0 ||
      n_particles <= 0 || (param_stride != 0 && param_stride != 1))
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const int64_t chunks = (n_particles + kApertureChunk - 1) / kApertureChunk;
  if (batch * chunks > 0x7fffffffLL) return fail(ctx, LYNX_ERR_INVALID, "grid too large");
  if (dtype == LYNX_F64)
    hipLaunchKernelGGL(k_aperture_mask<double>, dim3((unsigned)(batch * chunks)), dim3(256), 0, ctx->stream,
                       (const double*)d_p, n_particles, (int)chunks, (const double*)d_x_max, (const double*)d_y_max,
                       (int)param_stride, (int)elliptical, d_mask, d_counts);
  else
    hipLaunchKernelGGL(k_aperture_mask<float>, dim3((unsigned)(batch * chunks)), dim3(256), 0, ctx->stream,
                       (const float*)d_p, n_particles, (int)chunks, (const float*)d_x_max, (const float*)d_y_max,
                       (int)param_stride, (int)elliptical, d_mask, d_counts);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_aperture_scan, dim3((unsigned)batch), dim3(256), 0, ctx->stream, d_counts, (int)chunks, d_offsets,
                     d_totals);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_aperture_compact(lynx_ctx* ctx, int dtype, int64_t n_particles, const void* d_p, const unsigned char* d_mask,
                          const int64_t* d_offsets, void* d_kept, void* d_lost) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_p || !d_mask || !d_offsets || !d_kept || !d_lost || n_particles <= 0)
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const int64_t chunks = (n_particles + kApertureChunk - 1) / kApertureChunk;
  if (dtype == LYNX_F64)
    hipLaunchKernelGGL(k_aperture_compact<double>, dim3((unsigned)chunks), dim3(256), 0, ctx->stream, (const double*)d_p,
                       n_particles, d_mask, d_offsets, (double*)d_kept, (double*)d_lost);
  else
    hipLaunchKernelGGL(k_aperture_compact<float>, dim3((unsigned)chunks), dim3(256), 0, ctx->stream, (const float*)d_p,
                       n_particles, d_mask, d_offsets, (float*)d_kept, (float*)d_lost);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

int lynx_gaussian_image(lynx_ctx* ctx, int dtype, int64_t batch, const void* d_mu, const void* d_cov,
                        const void* d_xs, const void* d_ys, int32_t nx, int32_t ny, void* d_image) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_mu || !d_cov || !d_xs || !d_ys || !d_image || batch <= 0 || nx <= 0 || ny <= 0 || batch > 65535)
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const unsigned gx = (unsigned)(((int64_t)nx * ny + 255) / 256);
  if (dtype == LYNX_F64)
    hipLaunchKernelGGL(k_gaussian_image<double>, dim3(gx, (unsigned)batch), dim3(256), 0, ctx->stream,
                       (const double*)d_mu, (const double*)d_cov, (const double*)d_xs, (const double*)d_ys, nx, ny,
                       (double*)d_image);
  else
    hipLaunchKernelGGL(k_gaussian_image<float>, dim3(gx, (unsigned)batch), dim3(256), 0, ctx->stream,
                       (const float*)d_mu, (const float*)d_cov, (const float*)d_xs, (const float*)d_ys, nx, ny,
                       (float*)d_image);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

// ---- synthetic beams -------------------------------------------------------------------

int lynx_fill_gaussian(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const double* mu,
                       const double* sigma, uint64_t seed, void* d_p) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !d_p || !mu || !sigma || batch <= 0 || n_particles <= 0)
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  GaussArgs g;
  for (int i = 0; i < 6; ++i) {
    g.mu[i] = mu[i];
    g.sigma[i] = sigma[i];
  }
  const int64_t total = batch * n_particles * 7;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  if (dtype == LYNX_F64)
    hipLaunchKernelGGL(k_fill_gaussian<double>, dim3(grid), dim3(256), 0, ctx->stream, (double*)d_p, total, seed, g);
  else
    hipLaunchKernelGGL(k_fill_gaussian<float>, dim3(grid), dim3(256), 0, ctx->stream, (float*)d_p, total, seed, g);
  HIP_TRY(ctx, hipGetLastError());
  return LYNX_OK;
}

// ---- diagnostics: practical HBM ceiling ------------------------------------------------------

int lynx_diag_copy(lynx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int repeats, int vec_per_thread,
                   float* avg_ms) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  const int per_thread = vec_per_thread >= 100 ? vec_per_thread - 100 : vec_per_thread;  // >= 100: non-temporal stores
  if (!ctx || !d_dst || !d_src || bytes % 16 || repeats <= 0 || vec_per_thread < 0 || per_thread > 64 ||
      (vec_per_thread >= 100 && per_thread == 0))
    return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  HIP_TRY(ctx, use_device(ctx));
  const int64_t n_vec = (int64_t)(bytes / 16);
  const int vpt = vec_per_thread;  // 0 = grid-stride
  const unsigned grid = per_thread > 0 ? (unsigned)((n_vec + 256LL * per_thread - 1) / (256LL * per_thread))
                                : (unsigned)std::min<int64_t>((n_vec + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_diag_copy, dim3(grid), dim3(256), 0, ctx->stream, (const lynx_f32x4*)d_src,
                     (lynx_f32x4*)d_dst, n_vec, vpt);  // warm-up
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  for (int r = 0; r < repeats; ++r)
    hipLaunchKernelGGL(k_diag_copy, dim3(grid), dim3(256), 0, ctx->stream, (const lynx_f32x4*)d_src,
                       (lynx_f32x4*)d_dst, n_vec, vpt);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
  *avg_ms = ms / repeats;
  return LYNX_OK;
}

// ---- RCCL ------------------------------------------------------------------------------

int lynx_comm_unique_id(char* id_out) {
  static_assert(sizeof(ncclUniqueId) <= LYNX_UNIQUE_ID_BYTES, "unique id does not fit");
  ncclUniqueId id;
  NCCL_TRY(nullptr, ncclGetUniqueId(&id));
  memset(id_out, 0, LYNX_UNIQUE_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return LYNX_OK;
}

int lynx_comm_init(lynx_ctx* ctx, int n_ranks, int rank, const char* id) {
  if (!ctx || !id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(ctx, LYNX_ERR_INVALID, "bad argument");
  if (ctx->comm) return fail(ctx, LYNX_ERR_INVALID, "communicator already initialised");
  HIP_TRY(ctx, use_device(ctx));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NCCL_TRY(ctx, ncclCommInitRank(&ctx->comm, n_ranks, uid, rank));
  ctx->comm_ranks = n_ranks;
  if (n_ranks > 1 && !ctx->plain_events) {  // (a job that did not say WORLD_SIZE: its events get HIP's default fence now)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_build));
    HIP_TRY(ctx, sync_main(ctx));
    const int rc = wait_for_side(ctx);
    if (rc) return rc;
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      retire_side_ops(ctx, true);
    }
    ctx->plain_events = true;
    ctx->last_stream_stop = nullptr;
    return make_sync_events(ctx);
  }
  return LYNX_OK;
}

int lynx_comm_destroy(lynx_ctx* ctx) {
  if (ctx && ctx->comm) {
    HIP_TRY(ctx, sync_main(ctx));
    int rc = wait_for_side(ctx);
    if (rc) return rc;
    NCCL_TRY(ctx, ncclCommDestroy(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
  }
  return LYNX_OK;
}

int lynx_comm_info(lynx_ctx* ctx, int32_t* rccl_version, int32_t* n_ranks, int32_t* rank) {
  if (!ctx || !rccl_version || !n_ranks || !rank) return fail(ctx, LYNX_ERR_INVALID, "null argument");
  int v = 0;
  NCCL_TRY(ctx, ncclGetVersion(&v));
  *rccl_version = v;
  *n_ranks = 0;
  *rank = -1;
  if (ctx->comm) {
    int n = 0, r = -1;
    NCCL_TRY(ctx, ncclCommCount(ctx->comm, &n));
    NCCL_TRY(ctx, ncclCommUserRank(ctx->comm, &r));
    *n_ranks = n;
    *rank = r;
  }
  return LYNX_OK;
}

int lynx_gather_moments(lynx_ctx* ctx, const double* d_send, double* d_recv, int64_t count) {
  ctx->main_idle = false;  // (something is about to be enqueued on the main stream)
  if (!ctx || !ctx->comm) return fail(ctx, LYNX_ERR_INVALID, "communicator not initialised");
  HIP_TRY(ctx, use_device(ctx));
  // Default with more than one rank: the side stream.  With one rank (LYNX_FORCE_COMM rehearsals) the "gather" is
  // a copy; it follows the reduction wherever that ran.
  const bool produced_on_side = ctx->side_wrote && ctx->side_wrote == (const void*)d_send;
  if (knob(ctx->knobs.gather_overlap, (ctx->comm_ranks > 1 || produced_on_side) ? 1 : 0) == 0) {  // in line, on the main stream
    int rc = join_side(ctx);  // the records may come from a reduction on the side stream
    if (rc) return rc;
    hipEvent_t g0 = nullptr, g1 = nullptr;
    if (ctx->profiling) {
      HIP_TRY(ctx, hipEventCreateWithFlags(&g0, timing_event_flags(ctx)));
      HIP_TRY(ctx, hipEventCreateWithFlags(&g1, timing_event_flags(ctx)));
      HIP_TRY(ctx, hipEventRecord(g0, ctx->stream));
    }
    NCCL_TRY(ctx, ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, ctx->comm, ctx->stream));
    if (g1) {
      HIP_TRY(ctx, hipEventRecord(g1, ctx->stream));
      ctx->prof_gather_events.emplace_back(g0, g1);
    }
    return LYNX_OK;
  }
  // The gather is latency (a few hundred KB over xGMI, and it couples this GPU to the slowest rank of the step) and
  // nothing on this GPU waits for its result but the host: it runs on the side stream, underneath the next call's
  // streaming kernel, instead of between two of them.  In line every rank would run in lockstep with the slowest
  // one, step by step; decoupled only the end of the job waits for everybody.
  //   start : when the records are final -- stream order when the side stream reduced them itself, else a marker
  //           behind everything enqueued on the main stream so far;
  //   end   : a "done" event; the main stream never waits for it.  The two blocks stay out of the allocator until it
  //           has fired (ctx_free defers them), the host sees the result through lynx_buf_d2h / lynx_sync, which
  //           wait for this stream.
  if (!produced_on_side) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_main_mark, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_side, ctx->ev_main_mark, 0));
  }
  hipEvent_t done = nullptr;
  int rc = side_event(ctx, &done);
  if (rc) return rc;
  hipEvent_t g0 = nullptr, g1 = nullptr;
  if (ctx->profiling) {  // the gather's own duration on its stream (bench.py at N > 1: how long a rank waits for the others)
    HIP_TRY(ctx, hipEventCreateWithFlags(&g0, timing_event_flags(ctx)));
    HIP_TRY(ctx, hipEventCreateWithFlags(&g1, timing_event_flags(ctx)));
    HIP_TRY(ctx, hipEventRecord(g0, ctx->s_side));
  }
  NCCL_TRY(ctx, ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, ctx->comm, ctx->s_side));
  if (g1) {
    HIP_TRY(ctx, hipEventRecord(g1, ctx->s_side));
    ctx->prof_gather_events.emplace_back(g0, g1);
  }
  HIP_TRY(ctx, hipEventRecord(done, ctx->s_side));
  side_op_issued(ctx, done, d_send, d_recv);
  return LYNX_OK;
}

