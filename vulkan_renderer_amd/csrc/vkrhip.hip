// vkrhip.hip - C ABI of the gfx950 scene-processing library: context, HBM
// buffers, uploads, the per-frame launch sequence and result read-back.
// Entry points are declared (with the reference interfaces they stand behind)
// in include/vkr_hip_scene.h.
#include "../../include/vkr_hip_scene.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "device_math.hpp"
#include "k_cull.hpp"
#include "k_dirty.hpp"
#include "k_fused.hpp"
#include "k_pack.hpp"
#include "k_exchange.hpp"
#include "k_epoch.hpp"
#include "k_sort.hpp"
#include "k_transform.hpp"

using namespace vkr;

static_assert(sizeof(vkrhip_compaction_state) == sizeof(CompactionState), "state ABI");

namespace {

thread_local std::string g_create_error;

struct EventPair {
  hipEvent_t start, stop;
  int timer;
  int samples;  // what the pair adds to the timer's sample count (0: a further launch of a stage)
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0;
};

}  // namespace

// The tail of a frame enqueued by a second host thread. With the tail on the second
// stream a frame costs the host a stream-wait, three or more launches and an event
// record beyond the level launches - 38 us per frame where the device needs 31 (1M
// nodes) - and none of those calls depends on anything the calling thread does
// next. The calling thread therefore only DESCRIBES them (kernel, grid, a copy of
// the arguments) and a worker thread of the context issues them on the second
// stream, in order, while the caller is already preparing the next frame. Every
// other entry point waits for the worker to be idle first (bind_device), so
// outside vkrhip_frame the second stream is exactly where a single thread would
// have left it.
struct TailLaunch {
  const void *fn = nullptr;
  dim3 grid, block;
  uint32_t lds = 0, nparams = 0;
  uint32_t offset[3] = {0, 0, 0};
  alignas(16) unsigned char blob[448];
};
struct TailJob {
  hipEvent_t wait = nullptr;    // the second stream waits for this first
  hipEvent_t record = nullptr;  // ... and records this behind the launches
  uint32_t n = 0;
  bool overflow = false;  // more launches than the job holds: the frame fails (never seen)
  static constexpr uint32_t kLaunches = 12;  // scan, gather, histogram, four passes, emission, signal
  TailLaunch launch[kLaunches];
};
struct TailWorker {
  std::thread thread;
  std::mutex m;
  std::condition_variable cv;
  TailJob ring[4];
  std::atomic<uint64_t> pushed{0}, done{0};
  std::atomic<bool> stop{false}, sleeping{false};
  std::atomic<int> error{0};  // the first hipError_t a job met
  hipStream_t stream = nullptr;
  int device = 0;
  uint64_t busy_ns = 0, jobs = 0;  // (read after the thread has ended)
};

struct vkrhip_ctx {
  vkrhip_config cfg{};
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // scene
  uint64_t n_nodes = 0;
  std::vector<uint64_t> level_offsets;
  DevBuf<uint32_t> parent;
  // parents of stepping levels as (first parent, step mask) per 64 nodes, see
  // TransformArgs::parent_steps; steps_entry[l]: first entry of level l, -1: not stepping
  DevBuf<uint32_t> parent_steps;
  std::vector<int64_t> steps_entry;
  bool steps_enabled = true;  // VKRHIP_PARENT_STEPS=0 keeps the index per node (A/B runs)
  DevBuf<float4> pos, quat, scale;  // lazily allocated
  DevBuf<float4> local, world;
  // SceneTransform.flags, held twice: a frame reads the nodes' flags from one copy
  // and writes the updated flags to the other, so that blocks which recompute
  // their ancestors (the merged top levels) see the flags as they were before the
  // frame whatever the owners of those nodes have already written. flags_cur names
  // the copy that holds the current state.
  DevBuf<uint8_t> flags2[2];
  uint32_t flags_cur = 0;
  uint8_t *flags() const { return flags2[flags_cur].p; }
  uint8_t *flags_other() const { return flags2[flags_cur ^ 1u].p; }
  // Merged top levels (see TopPlan in k_fused.hpp): levels [0, top_levels) run as
  // ONE launch whose blocks recompute their ancestors; 0 = every level its own launch.
  uint32_t top_levels = 0;
  uint32_t top_blocks = 0;
  uint32_t top_anc_cap = 0;  // largest ancestor count of a block (rounded up to 64)
  DevBuf<TopDesc> top_desc;
  DevBuf<uint32_t> top_anc_node, top_anc_info;
  bool top_enabled = true;  // VKRHIP_TOP_LEVELS=0 switches the merged launch off (A/B runs)
  bool top_balance = true;  // VKRHIP_TOP_BALANCE=0: 256-node blocks whatever their number
  // Streaming loads and stores in the level launches of the fused kernel, for frames
  // that do not fit the last-level cache. VKRHIP_STREAMING: 0 never, 1 always,
  // unset: by the frame's footprint.
  int streaming_mode = -1;
  uint32_t force_flags = 0;
  uint32_t fused_rounds = 0;  // tuning override (VKRHIP_FUSED_ROUNDS = 1, 2 or 4); 0 = by level size
  bool have_trs = false;

  // candidates
  uint64_t n_cand = 0;
  bool node_identity = true;
  uint64_t node_limit = 0;  // 1 + the largest node index a candidate row names (explicit mapping)
  DevBuf<uint32_t> node;
  DevBuf<uint4> ident;
  DevBuf<float4> sphere;
  DevBuf<uint32_t> cflags;
  DevBuf<uint16_t> cflags16;
  uint64_t blend_rows = 0;  // rows that can enter the transparent list
  uint64_t tcap = 0;        // allocated transparent capacity

  // views
  uint32_t views = 0;
  DevBuf<float4> planes;
  DevBuf<float4> planes_paired;  // the same planes, two at a time (see sphere_visible)
  DevBuf<float> tie_eps;
  DevBuf<uint32_t> required;
  DevBuf<float4> camera_view;

  // depth pyramid of view 0 (vkrhip_set_hzb); hzb_mips == 0: occlusion test off
  DevBuf<float> hzb;
  DevBuf<uint32_t> hzb_offset;
  DevBuf<float4> hzb_view_projection;
  uint32_t hzb_width = 0, hzb_height = 0, hzb_mips = 0;
  float hzb_epsilon = 0.f;

  // cull intermediates / outputs
  DevBuf<uint8_t> tvis;
  DevBuf<CompactionState> states;
  DevBuf<uint32_t> row_totals;  // survivors per count row, written by k_scan
  // Segment tables of the compaction: a segment is the stretch of candidates one
  // classify block handled (at most 1024). Sized for the hierarchy in use.
  // What the classify blocks write and the compaction reads exists twice once a
  // frame's compaction runs beside the next frame's classify (see frame_tail): a
  // frame uses copy `par`.
  DevBuf<unsigned long long> seg_masks2[2];  // [seg_stride][seg_record_words(max_views)] survivor masks
  DevBuf<uint32_t> seg_counts2[2];  // [max_views * 4 + 1][seg_stride]
  DevBuf<uint32_t> seg_offsets;     // same shape, exclusive prefix
  DevBuf<uint32_t> seg_first2[2];   // [seg_stride]
  uint32_t par = 0;           // copy of the frame being enqueued
  bool second_copy = false;   // copy 1 of the tables above is allocated
  bool want_second = false;   // ... or will be, the next time the tables are (re)allocated
  uint64_t seg_alloc_stride = 0;  // columns the tables were allocated for (seg_stride: those in use)
  DevBuf<unsigned long long> &seg_masks() { return seg_masks2[par]; }
  DevBuf<uint32_t> &seg_counts() { return seg_counts2[par]; }
  DevBuf<uint32_t> &seg_first() { return seg_first2[par]; }
  uint64_t seg_stride = 0;
  uint64_t segments = 0;  // segments of the frame being enqueued
  // Whether the classify blocks of the fused kernel expand their own survivors
  // (segment_finish_staged, k_place). VKRHIP_STAGED: 0 never, 1 flat scenes of at
  // least kStagedMinSegments segments (default), 2 every frame the variants exist for.
  int staged_compaction = 1;
  bool frame_staged = false;      // the frame's classify blocks staged their survivors (k_place follows)
  // Per-frame counters: three blocks of [sharded counters | tie counters | sort
  // state | caster box]. Frames take them in turn; the scan kernel of a frame
  // zeroes the block of the frame after the next one (the previous frame's block,
  // whose readers are done by then: they precede the scan on its stream, or the
  // frame's stream has joined them), so steady-state frames start without a fill.
  // block_clean falls back to a fill whenever a frame ran without that kernel.
  static constexpr uint32_t kBlocks = 3;
  DevBuf<uint8_t> frame_zero;
  uint32_t cur_block = 0;
  bool block_clean[kBlocks] = {true, true, true};
  static constexpr size_t kCountersBytes = sizeof(uint32_t) * CNT_COUNT * STAT_SHARDS;
  static constexpr size_t kTiesBytes = sizeof(unsigned long long) * VKRHIP_MAX_VIEWS * TIE_SHARDS;
  static constexpr size_t kSortStateBytes = sizeof(SortState);
  static constexpr size_t kCasterBoxBytes = sizeof(uint32_t) * 8;  // six keys, padded
  static constexpr size_t kBlockBytes = kCountersBytes + kTiesBytes + kSortStateBytes + kCasterBoxBytes;
  uint32_t cu_count = 0;
  uint32_t *caster_box() const {
    return reinterpret_cast<uint32_t *>(counter_block(cur_block) + kCountersBytes + kTiesBytes + kSortStateBytes);
  }
  uint32_t frame_stages = 0;  // the stages of the frame being enqueued
  SortState *sort_state() const {
    return reinterpret_cast<SortState *>(counter_block(cur_block) + kCountersBytes + kTiesBytes);
  }
  uint8_t *counter_block(uint32_t i) const { return frame_zero.p + (size_t)i * kBlockBytes; }
  uint32_t *counters() const { return reinterpret_cast<uint32_t *>(counter_block(cur_block)); }
  unsigned long long *ties() const {
    return reinterpret_cast<unsigned long long *>(counter_block(cur_block) + kCountersBytes);
  }
  DevBuf<uint64_t> tkey2[2];  // copy `par`, like the segment tables
  DevBuf<uint64_t> &tkey() { return tkey2[par]; }
  // The world matrix of every visible blend row under its candidate index, written
  // by the classify blocks of a frame whose whole tail runs on the second stream.
  DevBuf<float4> tworld_sparse[2];
  DevBuf<uint32_t> visible;
  uint64_t visible_stride = 0;
  DevBuf<float4> world_bounds;
  DevBuf<uint4> cand_rows;
  DevBuf<uint32_t> transmission_list;  // source rows with a transmissive material, ascending
  uint64_t n_transmission = 0;
  DevBuf<uint4> transmission_rows;

  // render-dirty list (allocated when the stage first runs)
  DevBuf<uint32_t> dirty_list, dirty_block_counts, dirty_block_offsets, dirty_total;

  // backend pack / encode (allocated when a publication is set / the stage first runs)
  bool published = false;
  DevBuf<GeometryEntry> pub_geometries;
  DevBuf<SubmeshRange> pub_submeshes;
  DevBuf<MaterialEntry> pub_materials;
  DevBuf<uint8_t> pack_status;
  DevBuf<uint32_t> packed_index;
  DevBuf<uint4> pack_a;
  DevBuf<int32_t> pack_vo;
  DevBuf<uint32_t> pack_counts;  // 4 words
  uint32_t pack_counts_host[4] = {0, 0, 0, 0};
  DevBuf<uint4> packed_rows;
  DevBuf<float4> packed_instances;
  DevBuf<uint4> visible_rows;
  DevBuf<uint32_t> commands;

  // transparent
  DevBuf<uint64_t> keys_a, keys_b, out_keys;
  DevBuf<uint32_t> out_source;
  DevBuf<unsigned long long> sort_status;  // look-back words of the radix passes
  uint32_t sort_tiles_stride = 0;
  uint32_t sort_serial = 0;  // tags the status words of one sort (never 0)
  uint32_t sort_block = 0;   // counter block of the last radix sort (its stall flag lives there)
  DevBuf<uint4> draws;
  DevBuf<float4> instances;

  // pinned host word the scan kernel writes (transparent count + 1; 0 = unknown)
  uint32_t *host_hint = nullptr;
  // pinned host block the frame's counters are published to when the candidate rows
  // are produced (word 0: frame serial, written last; then CNT_COUNT totals), and the
  // pinned arrays vkrhip_fill_world_payload hands out
  uint32_t *host_counts = nullptr;
  uint32_t publish_serial = 0;
  void *pay_candidates = nullptr, *pay_transmission = nullptr, *pay_draws = nullptr, *pay_instances = nullptr;
  size_t pay_candidates_bytes = 0, pay_transmission_bytes = 0, pay_draws_bytes = 0, pay_instances_bytes = 0;

  // merge scratch
  DevBuf<const uint64_t *> merge_ptrs;
  DevBuf<uint64_t> merge_lens;
  DevBuf<uint64_t> concat_desc;
  // candidate epoch hash workspace
  DevBuf<uint64_t> epoch_maps, epoch_maps_low, epoch_wave_maps, epoch_pairs, epoch_result;
  DevBuf<uint8_t> epoch_enter_low, epoch_enter_high;
  DevBuf<uint32_t> epoch_wave_occluders;
  // native exchange: RCCL communicator, second stream, two buffer sets
  void *comm = nullptr;  // ncclComm_t
  uint32_t comm_world = 1, comm_rank = 0;
  hipStream_t xstream = nullptr;
  // Sort and emission of the transparent draws run on a stream of their own, beside
  // the next frame's transform and cull (they read nothing those overwrite).
  hipStream_t tstream = nullptr;
  hipEvent_t attach_stop = nullptr;  // recorded by the next heavy launch itself, as its completion signal
  hipEvent_t ev_gathered = nullptr;  // main stream: what the side stream's work of the frame reads is complete
  hipEvent_t ev_gathered2[2] = {nullptr, nullptr};  // (frames alternate: a worker may still have to enqueue the
  uint32_t gathered_turn = 0;                       //  wait for the previous frame's)
  TailWorker *worker = nullptr;
  TailJob *building = nullptr;     // launches on the second stream are described into this job
  int worker_mode = 1;             // VKRHIP_TAIL_WORKER=0: the calling thread enqueues the tail itself
  // Batches of second-stream work are numbered; side_flag[k] holds the serial of the
  // last batch behind ev_side[k] that is over (k_side_signal / k_side_gate, k_transform.hpp).
  // VKRHIP_SIDE_GATE=0: the context's stream waits for the event instead.
  DevBuf<uint64_t> side_flag;
  uint64_t side_serial = 0;
  uint64_t side_expected[2] = {0, 0};
  int side_gate = 1;
  // Frames whose tail the worker enqueues also tell the second stream that the classify
  // is over with a signal kernel behind the last level launch and a gate at the head of
  // the tail (side_flag[3]): a launch that carries a completion event costs the calling
  // thread 10 us and the wait for it the worker 9 of its 32 per frame, which set the
  // pace of a 1M-node frame (38.6 -> 35.8 us; large scenes do not notice either way).
  bool gathered_by_gate = false;
  uint64_t gathered_serial = 0;
  uint64_t side_job[2] = {0, 0};   // the job that records ev_side[k] (0: recorded by the calling thread)
  // Side stream: everything a frame put there is done. One event per copy `par`, so
  // that a wait that is still queued never sees its event recorded again.
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  bool side_pending[2] = {false, false};  // ev_side[k] has not been waited for yet
  int side_last = -1;                     // the event recorded last (the side stream runs in order)
  uint32_t side_turn = 0;                 // the event the next batch of side-stream work records
  // Side-stream work that must be over before the main stream touches something
  // again, as the index of the event recorded behind it (-1: none): the batch that
  // read copy k of the segment tables, and the batch whose scan zeroes counter block b.
  int copy_event[2] = {-1, -1};
  int block_event[3] = {-1, -1, -1};
  int overlap_sort = 1;             // VKRHIP_OVERLAP_SORT: 0 one stream, 1 large hierarchies, 2 always
  // The whole tail of a frame (scan, gather, sort, emission) on the second stream:
  // VKRHIP_OVERLAP_TAIL 0 never, 1 large hierarchies (default), 2 whenever the
  // fused kernel runs.
  int overlap_tail = 1;
  bool frame_tail = false;  // the frame being enqueued does that
  int xdebug = 0;           // VKRHIP_XDEBUG: 16 logs the exchange's stream order, 32 makes
                            // vkrhip_exchange_finish poll and name what is pending when it stalls
  DevBuf<uint32_t> tsrc;
  DevBuf<float4> tworld;
  struct ExchangeSet {
    DevBuf<uint32_t> send, recv, merged_visible;
    DevBuf<uint64_t> merged_keys, result;
    hipEvent_t packed = nullptr;  // the set's block is packed (the pack launch's own completion signal)
    hipEvent_t done = nullptr;    // the set's merged outputs are complete
    bool in_flight = false;
  } xset[2];
  uint32_t x_world = 0, x_views = 0, x_vis_cap = 0, x_key_cap = 0, x_turn = 0;
  int x_last = -1;
  // device-planned exchange
  DevBuf<uint64_t> exch_desc, exch_rank_view_start, exch_prefix, exch_runs;

  // timing
  uint32_t profiling = 0;  // bit i: time stage i
  std::vector<EventPair> pending;
  std::vector<EventPair> free_events;
  double timer_ms[VKRHIP_TIMER_COUNT] = {};
  uint64_t timer_n[VKRHIP_TIMER_COUNT] = {};

  bool frame_ran = false;
  uint32_t last_stages = 0;
};

namespace {

int fail(vkrhip_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), VKRHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

template <typename T>
int dev_alloc(vkrhip_ctx *c, DevBuf<T> &b, size_t count) {
  if (b.p && b.count >= count) return VKRHIP_OK;
  if (b.p) {
    HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.count = 0;
  }
  if (count == 0) count = 1;
  HIP_TRY(c, hipMalloc((void **)&b.p, count * sizeof(T)));
  b.count = count;
  return VKRHIP_OK;
}

template <typename T>
void dev_free(DevBuf<T> &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.count = 0;
}

void cpu_relax() { __builtin_ia32_pause(); }

void tail_worker_main(TailWorker *w) {
  (void)hipSetDevice(w->device);
  uint64_t next = 0;
  uint32_t idle = 0;
  for (;;) {
    if (w->pushed.load(std::memory_order_acquire) > next) {
      const TailJob &job = w->ring[next & 3u];
      timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      hipError_t e = hipSuccess;
      if (job.wait) e = hipStreamWaitEvent(w->stream, job.wait, 0);
      for (uint32_t k = 0; k < job.n && e == hipSuccess; ++k) {
        const TailLaunch &l = job.launch[k];
        void *argv[3] = {(void *)(l.blob + l.offset[0]), (void *)(l.blob + l.offset[1]),
                         (void *)(l.blob + l.offset[2])};
        e = hipLaunchKernel(l.fn, l.grid, l.block, argv, l.lds, w->stream);
      }
      if (e == hipSuccess && job.record) e = hipEventRecord(job.record, w->stream);
      if (e != hipSuccess) {
        int none = 0;
        w->error.compare_exchange_strong(none, (int)e);
      }
      clock_gettime(CLOCK_MONOTONIC, &t1);
      w->busy_ns += (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
      ++w->jobs;
      ++next;
      w->done.store(next, std::memory_order_release);
      idle = 0;
      continue;
    }
    if (w->stop.load(std::memory_order_acquire)) return;
    if (++idle < 200000u) {  // frames arrive tens of microseconds apart: spin between them
      cpu_relax();
      continue;
    }
    std::unique_lock<std::mutex> lock(w->m);  // nothing for a while: sleep
    w->sleeping.store(true, std::memory_order_seq_cst);
    w->cv.wait_for(lock, std::chrono::milliseconds(20), [&] {
      return w->pushed.load(std::memory_order_acquire) > next || w->stop.load(std::memory_order_acquire);
    });
    w->sleeping.store(false, std::memory_order_seq_cst);
    idle = 0;
  }
}

int worker_error(vkrhip_ctx *c) {
  const int e = c->worker->error.exchange(0);
  if (e)
    return fail(c, VKRHIP_ERR_HIP, "a launch on the second stream failed: %s", hipGetErrorString((hipError_t)e));
  return VKRHIP_OK;
}

// Waits until the worker has issued job `serial` (jobs count from 1).
int worker_wait(vkrhip_ctx *c, uint64_t serial) {
  if (!c->worker) return VKRHIP_OK;
  while (c->worker->done.load(std::memory_order_acquire) < serial) cpu_relax();
  return worker_error(c);
}
int worker_flush(vkrhip_ctx *c) {
  return c->worker ? worker_wait(c, c->worker->pushed.load(std::memory_order_relaxed)) : VKRHIP_OK;
}

int worker_start(vkrhip_ctx *c) {
  if (c->worker) return VKRHIP_OK;
  c->worker = new (std::nothrow) TailWorker();
  if (!c->worker) return fail(c, VKRHIP_ERR_HIP, "out of host memory");
  c->worker->stream = c->tstream;
  c->worker->device = c->device;
  c->worker->thread = std::thread(tail_worker_main, c->worker);
  return VKRHIP_OK;
}

// The job the calling thread fills next (at most two are ahead of the worker).
TailJob *worker_begin(vkrhip_ctx *c) {
  TailWorker *w = c->worker;
  const uint64_t pushed = w->pushed.load(std::memory_order_relaxed);
  while (pushed - w->done.load(std::memory_order_acquire) >= 2u) cpu_relax();
  TailJob *job = &w->ring[pushed & 3u];
  job->wait = job->record = nullptr;
  job->n = 0;
  job->overflow = false;
  return job;
}
void worker_push(vkrhip_ctx *c) {
  TailWorker *w = c->worker;
  w->pushed.fetch_add(1, std::memory_order_release);
  if (w->sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lock(w->m);
    w->cv.notify_one();
  }
}
void worker_stop(vkrhip_ctx *c) {
  if (!c->worker) return;
  while (c->worker->done.load(std::memory_order_acquire) < c->worker->pushed.load(std::memory_order_relaxed))
    cpu_relax();
  c->worker->stop.store(true, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lock(c->worker->m);
    c->worker->cv.notify_one();
  }
  if (c->worker->thread.joinable()) c->worker->thread.join();
  if ((c->xdebug & 1) && c->worker->jobs)
    std::fprintf(stderr, "tail worker: %llu jobs, %.1f us each\n", (unsigned long long)c->worker->jobs,
                 c->worker->busy_ns / 1e3 / c->worker->jobs);
  delete c->worker;
  c->worker = nullptr;
}

// `flush`: the worker thread has issued everything it was given (every entry point
// but vkrhip_frame: what they do with the streams comes behind it).
int bind_device(vkrhip_ctx *c, bool flush = true) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (flush && c->worker) return worker_flush(c);
  return VKRHIP_OK;
}

// A kernel launch on `on`; while a job for the worker is being described and `on`
// is the second stream, the launch goes into the job instead.
template <typename T>
struct as_given {
  using type = T;
};
template <typename... P>
void launch_on(vkrhip_ctx *c, hipStream_t on, void (*kernel)(P...), dim3 grid, dim3 block, size_t lds,
               typename as_given<P>::type... args) {
  static_assert(sizeof...(P) >= 1 && sizeof...(P) <= 3, "launches described for the worker");
  if (c->building && on == c->tstream) {
    if (c->building->n >= TailJob::kLaunches) {
      c->building->overflow = true;
      return;
    }
    TailLaunch &l = c->building->launch[c->building->n++];
    l.fn = reinterpret_cast<const void *>(kernel);
    l.grid = grid;
    l.block = block;
    l.lds = (uint32_t)lds;
    l.nparams = (uint32_t)sizeof...(P);
    size_t off = 0;
    uint32_t k = 0;
    auto put = [&](const void *src, size_t size, size_t align) {
      off = (off + align - 1) / align * align;
      std::memcpy(l.blob + off, src, size);
      l.offset[k++] = (uint32_t)off;
      off += size;
    };
    static_assert((sizeof(P) + ...) + 32 <= sizeof(l.blob), "argument copy");
    (put(&args, sizeof(P), alignof(P)), ...);
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, on, args...);
}

#define TRY(expr)                    \
  do {                               \
    int r_ = (expr);                 \
    if (r_ != VKRHIP_OK) return r_;  \
  } while (0)

// RCCL is bound at run time (the library has no link-time dependency on it): a
// process that already has it loaded (e.g. under PyTorch) is given that copy.
struct RcclApi {
  int (*GetUniqueId)(void *id128) = nullptr;
  // ncclUniqueId is passed by value: a 128-byte struct
  struct UniqueId {
    char internal[128];
  };
  int (*CommInitRank)(void **comm, int nranks, UniqueId id, int rank) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int datatype, void *comm,
                   hipStream_t stream) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*CommCount)(void *comm, int *count) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool tried = false, ok = false;
};
RcclApi g_rccl;
constexpr int kNcclUint32 = 3;  // ncclDataType_t: ncclUint32

bool load_rccl() {
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  void *lib = nullptr;
  const char *names[] = {"librccl.so.1", "librccl.so"};
  for (const char *n : names)
    if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char *n : names)
    if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return false;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(lib, "ncclCommCount"));
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllGather && g_rccl.CommDestroy;
  return g_rccl.ok;
}

void release_exchange(vkrhip_ctx *c) {
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  for (auto &x : c->xset) {
    dev_free(x.send);
    dev_free(x.recv);
    dev_free(x.merged_visible);
    dev_free(x.merged_keys);
    dev_free(x.result);
    if (x.done) (void)hipEventDestroy(x.done);
    if (x.packed) (void)hipEventDestroy(x.packed);
    x.done = x.packed = nullptr;
    x.in_flight = false;
  }
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  c->xstream = nullptr;
}

// No exception leaves the library: a failed host allocation inside an entry point
// becomes an error code like any other failure.
int host_exception(vkrhip_ctx *c) {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    return fail(c, VKRHIP_ERR_CAPACITY, "out of host memory");
  } catch (const std::exception &e) {
    return fail(c, VKRHIP_ERR_STATE, "unexpected exception: %s", e.what());
  } catch (...) {
    return fail(c, VKRHIP_ERR_STATE, "unexpected exception");
  }
}

// Timer scope: records an event pair around a stage when profiling is on.
struct StageTimer {
  vkrhip_ctx *c;
  EventPair ev{};
  bool on = false;
  hipStream_t stream;
  StageTimer(vkrhip_ctx *ctx, int timer, hipStream_t on_stream = nullptr)
      : c(ctx), stream(on_stream ? on_stream : ctx->stream) {
    if (!(c->profiling & (1u << timer))) return;
    if (!c->free_events.empty()) {
      ev = c->free_events.back();
      c->free_events.pop_back();
    } else {
      if (hipEventCreate(&ev.start) != hipSuccess) return;
      if (hipEventCreate(&ev.stop) != hipSuccess) {
        (void)hipEventDestroy(ev.start);
        return;
      }
    }
    ev.timer = timer;
    ev.samples = 1;
    on = hipEventRecord(ev.start, stream) == hipSuccess;
    if (!on) c->free_events.push_back(ev);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(ev.stop, stream);
    c->pending.push_back(ev);
  }
};

// Orders the main stream behind the side stream's last emission (a device-side
// wait; the host does not block). Called wherever the main stream, or the host
// through it, is about to read the draws or to change what the emission reads.
int join_side(vkrhip_ctx *c) {
  TRY(worker_flush(c));
  // (the side stream runs in order: behind its last event nothing of it is left)
  if (c->side_last >= 0 && c->side_pending[c->side_last]) {
    const int e = c->side_last;
    if (c->side_expected[e]) {
      hipLaunchKernelGGL(k_side_gate, dim3(1), dim3(64), 0, c->stream, (const uint64_t *)c->side_flag.p + e,
                         c->side_expected[e], c->side_flag.p + 2);
      HIP_TRY(c, hipGetLastError());
    } else {
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side[e], 0));
    }
    c->side_pending[0] = c->side_pending[1] = false;
  }
  return VKRHIP_OK;
}

// Orders the main stream behind one batch of side-stream work (see copy_event,
// block_event): normally the batch of two frames ago, long finished. A later
// record of the same event covers the earlier one.
int wait_side_event(vkrhip_ctx *c, int e) {
  if (e < 0) return VKRHIP_OK;
  if (c->side_pending[e]) {
    if (c->side_expected[e]) {
      // (needs nothing of the worker: the gate passes once the batch's signal has run,
      // whenever that is enqueued)
      hipLaunchKernelGGL(k_side_gate, dim3(1), dim3(64), 0, c->stream, (const uint64_t *)c->side_flag.p + e,
                         c->side_expected[e], c->side_flag.p + 2);
      HIP_TRY(c, hipGetLastError());
    } else {
      TRY(worker_wait(c, c->side_job[e]));  // (the record has to be enqueued before the wait)
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side[e], 0));
    }
    c->side_pending[e] = false;
    if (c->side_last == e) c->side_pending[e ^ 1] = false;  // recorded earlier: covered as well
  }
  return VKRHIP_OK;
}

// Marks the end of a batch of side-stream work. Returns the event's index.
int record_side(vkrhip_ctx *c, int *out_event) {
  const int e = (int)c->side_turn;
  c->side_expected[e] = 0;
  c->side_job[e] = c->building ? c->worker->pushed.load(std::memory_order_relaxed) + 1u : 0u;
  if (c->side_gate && c->side_flag.p) {  // the batch ends with its signal (no event)
    c->side_expected[e] = ++c->side_serial;
    launch_on(c, c->tstream, k_side_signal, dim3(1), dim3(64), 0, c->side_flag.p + e, c->side_expected[e]);
    if (!c->building) HIP_TRY(c, hipGetLastError());
  } else if (c->building) {  // the worker records the event behind the job's launches
    c->building->record = c->ev_side[e];
  } else {
    HIP_TRY(c, hipEventRecord(c->ev_side[e], c->tstream));
  }
  c->side_pending[e] = true;
  c->side_last = e;
  c->side_turn ^= 1u;
  if (out_event) *out_event = e;
  return VKRHIP_OK;
}

// The host waits for the side stream (entry points that free or rewrite what its
// kernels read).
int drain_side(vkrhip_ctx *c) {
  TRY(worker_flush(c));
  if (c->tstream) HIP_TRY(c, hipStreamSynchronize(c->tstream));
  c->side_pending[0] = c->side_pending[1] = false;
  return VKRHIP_OK;
}

int drain_timers(vkrhip_ctx *c) {
  for (const EventPair &ev : c->pending) {
    float ms = 0.f;
    HIP_TRY(c, hipEventSynchronize(ev.stop));
    HIP_TRY(c, hipEventElapsedTime(&ms, ev.start, ev.stop));
    c->timer_ms[ev.timer] += ms;
    c->timer_n[ev.timer] += (uint64_t)ev.samples;
    c->free_events.push_back(ev);
  }
  c->pending.clear();
  return VKRHIP_OK;
}

// Launch of one of the heavy kernels (transform, classify, fused). When its
// timer is on, the launch carries its own event pair, which brackets the kernel
// itself (dispatch start/end timestamps) rather than a stretch of the stream; a
// stage of several launches (one per hierarchy level) accumulates into one
// sample.
template <typename K, typename A>
void launch_timed(vkrhip_ctx *c, int timer, bool first_of_stage, K kernel, dim3 grid, dim3 block,
                  size_t lds, const A &args) {
  if (c->profiling & (1u << timer)) {
    EventPair ev{};
    bool have = false;
    if (!c->free_events.empty()) {
      ev = c->free_events.back();
      c->free_events.pop_back();
      have = true;
    } else if (hipEventCreate(&ev.start) == hipSuccess) {
      if (hipEventCreate(&ev.stop) == hipSuccess)
        have = true;
      else
        (void)hipEventDestroy(ev.start);
    }
    if (have) {
      ev.timer = timer;
      ev.samples = first_of_stage ? 1 : 0;
      hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, c->stream, ev.start, ev.stop, 0u, args);
      c->pending.push_back(ev);
      if (c->attach_stop) (void)hipEventRecord(c->attach_stop, c->stream);
      c->attach_stop = nullptr;
      return;
    }
  }
  if (c->attach_stop) {  // the launch's own completion signal is the event: no packet of its own
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, c->stream, nullptr, c->attach_stop, 0u, args);
    c->attach_stop = nullptr;
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, c->stream, args);
}

uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

int check_range(vkrhip_ctx *c, uint64_t first, uint64_t count, uint64_t limit, const char *what) {
  if (first > limit || count > limit - first)
    return fail(c, VKRHIP_ERR_ARG, "%s: range [%llu, +%llu) exceeds %llu", what,
                (unsigned long long)first, (unsigned long long)count, (unsigned long long)limit);
  return VKRHIP_OK;
}


// Plan of the merged top levels. The first levels of a hierarchy hold few
// nodes; a launch per level spends most of its time on the chain of dependent
// cold loads of one block. Instead ONE launch covers levels [0, K): a block
// (one 1024-node tile of one level) first recomputes the world matrices of its
// nodes' ancestors - from the same inputs, with the same operations in the same
// order as their owners, so bit for bit the same values - and therefore waits
// for no other block. In BFS order the ancestors of a tile are a short
// contiguous range per level; this builds the tables (per block: that range
// list flattened to one entry per ancestor, root side first). Hierarchies
// whose ranges are wide (parents not ascending with their children) get no
// plan and run a launch per level.
constexpr uint64_t kStreamingFootprint = 300ull << 20;  // bytes (measured crossover: about 2M nodes; the last-level cache holds 256 MB)
// The merged launch covers at most this many tiles. A level that joins it saves a
// launch boundary (the drain of one grid and the cold start of the next, about 10 us
// on a busy device) and pays for its blocks' ancestor chains (about a tenth of a
// block's time): the 2049 tiles of a 2M-node level still gain (16M-node hierarchy
// 552 -> 550 us), all levels in one launch lose (632 us).
constexpr uint32_t kTopMaxTiles = 3072;
constexpr uint32_t kTopMaxAncestors = TOP_MAX_ANCESTORS;
constexpr uint64_t kGatherSplitSegments = 4096;  // scenes up to this many segments gather with a wave per (segment, view)
constexpr uint64_t kStepsMinNodes = 1ull << 20;  // levels below this keep their parent indices
constexpr uint32_t kTopMaxSteps = 24;

struct TopPlanHost {
  uint32_t levels = 0;
  uint32_t anc_cap = 0;
  std::vector<TopDesc> desc;
  std::vector<uint32_t> anc_node, anc_info;
};

// `resident`: blocks of the merged launch the device holds at once (0: unknown).
TopPlanHost build_top_plan_spans(const uint32_t *parent, const std::vector<uint64_t> &lo, uint32_t levels,
                                 uint64_t resident, uint32_t max_span, uint32_t max_levels) {
  TopPlanHost plan;
  if (!parent || levels < 2) return plan;
  // how many levels fit the tile budget
  uint32_t k = 0;
  uint64_t tiles = 0;
  while (k < levels && k < max_levels) {
    const uint64_t b = lo[k], e = lo[k + 1];
    const uint64_t t = e > b ? ceil_div(e, CULL_TILE) - b / CULL_TILE : 0;
    if (tiles + t > kTopMaxTiles) break;
    tiles += t;
    ++k;
  }
  if (k < 2) return plan;  // nothing to merge
  struct Range {
    uint32_t first, count;
  };
  // One block: nodes [nb, ne) of one level, inside one tile. False: the hierarchy
  // does not fit the scheme (the caller drops what it was building).
  auto add_block = [&](uint64_t nb, uint64_t ne) -> bool {
    const uint64_t tile_base = nb / CULL_TILE * CULL_TILE;
    // ranges of ancestors, nearest level first
    std::vector<Range> chain;
    uint64_t rb = nb, re = ne;
    for (;;) {
      uint32_t pmin = VKRHIP_NO_PARENT, pmax = 0;
      for (uint64_t i = rb; i < re; ++i) {
        const uint32_t p = parent[i];
        if (p == VKRHIP_NO_PARENT) continue;
        pmin = std::min(pmin, p);
        pmax = std::max(pmax, p);
      }
      if (pmin == VKRHIP_NO_PARENT) break;  // only roots left
      // a range must stay inside one level, above the level it was derived from:
      // its nodes are recomputed side by side in one step of the chain
      const size_t lvl = (size_t)(std::upper_bound(lo.begin(), lo.end(), (uint64_t)pmin) - lo.begin()) - 1u;
      if ((uint64_t)pmax >= lo[lvl + 1] || lo[lvl + 1] > rb) return false;
      chain.push_back({pmin, pmax - pmin + 1u});
      if (chain.size() > kTopMaxSteps) return false;
      rb = pmin;
      re = (uint64_t)pmax + 1u;
    }
    uint64_t total = 0;
    for (const Range &r : chain) total += r.count;
    if (total > kTopMaxAncestors) return false;
    TopDesc d{};
    d.tile_base = (uint32_t)tile_base;
    d.lo_hi = (uint32_t)(nb - tile_base) | ((uint32_t)(ne - tile_base) << 16);
    d.anc_off = (uint32_t)plan.anc_node.size();
    d.anc_count = (uint32_t)total;
    d.steps = (uint32_t)chain.size();
    // root side first: step s holds chain[steps - 1 - s]
    std::vector<uint32_t> slot_base(chain.size());
    uint32_t slot = 0;
    for (size_t s2 = 0; s2 < chain.size(); ++s2) {
      slot_base[chain.size() - 1 - s2] = slot;
      slot += chain[chain.size() - 1 - s2].count;
    }
    for (size_t s2 = 0; s2 < chain.size(); ++s2) {
      const size_t ci = chain.size() - 1 - s2;  // index into chain (0 = the parents' level)
      const Range &r = chain[ci];
      for (uint32_t j = 0; j < r.count; ++j) {
        const uint32_t g = r.first + j;
        const uint32_t p = parent[g];
        uint32_t pslot = 0xffffu;
        if (p != VKRHIP_NO_PARENT) pslot = slot_base[ci + 1] + (p - chain[ci + 1].first);
        plan.anc_node.push_back(g);
        plan.anc_info.push_back(pslot | ((uint32_t)s2 << 16));
      }
    }
    if (!chain.empty()) {
      d.par_first = chain[0].first;
      d.par_slot = slot_base[0];
    }
    plan.anc_cap = std::max(plan.anc_cap, (uint32_t)total);
    plan.desc.push_back(d);
    return true;
  };
  // Nodes per block. The levels are small, so many short blocks hide one
  // another's load latency better than a few long ones: a block per ROUND of a
  // tile (256 nodes) - unless that makes more blocks than the device holds at
  // once, where a second wave of blocks would cost a second chain of dependent
  // loads: then the levels with the most blocks get longer ones.
  std::vector<uint32_t> span(k, FUSED_BLOCK);
  auto level_blocks = [&](uint32_t l) -> uint64_t {
    return lo[l + 1] > lo[l] ? ceil_div(lo[l + 1], span[l]) - lo[l] / span[l] : 0u;
  };
  while (resident) {
    uint64_t total = 0;
    uint32_t widest = k;
    for (uint32_t l = 0; l < k; ++l) {
      total += level_blocks(l);
      if (span[l] < max_span && (widest == k || level_blocks(l) > level_blocks(widest))) widest = l;
    }
    if (total <= resident || widest == k) break;
    span[widest] *= 2u;
  }
  for (uint32_t l = 0; l < k; ++l) {
    const uint64_t b = lo[l], e = lo[l + 1];
    if (e == b) continue;
    for (uint64_t chunk = b / span[l]; chunk < ceil_div(e, span[l]); ++chunk)
      if (!add_block(std::max<uint64_t>(b, chunk * span[l]), std::min<uint64_t>(e, (chunk + 1) * span[l])))
        return TopPlanHost();
  }
  plan.levels = k;
  plan.anc_cap = (plan.anc_cap + 63u) & ~63u;
  return plan;
}

// Longer blocks have more ancestors: where a level with a low fan-out does not fit
// the ancestor table with whole-tile blocks (1M nodes in a level of 700K under one of
// 262K: 384 parents per tile), the plan is built again with shorter ones (that
// hierarchy: 90 us in a launch per level, 58 us merged).
TopPlanHost build_top_plan(const uint32_t *parent, const std::vector<uint64_t> &lo, uint32_t levels,
                           uint64_t resident) {
  // ... and where even single rounds have too many ancestors (a level smaller than the
  // one above it), with that level and those below it left to launches of their own.
  for (uint32_t max_levels = levels; max_levels >= 2u; --max_levels)
    for (uint32_t max_span = (uint32_t)CULL_TILE; max_span >= (uint32_t)FUSED_BLOCK; max_span /= 2u) {
      TopPlanHost plan = build_top_plan_spans(parent, lo, levels, resident, max_span, max_levels);
      if (plan.levels) return plan;
    }
  return TopPlanHost();
}

}  // namespace

extern "C" {

const char *vkrhip_last_error(const vkrhip_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int vkrhip_create(const vkrhip_config *config, vkrhip_ctx **out_ctx) {
  if (!out_ctx) return fail(nullptr, VKRHIP_ERR_ARG, "out_ctx is null");
  *out_ctx = nullptr;
  if (!config || config->struct_size != sizeof(vkrhip_config))
    return fail(nullptr, VKRHIP_ERR_ARG, "config is null or struct_size mismatch");
  if (config->max_views == 0 || config->max_views > VKRHIP_MAX_VIEWS)
    return fail(nullptr, VKRHIP_ERR_ARG, "max_views must be 1..%u", VKRHIP_MAX_VIEWS);
  if (config->max_nodes == 0 || config->max_nodes > 0xfffffffeull ||
      config->max_candidates > 0xfffffffeull)
    return fail(nullptr, VKRHIP_ERR_ARG, "max_nodes/max_candidates must be 1..2^32-2");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, VKRHIP_ERR_HIP, "no HIP device available");
  if (config->device < 0 || config->device >= ndev)
    return fail(nullptr, VKRHIP_ERR_ARG, "device %d out of range (0..%d)", config->device, ndev - 1);
  vkrhip_ctx *c = new (std::nothrow) vkrhip_ctx();
  if (!c) return fail(nullptr, VKRHIP_ERR_HIP, "out of host memory");
  c->cfg = *config;
  c->device = config->device;
  if (c->cfg.visible_capacity == 0) c->cfg.visible_capacity = c->cfg.max_candidates;
  if (const char *env = std::getenv("VKRHIP_FUSED_ROUNDS")) {
    const int r = std::atoi(env);
    if (r == 1 || r == 2 || r == 4) c->fused_rounds = (uint32_t)r;
  }
  if (const char *env = std::getenv("VKRHIP_TOP_LEVELS")) c->top_enabled = std::atoi(env) != 0;
  if (const char *env = std::getenv("VKRHIP_TOP_BALANCE")) c->top_balance = std::atoi(env) != 0;
  if (const char *env = std::getenv("VKRHIP_PARENT_STEPS")) c->steps_enabled = std::atoi(env) != 0;
  if (const char *env = std::getenv("VKRHIP_STREAMING")) c->streaming_mode = std::atoi(env) != 0 ? 1 : 0;
  if (const char *env = std::getenv("VKRHIP_OVERLAP_SORT")) c->overlap_sort = std::atoi(env);
  if (const char *env = std::getenv("VKRHIP_STAGED")) c->staged_compaction = std::atoi(env);
  if (const char *env = std::getenv("VKRHIP_OVERLAP_TAIL")) c->overlap_tail = std::atoi(env);
  if (const char *env = std::getenv("VKRHIP_TAIL_WORKER")) c->worker_mode = std::atoi(env);
  // A gate kernel waits for a kernel of another queue, so the two must be able to run
  // at the same time: a tool that serialises all kernels of the device (hardware
  // counter collection, AMD_SERIALIZE_KERNEL) would leave every gate spinning until it
  // gives up. Those that announce themselves in the environment switch the gates off.
  for (const char *name : {"ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "ROCP_METRICS", "ROCP_INPUT",
                           "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"}) {
    const char *v = std::getenv(name);
    if (v && *v && std::strcmp(v, "0") != 0) c->side_gate = 0;
  }
  if (const char *env = std::getenv("VKRHIP_SIDE_GATE")) c->side_gate = std::atoi(env);
  if (const char *env = std::getenv("VKRHIP_XDEBUG")) c->xdebug = std::atoi(env);
  if (c->cfg.transparent_capacity == 0) c->cfg.transparent_capacity = c->cfg.max_candidates;
  int rc = VKRHIP_OK;
  auto run = [&]() -> int {
    TRY(bind_device(c));
    {
      int cus = 0;
      HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
      c->cu_count = cus > 0 ? (uint32_t)cus : 0u;
    }
    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->tstream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto &e : c->ev_gathered2) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_gathered = c->ev_gathered2[0];
    for (auto &e : c->ev_side) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint64_t mn = c->cfg.max_nodes, mc = std::max<uint64_t>(c->cfg.max_candidates, 1);
    TRY(dev_alloc(c, c->local, mn * 4));
    TRY(dev_alloc(c, c->world, mn * 4));
    for (auto &f : c->flags2) {
      TRY(dev_alloc(c, f, mn + 16));
      HIP_TRY(c, hipMemsetAsync(f.p, 0, mn + 16, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->world.p, 0, mn * 64, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->local.p, 0, mn * 64, c->stream));
    TRY(dev_alloc(c, c->ident, mc * 2));
    TRY(dev_alloc(c, c->sphere, mc));
    TRY(dev_alloc(c, c->cflags, mc));
    TRY(dev_alloc(c, c->cflags16, mc));
    TRY(dev_alloc(c, c->planes, VKRHIP_MAX_VIEWS * 6));
    TRY(dev_alloc(c, c->planes_paired, VKRHIP_MAX_VIEWS * 6));
    TRY(dev_alloc(c, c->tie_eps, VKRHIP_MAX_VIEWS * 6));
    TRY(dev_alloc(c, c->required, VKRHIP_MAX_VIEWS));
    TRY(dev_alloc(c, c->camera_view, 4));
    TRY(dev_alloc(c, c->tvis, mc));
    TRY(dev_alloc(c, c->side_flag, 4));  // (two batch flags, the gates' stall mark, the classify's flag)
    HIP_TRY(c, hipMemsetAsync(c->side_flag.p, 0, 4 * sizeof(uint64_t), c->stream));
    TRY(dev_alloc(c, c->frame_zero, vkrhip_ctx::kBlocks * vkrhip_ctx::kBlockBytes));
    HIP_TRY(c, hipMemsetAsync(c->frame_zero.p, 0, c->frame_zero.count, c->stream));
    TRY(dev_alloc(c, c->tkey2[0], mc));
    TRY(dev_alloc(c, c->states, VKRHIP_MAX_VIEWS));
    TRY(dev_alloc(c, c->row_totals, CULL_MAX_ROWS));
    HIP_TRY(c, hipMemsetAsync(c->states.p, 0, sizeof(CompactionState) * VKRHIP_MAX_VIEWS, c->stream));
    HIP_TRY(c, hipHostMalloc((void **)&c->host_hint, 64, hipHostMallocMapped));
    c->host_hint[0] = 0u;
    HIP_TRY(c, hipHostMalloc((void **)&c->host_counts, 256, hipHostMallocMapped));
    std::memset(c->host_counts, 0, 256);
    c->visible_stride = c->cfg.visible_capacity;
    TRY(dev_alloc(c, c->visible, (size_t)c->cfg.max_views * c->visible_stride));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VKRHIP_OK;
  };
  rc = run();
  if (rc != VKRHIP_OK) {
    g_create_error = c->err;
    vkrhip_destroy(c);
    return rc;
  }
  *out_ctx = c;
  return VKRHIP_OK;
}

void vkrhip_destroy(vkrhip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  worker_stop(c);
  if (c->tstream) (void)hipStreamSynchronize(c->tstream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->tstream) (void)hipStreamDestroy(c->tstream);
  for (auto &e : c->ev_gathered2) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  for (auto &e : c->ev_side) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  c->tstream = nullptr;
  c->ev_gathered = nullptr;
  dev_free(c->tsrc);
  dev_free(c->side_flag);
  dev_free(c->tworld);
  for (EventPair &ev : c->pending) {
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  for (EventPair &ev : c->free_events) {
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  dev_free(c->parent);
  dev_free(c->parent_steps);
  dev_free(c->pos);
  dev_free(c->quat);
  dev_free(c->scale);
  dev_free(c->local);
  dev_free(c->world);
  dev_free(c->flags2[0]);
  dev_free(c->flags2[1]);
  dev_free(c->top_desc);
  dev_free(c->top_anc_node);
  dev_free(c->top_anc_info);
  dev_free(c->node);
  dev_free(c->ident);
  dev_free(c->sphere);
  dev_free(c->cflags);
  dev_free(c->cflags16);
  dev_free(c->planes);
  dev_free(c->planes_paired);
  dev_free(c->tie_eps);
  dev_free(c->required);
  dev_free(c->camera_view);
  for (int k = 0; k < 2; ++k) {
    dev_free(c->seg_masks2[k]);
    dev_free(c->seg_counts2[k]);
    dev_free(c->seg_first2[k]);
    dev_free(c->tkey2[k]);
    dev_free(c->tworld_sparse[k]);
  }
  dev_free(c->dirty_list);
  dev_free(c->dirty_block_counts);
  dev_free(c->dirty_block_offsets);
  dev_free(c->dirty_total);
  dev_free(c->pub_geometries);
  dev_free(c->pub_submeshes);
  dev_free(c->pub_materials);
  dev_free(c->pack_status);
  dev_free(c->packed_index);
  dev_free(c->pack_a);
  dev_free(c->pack_vo);
  dev_free(c->pack_counts);
  dev_free(c->packed_rows);
  dev_free(c->packed_instances);
  dev_free(c->visible_rows);
  dev_free(c->commands);
  dev_free(c->seg_offsets);
  dev_free(c->tvis);
  dev_free(c->frame_zero);
  dev_free(c->states);
  dev_free(c->row_totals);
  dev_free(c->visible);
  dev_free(c->world_bounds);
  dev_free(c->cand_rows);
  dev_free(c->keys_a);
  dev_free(c->keys_b);
  dev_free(c->out_keys);
  dev_free(c->out_source);
  dev_free(c->sort_status);
  dev_free(c->draws);
  dev_free(c->instances);
  dev_free(c->merge_ptrs);
  dev_free(c->merge_lens);
  dev_free(c->concat_desc);
  release_exchange(c);
  dev_free(c->hzb);
  dev_free(c->hzb_offset);
  dev_free(c->hzb_view_projection);
  dev_free(c->exch_desc);
  dev_free(c->transmission_list);
  dev_free(c->transmission_rows);
  dev_free(c->epoch_maps);
  dev_free(c->epoch_maps_low);
  dev_free(c->epoch_wave_maps);
  dev_free(c->epoch_pairs);
  dev_free(c->epoch_result);
  dev_free(c->epoch_enter_low);
  dev_free(c->epoch_enter_high);
  dev_free(c->epoch_wave_occluders);
  dev_free(c->exch_rank_view_start);
  dev_free(c->exch_prefix);
  dev_free(c->exch_runs);
  if (c->host_hint) (void)hipHostFree(c->host_hint);
  if (c->host_counts) (void)hipHostFree(c->host_counts);
  for (void *p : {c->pay_candidates, c->pay_transmission, c->pay_draws, c->pay_instances})
    if (p) (void)hipHostFree(p);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int vkrhip_set_stream(vkrhip_ctx *c, void *hip_stream) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  TRY(join_side(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return VKRHIP_OK;
}

void *vkrhip_get_stream(const vkrhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int vkrhip_join(vkrhip_ctx *c) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  return join_side(c);
}

int vkrhip_sync(vkrhip_ctx *c) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  TRY(join_side(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// scene state
// ---------------------------------------------------------------------------
int vkrhip_set_hierarchy(vkrhip_ctx *c, const uint32_t *parent, uint64_t n_nodes,
                         const uint64_t *level_offsets, uint32_t levels) try {
  if (!c) return VKRHIP_ERR_ARG;
  if (n_nodes > c->cfg.max_nodes)
    return fail(c, VKRHIP_ERR_CAPACITY, "n_nodes %llu exceeds max_nodes %llu",
                (unsigned long long)n_nodes, (unsigned long long)c->cfg.max_nodes);
  if (levels == 0 && n_nodes != 0) return fail(c, VKRHIP_ERR_ARG, "levels is 0");
  if (n_nodes && !level_offsets) return fail(c, VKRHIP_ERR_ARG, "level_offsets is null");
  if (levels > 1 && !parent) return fail(c, VKRHIP_ERR_ARG, "parent is null with %u levels", levels);
  std::vector<uint64_t> lo;
  if (n_nodes) {
    lo.assign(level_offsets, level_offsets + levels + 1);
    if (lo[0] != 0 || lo[levels] != n_nodes)
      return fail(c, VKRHIP_ERR_ARG, "level_offsets must start at 0 and end at n_nodes");
    for (uint32_t l = 0; l < levels; ++l)
      if (lo[l] > lo[l + 1]) return fail(c, VKRHIP_ERR_ARG, "level_offsets not ascending");
    // every node's parent must sit in an earlier level: the level launches
    // rely on it both for correctness and for memory safety
    if (parent) {
      for (uint32_t l = 0; l < levels; ++l) {
        for (uint64_t i = lo[l]; i < lo[l + 1]; ++i) {
          const uint32_t p = parent[i];
          if (p == VKRHIP_NO_PARENT) continue;
          if (l == 0 || p >= lo[l])
            return fail(c, VKRHIP_ERR_ARG, "node %llu (level %u) has parent %u outside earlier levels",
                        (unsigned long long)i, l, p);
        }
      }
    }
  }
  TRY(bind_device(c));
  if (parent && n_nodes) {
    TRY(dev_alloc(c, c->parent, c->cfg.max_nodes));
    HIP_TRY(c, hipMemcpyAsync(c->parent.p, parent, n_nodes * sizeof(uint32_t),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (n_nodes != c->n_nodes) {
    // A frame rewrites the flags of the bound nodes only (one copy read, the other
    // written): the nodes that enter or leave the hierarchy here are current in the
    // copy in use and must read the same in the other one from now on.
    const uint64_t from = std::min(n_nodes, c->n_nodes);
    const uint64_t upto = std::max(n_nodes, c->n_nodes);
    HIP_TRY(c, hipMemcpyAsync(c->flags_other() + from, c->flags() + from, upto - from,
                              hipMemcpyDeviceToDevice, c->stream));
  }
  // stepping levels (large ones only: the table saves bytes, and small levels are not bound by theirs)
  c->steps_entry.assign(levels, -1);
  if (parent && n_nodes && c->steps_enabled) {
    std::vector<uint32_t> table;
    for (uint32_t l = 1; l < levels; ++l) {
      const uint64_t b = lo[l], e = lo[l + 1];
      if (e - b < kStepsMinNodes) continue;
      bool stepping = parent[b] != VKRHIP_NO_PARENT;
      for (uint64_t i = b + 1; i < e && stepping; ++i)
        stepping = parent[i] != VKRHIP_NO_PARENT && parent[i] - parent[i - 1] <= 1u;
      if (!stepping) continue;
      c->steps_entry[l] = (int64_t)(table.size() / 4u);
      for (uint64_t g = b / 64u; g <= (e - 1u) / 64u; ++g) {
        const uint64_t first = std::max(b, g * 64u), end = std::min(e, (g + 1u) * 64u);
        uint64_t mask = 0;
        for (uint64_t i = first + 1u; i < end; ++i)
          if (parent[i] != parent[i - 1]) mask |= 1ull << (i - g * 64u - 1u);
        table.push_back(parent[first]);
        table.push_back((uint32_t)mask);
        table.push_back((uint32_t)(mask >> 32));
        table.push_back(0u);
      }
    }
    if (!table.empty()) {
      TRY(dev_alloc(c, c->parent_steps, (c->cfg.max_nodes / 64u + 2u * levels + 2u) * 4u));
      HIP_TRY(c, hipMemcpyAsync(c->parent_steps.p, table.data(), table.size() * 4u, hipMemcpyHostToDevice,
                                c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  c->n_nodes = n_nodes;
  c->level_offsets = lo;
  c->top_levels = 0;
  c->top_blocks = 0;
  if (c->top_enabled && n_nodes) {
    const TopPlanHost plan =
        build_top_plan(parent, lo, levels, c->top_balance ? (uint64_t)c->cu_count * 4u : 0u);
    if (c->xdebug & 2)
      std::fprintf(stderr, "top plan: %u of %u levels, %zu blocks, %zu ancestor entries (cap %u)\n", plan.levels,
                   levels, plan.desc.size(), plan.anc_node.size(), plan.anc_cap);
    if (plan.levels) {
      TRY(dev_alloc(c, c->top_desc, plan.desc.size()));
      TRY(dev_alloc(c, c->top_anc_node, std::max<size_t>(plan.anc_node.size(), 1)));
      TRY(dev_alloc(c, c->top_anc_info, std::max<size_t>(plan.anc_info.size(), 1)));
      HIP_TRY(c, hipMemcpyAsync(c->top_desc.p, plan.desc.data(), plan.desc.size() * sizeof(TopDesc),
                                hipMemcpyHostToDevice, c->stream));
      if (!plan.anc_node.empty()) {
        HIP_TRY(c, hipMemcpyAsync(c->top_anc_node.p, plan.anc_node.data(), plan.anc_node.size() * 4,
                                  hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->top_anc_info.p, plan.anc_info.data(), plan.anc_info.size() * 4,
                                  hipMemcpyHostToDevice, c->stream));
      }
      HIP_TRY(c, hipStreamSynchronize(c->stream));  // the plan's vectors go out of scope
      c->top_levels = plan.levels;
      c->top_blocks = (uint32_t)plan.desc.size();
      c->top_anc_cap = plan.anc_cap;
    }
  }
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_upload_trs(vkrhip_ctx *c, uint64_t first, uint64_t count, const float *pos4,
                      const float *quat4, const float *scale4) try {
  if (!c || !pos4 || !quat4 || !scale4) return c ? fail(c, VKRHIP_ERR_ARG, "null TRS array") : VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "upload_trs"));
  TRY(bind_device(c));
  TRY(dev_alloc(c, c->pos, c->cfg.max_nodes));
  TRY(dev_alloc(c, c->quat, c->cfg.max_nodes));
  TRY(dev_alloc(c, c->scale, c->cfg.max_nodes));
  c->have_trs = true;
  if (count == 0) return VKRHIP_OK;
  HIP_TRY(c, hipMemcpyAsync(c->pos.p + first, pos4, count * 16, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->quat.p + first, quat4, count * 16, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->scale.p + first, scale4, count * 16, hipMemcpyHostToDevice, c->stream));
  // transform setters set DIRTY_LOCAL (vkr_scene_system.h:70-72)
  std::vector<uint8_t> f(count);
  HIP_TRY(c, hipMemcpyAsync(f.data(), c->flags() + first, count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto &x : f) x |= VKRHIP_DIRTY_LOCAL;
  // into both copies: nodes outside the bound hierarchy are not rewritten by a frame
  // and must read the same whichever copy is current
  for (auto &b : c->flags2)
    HIP_TRY(c, hipMemcpyAsync(b.p + first, f.data(), count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_upload_locals(vkrhip_ctx *c, uint64_t first, uint64_t count, const float *local16) try {
  if (!c || !local16) return c ? fail(c, VKRHIP_ERR_ARG, "null locals") : VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "upload_locals"));
  if (count == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  HIP_TRY(c, hipMemcpyAsync(c->local.p + first * 4, local16, count * 64, hipMemcpyHostToDevice,
                            c->stream));
  std::vector<uint8_t> f(count);
  HIP_TRY(c, hipMemcpyAsync(f.data(), c->flags() + first, count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto &x : f) x |= VKRHIP_DIRTY_WORLD;
  for (auto &b : c->flags2)
    HIP_TRY(c, hipMemcpyAsync(b.p + first, f.data(), count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_upload_worlds(vkrhip_ctx *c, uint64_t first, uint64_t count, const float *world16) {
  if (!c || !world16) return c ? fail(c, VKRHIP_ERR_ARG, "null worlds") : VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "upload_worlds"));
  if (count == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  HIP_TRY(c, hipMemcpyAsync(c->world.p + first * 4, world16, count * 64, hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
}

int vkrhip_upload_node_flags(vkrhip_ctx *c, uint64_t first, uint64_t count, const uint8_t *flags) {
  if (!c || !flags) return c ? fail(c, VKRHIP_ERR_ARG, "null flags") : VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "upload_node_flags"));
  if (count == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  for (auto &b : c->flags2)
    HIP_TRY(c, hipMemcpyAsync(b.p + first, flags, count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
}

int vkrhip_mark_all(vkrhip_ctx *c, uint8_t flags) {
  if (!c) return VKRHIP_ERR_ARG;
  // folded into the next transform launch instead of a separate pass over flags[]
  c->force_flags |= flags;
  return VKRHIP_OK;
}

int vkrhip_mark_nodes(vkrhip_ctx *c, uint64_t first, uint64_t count, uint8_t flags) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "mark_nodes"));
  if (count == 0 || flags == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  // (both copies, like the uploads: a frame reads one and writes the other)
  hipLaunchKernelGGL(k_or_flags, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, c->stream,
                     c->flags2[0].p, c->flags2[1].p, first, count, flags);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// candidates
// ---------------------------------------------------------------------------
int vkrhip_set_candidates(vkrhip_ctx *c, uint64_t n, const uint32_t *node, const uint32_t *ident,
                          const uint32_t *matbits, const float *bounds) try {
  if (!c) return VKRHIP_ERR_ARG;
  if (n > c->cfg.max_candidates)
    return fail(c, VKRHIP_ERR_CAPACITY, "n %llu exceeds max_candidates %llu", (unsigned long long)n,
                (unsigned long long)c->cfg.max_candidates);
  if (n && (!ident || !matbits || !bounds)) return fail(c, VKRHIP_ERR_ARG, "null candidate array");
  if (!node && n > c->cfg.max_nodes)
    return fail(c, VKRHIP_ERR_ARG, "identity node mapping needs n <= max_nodes");
  uint64_t node_limit = 0;
  if (node) {
    // a row must name a node of the bound hierarchy (any node slot before one is bound)
    const uint64_t nodes = c->level_offsets.empty() ? c->cfg.max_nodes : c->n_nodes;
    for (uint64_t i = 0; i < n; ++i) {
      if (node[i] >= nodes)
        return fail(c, VKRHIP_ERR_ARG, "node[%llu] = %u is not a node of the scene (%llu nodes)",
                    (unsigned long long)i, node[i], (unsigned long long)nodes);
      node_limit = std::max<uint64_t>(node_limit, (uint64_t)node[i] + 1u);
    }
  }
  TRY(bind_device(c));
  // the side stream may still emit the last frame's draws from the identity words
  // and the key buffers that are replaced below
  TRY(drain_side(c));
  uint64_t blend = 0;
  std::vector<uint32_t> transmissive;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t m = matbits[i];
    if ((m & VKRHIP_MAT_ALPHA_MASK) == 2u && !(m & VKRHIP_MAT_TRANSMISSIVE)) blend++;
    if (m & VKRHIP_MAT_TRANSMISSIVE) transmissive.push_back((uint32_t)i);
  }
  c->n_transmission = transmissive.size();
  if (!transmissive.empty()) {
    TRY(dev_alloc(c, c->transmission_list, transmissive.size()));
    HIP_TRY(c, hipMemcpyAsync(c->transmission_list.p, transmissive.data(), transmissive.size() * 4,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the vector goes out of scope
  }
  c->n_cand = n;
  c->node_identity = node == nullptr;
  c->node_limit = node_limit;
  c->blend_rows = blend;
  c->published = false;  // the candidate words are rebuilt below; vkrhip_set_publication must follow
  if (n) {
    if (node) {
      TRY(dev_alloc(c, c->node, c->cfg.max_candidates));
      HIP_TRY(c, hipMemcpyAsync(c->node.p, node, n * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->ident.p, ident, n * 32, hipMemcpyHostToDevice, c->stream));
    DevBuf<uint32_t> d_mat;
    DevBuf<float4> d_bounds;
    TRY(dev_alloc(c, d_mat, n));
    int rc = dev_alloc(c, d_bounds, n * 3);
    if (rc != VKRHIP_OK) {
      dev_free(d_mat);
      return rc;
    }
    hipError_t e = hipMemcpyAsync(d_mat.p, matbits, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(d_bounds.p, bounds, n * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_prepare_candidates, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                         c->stream, d_mat.p, d_bounds.p, c->sphere.p, c->cflags.p, c->cflags16.p, n);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_mat);
    dev_free(d_bounds);
    if (e != hipSuccess)
      return fail(c, VKRHIP_ERR_HIP, "candidate upload failed: %s", hipGetErrorString(e));
  }
  // transparent buffers sized by the rows that can ever enter the blend list
  const uint64_t tcap = std::max<uint64_t>(1, std::min<uint64_t>(blend, c->cfg.transparent_capacity));
  if (tcap > c->tcap || !c->keys_a.p) {
    TRY(dev_alloc(c, c->keys_a, tcap));
    TRY(dev_alloc(c, c->keys_b, tcap));
    TRY(dev_alloc(c, c->out_keys, tcap));
    TRY(dev_alloc(c, c->out_source, tcap));
    TRY(dev_alloc(c, c->tsrc, tcap));
    TRY(dev_alloc(c, c->tworld, tcap * 4));
    if (tcap >= (1ull << 30))
      return fail(c, VKRHIP_ERR_CAPACITY, "more than 2^30 sortable blend rows on one GPU");
    c->sort_tiles_stride = (uint32_t)ceil_div(tcap, SORT_TILE);
    // (zeroed once: a status word of another sort must never look like one of ours)
    const size_t status_words = (size_t)SORT_PASSES * c->sort_tiles_stride * SORT_RADIX;
    TRY(dev_alloc(c, c->sort_status, status_words));
    HIP_TRY(c, hipMemsetAsync(c->sort_status.p, 0, c->sort_status.count * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sort_serial = 0;
    TRY(dev_alloc(c, c->draws, tcap * 3));
    TRY(dev_alloc(c, c->instances, tcap * 5));
    c->tcap = tcap;
  }
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_set_views(vkrhip_ctx *c, uint32_t views, const float *planes,
                     const uint32_t *required_flags, const float *camera_view) try {
  if (!c) return VKRHIP_ERR_ARG;
  if (views == 0 || views > c->cfg.max_views)
    return fail(c, VKRHIP_ERR_CAPACITY, "views %u outside 1..max_views %u", views, c->cfg.max_views);
  if (!planes || !required_flags || !camera_view) return fail(c, VKRHIP_ERR_ARG, "null view array");
  TRY(bind_device(c));
  std::vector<float> eps((size_t)views * 6, 0.f);
  for (uint32_t k = 0; k < views * 6; ++k) {
    const float d = std::fabs(planes[4 * k + 3]);
    eps[k] = c->cfg.tie_epsilon * (d > 1.0f ? d : 1.0f);
  }
  // planes 2j and 2j+1 of a view interleaved component by component
  // (nx, nx', ny, ny', nz, nz', d, d'): the operand pairs of the packed plane test
  std::vector<float> paired((size_t)views * 24);
  for (uint32_t v = 0; v < views; ++v)
    for (uint32_t j = 0; j < 3; ++j)
      for (uint32_t k = 0; k < 4; ++k) {
        paired[v * 24 + j * 8 + k * 2 + 0] = planes[v * 24 + (2 * j) * 4 + k];
        paired[v * 24 + j * 8 + k * 2 + 1] = planes[v * 24 + (2 * j + 1) * 4 + k];
      }
  HIP_TRY(c, hipMemcpyAsync(c->planes.p, planes, (size_t)views * 96, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->planes_paired.p, paired.data(), (size_t)views * 96, hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->tie_eps.p, eps.data(), eps.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->required.p, required_flags, (size_t)views * 4, hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->camera_view.p, camera_view, 64, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->views = views;
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_set_publication(vkrhip_ctx *c, const vkrhip_geometry_entry *geometries,
                           uint32_t geometry_capacity, const vkrhip_submesh_range *submeshes,
                           uint32_t submesh_count, const vkrhip_material_entry *materials,
                           uint32_t material_capacity) {
  if (!c) return VKRHIP_ERR_ARG;
  if ((geometry_capacity && !geometries) || (submesh_count && !submeshes) ||
      (material_capacity && !materials))
    return fail(c, VKRHIP_ERR_ARG, "null publication table");
  static_assert(sizeof(vkrhip_geometry_entry) == sizeof(GeometryEntry) &&
                    sizeof(vkrhip_submesh_range) == sizeof(SubmeshRange) &&
                    sizeof(vkrhip_material_entry) == sizeof(MaterialEntry),
                "publication table ABI");
  for (uint32_t g = 0; g < geometry_capacity; ++g)
    if ((uint64_t)geometries[g].submesh_offset + geometries[g].submesh_count > submesh_count)
      return fail(c, VKRHIP_ERR_ARG, "geometry %u names submeshes beyond the submesh table", g + 1u);
  TRY(bind_device(c));
  TRY(drain_side(c));  // a gather on the second stream may still read the row index map
  const uint64_t mc = std::max<uint64_t>(c->cfg.max_candidates, 1);
  TRY(dev_alloc(c, c->pub_geometries, std::max<uint32_t>(geometry_capacity, 1)));
  TRY(dev_alloc(c, c->pub_submeshes, std::max<uint32_t>(submesh_count, 1)));
  TRY(dev_alloc(c, c->pub_materials, std::max<uint32_t>(material_capacity, 1)));
  if (!c->pack_status.p) {
    TRY(dev_alloc(c, c->pack_status, mc));
    TRY(dev_alloc(c, c->packed_index, mc));
    TRY(dev_alloc(c, c->pack_a, mc));
    TRY(dev_alloc(c, c->pack_vo, mc));
    TRY(dev_alloc(c, c->pack_counts, 4));
  }
  if (geometry_capacity)
    HIP_TRY(c, hipMemcpyAsync(c->pub_geometries.p, geometries, (size_t)geometry_capacity * 32,
                              hipMemcpyHostToDevice, c->stream));
  if (submesh_count)
    HIP_TRY(c, hipMemcpyAsync(c->pub_submeshes.p, submeshes, (size_t)submesh_count * 16,
                              hipMemcpyHostToDevice, c->stream));
  if (material_capacity)
    HIP_TRY(c, hipMemcpyAsync(c->pub_materials.p, materials, (size_t)material_capacity * 16,
                              hipMemcpyHostToDevice, c->stream));
  const uint64_t n = c->n_cand;
  if (n) {
    PackResolveArgs a{};
    a.ident = c->ident.p;
    a.geometries = c->pub_geometries.p;
    a.submeshes = c->pub_submeshes.p;
    a.materials = c->pub_materials.p;
    a.cflags = c->cflags.p;
    a.cflags16 = c->cflags16.p;
    a.pack_a = c->pack_a.p;
    a.pack_vo = c->pack_vo.p;
    a.status = c->pack_status.p;
    a.n = n;
    a.geometry_capacity = geometry_capacity;
    a.material_capacity = material_capacity;
    hipLaunchKernelGGL(k_pack_resolve, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream, a);
  }
  hipLaunchKernelGGL(k_pack_index, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, c->pack_status.p,
                     c->packed_index.p, n, c->pack_counts.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->pack_counts_host, c->pack_counts.p, 16, hipMemcpyDeviceToHost,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->published = true;
  return VKRHIP_OK;
}

}  // extern "C"

static int launch_pack(vkrhip_ctx *c) {
  const uint64_t mc = std::max<uint64_t>(c->cfg.max_candidates, 1);
  if (!c->packed_rows.p) {
    TRY(dev_alloc(c, c->packed_rows, mc * 3));
    TRY(dev_alloc(c, c->packed_instances, mc * 5));
  }
  if (!c->n_cand) return VKRHIP_OK;
  PackEmitArgs a{};
  a.status = c->pack_status.p;
  a.packed_index = c->packed_index.p;
  a.pack_a = c->pack_a.p;
  a.pack_vo = c->pack_vo.p;
  a.cflags = c->cflags.p;
  a.sphere = c->sphere.p;
  a.ident = c->ident.p;
  a.world = c->world.p;
  a.node = c->node_identity ? nullptr : c->node.p;
  a.rows = c->packed_rows.p;
  a.instances = c->packed_instances.p;
  a.n = c->n_cand;
  hipLaunchKernelGGL(k_emit_packed, dim3((unsigned)ceil_div(c->n_cand, 32)), dim3(256), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

static int launch_encode(vkrhip_ctx *c) {
  if (!c->visible_rows.p)
    TRY(dev_alloc(c, c->visible_rows, (size_t)c->cfg.max_views * c->visible_stride * 2));
  if (c->cfg.partition_capacity && !c->commands.p)
    TRY(dev_alloc(c, c->commands, (size_t)c->cfg.max_views * 4u * c->cfg.partition_capacity * 5u));
  if (!c->n_cand) return VKRHIP_OK;
  EncodeArgs a{};
  a.visible = c->visible.p;
  a.states = c->states.p;
  a.rows = c->packed_rows.p;
  a.visible_rows = c->visible_rows.p;
  a.commands = c->commands.p;
  a.visible_stride = c->visible_stride;
  a.views = c->views;
  a.partition_capacity = c->cfg.partition_capacity;
  // visible counts live on the device: cover what the packed rows can yield
  const uint64_t slots = std::min<uint64_t>(c->pack_counts_host[0], c->visible_stride);
  if (!slots) return VKRHIP_OK;
  hipLaunchKernelGGL(k_encode_visible, dim3((unsigned)ceil_div(slots * 2, 256), c->views), dim3(256), 0,
                     c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// the frame
// ---------------------------------------------------------------------------
static TransformArgs transform_args(vkrhip_ctx *c, uint32_t level) {
  TransformArgs a{};
  a.parent = c->parent.p;
  a.parent_steps = nullptr;
  if (level < c->steps_entry.size() && c->steps_entry[level] >= 0)  // indexed by the global chunk
    a.parent_steps = c->parent_steps.p + (c->steps_entry[level] - (int64_t)(c->level_offsets[level] / 64u)) * 4;
  a.pos = c->have_trs ? c->pos.p : nullptr;
  a.quat = c->quat.p;
  a.scale = c->scale.p;
  a.local = c->local.p;
  a.world = c->world.p;
  a.flags_in = c->flags();
  a.flags_out = c->flags_other();
  a.counters = c->counters();
  a.begin = c->level_offsets[level];
  a.end = c->level_offsets[level + 1];
  a.force = c->force_flags;
  return a;
}

static CullArgs cull_args(vkrhip_ctx *c) {
  CullArgs a{};
  a.world = c->world.p;
  a.node = c->node_identity ? nullptr : c->node.p;
  a.sphere = c->sphere.p;
  a.cflags = c->cflags.p;
  a.cflags16 = c->cflags16.p;
  a.planes = c->planes.p;
  a.planes_paired = c->planes_paired.p;
  a.tie_eps = c->tie_eps.p;
  a.required = c->required.p;
  a.view = c->camera_view.p;
  a.tvis = c->tvis.p;
  a.tkey = c->tkey().p;
  a.tworld_sparse = c->frame_tail ? c->tworld_sparse[c->par].p : nullptr;
  a.world_bounds = c->world_bounds.p;
  a.seg_masks = c->seg_masks().p;
  a.seg_counts = c->seg_counts().p;
  a.seg_first = c->seg_first().p;
  a.counters = c->counters();
  a.ties = c->ties();
  a.caster_bounds = (c->frame_stages & VKRHIP_STAGE_CASTER_BOUNDS) ? c->caster_box() : nullptr;
  a.n = c->n_cand;
  a.seg_stride = c->seg_stride;
  a.index_base = c->cfg.source_index_base;
  a.views = c->views;
  a.hzb = c->hzb.p;
  a.hzb_offset = c->hzb_offset.p;
  a.hzb_view_projection = c->hzb_view_projection.p;
  a.hzb_width = c->hzb_width;
  a.hzb_height = c->hzb_height;
  a.hzb_mips = c->hzb_mips;
  a.hzb_epsilon = c->hzb_epsilon;
  return a;
}

static uint64_t fused_level_grid(const vkrhip_ctx *c, uint32_t l, uint32_t *rounds);
static bool frame_streams(const vkrhip_ctx *c);
constexpr uint64_t kStagedMinSegments = 4096;  // see launch_fused

// The merged launch of the top levels: ancestor tables, dynamic LDS behind the
// cull tables.
static size_t top_lds_bytes(const vkrhip_ctx *c) {
  return (size_t)c->top_anc_cap * (64u + 4u);
}
static void top_args(vkrhip_ctx *c, FusedArgs &a, size_t cull_lds) {
  a.t = transform_args(c, 0);
  a.t.begin = 0;
  a.t.end = c->level_offsets[c->top_levels];
  a.rounds = FUSED_ROUNDS;
  a.top = c->top_desc.p;
  a.top_anc_node = c->top_anc_node.p;
  a.top_anc_info = c->top_anc_info.p;
  a.top_lds_offset = (uint32_t)((cull_lds + 15u) & ~(size_t)15u);
  a.top_anc_cap = c->top_anc_cap;
}

// Transform only: the fused kernel without its cull phase.
static int launch_transform(vkrhip_ctx *c) {
  bool first = true;
  const uint32_t levels = c->level_offsets.empty() ? 0u : (uint32_t)c->level_offsets.size() - 1u;
  uint32_t l0 = 0;
  if (c->top_levels && c->parent.p) {
    FusedArgs a{};
    top_args(c, a, 0);
    launch_timed(c, VKRHIP_TIMER_TRANSFORM, first, k_propagate_classify<true, false, false, false, true>,
                 dim3(c->top_blocks), dim3(FUSED_BLOCK), a.top_lds_offset + top_lds_bytes(c), a);
    first = false;
    l0 = c->top_levels;
  }
  for (uint32_t l = l0; l < levels; ++l) {
    FusedArgs a{};
    const uint64_t grid = fused_level_grid(c, l, &a.rounds);
    if (!grid) continue;
    a.t = transform_args(c, l);
    const bool root = l == 0 || !c->parent.p;
    const dim3 g((unsigned)grid), b(FUSED_BLOCK);
    if (frame_streams(c)) {
      if (root)
        launch_timed(c, VKRHIP_TIMER_TRANSFORM, first,
                     k_propagate_classify<false, false, false, false, false, true>, g, b, 0, a);
      else
        launch_timed(c, VKRHIP_TIMER_TRANSFORM, first,
                     k_propagate_classify<true, false, false, false, false, true>, g, b, 0, a);
    } else {
      if (root)
        launch_timed(c, VKRHIP_TIMER_TRANSFORM, first, k_propagate_classify<false, false, false, false>, g, b, 0, a);
      else
        launch_timed(c, VKRHIP_TIMER_TRANSFORM, first, k_propagate_classify<true, false, false, false>, g, b, 0, a);
    }
    first = false;
  }
  c->force_flags = 0;
  if (!first) c->flags_cur ^= 1u;  // the launches wrote every node's flags to the other copy
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

// Grows the segment tables to hold `segments` columns (whole uint4s per row), and
// gives them their second copy once a frame wants one.
static int ensure_segments(vkrhip_ctx *c, uint64_t segments) {
  const uint64_t need = (segments + 3u) & ~(uint64_t)3u;
  const bool add_copy = c->want_second && !c->second_copy;
  if (need <= c->seg_stride && !add_copy) return VKRHIP_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  TRY(drain_side(c));
  uint64_t stride = c->seg_alloc_stride;
  if (need > c->seg_stride) {
    for (int k = 0; k < 2; ++k) {
      dev_free(c->seg_counts2[k]);
      dev_free(c->seg_first2[k]);
      dev_free(c->seg_masks2[k]);
    }
    dev_free(c->seg_offsets);
    c->seg_stride = 0;
    c->second_copy = false;
    stride = need + need / 4u;  // head room for a growing scene
  }
  const size_t rows = (size_t)c->cfg.max_views * 4u + 1u;
  const uint64_t kept = stride & ~(uint64_t)3u;
  // (large enough for either use: the segments' records, or the blend-row masks and
  // the staging lists of segment_finish_staged)
  const size_t mask_words = std::max<size_t>(
      stride * seg_record_words(c->cfg.max_views),
      c->staged_compaction ? staged_table_words(kept, std::min<uint32_t>(c->cfg.max_views, STAGED_MAX_VIEWS)) : 0u);
  const int copies = c->want_second ? 2 : 1;
  for (int k = 0; k < copies; ++k) {
    if (c->seg_masks2[k].p) continue;  // copy 0 when only the second copy is added
    TRY(dev_alloc(c, c->seg_counts2[k], rows * stride));
    TRY(dev_alloc(c, c->seg_first2[k], stride));
    TRY(dev_alloc(c, c->seg_masks2[k], mask_words));
    HIP_TRY(c, hipMemsetAsync(c->seg_masks2[k].p, 0, c->seg_masks2[k].count * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->seg_first2[k].p, 0, c->seg_first2[k].count * sizeof(uint32_t), c->stream));
  }
  TRY(dev_alloc(c, c->seg_offsets, rows * stride));
  c->seg_alloc_stride = stride;
  c->seg_stride = kept;
  c->second_copy = copies == 2;
  if (!c->second_copy) c->par = 0;
  return VKRHIP_OK;
}

// What a frame needs whose whole tail runs on the second stream: the sparse keys
// and the set-aside matrices twice (the segment tables get their second copy in
// ensure_segments).
static int ensure_tail_buffers(vkrhip_ctx *c) {
  const uint64_t mc = std::max<uint64_t>(c->cfg.max_candidates, 1);
  TRY(dev_alloc(c, c->tkey2[1], mc));
  for (auto &b : c->tworld_sparse) TRY(dev_alloc(c, b, mc * 4u));
  c->want_second = true;
  return VKRHIP_OK;
}

// Blocks (= cull segments) the fused kernel launches for level l, and the rounds
// of a tile each block walks: whole 1024-node tiles only for large levels, so
// that small levels still give every CU several blocks.
static uint64_t fused_level_grid(const vkrhip_ctx *c, uint32_t l, uint32_t *rounds) {
  const uint64_t begin = c->level_offsets[l], end = c->level_offsets[l + 1];
  if (end == begin) return 0;
  const uint64_t tiles = ceil_div(end, CULL_TILE) - begin / CULL_TILE;
  *rounds = c->fused_rounds ? c->fused_rounds : (tiles >= 1024 ? 4u : tiles >= 512 ? 2u : 1u);
  return tiles * (FUSED_ROUNDS / *rounds);
}

// Whether the level launches of a frame mark their loads and stores as streaming:
// when what the frame reads and writes per node (about 160 bytes) does not fit
// the last-level cache, so that nothing of it would be found there by the next
// frame anyway.
static bool frame_streams(const vkrhip_ctx *c) {
  if (c->streaming_mode >= 0) return c->streaming_mode != 0;
  return c->n_nodes * 160u > kStreamingFootprint;
}

template <bool HAS_PARENT, bool TOP = false, bool STREAM = false>
static void launch_fused_variant(vkrhip_ctx *c, const FusedArgs &a, unsigned grid, bool ties,
                                 bool bounds_out, bool first, bool keep = false) {
  const dim3 g(grid), b(FUSED_BLOCK);
  const size_t lds = TOP ? a.top_lds_offset + top_lds_bytes(c)
                         : cull_shared_bytes(a.c.views);
  // (root level of a frame whose tail runs beside the next frame: see fused_waves)
  constexpr bool kRootLevel = !HAS_PARENT && !TOP;
  if (c->frame_staged) {  // (never with tie counting or world bounds)
    if (kRootLevel && c->frame_tail)
      launch_timed(c, VKRHIP_TIMER_FUSED, first,
                   k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, true, kRootLevel>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_FUSED, first,
                   k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, true>, g, b, lds, a);
    return;
  }
  if (kRootLevel && c->frame_tail && !ties && !bounds_out) {
    launch_timed(c, VKRHIP_TIMER_FUSED, first,
                 k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, false, kRootLevel>, g, b, lds, a);
    return;
  }
  if (STREAM && !TOP && HAS_PARENT && a.t.parent_steps && !ties && !bounds_out) {  // (see STEPS in k_fused.hpp)
    constexpr bool kSteps = STREAM && !TOP && HAS_PARENT;
    if (keep)
      launch_timed(c, VKRHIP_TIMER_FUSED, first,
                   k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, false, false, kSteps, kSteps>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_FUSED, first,
                   k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, false, false, false, kSteps>, g, b, lds, a);
    return;
  }
  if (STREAM && keep && !ties && !bounds_out) {  // (see KEEP in k_fused.hpp)
    launch_timed(c, VKRHIP_TIMER_FUSED, first,
                 k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM, false, false, STREAM>, g, b, lds, a);
    return;
  }
  if (ties) {
    if (bounds_out)
      launch_timed(c, VKRHIP_TIMER_FUSED, first, k_propagate_classify<HAS_PARENT, true, true, true, TOP, STREAM>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_FUSED, first, k_propagate_classify<HAS_PARENT, true, false, true, TOP, STREAM>, g, b, lds, a);
  } else {
    if (bounds_out)
      launch_timed(c, VKRHIP_TIMER_FUSED, first, k_propagate_classify<HAS_PARENT, false, true, true, TOP, STREAM>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_FUSED, first, k_propagate_classify<HAS_PARENT, false, false, true, TOP, STREAM>, g, b, lds, a);
  }
}

// Transform and classify in one launch per level (candidate i <-> node i); the
// small top levels of a hierarchy share one launch when a plan exists.
static int launch_fused(vkrhip_ctx *c, bool bounds_out) {
  const uint32_t levels = (uint32_t)c->level_offsets.size() - 1u;
  const uint32_t l0 = c->top_levels && c->parent.p ? c->top_levels : 0u;
  uint64_t segments = l0 ? c->top_blocks : 0u;
  for (uint32_t l = l0; l < levels; ++l) {
    uint32_t rounds = 4;
    segments += fused_level_grid(c, l, &rounds);
  }
  const bool ties = c->cfg.tie_epsilon > 0.f;
  // the blocks expand their own survivors where that is built: the plain cull
  // variants, up to STAGED_MAX_VIEWS views
  // By default only large flat scenes do it. Measured on one box (frame, fused kernel,
  // compaction behind it; gather / staged): 32M roots with 1, 2, 3 views 1010 / 977,
  // 1027 / 990, 1026 / 993 us - the root-level variant of the kernel even runs
  // faster with the expansion at its blocks' end than with the stores of the
  // segment records (912 / 895 us), and k_place takes 24 us where k_gather takes
  // 35; a 16M-node hierarchy with 1, 2, 5 views 560 / 570, 583 / 596, 599 / 622 us -
  // the below-root variant (128 registers, four waves per SIMD) pays 5 to 19 us for
  // it and k_place is no shorter than k_gather there (59 / 53 us at 5 views: its
  // waves are few and long). A frame of few segments is one resident set of blocks
  // or less: the expansion then lengthens the launch by its full duration (1M
  // roots: 42.5 / 47.3 us).
  const bool flat = levels == 1u || !c->parent.p;
  c->frame_staged = !ties && !bounds_out && c->views <= STAGED_MAX_VIEWS &&
                    (c->staged_compaction >= 2 ||
                     (c->staged_compaction == 1 && flat && segments >= kStagedMinSegments));
  TRY(ensure_segments(c, segments));
  bool first = true;
  uint64_t seg_base = 0;
  // the frame's last launch (its completion releases the tail on the second stream)
  int last_level = -1;
  for (uint32_t l = l0; l < levels; ++l) {
    uint32_t rounds = 4;
    if (fused_level_grid(c, l, &rounds)) last_level = (int)l;
  }
  if (l0) {
    FusedArgs a{};
    a.c = cull_args(c);
    top_args(c, a, cull_shared_bytes(c->views));
    a.seg_base = 0;
    if (c->frame_tail && !c->gathered_by_gate && last_level < 0) c->attach_stop = c->ev_gathered;
    // (a large frame streams through the merged levels as well: 16M-node hierarchy 555 -> 552 us;
    // the launches behind it read the last merged level's world matrices as their parents')
    if (frame_streams(c) && !ties && !bounds_out && !c->frame_staged && last_level >= 0)
      launch_fused_variant<true, true, true>(c, a, c->top_blocks, ties, bounds_out, first, true);
    else
      launch_fused_variant<true, true>(c, a, c->top_blocks, ties, bounds_out, first);
    first = false;
    seg_base = c->top_blocks;
  }
  for (uint32_t l = l0; l < levels; ++l) {
    FusedArgs a{};
    const uint64_t grid = fused_level_grid(c, l, &a.rounds);
    if (!grid) continue;
    a.t = transform_args(c, l);
    a.c = cull_args(c);
    a.seg_base = (uint32_t)seg_base;
    const bool root = l == 0 || !c->parent.p;
    // (a later launch of the frame reads this level's world matrices as its parents')
    const bool keep = (int)l < last_level && c->parent.p;
    if (c->frame_tail && !c->gathered_by_gate && (int)l == last_level) c->attach_stop = c->ev_gathered;
    if (frame_streams(c)) {
      if (root)
        launch_fused_variant<false, false, true>(c, a, (unsigned)grid, ties, bounds_out, first, keep);
      else
        launch_fused_variant<true, false, true>(c, a, (unsigned)grid, ties, bounds_out, first, keep);
    } else {
      if (root)
        launch_fused_variant<false>(c, a, (unsigned)grid, ties, bounds_out, first);
      else
        launch_fused_variant<true>(c, a, (unsigned)grid, ties, bounds_out, first);
    }
    first = false;
    seg_base += grid;
  }
  c->segments = segments;
  c->force_flags = 0;
  if (!first) c->flags_cur ^= 1u;  // the launches wrote every node's flags to the other copy
  if (c->frame_tail && !c->gathered_by_gate && first)
    HIP_TRY(c, hipEventRecord(c->ev_gathered, c->stream));  // (no launch at all)
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

template <bool IDENTITY>
static void launch_classify_variant(vkrhip_ctx *c, const CullArgs &a, unsigned grid, bool ties,
                                    bool bounds_out) {
  const dim3 g(grid), b(CULL_BLOCK);
  const size_t lds = cull_shared_bytes(a.views);
  if (a.hzb_mips) {  // the occlusion variants
    if (ties) {
      if (bounds_out)
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, true, true>, g, b, lds, a);
      else
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, false, true>, g, b, lds, a);
    } else {
      if (bounds_out)
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, true, true>, g, b, lds, a);
      else
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, false, true>, g, b, lds, a);
    }
    return;
  }
  // 85 bytes move per row: streaming loads once that exceeds the last-level cache
  const bool stream = c->streaming_mode >= 0 ? c->streaming_mode != 0 : c->n_cand * 85u > kStreamingFootprint;
  if (stream) {
    if (ties) {
      if (bounds_out)
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, true, false, true>, g, b, lds, a);
      else
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, false, false, true>, g, b, lds, a);
    } else {
      if (bounds_out)
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, true, false, true>, g, b, lds, a);
      else
        launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, false, false, true>, g, b, lds, a);
    }
    return;
  }
  if (ties) {
    if (bounds_out)
      launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, true>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, true, false>, g, b, lds, a);
  } else {
    if (bounds_out)
      launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, true>, g, b, lds, a);
    else
      launch_timed(c, VKRHIP_TIMER_CLASSIFY, true, k_classify<IDENTITY, false, false>, g, b, lds, a);
  }
}

static int launch_classify(vkrhip_ctx *c, bool bounds_out) {
  c->frame_staged = false;
  c->segments = ceil_div(c->n_cand, CULL_TILE);
  if (!c->n_cand) return VKRHIP_OK;
  TRY(ensure_segments(c, c->segments));
  const CullArgs a = cull_args(c);
  const bool ties = c->cfg.tie_epsilon > 0.f;
  if (c->node_identity)
    launch_classify_variant<true>(c, a, (unsigned)c->segments, ties, bounds_out);
  else
    launch_classify_variant<false>(c, a, (unsigned)c->segments, ties, bounds_out);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

// `on`: the frame's stream, or the second stream when the whole tail runs there.
static int launch_compaction(vkrhip_ctx *c, bool set_aside, hipStream_t on) {
  const uint64_t segments = c->n_cand ? c->segments : 0;
  if (!segments) {  // nothing to compact: empty states, zero transparent rows (counters are zeroed)
    HIP_TRY(c, hipMemsetAsync(c->states.p, 0, sizeof(CompactionState) * VKRHIP_MAX_VIEWS, on));
    return VKRHIP_OK;
  }
  const uint32_t next_block = (c->cur_block + 2u) % vkrhip_ctx::kBlocks;
  {
    StageTimer timer(c, VKRHIP_TIMER_SCAN, on);
    ScanArgs s{};
    s.seg_counts = c->seg_counts().p;
    s.seg_offsets = c->seg_offsets.p;
    s.row_totals = c->row_totals.p;
    s.counters = c->counters();
    s.segments = segments;
    s.seg_stride = c->seg_stride;
    s.views = c->views;
    s.transparent_capacity = c->tcap;
    s.host_hint = c->host_hint;
    s.clear = reinterpret_cast<uint32_t *>(c->counter_block(next_block));
    s.clear_words = (uint32_t)(vkrhip_ctx::kBlockBytes / sizeof(uint32_t));
    if (segments <= 4096u)
      launch_on(c, on, k_scan<256>, dim3(c->views * 4u + 1u), dim3(256), 0, s);
    else
      launch_on(c, on, k_scan<SCAN_BLOCK>, dim3(c->views * 4u + 1u), dim3(SCAN_BLOCK), 0, s);
    HIP_TRY(c, hipGetLastError());
    c->block_clean[next_block] = true;
  }
  {
    StageTimer timer(c, VKRHIP_TIMER_SCATTER, on);
    GatherArgs g{};
    g.seg_masks = c->seg_masks().p;
    g.seg_counts = c->seg_counts().p;
    g.seg_offsets = c->seg_offsets.p;
    g.seg_first = c->seg_first().p;
    g.row_totals = c->row_totals.p;
    g.tkey = c->tkey().p;
    g.remap = c->published ? c->packed_index.p : nullptr;
    g.states = c->states.p;
    g.visible = c->visible.p;
    g.keys = c->keys_a.p;
    // (a tail on the second stream finds the rows' matrices under the candidate index:
    // the keys carry the candidate and nothing is looked up by place)
    g.tsrc = c->frame_tail ? nullptr : c->tsrc.p;
    g.world = c->world.p;
    g.node = c->node_identity ? nullptr : c->node.p;
    g.tworld = set_aside ? c->tworld.p : nullptr;
    g.segments = segments;
    g.seg_stride = c->seg_stride;
    g.visible_stride = c->visible_stride;
    g.transparent_capacity = c->tcap;
    g.index_base = c->cfg.source_index_base;
    g.candidates = c->n_cand;
    g.views = c->views;
    g.partition_capacity = c->cfg.partition_capacity;
    // (a wave of k_gather takes one segment, a wave of k_place PLACE_SEGS)
    // (few segments: a wave per (segment, view), see GatherArgs::view_parts)
    const bool plain_gather = !c->frame_staged && c->views <= GATHER_MANY_VIEWS;
    g.view_parts = plain_gather && c->views > 1u && segments <= kGatherSplitSegments ? c->views : 1u;
    const unsigned grid = (unsigned)ceil_div(segments * g.view_parts,
                                             (GATHER_BLOCK / 64) * (c->frame_staged ? PLACE_SEGS : 1u));
    launch_on(c, on, c->frame_staged ? k_place : (c->views > GATHER_MANY_VIEWS ? k_gather_many : k_gather),
              dim3(grid), dim3(GATHER_BLOCK), 0, g);
    HIP_TRY(c, hipGetLastError());
  }
  return VKRHIP_OK;
}

// Hierarchies below this size keep the transparent sort on the frame's stream.
constexpr uint64_t OVERLAP_MIN_NODES = 4ull << 20;
// ... and below this size the whole tail. With the worker thread enqueuing the tail and
// kernels instead of events between the streams a frame costs the calling thread and
// the worker about 11 us each, and the second stream wins at every size measured (flat,
// one view: 20K nodes 25.3 us on one stream / 20.0 on two, 131K 24.5 / 18.4, 1M 40.8 /
// 35.8; depth 8, five views: 20K 37.3 / 29.7, 65K 52.4 / 36.2, 524K 119 / 90, 3M 275 /
// 206).
constexpr uint64_t TAIL_MIN_NODES = 1ull << 14;
constexpr uint64_t TAIL_MIN_NODES_EVENTS = 1000000ull;

// Whether the frame's transparent sort and emission go to the second stream,
// beside the next frame's transform. They must then read nothing that transform
// overwrites: the gather sets the rows' world matrices aside, everything else
// they read the gather wrote for them anyway. The two cross-stream waits cost
// about a quarter of a small frame, so small hierarchies stay on one stream; so
// does a caller's own stream (vkrhip_set_stream promises all work on it). With the
// multi-GPU exchange configured the pack follows the sort on the second stream
// (vkrhip_exchange_run), so the context's stream does not wait for it either.
static bool second_stream_usable(const vkrhip_ctx *c) { return c->stream == c->own_stream; }
static bool sort_on_second_stream(const vkrhip_ctx *c) {
  return second_stream_usable(c) &&
         (c->overlap_sort >= 2 || (c->overlap_sort == 1 && c->n_nodes >= OVERLAP_MIN_NODES));
}

// Whether the WHOLE tail of the frame - scan, gather, sort, emission - goes to the
// second stream, beside the next frame's transform and classify. The next frame
// then starts right behind this frame's last classify launch. Nothing the tail
// reads may be rewritten by that frame: the segment tables and sparse keys exist
// twice, and the classify blocks set the visible blend rows' world matrices aside
// themselves (CullArgs::tworld_sparse), so the fused kernel is a precondition.
// A block of another stream is placed beside a running grid wherever its waves fit
// next to that grid's: the large-level variant of the fused kernel holds four waves
// of 128 registers per SIMD, i.e. the whole register file, and every block it
// retires frees room for one wave of at most 128 registers per SIMD - which is why
// every kernel of the tail is kept at or below that (a radix pass of 204 registers
// waited for the grid's last blocks: 572 us per frame of the 16M-node hierarchy
// against 564 us). Measured per frame, tail here / on the frame's stream: 16M-node
// hierarchy with five views 564 / 597 us, 32M flat nodes with one view 921 / 936
// us; with 64 views the fused kernel is bound by its vector instructions and the
// gather's compete with them (1336 / 1304 us), and a frame of 1M nodes is too short
// for the two cross-stream waits (43.8 / 42.6 us): many views and small scenes keep
// the tail where it is. Stages that consume the frame's lists or counters right
// behind it (candidate rows' published totals, pack, encode) keep it there as well.
constexpr uint32_t TAIL_MAX_VIEWS = 8;
static bool tail_on_second_stream(const vkrhip_ctx *c, uint32_t stages, bool fused) {
  if (!fused || !second_stream_usable(c) || !(stages & VKRHIP_STAGE_TRANSPARENT)) return false;
  if (stages & (VKRHIP_STAGE_CANDIDATES | VKRHIP_STAGE_PACK | VKRHIP_STAGE_ENCODE)) return false;
  if (c->overlap_tail >= 2) return true;
  // (with events between the streams the wait packet costs small frames more than the
  // second stream gives them: flat, one view, 524K nodes 29.8 us on one stream / 33.2 on two)
  // (with the multi-GPU exchange configured the host enqueues a pack, an all-gather and a
  // merge per frame as well and small frames are bound by that: 1M nodes 58 us per frame
  // on one stream, 78 on two)
  const uint64_t min_nodes =
      c->x_world ? OVERLAP_MIN_NODES : (c->side_gate ? TAIL_MIN_NODES : TAIL_MIN_NODES_EVENTS);
  return c->overlap_tail == 1 && c->overlap_sort != 0 && c->n_nodes >= min_nodes &&
         c->views <= TAIL_MAX_VIEWS;
}

static EmitArgs emit_args(vkrhip_ctx *c, bool set_aside) {
  EmitArgs e{};
  e.keys = c->keys_a.p;
  e.counters = c->counters();
  e.tworld = set_aside && !c->frame_tail ? c->tworld.p : nullptr;
  e.tsrc = c->tsrc.p;
  e.world = c->world.p;
  e.node = c->node_identity ? nullptr : c->node.p;
  if (c->frame_tail) {  // the classify blocks set the matrices aside, under the candidate index
    e.world = c->tworld_sparse[c->par].p;
    e.node = nullptr;
    e.tsrc = nullptr;
  }
  e.ident = c->ident.p;
  e.draws = c->draws.p;
  e.instances = c->instances.p;
  e.out_keys = c->out_keys.p;
  e.out_source = c->out_source.p;
  e.index_base = c->cfg.source_index_base;
  e.capacity = (uint32_t)c->tcap;
  return e;
}

static int launch_transparent(vkrhip_ctx *c, bool overlap) {
  const uint64_t cap = std::min<uint64_t>(c->tcap, std::max<uint64_t>(c->blend_rows, 1));
  // The element count is only known on the device; the host has the static bound
  // `cap` and the count the scan kernel reported for an earlier frame. When
  // either says the count is small, one launch ranks every key against all
  // others and emits the rows (k_rank_emit). Otherwise the multi-pass radix
  // kernels run on the depth bits (the low word is already ordered by
  // construction) and a separate emission launch follows. Every kernel reads
  // the real count on the device and both paths sort any count, so a wrong
  // guess costs time, never correctness.
  const uint32_t hint = *reinterpret_cast<volatile uint32_t *>(c->host_hint);
  uint64_t expected = cap;
  if (hint != 0u) expected = std::min<uint64_t>(cap, (uint64_t)(hint - 1u) * 5u / 4u + 256u);
  hipStream_t on = c->stream;
  if (overlap) {
    on = c->tstream;
    if (!c->frame_tail) {  // (the whole tail: the side stream already waits behind the classify)
      HIP_TRY(c, hipEventRecord(c->ev_gathered, c->stream));
      HIP_TRY(c, hipStreamWaitEvent(on, c->ev_gathered, 0));
    }
  }
  if (expected <= RANK_LIMIT) {
    StageTimer timer(c, VKRHIP_TIMER_SORT, on);
    const unsigned grid = (unsigned)std::max<uint64_t>(1, ceil_div(expected, (uint64_t)RANK_BLOCK_KEYS));
    launch_on(c, on, k_rank_emit, dim3(grid), dim3(SMALL_THREADS), 0, emit_args(c, overlap), c->keys_a.p,
              c->keys_b.p);
    HIP_TRY(c, hipGetLastError());
  } else {
    {
      StageTimer timer(c, VKRHIP_TIMER_SORT, on);
      // a block per expected tile, never more than the device holds at once (four
      // 256-thread blocks per CU); tiles beyond the grid are handed out by ticket
      const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(expected, SORT_TILE), 1024);
      if (++c->sort_serial == 0u) c->sort_serial = 1u;
      SortArgs s{};
      s.state = c->sort_state();
      c->sort_block = c->cur_block;
      s.status = c->sort_status.p;
      s.counters = c->counters();
      s.tiles_stride = c->sort_tiles_stride;
      s.serial = c->sort_serial;
      s.capacity = (uint32_t)c->tcap;
      s.keys_in = c->keys_a.p;
      launch_on(c, on, k_sort_hist, dim3(grid), dim3(SORT_BLOCK), 0, s);
      bool in_a = true;
      for (uint32_t pass = 0; pass < (uint32_t)SORT_PASSES; ++pass) {  // four passes end in keys_a again
        s.keys_in = in_a ? c->keys_a.p : c->keys_b.p;
        s.keys_out = in_a ? c->keys_b.p : c->keys_a.p;
        s.pass = pass;
        s.descending = pass + 1u == (uint32_t)SORT_PASSES;  // the draw order is back to front
        launch_on(c, on, k_sort_pass, dim3(grid), dim3(SORT_BLOCK), 0, s);
        in_a = !in_a;
      }
      HIP_TRY(c, hipGetLastError());
    }
    {
      StageTimer timer(c, VKRHIP_TIMER_EMIT, on);
      const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(expected, EMIT_BLOCK_ROWS), 4096);
      EmitArgs e = emit_args(c, overlap);
      e.keys_descending = 1u;
      launch_on(c, on, k_emit_transparent, dim3(grid), dim3(256), 0, e);
      HIP_TRY(c, hipGetLastError());
    }
  }
  if (overlap) {
    int e = -1;
    TRY(record_side(c, &e));
    if (c->frame_tail) {
      c->copy_event[c->par] = e;  // the batch read this copy of the segment tables ...
      c->block_event[(c->cur_block + 2u) % vkrhip_ctx::kBlocks] = e;  // ... and its scan zeroes this block
    }
  }
  return VKRHIP_OK;
}

// The nodes the frame's transform update rewrote, ascending (= topological order).
static int launch_dirty_list(vkrhip_ctx *c) {
  const uint64_t mn = c->cfg.max_nodes;
  const uint64_t max_blocks = ceil_div(mn, DIRTY_NODES);
  TRY(dev_alloc(c, c->dirty_list, mn));
  TRY(dev_alloc(c, c->dirty_block_counts, max_blocks));
  TRY(dev_alloc(c, c->dirty_block_offsets, max_blocks));
  TRY(dev_alloc(c, c->dirty_total, 1));
  DirtyArgs a{};
  a.flags = c->flags();
  a.block_counts = c->dirty_block_counts.p;
  a.block_offsets = c->dirty_block_offsets.p;
  a.list = c->dirty_list.p;
  a.total = c->dirty_total.p;
  a.n = c->n_nodes;
  a.blocks = (uint32_t)ceil_div(c->n_nodes, DIRTY_NODES);
  if (a.blocks) hipLaunchKernelGGL(k_dirty_count, dim3(a.blocks), dim3(DIRTY_BLOCK), 0, c->stream, a);
  hipLaunchKernelGGL(k_dirty_scan, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, a);
  if (a.blocks) hipLaunchKernelGGL(k_dirty_write, dim3(a.blocks), dim3(DIRTY_BLOCK), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

static int launch_candidates(vkrhip_ctx *c) {
  TRY(dev_alloc(c, c->cand_rows, std::max<uint64_t>(c->cfg.max_candidates, 1) * 9));
  if (!c->n_cand) return VKRHIP_OK;
  StageTimer timer(c, VKRHIP_TIMER_CANDIDATES);
  CandidateEmitArgs a{};
  a.world = c->world.p;
  a.node = c->node_identity ? nullptr : c->node.p;
  a.ident = c->ident.p;
  a.sphere = c->sphere.p;
  a.cflags = c->cflags.p;
  a.rows = c->cand_rows.p;
  a.n = c->n_cand;
  // 260 bytes move per row: streaming loads and stores once that exceeds the last-level cache
  const bool stream = c->streaming_mode >= 0 ? c->streaming_mode != 0 : c->n_cand * 260u > kStreamingFootprint;
  const auto kernel = stream ? k_emit_candidates<true> : k_emit_candidates<false>;
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(c->n_cand, EMIT_ROWS)), dim3(EMIT_ROWS), 0, c->stream, a);
  if (c->n_transmission) {
    TRY(dev_alloc(c, c->transmission_rows, c->n_transmission * 9));
    a.rows = c->transmission_rows.p;
    a.n = c->n_transmission;
    a.list = c->transmission_list.p;
    hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(c->n_transmission, EMIT_ROWS)), dim3(EMIT_ROWS), 0,
                       c->stream, a);
  }
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

extern "C" {

int vkrhip_frame(vkrhip_ctx *c, uint32_t stages) {
  if (!c) return VKRHIP_ERR_ARG;
  if ((stages & (VKRHIP_STAGE_CULL | VKRHIP_STAGE_TRANSPARENT)) && c->views == 0)
    return fail(c, VKRHIP_ERR_STATE, "vkrhip_set_views has not been called");
  if ((stages & VKRHIP_STAGE_TRANSPARENT) && !(stages & VKRHIP_STAGE_CULL))
    return fail(c, VKRHIP_ERR_STATE, "STAGE_TRANSPARENT needs STAGE_CULL in the same frame");
  if ((stages & VKRHIP_STAGE_WORLD_BOUNDS) && !(stages & VKRHIP_STAGE_CULL))
    return fail(c, VKRHIP_ERR_STATE, "STAGE_WORLD_BOUNDS needs STAGE_CULL in the same frame");
  if ((stages & VKRHIP_STAGE_CASTER_BOUNDS) && !(stages & VKRHIP_STAGE_CULL))
    return fail(c, VKRHIP_ERR_STATE, "STAGE_CASTER_BOUNDS needs STAGE_CULL in the same frame");
  if ((stages & (VKRHIP_STAGE_PACK | VKRHIP_STAGE_ENCODE)) && !c->published)
    return fail(c, VKRHIP_ERR_STATE, "STAGE_PACK / STAGE_ENCODE need vkrhip_set_publication");
  if ((stages & VKRHIP_STAGE_ENCODE) &&
      (stages & (VKRHIP_STAGE_CULL | VKRHIP_STAGE_PACK)) != (VKRHIP_STAGE_CULL | VKRHIP_STAGE_PACK))
    return fail(c, VKRHIP_ERR_STATE, "STAGE_ENCODE needs STAGE_CULL and STAGE_PACK in the same frame");
  // every candidate row must name a node of the hierarchy as it is now (the
  // hierarchy may have shrunk since the rows were bound)
  if (stages & (VKRHIP_STAGE_CULL | VKRHIP_STAGE_CANDIDATES | VKRHIP_STAGE_PACK)) {
    const uint64_t nodes = c->level_offsets.empty() ? c->cfg.max_nodes : c->n_nodes;
    const uint64_t named = c->node_identity ? c->n_cand : c->node_limit;
    if (named > nodes)
      return fail(c, VKRHIP_ERR_STATE,
                  "candidate rows name node %llu but the scene has %llu nodes: bind the rows again",
                  (unsigned long long)(named - 1u), (unsigned long long)nodes);
  }
  TRY(bind_device(c, false));
  c->frame_stages = stages;
  const bool cull = (stages & VKRHIP_STAGE_CULL) != 0;
  const bool bounds_out = (stages & VKRHIP_STAGE_WORLD_BOUNDS) != 0;
  if (bounds_out) TRY(dev_alloc(c, c->world_bounds, std::max<uint64_t>(c->cfg.max_candidates, 1)));
  {
    StageTimer frame_timer(c, VKRHIP_TIMER_FRAME);
    // this frame's counter block is normally already zero (cleared by the
    // previous frame's scan kernel)
    c->cur_block = (c->cur_block + 1u) % vkrhip_ctx::kBlocks;
    // a scan on the second stream may be the one that zeroes this block (two frames ago)
    TRY(wait_side_event(c, c->block_event[c->cur_block]));
    c->block_event[c->cur_block] = -1;
    if (!c->block_clean[c->cur_block]) {
      // the block was last used three frames ago, and that frame's sort and emission
      // (side stream) read its count, histograms and tickets: they must be done
      // before it is zeroed (a frame that culls joins anyway, behind its classify)
      TRY(join_side(c));
      HIP_TRY(c, hipMemsetAsync(c->counter_block(c->cur_block), 0, vkrhip_ctx::kBlockBytes, c->stream));
    }
    c->block_clean[c->cur_block] = false;
    // the fused kernel serves scenes that draw node i with candidate i
    // (the occlusion test lives in the stand-alone classify kernel only)
    const bool fused = cull && c->hzb_mips == 0 && (stages & VKRHIP_STAGE_TRANSFORM) && c->node_identity &&
                       c->n_cand > 0 && c->n_cand <= c->n_nodes && c->level_offsets.size() > 1;
    c->frame_tail = tail_on_second_stream(c, stages, fused);
    // A frame that does not leave its tail to the worker thread finds the second
    // stream as a single thread would have left it.
    constexpr uint32_t kTailTimers = (1u << VKRHIP_TIMER_SCAN) | (1u << VKRHIP_TIMER_SCATTER) |
                                     (1u << VKRHIP_TIMER_SORT) | (1u << VKRHIP_TIMER_EMIT);
    // (with the exchange configured vkrhip_exchange_run follows every frame and puts its
    // pack behind the tail: it would only wait for the worker, 78 against 58 us per
    // frame of 1M nodes)
    const bool to_worker =
        c->frame_tail && c->worker_mode != 0 && !(c->profiling & kTailTimers) && c->x_world == 0;
    if (!to_worker) TRY(worker_flush(c));
    c->gathered_by_gate = to_worker && c->side_gate != 0;
    if (c->frame_tail) {
      TRY(ensure_tail_buffers(c));
      // The frame's last level launch records the event that the tail of two frames
      // ago waited for: that wait must have been enqueued (a worker that lags two jobs
      // would make that tail wait for THIS frame, which waits for that tail's signal).
      if (c->worker) {
        const uint64_t pushed = c->worker->pushed.load(std::memory_order_relaxed);
        if (pushed > 1u) TRY(worker_wait(c, pushed - 1u));
      }
      c->gathered_turn ^= 1u;
      c->ev_gathered = c->ev_gathered2[c->gathered_turn];
    }
    if (cull) {
      // frames alternate between the copies of the segment tables; the batch of
      // side-stream work that read this frame's copy must be over before the
      // classify blocks rewrite it
      if (c->second_copy) c->par ^= 1u;
      TRY(wait_side_event(c, c->copy_event[c->par]));
      c->copy_event[c->par] = -1;
    }
    if (fused) {
      TRY(launch_fused(c, bounds_out));
    } else {
      if (stages & VKRHIP_STAGE_TRANSFORM) TRY(launch_transform(c));
      if (cull) TRY(launch_classify(c, bounds_out));
    }
    if (stages & VKRHIP_STAGE_DIRTY_LIST) TRY(launch_dirty_list(c));
    if (c->frame_tail) {
      // the tail waits for the classify on the second stream (ev_gathered is the last
      // classify launch's own completion signal); the frame's stream goes on
      if (to_worker) {
        TRY(worker_start(c));
        TailJob *job = worker_begin(c);
        c->building = job;
        if (c->gathered_by_gate) {
          ++c->gathered_serial;
          hipLaunchKernelGGL(k_side_signal, dim3(1), dim3(64), 0, c->stream, c->side_flag.p + 3, c->gathered_serial);
          HIP_TRY(c, hipGetLastError());
          launch_on(c, c->tstream, k_side_gate, dim3(1), dim3(64), 0, (const uint64_t *)c->side_flag.p + 3,
                    c->gathered_serial, c->side_flag.p + 2);
        } else {
          job->wait = c->ev_gathered;
        }
        int rc = launch_compaction(c, false, c->tstream);
        if (rc == VKRHIP_OK) rc = launch_transparent(c, true);
        c->building = nullptr;
        if (rc == VKRHIP_OK && job->overflow) rc = fail(c, VKRHIP_ERR_STATE, "too many launches in one tail job");
        if (rc != VKRHIP_OK) {  // the job is dropped: nothing may wait for its signal
          const uint64_t dropped = c->worker->pushed.load(std::memory_order_relaxed) + 1u;
          for (int k = 0; k < 2; ++k)
            if (c->side_job[k] == dropped) {
              c->side_job[k] = 0;
              c->side_expected[k] = 0;
              c->side_pending[k] = false;
            }
          return rc;
        }
        worker_push(c);
      } else {
        HIP_TRY(c, hipStreamWaitEvent(c->tstream, c->ev_gathered, 0));
        TRY(launch_compaction(c, false, c->tstream));
        TRY(launch_transparent(c, true));
      }
    } else {
      // the previous frame's work on the side stream still owns the previous counter
      // block, the key buffers, the lists and the set-aside rows: the scan zeroes
      // that block and the gather refills the buffers
      if (cull) TRY(join_side(c));
      const bool overlap = (stages & VKRHIP_STAGE_TRANSPARENT) && sort_on_second_stream(c);
      if (cull) TRY(launch_compaction(c, overlap, c->stream));
      if (stages & VKRHIP_STAGE_TRANSPARENT) TRY(launch_transparent(c, overlap));
    }
    if (stages & VKRHIP_STAGE_CANDIDATES) {
      TRY(launch_candidates(c));
      // the frame's totals, for vkrhip_fill_world_payload: summed on the device and
      // stored to pinned host memory behind everything else of the frame
      if (++c->publish_serial == 0u) c->publish_serial = 1u;
      hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(64), 0, c->stream,
                         (const uint32_t *)c->counters(), c->host_counts, c->publish_serial);
      HIP_TRY(c, hipGetLastError());
    }
    if (stages & VKRHIP_STAGE_PACK) TRY(launch_pack(c));
    if (stages & VKRHIP_STAGE_ENCODE) TRY(launch_encode(c));
  }
  c->frame_ran = true;
  c->last_stages = stages;
  return VKRHIP_OK;
}

int vkrhip_get_frame_info(vkrhip_ctx *c, vkrhip_frame_info *out) try {
  if (!c || !out) return VKRHIP_ERR_ARG;
  if (!c->frame_ran) return fail(c, VKRHIP_ERR_STATE, "no frame has run");
  TRY(bind_device(c));
  TRY(join_side(c));  // (the sort's stall flag)
  static_assert(TF_STAT_SHARDS == STAT_SHARDS && TF_CNT_STRIDE == CNT_COUNT, "shard layout");
  std::vector<uint32_t> shards(CNT_COUNT * STAT_SHARDS);
  std::vector<uint64_t> tie_shards((size_t)VKRHIP_MAX_VIEWS * TIE_SHARDS);
  std::memset(out, 0, sizeof(*out));
  HIP_TRY(c, hipMemcpyAsync(shards.data(), c->counters(), shards.size() * 4, hipMemcpyDeviceToHost,
                            c->stream));
  HIP_TRY(c, hipMemcpyAsync(out->views, c->states.p, sizeof(CompactionState) * VKRHIP_MAX_VIEWS,
                            hipMemcpyDeviceToHost, c->stream));
  if (c->cfg.tie_epsilon > 0.f)
    HIP_TRY(c, hipMemcpyAsync(tie_shards.data(), c->ties(), tie_shards.size() * 8,
                              hipMemcpyDeviceToHost, c->stream));
  uint32_t sort_stalled[3] = {0, 0, 0};
  // (the block the last radix sort ran in: a later frame without a sort has moved on)
  HIP_TRY(c, hipMemcpyAsync(sort_stalled,
                            c->counter_block(c->sort_block) + vkrhip_ctx::kCountersBytes +
                                vkrhip_ctx::kTiesBytes + offsetof(SortState, stalled),
                            sizeof(sort_stalled), hipMemcpyDeviceToHost, c->stream));
  uint64_t gate_stalled = 0;
  HIP_TRY(c, hipMemcpyAsync(&gate_stalled, c->side_flag.p + 2, sizeof(gate_stalled), hipMemcpyDeviceToHost,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (gate_stalled)
    return fail(c, VKRHIP_ERR_STATE,
                "a level launch gave up waiting for batch %llu of second-stream work and ran ahead of it",
                (unsigned long long)gate_stalled);
  if (sort_stalled[0])
    return fail(c, VKRHIP_ERR_STATE,
                "the draw-key sort gave up waiting for another tile's prefix (pass %u, digit %u: tile %u "
                "waited for tile %u)",
                (sort_stalled[0] >> 8) & 0xffu, sort_stalled[0] >> 16, sort_stalled[1], sort_stalled[2]);
  uint32_t cnt[CNT_COUNT] = {};
  for (uint32_t sh = 0; sh < STAT_SHARDS; ++sh)
    for (int k = 0; k < CNT_COUNT; ++k) cnt[k] += shards[sh * CNT_COUNT + k];
  for (uint32_t sh = 0; sh < TIE_SHARDS; ++sh)
    for (uint32_t v = 0; v < VKRHIP_MAX_VIEWS; ++v)
      out->near_plane_ties[v] += tie_shards[(size_t)sh * VKRHIP_MAX_VIEWS + v];
  const bool culled = (c->last_stages & VKRHIP_STAGE_CULL) != 0;
  if (c->views) out->views[0].occlusion_culled_count = cnt[CNT_OCCLUDED];
  out->stats.objects_tested = culled ? (uint32_t)c->n_cand : 0u;
  out->stats.objects_culled_camera = cnt[CNT_CULLED_CAMERA];
  out->stats.objects_without_bounds = cnt[CNT_WITHOUT_BOUNDS];
  out->gpu_candidate_count = (uint32_t)c->n_cand;
  out->gpu_camera_opaque_candidate_count = cnt[CNT_CAMERA_OPAQUE];
  out->transmission_gpu_candidate_count = cnt[CNT_TRANSMISSION];
  out->transparent_draw_count = cnt[CNT_TRANSPARENT];
  out->transparent_overflow = cnt[CNT_TRANSPARENT_TOTAL] - cnt[CNT_TRANSPARENT];
  out->nodes_updated = cnt[CNT_NODES_UPDATED];
  out->view_count = c->views;
  if (c->published) {
    out->packed_count = c->pack_counts_host[0];
    out->unpublished_geometry_count = c->pack_counts_host[1];
    out->unpublished_material_count = c->pack_counts_host[2];
    out->invalid_submesh_count = c->pack_counts_host[3];
  }
  for (uint32_t v = c->views; v < VKRHIP_MAX_VIEWS; ++v) {
    std::memset(&out->views[v], 0, sizeof(out->views[v]));
    out->near_plane_ties[v] = 0;
  }
  return VKRHIP_OK;
} catch (...) {
  return host_exception(c);
}

int vkrhip_get_device_outputs(vkrhip_ctx *c, vkrhip_device_outputs *out) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  // what the caller enqueues on the context's stream from here on is ordered behind
  // the draws of the last frame, which a second stream wrote
  if (bind_device(c) == VKRHIP_OK) (void)join_side(c);
  std::memset(out, 0, sizeof(*out));
  out->world = reinterpret_cast<const float *>(c->world.p);
  out->node_flags = c->flags();
  out->visible = c->visible.p;
  out->visible_stride = c->visible_stride;
  out->transparent_visible = c->tvis.p;
  out->transparent_draws = c->draws.p;
  out->transparent_instances = c->instances.p;
  out->transparent_keys = c->out_keys.p;
  out->transparent_source = c->out_source.p;
  out->gpu_candidates = c->cand_rows.p;
  out->transmission_gpu_candidates = c->n_transmission ? c->transmission_rows.p : nullptr;
  out->world_bounds = reinterpret_cast<const float *>(c->world_bounds.p);
  out->device_counts = c->counters();
  out->packed_rows = c->packed_rows.p;
  out->packed_instances = c->packed_instances.p;
  out->visible_rows = c->visible_rows.p;
  out->commands = c->commands.p;
  out->dirty_nodes = c->dirty_list.p;
  out->dirty_node_count = c->dirty_total.p;
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// the reference's world-pass payload
// ---------------------------------------------------------------------------
static int ensure_pinned(vkrhip_ctx *c, void **p, size_t *have, size_t need) {
  if (*have >= need && *p) return VKRHIP_OK;
  if (*p) HIP_TRY(c, hipHostFree(*p));
  *p = nullptr;
  *have = 0;
  need = std::max<size_t>(need + need / 4u, 4096);
  HIP_TRY(c, hipHostMalloc(p, need, hipHostMallocDefault));
  *have = need;
  return VKRHIP_OK;
}

int vkrhip_fill_world_payload(vkrhip_ctx *c, vkrhip_world_payload *out, vkrhip_visibility_stats *stats) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  static_assert(sizeof(vkrhip_world_payload) == 88, "VkrWorldPassPayload is 88 bytes");
  TRY(bind_device(c));
  TRY(join_side(c));
  std::memset(out, 0, sizeof(*out));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (c->n_cand > VKRHIP_DRAW_CANDIDATE_CAPACITY) {  // application.h:703-711
    out->gpu_candidate_count = VKRHIP_DRAW_CANDIDATE_CAPACITY + 1u;
    if (stats) stats->objects_tested = VKRHIP_DRAW_CANDIDATE_CAPACITY + 1u;
    return VKRHIP_OK;
  }
  const uint32_t need = VKRHIP_STAGE_CULL | VKRHIP_STAGE_TRANSPARENT | VKRHIP_STAGE_CANDIDATES;
  if (!c->frame_ran || (c->last_stages & need) != need)
    return fail(c, VKRHIP_ERR_STATE,
                "the last frame did not run STAGE_CULL | STAGE_TRANSPARENT | STAGE_CANDIDATES");
  TRY(bind_device(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the frame, and with it the published totals
  const volatile uint32_t *hc = c->host_counts;
  if (hc[0] != c->publish_serial)
    return fail(c, VKRHIP_ERR_STATE, "the frame's totals were not published (serial %u, expected %u)",
                hc[0], c->publish_serial);
  const uint32_t n = (uint32_t)c->n_cand;
  const uint32_t nt = (uint32_t)c->n_transmission;
  const uint32_t t = hc[1 + CNT_TRANSPARENT];
  TRY(ensure_pinned(c, &c->pay_candidates, &c->pay_candidates_bytes, (size_t)n * 144u));
  TRY(ensure_pinned(c, &c->pay_transmission, &c->pay_transmission_bytes, (size_t)nt * 144u));
  TRY(ensure_pinned(c, &c->pay_draws, &c->pay_draws_bytes, (size_t)t * 48u));
  TRY(ensure_pinned(c, &c->pay_instances, &c->pay_instances_bytes, (size_t)t * 80u));
  if (n) HIP_TRY(c, hipMemcpyAsync(c->pay_candidates, c->cand_rows.p, (size_t)n * 144u, hipMemcpyDeviceToHost, c->stream));
  if (nt) HIP_TRY(c, hipMemcpyAsync(c->pay_transmission, c->transmission_rows.p, (size_t)nt * 144u, hipMemcpyDeviceToHost, c->stream));
  if (t) {
    HIP_TRY(c, hipMemcpyAsync(c->pay_draws, c->draws.p, (size_t)t * 48u, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->pay_instances, c->instances.p, (size_t)t * 80u, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out->gpu_candidates = n ? c->pay_candidates : nullptr;
  out->gpu_candidate_count = n;
  out->gpu_camera_opaque_candidate_count = hc[1 + CNT_CAMERA_OPAQUE];
  out->gpu_shadow_candidate_count = n;  // application.h:950
  out->transmission_gpu_candidates = nt ? c->pay_transmission : nullptr;
  out->transmission_gpu_candidate_count = hc[1 + CNT_TRANSMISSION];
  out->transparent_draws = t ? c->pay_draws : nullptr;
  out->transparent_draw_count = t;
  out->instances = t ? c->pay_instances : nullptr;
  out->instance_count = t;
  if (stats) {
    stats->objects_tested = n;
    stats->objects_culled_camera = hc[1 + CNT_CULLED_CAMERA];
    stats->objects_without_bounds = hc[1 + CNT_WITHOUT_BOUNDS];
  }
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// read-back helpers
// ---------------------------------------------------------------------------
static int read_back(vkrhip_ctx *c, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  TRY(join_side(c));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VKRHIP_OK;
}

int vkrhip_host_alloc(vkrhip_ctx *c, uint64_t bytes, void **out_pointer) {
  if (!c || !out_pointer) return c ? fail(c, VKRHIP_ERR_ARG, "host_alloc: null out_pointer") : VKRHIP_ERR_ARG;
  *out_pointer = nullptr;
  if (bytes == 0) return fail(c, VKRHIP_ERR_ARG, "host_alloc: zero bytes");
  TRY(bind_device(c));
  HIP_TRY(c, hipHostMalloc(out_pointer, bytes, hipHostMallocDefault));
  return VKRHIP_OK;
}

int vkrhip_host_free(vkrhip_ctx *c, void *pointer) {
  if (!c) return VKRHIP_ERR_ARG;
  if (!pointer) return VKRHIP_OK;
  TRY(bind_device(c));
  HIP_TRY(c, hipHostFree(pointer));
  return VKRHIP_OK;
}

int vkrhip_read_world(vkrhip_ctx *c, uint64_t first, uint64_t count, float *out16) {
  if (!c || !out16) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "read_world"));
  return read_back(c, out16, c->world.p + first * 4, count * 64);
}

int vkrhip_read_locals(vkrhip_ctx *c, uint64_t first, uint64_t count, float *out16) {
  if (!c || !out16) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "read_locals"));
  return read_back(c, out16, c->local.p + first * 4, count * 64);
}

int vkrhip_read_node_flags(vkrhip_ctx *c, uint64_t first, uint64_t count, uint8_t *out) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, c->cfg.max_nodes, "read_node_flags"));
  return read_back(c, out, c->flags() + first, count);
}

int vkrhip_read_visible(vkrhip_ctx *c, uint32_t view, uint64_t count, uint32_t *out) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  if (view >= c->cfg.max_views) return fail(c, VKRHIP_ERR_ARG, "view %u out of range", view);
  TRY(check_range(c, 0, count, c->visible_stride, "read_visible"));
  return read_back(c, out, c->visible.p + (uint64_t)view * c->visible_stride, count * 4);
}

int vkrhip_read_transparent_visible(vkrhip_ctx *c, uint64_t first, uint64_t count, uint8_t *out) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, std::max<uint64_t>(c->cfg.max_candidates, 1), "read_transparent_visible"));
  return read_back(c, out, c->tvis.p + first, count);
}

int vkrhip_read_transparent(vkrhip_ctx *c, uint64_t count, void *draws48, void *instances80,
                            uint64_t *keys, uint32_t *source) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(check_range(c, 0, count, c->tcap, "read_transparent"));
  if (draws48) TRY(read_back(c, draws48, c->draws.p, count * 48));
  if (instances80) TRY(read_back(c, instances80, c->instances.p, count * 80));
  if (keys) TRY(read_back(c, keys, c->out_keys.p, count * 8));
  if (source) TRY(read_back(c, source, c->out_source.p, count * 4));
  return VKRHIP_OK;
}

int vkrhip_read_candidates(vkrhip_ctx *c, uint64_t first, uint64_t count, void *rows144) {
  if (!c || !rows144) return VKRHIP_ERR_ARG;
  if (!c->cand_rows.p) return fail(c, VKRHIP_ERR_STATE, "STAGE_CANDIDATES has not run");
  TRY(check_range(c, first, count, std::max<uint64_t>(c->cfg.max_candidates, 1), "read_candidates"));
  return read_back(c, rows144, c->cand_rows.p + first * 9, count * 144);
}

int vkrhip_read_transmission_candidates(vkrhip_ctx *c, uint64_t first, uint64_t count,
                                        void *rows144) {
  if (!c || !rows144) return VKRHIP_ERR_ARG;
  if (!c->cand_rows.p) return fail(c, VKRHIP_ERR_STATE, "STAGE_CANDIDATES has not run");
  TRY(check_range(c, first, count, c->n_transmission, "read_transmission_candidates"));
  if (count == 0) return VKRHIP_OK;
  return read_back(c, rows144, c->transmission_rows.p + first * 9, count * 144);
}

int vkrhip_read_packed(vkrhip_ctx *c, uint64_t count, void *rows48, void *instances80) {
  if (!c) return VKRHIP_ERR_ARG;
  if (!c->packed_rows.p) return fail(c, VKRHIP_ERR_STATE, "STAGE_PACK has not run");
  TRY(check_range(c, 0, count, std::max<uint64_t>(c->cfg.max_candidates, 1), "read_packed"));
  if (rows48) TRY(read_back(c, rows48, c->packed_rows.p, count * 48));
  if (instances80) TRY(read_back(c, instances80, c->packed_instances.p, count * 80));
  return VKRHIP_OK;
}

int vkrhip_read_visible_rows(vkrhip_ctx *c, uint32_t view, uint64_t count, void *rows32) {
  if (!c || !rows32) return VKRHIP_ERR_ARG;
  if (!c->visible_rows.p) return fail(c, VKRHIP_ERR_STATE, "STAGE_ENCODE has not run");
  if (view >= c->cfg.max_views) return fail(c, VKRHIP_ERR_ARG, "view %u out of range", view);
  TRY(check_range(c, 0, count, c->visible_stride, "read_visible_rows"));
  return read_back(c, rows32, c->visible_rows.p + (size_t)view * c->visible_stride * 2, count * 32);
}

int vkrhip_read_commands(vkrhip_ctx *c, uint32_t view, uint32_t bucket, uint64_t count,
                         void *commands20) {
  if (!c || !commands20) return VKRHIP_ERR_ARG;
  if (!c->commands.p) return fail(c, VKRHIP_ERR_STATE, "no indirect commands (STAGE_ENCODE with a partition capacity)");
  if (view >= c->cfg.max_views || bucket >= 4u) return fail(c, VKRHIP_ERR_ARG, "view/bucket out of range");
  TRY(check_range(c, 0, count, c->cfg.partition_capacity, "read_commands"));
  return read_back(c, commands20,
                   c->commands.p + ((size_t)view * 4u + bucket) * c->cfg.partition_capacity * 5u,
                   count * 20);
}

int vkrhip_read_dirty_list(vkrhip_ctx *c, uint64_t capacity, uint32_t *out_nodes, uint64_t *out_count) {
  if (!c || !out_count || (capacity && !out_nodes)) return VKRHIP_ERR_ARG;
  *out_count = 0;
  if (!c->dirty_list.p || !(c->last_stages & VKRHIP_STAGE_DIRTY_LIST))
    return fail(c, VKRHIP_ERR_STATE, "the last frame did not run STAGE_DIRTY_LIST");
  uint32_t total = 0;
  TRY(read_back(c, &total, c->dirty_total.p, sizeof(total)));
  *out_count = total;
  return read_back(c, out_nodes, c->dirty_list.p, std::min<uint64_t>(total, capacity) * sizeof(uint32_t));
}

int vkrhip_read_world_bounds(vkrhip_ctx *c, uint64_t first, uint64_t count, float *out4) {
  if (!c || !out4) return VKRHIP_ERR_ARG;
  if (!c->world_bounds.p) return fail(c, VKRHIP_ERR_STATE, "STAGE_WORLD_BOUNDS has not run");
  TRY(check_range(c, first, count, std::max<uint64_t>(c->cfg.max_candidates, 1), "read_world_bounds"));
  return read_back(c, out4, c->world_bounds.p + first, count * 16);
}

int vkrhip_get_caster_bounds(vkrhip_ctx *c, float *out_min3, float *out_max3) {
  if (!c || !out_min3 || !out_max3) return VKRHIP_ERR_ARG;
  const uint32_t need = VKRHIP_STAGE_CULL | VKRHIP_STAGE_CASTER_BOUNDS;
  if (!c->frame_ran || (c->last_stages & need) != need)
    return fail(c, VKRHIP_ERR_STATE, "the last frame did not run STAGE_CASTER_BOUNDS");
  TRY(bind_device(c));
  uint32_t keys[6] = {};
  TRY(read_back(c, keys, c->caster_box(), sizeof(keys)));
  int found = 1;
  for (int k = 0; k < 6; ++k) found = found && keys[k] != 0u;
  for (int k = 0; k < 3; ++k) {
    // no caster: an empty box (min above max), which the cascade fit rejects
    out_min3[k] = found ? order_key_value(~keys[k]) : 3.402823466e+38f;
    out_max3[k] = found ? order_key_value(keys[3 + k]) : -3.402823466e+38f;
  }
  return VKRHIP_OK;
}

int vkrhip_read_candidate_prep(vkrhip_ctx *c, uint64_t first, uint64_t count, float *sphere4,
                               uint32_t *flags_bucket) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(check_range(c, first, count, std::max<uint64_t>(c->cfg.max_candidates, 1), "read_candidate_prep"));
  if (sphere4) TRY(read_back(c, sphere4, c->sphere.p + first, count * 16));
  if (flags_bucket) TRY(read_back(c, flags_bucket, c->cflags.p + first, count * 4));
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------
int vkrhip_set_profiling(vkrhip_ctx *c, uint32_t timer_mask) {
  if (!c) return VKRHIP_ERR_ARG;
  c->profiling = timer_mask;
  return VKRHIP_OK;
}

int vkrhip_get_timers(vkrhip_ctx *c, double *ms_out, uint64_t *samples_out, int reset) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  TRY(drain_timers(c));
  for (int i = 0; i < VKRHIP_TIMER_COUNT; ++i) {
    if (ms_out) ms_out[i] = c->timer_ms[i];
    if (samples_out) samples_out[i] = c->timer_n[i];
    if (reset) {
      c->timer_ms[i] = 0.0;
      c->timer_n[i] = 0;
    }
  }
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// multi-GPU draw-key merge
// ---------------------------------------------------------------------------
}  // extern "C"

namespace vkr {

// Rank-based G-way merge of descending runs with unique keys: the output slot of
// an element is its index in its own run plus, for every other run, the number
// of keys greater than it (found by binary search).
__global__ __launch_bounds__(256) void k_merge_runs(const uint64_t *const *runs,
                                                    const uint64_t *lengths, uint32_t g,
                                                    uint64_t total, uint64_t *out) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  uint32_t q = 0;
  uint64_t j = gid;
  while (q < g && j >= lengths[q]) {
    j -= lengths[q];
    ++q;
  }
  const uint64_t key = runs[q][j];
  uint64_t pos = j;
  for (uint32_t r = 0; r < g; ++r) {
    if (r == q) continue;
    const uint64_t *run = runs[r];
    uint64_t lo = 0, hi = lengths[r];
    while (lo < hi) {  // first index whose key is <= `key` in a descending run
      const uint64_t mid = (lo + hi) >> 1;
      if (run[mid] > key)
        lo = mid + 1;
      else
        hi = mid;
    }
    pos += lo;
  }
  out[pos] = key;
}

// Segmented copy: descriptor d = {src offset, dst offset, count} in elements.
__global__ __launch_bounds__(256) void k_concat_ranges(const uint32_t *src, uint32_t *dst,
                                                       const uint64_t *desc) {
  const uint64_t s = desc[blockIdx.x * 3u + 0];
  const uint64_t d = desc[blockIdx.x * 3u + 1];
  const uint64_t n = desc[blockIdx.x * 3u + 2];
  for (uint64_t k = (uint64_t)blockIdx.y * 256u + threadIdx.x; k < n; k += (uint64_t)gridDim.y * 256u)
    dst[d + k] = src[s + k];
}

}  // namespace vkr

extern "C" {

int vkrhip_concat_ranges(vkrhip_ctx *c, const uint32_t *src_dev, uint32_t *dst_dev,
                         const uint64_t *desc, uint32_t n_desc) {
  if (!c || !src_dev || !dst_dev || (!desc && n_desc)) return VKRHIP_ERR_ARG;
  if (n_desc == 0) return VKRHIP_OK;
  TRY(bind_device(c));
  TRY(dev_alloc(c, c->concat_desc, (size_t)n_desc * 3));
  HIP_TRY(c, hipMemcpyAsync(c->concat_desc.p, desc, (size_t)n_desc * 24, hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // desc is caller memory
  hipLaunchKernelGGL(k_concat_ranges, dim3(n_desc, 16), dim3(256), 0, c->stream, src_dev, dst_dev,
                     c->concat_desc.p);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

int vkrhip_merge_runs(vkrhip_ctx *c, uint32_t g, const uint64_t *const *runs_keys,
                      const uint64_t *lengths, uint64_t *out_keys_dev) {
  if (!c || !runs_keys || !lengths || !out_keys_dev) return VKRHIP_ERR_ARG;
  if (g == 0 || g > 64) return fail(c, VKRHIP_ERR_ARG, "run count %u outside 1..64", g);
  TRY(bind_device(c));
  uint64_t total = 0;
  for (uint32_t r = 0; r < g; ++r) total += lengths[r];
  if (total == 0) return VKRHIP_OK;
  TRY(dev_alloc(c, c->merge_ptrs, 64));
  TRY(dev_alloc(c, c->merge_lens, 64));
  HIP_TRY(c, hipMemcpyAsync(c->merge_ptrs.p, runs_keys, g * sizeof(void *), hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->merge_lens.p, lengths, g * sizeof(uint64_t), hipMemcpyHostToDevice,
                            c->stream));
  // the two small arrays above come from pageable host memory: make sure the
  // copies have been staged before the caller's arrays go out of scope
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  hipLaunchKernelGGL(k_merge_runs, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, c->stream,
                     c->merge_ptrs.p, c->merge_lens.p, g, total, out_keys_dev);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

int vkrhip_set_hzb(vkrhip_ctx *c, const float *mips, uint32_t width, uint32_t height,
                   uint32_t mip_count, const float *view_projection, float depth_epsilon) {
  if (!c) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!mips || mip_count == 0 || width == 0 || height == 0) {  // deferred.slang:243-245: test off
    c->hzb_mips = 0;
    return VKRHIP_OK;
  }
  if (!view_projection) return fail(c, VKRHIP_ERR_ARG, "null view_projection");
  if (mip_count > 32) return fail(c, VKRHIP_ERR_ARG, "mip_count %u > 32", mip_count);
  uint32_t offsets[32];
  uint64_t texels = 0;
  for (uint32_t m = 0; m < mip_count; ++m) {
    offsets[m] = (uint32_t)texels;
    texels += (uint64_t)std::max(width >> m, 1u) * std::max(height >> m, 1u);
    if (texels > 0xffffffffull) return fail(c, VKRHIP_ERR_CAPACITY, "depth pyramid too large");
  }
  TRY(dev_alloc(c, c->hzb, texels));
  TRY(dev_alloc(c, c->hzb_offset, 32));
  TRY(dev_alloc(c, c->hzb_view_projection, 4));
  HIP_TRY(c, hipMemcpy(c->hzb.p, mips, texels * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->hzb_offset.p, offsets, mip_count * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->hzb_view_projection.p, view_projection, 64, hipMemcpyHostToDevice));
  c->hzb_width = width;
  c->hzb_height = height;
  c->hzb_mips = mip_count;
  c->hzb_epsilon = depth_epsilon;
  return VKRHIP_OK;
}

int vkrhip_build_hzb(vkrhip_ctx *c, const float *depth, uint32_t source_width,
                     uint32_t source_height, uint32_t width, uint32_t height, uint32_t mip_count,
                     const float *view_projection, float depth_epsilon) {
  if (!c || !depth || !view_projection) return VKRHIP_ERR_ARG;
  if (!source_width || !source_height || !width || !height || !mip_count || mip_count > 32)
    return fail(c, VKRHIP_ERR_ARG, "empty depth image, empty pyramid or more than 32 levels");
  TRY(bind_device(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint32_t offsets[32];
  uint64_t texels = 0;
  for (uint32_t m = 0; m < mip_count; ++m) {
    offsets[m] = (uint32_t)texels;
    texels += (uint64_t)std::max(width >> m, 1u) * std::max(height >> m, 1u);
    if (texels > 0xffffffffull) return fail(c, VKRHIP_ERR_CAPACITY, "depth pyramid too large");
  }
  const uint64_t source_texels = (uint64_t)source_width * source_height;
  DevBuf<float> d_depth;
  TRY(dev_alloc(c, d_depth, source_texels));
  int rc = dev_alloc(c, c->hzb, texels);
  if (rc == VKRHIP_OK) rc = dev_alloc(c, c->hzb_offset, 32);
  if (rc == VKRHIP_OK) rc = dev_alloc(c, c->hzb_view_projection, 4);
  hipError_t e = hipSuccess;
  if (rc == VKRHIP_OK) {
    e = hipMemcpyAsync(d_depth.p, depth, source_texels * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(c->hzb_offset.p, offsets, mip_count * sizeof(uint32_t), hipMemcpyHostToDevice,
                         c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(c->hzb_view_projection.p, view_projection, 64, hipMemcpyHostToDevice, c->stream);
    for (uint32_t m = 0; m < mip_count && e == hipSuccess; ++m) {
      HzbBuildArgs a{};
      a.copy = m == 0 ? 1u : 0u;
      a.source = m == 0 ? d_depth.p : c->hzb.p + offsets[m - 1];
      a.source_width = m == 0 ? source_width : std::max(width >> (m - 1), 1u);
      a.source_height = m == 0 ? source_height : std::max(height >> (m - 1), 1u);
      a.width = std::max(width >> m, 1u);
      a.height = std::max(height >> m, 1u);
      a.destination = c->hzb.p + offsets[m];
      hipLaunchKernelGGL(k_hzb_build, dim3((unsigned)ceil_div(a.width, 16), (unsigned)ceil_div(a.height, 16)),
                         dim3(256), 0, c->stream, a);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // `offsets` and the caller's arrays
  }
  dev_free(d_depth);
  if (rc != VKRHIP_OK) return rc;
  HIP_TRY(c, e);
  c->hzb_width = width;
  c->hzb_height = height;
  c->hzb_mips = mip_count;
  c->hzb_epsilon = depth_epsilon;
  return VKRHIP_OK;
}

int vkrhip_read_hzb(vkrhip_ctx *c, uint32_t mip, float *out) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  if (mip >= c->hzb_mips) return fail(c, VKRHIP_ERR_ARG, "mip %u of %u", mip, c->hzb_mips);
  TRY(bind_device(c));
  uint64_t offset = 0;
  for (uint32_t m = 0; m < mip; ++m)
    offset += (uint64_t)std::max(c->hzb_width >> m, 1u) * std::max(c->hzb_height >> m, 1u);
  const uint64_t texels = (uint64_t)std::max(c->hzb_width >> mip, 1u) * std::max(c->hzb_height >> mip, 1u);
  return read_back(c, out, c->hzb.p + offset, texels * sizeof(float));
}

int vkrhip_candidate_epoch(vkrhip_ctx *c, uint64_t *out_hash, uint32_t *out_occluders) {
  if (!c || !out_hash) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  const uint64_t n = c->n_cand;
  const uint64_t waves = ceil_div(n, 64);
  TRY(dev_alloc(c, c->epoch_maps, std::max<uint64_t>(n, 1)));
  TRY(dev_alloc(c, c->epoch_maps_low, std::max<uint64_t>(n, 1)));
  TRY(dev_alloc(c, c->epoch_wave_maps, std::max<uint64_t>(waves, 1)));
  TRY(dev_alloc(c, c->epoch_pairs, std::max<uint64_t>(waves, 1) * 2));
  TRY(dev_alloc(c, c->epoch_result, 2));
  TRY(dev_alloc(c, c->epoch_enter_low, std::max<uint64_t>(waves, 1)));
  TRY(dev_alloc(c, c->epoch_enter_high, std::max<uint64_t>(waves, 1)));
  TRY(dev_alloc(c, c->epoch_wave_occluders, std::max<uint64_t>(waves, 1)));
  EpochArgs a{};
  a.ident = c->ident.p;
  a.world = c->world.p;
  a.node = c->node_identity ? nullptr : c->node.p;
  a.sphere = c->sphere.p;
  a.cflags = c->cflags.p;
  a.n = n;
  a.waves = waves;
  a.maps = c->epoch_maps.p;
  a.maps_low = c->epoch_maps_low.p;
  a.wave_maps = c->epoch_wave_maps.p;
  a.enter_low = c->epoch_enter_low.p;
  a.enter_high = c->epoch_enter_high.p;
  a.wave_occluders = c->epoch_wave_occluders.p;
  a.pairs = c->epoch_pairs.p;
  a.result = c->epoch_result.p;
  if (waves) {
    const unsigned grid = (unsigned)ceil_div(waves * 64, 256);
    hipLaunchKernelGGL(k_epoch_low, dim3(grid), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_epoch_scan, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, a.wave_maps, waves,
                       (uint32_t)(FNV_OFFSET & 0xfu), a.enter_low);
    hipLaunchKernelGGL(k_epoch_high, dim3(grid), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_epoch_scan, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, a.wave_maps, waves,
                       (uint32_t)((FNV_OFFSET >> 4) & 0xfu), a.enter_high);
    hipLaunchKernelGGL(k_epoch_affine, dim3(grid), dim3(256), 0, c->stream, a);
  }
  hipLaunchKernelGGL(k_epoch_final, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  uint64_t host[2] = {0, 0};
  HIP_TRY(c, hipMemcpyAsync(host, c->epoch_result.p, sizeof(host), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *out_hash = host[0];
  if (out_occluders) *out_occluders = (uint32_t)host[1];
  return VKRHIP_OK;
}

uint64_t vkrhip_exchange_block_words(uint32_t views, uint32_t vis_cap, uint32_t key_cap) {
  return exchange_block_words(views, vis_cap, key_cap);
}

uint64_t vkrhip_exchange_result_words(uint32_t views) { return exchange_result_words(views); }

}  // extern "C"

namespace {

// `on`: null = the context's stream, behind a join with the second stream; the
// second stream itself when the frame's tail was put there (the pack then follows
// that work in stream order and the context's stream is not held up).
int enqueue_exchange_pack(vkrhip_ctx *c, uint32_t *block_dev, uint32_t vis_cap, uint32_t key_cap,
                          hipEvent_t packed = nullptr, hipStream_t on = nullptr) {
  if (!c->frame_ran || !(c->last_stages & VKRHIP_STAGE_CULL))
    return fail(c, VKRHIP_ERR_STATE, "exchange needs a frame that ran the cull stage");
  // with a publication the lists hold rank-local packed row indices, which
  // collide between ranks when the lists are concatenated
  if (c->published)
    return fail(c, VKRHIP_ERR_STATE,
                "the exchange merges source-row indices; it cannot run while a publication is set");
  if (!on) {
    TRY(join_side(c));  // the pack reads the frame's sorted keys
    on = c->stream;
  }
  ExchangePackArgs a{};
  a.states = c->states.p;
  a.visible = c->visible.p;
  a.keys = c->out_keys.p;
  a.counters = c->counters();
  a.block = block_dev;
  a.visible_stride = c->visible_stride;
  a.views = c->views;
  a.vis_cap = vis_cap;
  a.key_cap = key_cap;
  a.have_keys = (c->last_stages & VKRHIP_STAGE_TRANSPARENT) && c->out_keys.p ? 1u : 0u;
  const uint64_t widest = std::max<uint64_t>(std::max(vis_cap, key_cap), 1);
  const unsigned blocks = (unsigned)std::min<uint64_t>(ceil_div(widest, 1024), 512);
  hipExtLaunchKernelGGL(k_exchange_pack, dim3(blocks, c->views + 1u), dim3(256), 0, on, nullptr,
                        packed, 0, a);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

// The planning tables k_exchange_plan fills and k_exchange_apply reads.
struct ExchangeTables {
  uint64_t *desc;             // [views * 4][world] x {src word, dst word, count}
  uint64_t *rank_view_start;  // [world][views]
  uint64_t *prefix;           // [views * 4 + 1]
  uint64_t *runs;             // [128]
};

// Plan + apply on `stream`; `done` (may be null) is recorded right behind the
// last kernel by the launch itself. Allocates nothing.
void launch_exchange_merge(const ExchangeTables &t, uint32_t views, const uint32_t *gathered_dev,
                           uint32_t world, uint32_t vis_cap, uint32_t key_cap,
                           uint32_t *merged_visible_dev, uint64_t *merged_keys_dev,
                           uint64_t *result_dev, hipStream_t stream, hipEvent_t done) {
  const uint32_t pairs = views * 4u;
  ExchangePlanArgs a{};
  a.gathered = gathered_dev;
  a.desc = t.desc;
  a.rank_view_start = t.rank_view_start;
  a.prefix = t.prefix;
  a.runs = t.runs;
  a.result = result_dev;
  a.world = world;
  a.views = views;
  a.vis_cap = vis_cap;
  a.key_cap = key_cap;
  hipLaunchKernelGGL(k_exchange_plan, dim3(1), dim3(SCAN_BLOCK), 0, stream, a);
  // copy rows: enough blocks per descriptor to cover the largest possible range
  const unsigned rows = (unsigned)std::min<uint64_t>(std::max<uint64_t>(ceil_div(vis_cap, 4096), 1), 64);
  // (a block merges 256 keys of one run: every run may end in a block that is not full)
  const unsigned key_blocks = (unsigned)ceil_div((uint64_t)world * key_cap, 256) + world;
  if (done)
    hipExtLaunchKernelGGL(k_exchange_apply, dim3(pairs * world * rows + key_blocks), dim3(256), 0, stream,
                          nullptr, done, 0, gathered_dev, (const uint64_t *)t.desc, pairs * world, rows,
                          merged_visible_dev, (const uint64_t *)t.runs, world, merged_keys_dev);
  else  // (a plain launch: also what a stream under graph capture records)
    hipLaunchKernelGGL(k_exchange_apply, dim3(pairs * world * rows + key_blocks), dim3(256), 0, stream,
                       gathered_dev, (const uint64_t *)t.desc, pairs * world, rows, merged_visible_dev,
                       (const uint64_t *)t.runs, world, merged_keys_dev);
}

int enqueue_exchange_merge(vkrhip_ctx *c, const uint32_t *gathered_dev, uint32_t world,
                           uint32_t vis_cap, uint32_t key_cap, uint32_t *merged_visible_dev,
                           uint64_t *merged_keys_dev, uint64_t *result_dev, hipStream_t stream,
                           hipEvent_t done = nullptr) {
  if (world == 0 || world > 64) return fail(c, VKRHIP_ERR_ARG, "world size %u outside 1..64", world);
  if (c->views == 0) return fail(c, VKRHIP_ERR_STATE, "no views set");
  const uint32_t pairs = c->views * 4u;
  TRY(dev_alloc(c, c->exch_desc, (size_t)pairs * world * 3));
  TRY(dev_alloc(c, c->exch_rank_view_start, (size_t)world * c->views));
  TRY(dev_alloc(c, c->exch_prefix, (size_t)pairs + 1));
  TRY(dev_alloc(c, c->exch_runs, 128));
  const ExchangeTables t{c->exch_desc.p, c->exch_rank_view_start.p, c->exch_prefix.p, c->exch_runs.p};
  launch_exchange_merge(t, c->views, gathered_dev, world, vis_cap, key_cap, merged_visible_dev,
                        merged_keys_dev, result_dev, stream, done);
  HIP_TRY(c, hipGetLastError());
  return VKRHIP_OK;
}

}  // namespace

extern "C" {

int vkrhip_exchange_pack(vkrhip_ctx *c, uint32_t *block_dev, uint32_t vis_cap, uint32_t key_cap) {
  if (!c || !block_dev) return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  return enqueue_exchange_pack(c, block_dev, vis_cap, key_cap);
}

int vkrhip_exchange_merge(vkrhip_ctx *c, const uint32_t *gathered_dev, uint32_t world,
                          uint32_t vis_cap, uint32_t key_cap, uint32_t *merged_visible_dev,
                          uint64_t *merged_keys_dev, uint64_t *result_dev, void *hip_stream) {
  if (!c || !gathered_dev || !merged_visible_dev || !merged_keys_dev || !result_dev)
    return VKRHIP_ERR_ARG;
  TRY(bind_device(c));
  return enqueue_exchange_merge(c, gathered_dev, world, vis_cap, key_cap, merged_visible_dev,
                                merged_keys_dev, result_dev,
                                hip_stream ? (hipStream_t)hip_stream : c->stream);
}

int vkrhip_comm_unique_id(uint8_t *out_id128) {
  if (!out_id128) return VKRHIP_ERR_ARG;
  if (!load_rccl()) return fail(nullptr, VKRHIP_ERR_STATE, "librccl.so could not be loaded");
  const int rc = g_rccl.GetUniqueId(out_id128);
  if (rc != 0) return fail(nullptr, VKRHIP_ERR_HIP, "ncclGetUniqueId failed (%d)", rc);
  return VKRHIP_OK;
}

int vkrhip_comm_init(vkrhip_ctx *c, const uint8_t *id128, uint32_t world, uint32_t rank) {
  if (!c || !id128 || world == 0 || world > 64 || rank >= world) return VKRHIP_ERR_ARG;
  if (c->comm) return fail(c, VKRHIP_ERR_STATE, "the context already has a communicator");
  if (!load_rccl()) return fail(c, VKRHIP_ERR_STATE, "librccl.so could not be loaded");
  TRY(bind_device(c));
  RcclApi::UniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  void *comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, (int)world, id, (int)rank);
  if (rc != 0)
    return fail(c, VKRHIP_ERR_HIP, "ncclCommInitRank failed: %s",
                g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  c->comm = comm;
  c->comm_world = world;
  c->comm_rank = rank;
  return VKRHIP_OK;
}

int vkrhip_comm_count(vkrhip_ctx *c, uint32_t *out_ranks) {
  if (!c || !out_ranks) return VKRHIP_ERR_ARG;
  *out_ranks = 0;
  if (!c->comm) return fail(c, VKRHIP_ERR_STATE, "the context has no communicator");
  if (!g_rccl.CommCount) return fail(c, VKRHIP_ERR_STATE, "ncclCommCount is not available");
  int n = 0;
  const int rc = g_rccl.CommCount(c->comm, &n);
  if (rc != 0)
    return fail(c, VKRHIP_ERR_HIP, "ncclCommCount failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  *out_ranks = (uint32_t)n;
  return VKRHIP_OK;
}

int vkrhip_exchange_configure(vkrhip_ctx *c, uint32_t world, uint32_t vis_cap, uint32_t key_cap) {
  if (!c || world == 0 || world > 64) return VKRHIP_ERR_ARG;
  if (c->comm ? world != c->comm_world : world != 1)
    return fail(c, VKRHIP_ERR_STATE, "world %u does not match the communicator (%u)", world,
                c->comm ? c->comm_world : 1u);
  if (c->views == 0) return fail(c, VKRHIP_ERR_STATE, "no views set");
  TRY(bind_device(c));
  if (!c->xstream) {
    // Default priority, on purpose. Created with the highest priority (so that its
    // few small kernels between the collectives would be placed ahead of the next
    // frame's large grids - which measured no difference), this stream stalled the
    // whole context once the process held more streams than hardware queues: with
    // eight streams of default priority and two of high priority alive elsewhere in
    // the process (PyTorch creates pools of 32 per priority on first use), the third
    // run of the exchange never finished and even hipStreamQuery on this stream did
    // not return, whenever the frame's sort ran on the second stream; with this
    // stream at default priority the same process completes (tools/
    // exchange_hang_probe.py reproduces both). The stream's first packet per run is
    // a barrier that waits for the pack on a default-priority queue: a high-priority
    // queue that holds its hardware slot while it waits for a queue that is not
    // scheduled is a priority inversion the queue scheduler does not resolve.
    HIP_TRY(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  }
  TRY(join_side(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  TRY(drain_side(c));  // (a pack on the second stream writes the blocks that are replaced below)
  HIP_TRY(c, hipStreamSynchronize(c->xstream));
  const uint64_t words = exchange_block_words(c->views, vis_cap, key_cap);
  for (auto &x : c->xset) {
    TRY(dev_alloc(c, x.send, words));
    TRY(dev_alloc(c, x.recv, words * world));
    TRY(dev_alloc(c, x.merged_visible, std::max<uint64_t>((uint64_t)world * vis_cap, 1)));
    TRY(dev_alloc(c, x.merged_keys, std::max<uint64_t>((uint64_t)world * key_cap, 1)));
    TRY(dev_alloc(c, x.result, exchange_result_words(VKRHIP_MAX_VIEWS)));
    if (!x.done) HIP_TRY(c, hipEventCreateWithFlags(&x.done, hipEventDisableTiming));
    if (!x.packed) HIP_TRY(c, hipEventCreateWithFlags(&x.packed, hipEventDisableTiming));
    x.in_flight = false;
  }
  c->x_world = world;
  c->x_views = c->views;
  c->x_vis_cap = vis_cap;
  c->x_key_cap = key_cap;
  c->x_last = -1;
  return VKRHIP_OK;
}

int vkrhip_exchange_run(vkrhip_ctx *c) {
  if (!c) return VKRHIP_ERR_ARG;
  if (!c->x_world) return fail(c, VKRHIP_ERR_STATE, "vkrhip_exchange_configure has not been called");
  if (c->views != c->x_views)  // the block layout depends on the view count
    return fail(c, VKRHIP_ERR_STATE, "view count changed from %u to %u: configure the exchange again",
                c->x_views, c->views);
  TRY(bind_device(c));
  const int k = (int)(c->x_turn++ & 1u);
  auto &x = c->xset[k];
  // Where the frame left work on the second stream (its sort and emission, or its
  // whole tail), the pack follows it there: the context's stream - the next frame's
  // transform and classify - then never waits for this frame's lists and keys, nor
  // for the exchange. Otherwise the pack goes to the context's stream.
  const bool on_side = second_stream_usable(c) && c->side_last >= 0 && c->side_pending[c->side_last];
  hipStream_t pack_stream = on_side ? c->tstream : c->stream;
  // the exchange that used this set two frames ago must be done with it
  // (normally long finished: a barrier packet is only queued when it is not)
  if (x.in_flight && hipEventQuery(x.done) != hipSuccess) {
    if (c->xdebug & 16) fprintf(stderr, "[x] run %u: stream %s waits for done[%d]\n", c->x_turn, on_side ? "side" : "main", k);
    HIP_TRY(c, hipStreamWaitEvent(pack_stream, x.done, 0));
  }
  if (c->xdebug & 16) fprintf(stderr, "[x] run %u: pack on %s, set %d\n", c->x_turn, on_side ? "side" : "main", k);
  {
    StageTimer timer(c, VKRHIP_TIMER_EXCHANGE_PACK, pack_stream);
    TRY(enqueue_exchange_pack(c, x.send.p, c->x_vis_cap, c->x_key_cap, x.packed, on_side ? c->tstream : nullptr));
  }
  // the batch of second-stream work now ends behind the pack (which reads the
  // frame's lists, keys and counters): its event is recorded again
  if (on_side) {
    if (c->side_expected[c->side_last]) {
      c->side_expected[c->side_last] = ++c->side_serial;
      hipLaunchKernelGGL(k_side_signal, dim3(1), dim3(64), 0, c->tstream, c->side_flag.p + c->side_last,
                         c->side_expected[c->side_last]);
      HIP_TRY(c, hipGetLastError());
    } else {
      HIP_TRY(c, hipEventRecord(c->ev_side[c->side_last], c->tstream));
    }
  }
  HIP_TRY(c, hipStreamWaitEvent(c->xstream, x.packed, 0));
  const uint64_t words = exchange_block_words(c->views, c->x_vis_cap, c->x_key_cap);
  if (c->comm) {
    StageTimer timer(c, VKRHIP_TIMER_EXCHANGE_GATHER, c->xstream);
    const int rc = g_rccl.AllGather(x.send.p, x.recv.p, words, kNcclUint32, c->comm, c->xstream);
    if (rc != 0)
      return fail(c, VKRHIP_ERR_HIP, "ncclAllGather failed: %s",
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  }
  // without a communicator the world is this rank: its own block is the gathered buffer
  {
    StageTimer timer(c, VKRHIP_TIMER_EXCHANGE_MERGE, c->xstream);
    TRY(enqueue_exchange_merge(c, c->comm ? x.recv.p : x.send.p, c->x_world, c->x_vis_cap, c->x_key_cap,
                               x.merged_visible.p, x.merged_keys.p, x.result.p, c->xstream, x.done));
  }
  x.in_flight = true;
  c->x_last = k;
  return VKRHIP_OK;
}

int vkrhip_debug_streams(vkrhip_ctx *c, int32_t *out, uint32_t n) {
  if (!c || !out) return VKRHIP_ERR_ARG;
  int32_t v[10];
  auto sq = [](hipStream_t s) { return s ? (int32_t)(hipStreamQuery(s) != hipSuccess) : -1; };
  auto eq = [](hipEvent_t e) { return e ? (int32_t)(hipEventQuery(e) != hipSuccess) : -1; };
  v[0] = sq(c->stream);
  v[1] = sq(c->tstream);
  v[2] = sq(c->xstream);
  v[3] = eq(c->ev_gathered);
  v[4] = eq(c->ev_side[0]);
  v[5] = eq(c->ev_side[1]);
  v[6] = eq(c->xset[0].packed);
  v[7] = eq(c->xset[0].done);
  v[8] = eq(c->xset[1].packed);
  v[9] = eq(c->xset[1].done);
  (void)hipGetLastError();
  for (uint32_t k = 0; k < n && k < 10u; ++k) out[k] = v[k];
  return VKRHIP_OK;
}

int vkrhip_exchange_finish(vkrhip_ctx *c, uint64_t *result_out, const uint32_t **merged_visible_dev,
                           const uint64_t **merged_keys_dev) {
  if (!c || !result_out) return VKRHIP_ERR_ARG;
  if (c->x_last < 0) return fail(c, VKRHIP_ERR_STATE, "vkrhip_exchange_run has not been called");
  TRY(bind_device(c));
  auto &x = c->xset[c->x_last];
  if (c->xdebug & 32) {  // diagnosis: poll instead of blocking, and name what is pending when it stalls
    for (int tries = 0; hipStreamQuery(c->xstream) != hipSuccess; ++tries) {
      if (tries > 500) {
        int32_t v[10];
        (void)vkrhip_debug_streams(c, v, 10);
        return fail(c, VKRHIP_ERR_STATE,
                    "exchange stalled: main %d side %d exchange %d | ev_gathered %d ev_side %d %d | set0 packed %d "
                    "done %d | set1 packed %d done %d (1 = pending)",
                    v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
      }
      struct timespec ts = {0, 1000000};
      nanosleep(&ts, nullptr);
    }
    (void)hipGetLastError();
  }
  HIP_TRY(c, hipMemcpyAsync(result_out, x.result.p, exchange_result_words(c->views) * 8,
                            hipMemcpyDeviceToHost, c->xstream));
  HIP_TRY(c, hipStreamSynchronize(c->xstream));
  if (merged_visible_dev) *merged_visible_dev = x.merged_visible.p;
  if (merged_keys_dev) *merged_keys_dev = x.merged_keys.p;
  return VKRHIP_OK;
}

// ---------------------------------------------------------------------------
// The exchange merge for callers that cannot allocate: the planning tables live
// in a device array of the library instead of buffers a context allocates.
// ---------------------------------------------------------------------------
}  // extern "C"

namespace {

constexpr uint32_t kStaticMaxWorld = VKRHIP_STATIC_MAX_WORLD;
constexpr uint32_t kStaticMaxViews = VKRHIP_MAX_VIEWS;
constexpr size_t kStaticDescWords = (size_t)kStaticMaxViews * 4u * kStaticMaxWorld * 3u;
constexpr size_t kStaticStartWords = (size_t)kStaticMaxWorld * kStaticMaxViews;
constexpr size_t kStaticPrefixWords = ((size_t)kStaticMaxViews * 4u + 1u + 3u) & ~(size_t)3u;
constexpr size_t kStaticRunsWords = 128;
constexpr size_t kStaticScratchWords =
    kStaticDescWords + kStaticStartWords + kStaticPrefixWords + kStaticRunsWords;
static_assert(kStaticScratchWords * sizeof(uint64_t) == VKRHIP_STATIC_SCRATCH_BYTES,
              "the scratch size the header documents");

// Every GPU the library runs on holds its own copy of this array, at its own
// address, until the process ends. In host code the name is only a handle for
// hipGetSymbolAddress: kernels and runtime calls get the per-device address
// kept in g_scratch_ptr.
static __device__ __align__(256) uint64_t scratch[kStaticScratchWords];

constexpr int kStaticMaxDevices = 64;
uint64_t *g_scratch_ptr[kStaticMaxDevices] = {};
std::mutex g_scratch_mutex;

// Address of `scratch` on `device` (looked up once per device, with that device
// current). Leaves `device` current.
int static_scratch(int device, uint64_t **out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, VKRHIP_ERR_HIP, "no HIP device available");
  if (device < 0 || device >= ndev || device >= kStaticMaxDevices)
    return fail(nullptr, VKRHIP_ERR_ARG, "device %d out of range (0..%d)", device,
                std::min(ndev, kStaticMaxDevices) - 1);
  HIP_TRY(nullptr, hipSetDevice(device));
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    HIP_TRY(nullptr, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(nullptr, VKRHIP_ERR_HIP, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = static_cast<uint64_t *>(p);
  }
  *out = g_scratch_ptr[device];
  return VKRHIP_OK;
}

}  // namespace

extern "C" {

int vkrhip_static_init(int device) {
  uint64_t *p = nullptr;
  return static_scratch(device, &p);
}

int vkrhip_exchange_merge_static(int device, uint32_t views, const uint32_t *gathered_dev,
                                 uint32_t world, uint32_t vis_cap, uint32_t key_cap,
                                 uint32_t *merged_visible_dev, uint64_t *merged_keys_dev,
                                 uint64_t *result_dev, void *hip_stream) {
  if (!gathered_dev || !merged_visible_dev || !merged_keys_dev || !result_dev)
    return fail(nullptr, VKRHIP_ERR_ARG, "null device pointer");
  if (views == 0 || views > kStaticMaxViews)
    return fail(nullptr, VKRHIP_ERR_CAPACITY, "views %u outside 1..%u", views, kStaticMaxViews);
  if (world == 0 || world > kStaticMaxWorld)
    return fail(nullptr, VKRHIP_ERR_CAPACITY,
                "world size %u outside 1..%u: the planning tables of this entry point are a fixed "
                "%u-byte device array (vkrhip_exchange_merge has no such limit)",
                world, kStaticMaxWorld, (unsigned)VKRHIP_STATIC_SCRATCH_BYTES);
  uint64_t *base = nullptr;
  TRY(static_scratch(device, &base));
  // every entry the apply kernel reads is written by the planning kernel of the
  // same call first, so nothing of the previous call has to be cleared
  ExchangeTables t{};
  t.desc = base;
  t.rank_view_start = t.desc + kStaticDescWords;
  t.prefix = t.rank_view_start + kStaticStartWords;
  t.runs = t.prefix + kStaticPrefixWords;
  launch_exchange_merge(t, views, gathered_dev, world, vis_cap, key_cap, merged_visible_dev,
                        merged_keys_dev, result_dev, (hipStream_t)hip_stream, nullptr);
  HIP_TRY(nullptr, hipGetLastError());
  return VKRHIP_OK;
}

}  // extern "C"
