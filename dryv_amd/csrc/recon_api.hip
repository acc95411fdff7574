// recon_api.hip — extern "C" boundary (include/dryv_recon.h) over the gfx950 reconstruction kernel.
//
// Host-side responsibilities only: validate the parameter block, derive the per-submit constant
// tables (LevelScale, prediction gather tables), move buffers, launch, time, report status.
// There is deliberately NO CPU implementation of the path in this library: without a HIP device
// every entry point that would compute fails with DRYV_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "band_launch.h"
#include "recon_params.h"
#include "output_pack.h"
#include "deblock_launch.h"
#include "deblock_params.h"

using dryv::KParams;
using dryv::params::build_params;

struct dryv_recon_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // the pair around the most recent timed launch (one of ev_ring)
  // a ring of event pairs: launches queued back to back (dryv_recon_submit_device_queued) are timed one by one
  static constexpr unsigned kEvRing = 64;
  hipEvent_t ev_ring[kEvRing][2] = {};
  unsigned ev_n = 0;   // timed launches recorded so far
  unsigned* d_status = nullptr;
  unsigned* h_status = nullptr;  // pinned
  // host-buffer path: grow-only device staging
  void* d_mbs = nullptr;
  void* d_coeffs = nullptr;
  void* d_yuv = nullptr;
  size_t cap_mbs = 0, cap_coeffs = 0, cap_yuv = 0;
  void* d_pack = nullptr;        // output stage (dryv_recon_wait_packed): grow-only
  void* d_dbwork = nullptr;      // deblocking workspace (task counter, progress words, side buffer): grow-only
  size_t cap_dbwork = 0;
  size_t cap_pack = 0;
  dryv_frame_params pending_fp;  // the parameters of the host-path batch in flight
  uint32_t pending_frames = 0;
  size_t pending_yuv_bytes = 0;
  bool in_flight = false;        // a submit has not been waited for
  bool in_flight_host = false;   // ... and it was a host-buffer submit
  bool timed = false;
  // kernel workspace (task counter, per-row progress, bottom-row modes): grow-only
  void* d_work = nullptr;
  size_t cap_work = 0;
  // the hand-off records between bands carry the launch's generation as their tag: they are zeroed when the workspace is
  // (re)allocated or laid out for other picture dimensions / batch sizes, never per launch
  unsigned launch_gen = 0;
  unsigned task_base = 0;          // what the workspace's task counter stands at (it is never reset between launches)
  const void* hand_ws = nullptr;   // the workspace and the layout the records were last zeroed for
  int hand_W = 0, hand_H = 0, hand_frames = 0;
  // Queue lanes (dryv_recon_set_queue_lanes): queued device batches rotate over `queue_lanes` streams, each with a workspace of
  // its own, and every launch takes half of the resident grid -- two launches are then resident side by side, half a launch
  // apart, and one's ramp and drain run beside the other's steady state. Lane 0 is the context's own stream and workspace (the
  // fields above and below); a launch on lane k > 0 swaps that lane's fields in for its duration (LaneScope).
  struct Lane {
    hipStream_t stream = nullptr;
    void* d_work = nullptr;
    size_t cap_work = 0;
    unsigned launch_gen = 0, task_base = 0;
    const void* hand_ws = nullptr;
    int hand_W = 0, hand_H = 0, hand_frames = 0;
  };
  std::vector<Lane> lanes;         // lanes 1 .. queue_lanes - 1
  int queue_lanes = 1;
  bool lane_launch = false;        // the launch being made is a queued one on a context with several lanes (half grid)
  hipEvent_t ev_queue = nullptr;   // recorded on the context's stream in front of a queue's first launch: the other lanes wait for it
  int num_cus = 256;
  int grid_override = 0;
  bool force_wide = false;  // DRYV_RECON_FORCE_WIDE (test hook): every launch with the WIDE build (64-bit residual arithmetic)
  // the launch in flight, kept so that a batch the fast band kernel flagged (status bit 1: a block beyond int32)
  // can be run again with the wide build before its status is reported
  KParams last_P;
  const void* last_mbs = nullptr;
  const void* last_coeffs = nullptr;
  void* last_yuv = nullptr;
  bool last_band = false;
  int wide_reruns = 0;
  int wide_rerun_batches = 0;   // batches that were launched again with the wide build
  // device-resident batches queued since the last sync (dryv_recon_submit_device_queued): kept for the wide re-run
  struct Queued { KParams P; const void* mbs; const void* coeffs; void* yuv; };
  std::vector<Queued> queued;
  // pipelined host path (dryv_recon_submit_host): copy-in and copy-out streams, per-chunk events
  hipStream_t s_in = nullptr, s_out = nullptr;
  std::vector<hipEvent_t> ev_in, ev_k;
  bool piped = false;          // the batch in flight was submitted with dryv_recon_submit_host
  const dryv_mb_desc* piped_mbs = nullptr;
  const int16_t* piped_coeffs = nullptr;
  uint8_t* piped_out = nullptr;
  bool deblock_pending = false;  // a deblocking launch has not been synchronised yet (its status word is unread)
  std::string last_error;
};

namespace {

int fail(dryv_recon_ctx* ctx, hipError_t e, const char* what) {
  if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return DRYV_E_DEVICE;
}

int ensure(dryv_recon_ctx* ctx, void** p, size_t* cap, size_t need) {
  if (*cap >= need) return DRYV_OK;
  if (*p) (void)hipFree(*p);
  if (p == &ctx->d_work) ctx->hand_ws = nullptr;   // (a new workspace: its hand-off records are not zeroed yet)
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess) {
    ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
    return DRYV_E_NOMEM;
  }
  *cap = need;
  return DRYV_OK;
}

size_t workspace_bytes(const KParams& P) {
  size_t b = dryv::band_workspace_bytes(P);
#if defined(DRYV_BAND_PROFILE) || defined(DRYV_BAND_TRACE) || defined(DRYV_BAND_TIMELINE)
  b = dryv::band_profile_offset(P) + (size_t)65536 * 16 * 8 + (size_t)65536 * 32 * 2;  /* phases, trace, timeline */
#endif
  return b;
}

// The per-launch part of the workspace (task counter, progress words) zeroed on the stream; the hand-off records zeroed if
// this is another workspace or another layout than the last launch's. Returns the launch's generation in *gen.
// Nothing in the workspace is reset per launch: the task counter keeps counting (a launch's tasks are its values from
// task_base on), progress words and hand-off records carry the launch's generation. Everything is zeroed when the workspace
// is new or laid out for other dimensions, or when the generation's low 21 bits (the progress words' tag) come round.
int prepare_workspace(dryv_recon_ctx* ctx, const KParams& P, unsigned* gen, unsigned* task_base) {
  hipError_t e;
  unsigned g = ctx->launch_gen + 1;
  if ((g & 0x1FFFFFu) == 0u) g++;   // (never a tag of 0; g == 0 is caught by the wrap below as well)
  const bool other = ctx->hand_ws != ctx->d_work || ctx->hand_W != P.W || ctx->hand_H != P.H || ctx->hand_frames != P.n_frames;
  if (other || (g >> 21) != (ctx->launch_gen >> 21)) {
    e = hipMemsetAsync(ctx->d_work, 0, dryv::band_reset_bytes(P), ctx->stream);
    if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(workspace)");
    if (dryv::band_handoff_bytes(P) != 0) {
      e = hipMemsetAsync((unsigned char*)ctx->d_work + dryv::band_handoff_offset(P), 0, dryv::band_handoff_bytes(P), ctx->stream);
      if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(hand-off records)");
    }
    ctx->hand_ws = ctx->d_work;
    ctx->hand_W = P.W;
    ctx->hand_H = P.H;
    ctx->hand_frames = P.n_frames;
    ctx->task_base = 0;
    if (g == 0) g = 1;
  }
  ctx->launch_gen = g;
  *gen = g;
  *task_base = ctx->task_base;
  return DRYV_OK;
}

// Swaps lane k's stream / workspace / hand-off state with the context's own for the lifetime of the object (k = 0: nothing)
struct LaneScope {
  dryv_recon_ctx* c;
  dryv_recon_ctx::Lane* l;
  LaneScope(dryv_recon_ctx* ctx, int k) : c(ctx), l(k > 0 ? &ctx->lanes[(size_t)k - 1] : nullptr) { swap(); }
  ~LaneScope() { swap(); }
  void swap() {
    if (!l) return;
    std::swap(c->stream, l->stream);
    std::swap(c->d_work, l->d_work);
    std::swap(c->cap_work, l->cap_work);
    std::swap(c->launch_gen, l->launch_gen);
    std::swap(c->task_base, l->task_base);
    std::swap(c->hand_ws, l->hand_ws);
    std::swap(c->hand_W, l->hand_W);
    std::swap(c->hand_H, l->hand_H);
    std::swap(c->hand_frames, l->hand_frames);
  }
};

// the event pair the next timed launch records
void next_events(dryv_recon_ctx* ctx) {
  const unsigned k = ctx->ev_n++ % dryv_recon_ctx::kEvRing;
  ctx->ev_start = ctx->ev_ring[k][0];
  ctx->ev_stop = ctx->ev_ring[k][1];
}

int launch_band(dryv_recon_ctx* ctx, const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, bool wide) {
  hipError_t e;
  const long long tasks = (long long)P.n_frames * ((P.H + 3) / 4);
  const int wpb = dryv::band_teams_per_block(P.transform8x8 != 0, wide);
  long long grid = ctx->grid_override > 0 ? ctx->grid_override : (long long)ctx->num_cus * dryv::band_blocks_per_cu(P.transform8x8 != 0, wide);
  // (queue lanes: half of the resident grid per launch, two launches side by side)
  if (ctx->lane_launch) grid = std::max(1ll, grid / 2);
  if (grid > (tasks + wpb - 1) / wpb) grid = (tasks + wpb - 1) / wpb;
  if (grid < 1) grid = 1;
  unsigned gen = 0, task_base = 0;
  if (int st = prepare_workspace(ctx, P, &gen, &task_base)) return st;
#if defined(DRYV_BAND_PROFILE) || defined(DRYV_BAND_TRACE) || defined(DRYV_BAND_TIMELINE)
  e = hipMemsetAsync((unsigned char*)ctx->d_work + dryv::band_profile_offset(P), 0, (size_t)65536 * 16 * 8 + (size_t)65536 * 32 * 2, ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(profile)");
#endif
  next_events(ctx);
  e = dryv::band_launch(P, d_mbs, d_coeffs, d_yuv, ctx->d_status, ctx->d_work, (int)grid, wide, (unsigned)ctx->queued.size(), gen, task_base, ctx->stream,
                        ctx->ev_start, ctx->ev_stop);
  if (e != hipSuccess) return fail(ctx, e, "band_kernel launch");
  ctx->task_base += dryv::band_claims_per_launch(P, (int)grid, wide);
  // (the status words are fetched where the host waits -- finish() --, once per wait: a 32-byte copy behind every launch of a
  // queue is a stream operation of its own between two kernels)
  ctx->timed = true;
  return DRYV_OK;
}


// One chunk of a pipelined host submit: workspace reset + kernel on the compute stream; status accumulates over the
// chunks (it is cleared once per submit), no events.
int launch_chunk(dryv_recon_ctx* ctx, const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv) {
  hipError_t e;
  ctx->last_band = true;
  const long long tasks = (long long)P.n_frames * ((P.H + 3) / 4);
  const int wpb = dryv::band_teams_per_block(P.transform8x8 != 0, false);
  long long grid = ctx->grid_override > 0 ? ctx->grid_override : (long long)ctx->num_cus * dryv::band_blocks_per_cu(P.transform8x8 != 0, false);
  grid = std::max(1ll, std::min(grid, (tasks + wpb - 1) / wpb));
  unsigned gen = 0, task_base = 0;
  if (int st = prepare_workspace(ctx, P, &gen, &task_base)) return st;
  e = dryv::band_launch(P, d_mbs, d_coeffs, d_yuv, ctx->d_status, ctx->d_work, (int)grid, false, 0u, gen, task_base, ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "band_kernel launch");
  ctx->task_base += dryv::band_claims_per_launch(P, (int)grid, false);
  return DRYV_OK;
}

// Waits for the launch in flight. A batch the fast band kernel flagged as needing 64-bit arithmetic is run again with
// the wide build (same buffers; the caller's inputs are still valid: they must be until wait/sync returns).
// the device's status words behind everything queued so far, then the wait
hipError_t sync_with_status(dryv_recon_ctx* ctx) {
  hipError_t e = hipMemcpyAsync(ctx->h_status, ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

int finish(dryv_recon_ctx* ctx) {
  hipError_t e = hipSuccess;
  for (dryv_recon_ctx::Lane& l : ctx->lanes)   // (queued batches on the other lanes: their status words are in d_status too)
    if (l.stream && (e = hipStreamSynchronize(l.stream)) != hipSuccess) return fail(ctx, e, "hipStreamSynchronize(queue lane)");
  e = sync_with_status(ctx);
  if (e != hipSuccess) return fail(ctx, e, "hipStreamSynchronize");
  if (ctx->piped) {
    ctx->piped = false;
    if ((e = hipStreamSynchronize(ctx->s_out)) != hipSuccess) return fail(ctx, e, "hipStreamSynchronize(copy-out)");
    if (*ctx->h_status & 2u) {
      // a block beyond int32: the whole batch again, unpipelined, with the band kernel's wide
      // build (never for a conformant stream)
      ctx->wide_reruns++;
      ctx->wide_rerun_batches++;
      ctx->last_band = true;
      const KParams& P = ctx->last_P;
      if ((e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipMemsetAsync(status)");
      int st = launch_band(ctx, P, ctx->d_mbs, ctx->d_coeffs, ctx->d_yuv, true);
      if (st != DRYV_OK) return st;
      e = hipMemcpyAsync(ctx->piped_out, ctx->d_yuv, (size_t)P.n_frames * P.W * P.H * 384, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = sync_with_status(ctx);
      if (e != hipSuccess) return fail(ctx, e, "wide re-run");
    }
  } else
  if (*ctx->h_status & 2u) {
    ctx->wide_reruns++;
    ctx->last_band = true;
    // What the batches that are NOT run again have reported stays reported: an unsupported record (bit 0) or a band that
    // gave up waiting (bit 2, with where) in front of the first flagged batch of a queue would otherwise vanish with the
    // status words that the re-run starts from
    const unsigned keep = ctx->h_status[0] & 5u, keep1 = ctx->h_status[1], keep2 = ctx->h_status[2], keep3 = ctx->h_status[3];
    e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream);
    if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(status)");
    int st = DRYV_OK;
    if (ctx->queued.size() > 1) {
      // several batches were queued behind each other: status word 4 names the first one that raised the flag (a launch
      // carries its position in the queue); that one and everything behind it again with the wide build, in order
      const unsigned first = std::min<unsigned>(~ctx->h_status[4], (unsigned)ctx->queued.size() - 1u);
      std::vector<dryv_recon_ctx::Queued> again(ctx->queued.begin() + first, ctx->queued.end());
      ctx->queued.clear();   // (the re-runs are launches of their own: sequence numbers from 0)
      ctx->wide_rerun_batches += (int)again.size();
      for (const dryv_recon_ctx::Queued& q : again)
        if ((st = launch_band(ctx, q.P, q.mbs, q.coeffs, q.yuv, true)) != DRYV_OK) break;
    } else {
      ctx->wide_rerun_batches++;
      st = launch_band(ctx, ctx->last_P, ctx->last_mbs, ctx->last_coeffs, ctx->last_yuv, true);
    }
    if (st != DRYV_OK) return st;
    e = sync_with_status(ctx);
    if (e != hipSuccess) return fail(ctx, e, "hipStreamSynchronize");
    if ((keep & 4u) && !(ctx->h_status[0] & 4u)) {
      ctx->h_status[1] = keep1;
      ctx->h_status[2] = keep2;
      ctx->h_status[3] = keep3;
    }
    ctx->h_status[0] |= keep;
  }
  if (ctx->last_band && (*ctx->h_status & 4u)) {
    char msg[160];
    snprintf(msg, sizeof msg, "band kernel: band task %u gave up waiting at step %u for the band above (needs %u macroblocks, saw %u)",
             ctx->h_status[1], ctx->h_status[2] >> 16, ctx->h_status[2] & 0xffffu, ctx->h_status[3]);
    ctx->last_error = msg;
    return DRYV_E_DEVICE;
  }
  return DRYV_OK;
}

int launch(dryv_recon_ctx* ctx, const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, bool reset_status = true) {
  int st = ensure(ctx, &ctx->d_work, &ctx->cap_work, workspace_bytes(P));
  if (st != DRYV_OK) return st;
  hipError_t e = hipSuccess;
  if (reset_status) e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(status)");
  // Persistent grid: every team keeps claiming 4-row bands until none are left, so a smaller grid is merely slower and
  // never incorrect.
  ctx->last_P = P;
  ctx->last_mbs = d_mbs;
  ctx->last_coeffs = d_coeffs;
  ctx->last_yuv = d_yuv;
  ctx->last_band = true;
  return launch_band(ctx, P, d_mbs, d_coeffs, d_yuv, ctx->force_wide);
}

}  // namespace

extern "C" {

size_t dryv_recon_frame_bytes(const dryv_frame_params* fp) {
  if (!fp) return 0;
  return (size_t)384 * fp->pic_width_in_mbs * fp->pic_height_in_mbs;
}

int dryv_recon_abi_version(void) { return DRYV_RECON_ABI_VERSION; }

int dryv_recon_check_params(const dryv_frame_params* fp, uint32_t n_frames) {
  KParams P;
  return build_params(fp, n_frames, &P);
}

int dryv_recon_create(dryv_recon_ctx** out, int device_ordinal) {
  if (!out) return DRYV_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DRYV_E_NODEVICE;
  if (device_ordinal < 0 || device_ordinal >= n) return DRYV_E_INVALID;
  if (hipSetDevice(device_ordinal) != hipSuccess) return DRYV_E_NODEVICE;
  dryv_recon_ctx* ctx = new (std::nothrow) dryv_recon_ctx();
  if (!ctx) return DRYV_E_NOMEM;
  ctx->device = device_ordinal;
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc((void**)&ctx->d_status, 32)) != hipSuccess ||
      (e = hipHostMalloc((void**)&ctx->h_status, 32, hipHostMallocDefault)) != hipSuccess) {
    dryv_recon_destroy(ctx);
    return DRYV_E_DEVICE;
  }
  *ctx->h_status = 0;
  for (unsigned k = 0; k < dryv_recon_ctx::kEvRing; k++)
    if ((e = hipEventCreate(&ctx->ev_ring[k][0])) != hipSuccess || (e = hipEventCreate(&ctx->ev_ring[k][1])) != hipSuccess) {
      dryv_recon_destroy(ctx);
      return DRYV_E_DEVICE;
    }
  ctx->ev_start = ctx->ev_ring[0][0];
  ctx->ev_stop = ctx->ev_ring[0][1];
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
      ctx->num_cus = prop.multiProcessorCount;
  }
  if (const char* s = getenv("DRYV_RECON_GRID")) ctx->grid_override = atoi(s);
  if (const char* s = getenv("DRYV_RECON_FORCE_WIDE")) ctx->force_wide = atoi(s) != 0;
  *out = ctx;
  return DRYV_OK;
}

void dryv_recon_destroy(dryv_recon_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_mbs) (void)hipFree(ctx->d_mbs);
  if (ctx->d_coeffs) (void)hipFree(ctx->d_coeffs);
  if (ctx->d_yuv) (void)hipFree(ctx->d_yuv);
  if (ctx->d_pack) (void)hipFree(ctx->d_pack);
  if (ctx->d_dbwork) (void)hipFree(ctx->d_dbwork);
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  for (dryv_recon_ctx::Lane& l : ctx->lanes) {
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (l.d_work) (void)hipFree(l.d_work);
    if (l.stream) (void)hipStreamDestroy(l.stream);
  }
  if (ctx->ev_queue) (void)hipEventDestroy(ctx->ev_queue);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  for (unsigned k = 0; k < dryv_recon_ctx::kEvRing; k++)
    for (int j = 0; j < 2; j++)
      if (ctx->ev_ring[k][j]) (void)hipEventDestroy(ctx->ev_ring[k][j]);
  for (hipEvent_t ev : ctx->ev_in) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_k) (void)hipEventDestroy(ev);
  if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
  if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int dryv_recon_submit(dryv_recon_ctx* ctx, const dryv_frame_params* fp, uint32_t n_frames, const dryv_mb_desc* mbs,
                      const int16_t* coeffs) {
  if (!ctx || !mbs || !coeffs || n_frames == 0) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;
  KParams P;
  int st = build_params(fp, n_frames, &P);
  if (st != DRYV_OK) return st;
  (void)hipSetDevice(ctx->device);
  const size_t n_mbs = (size_t)n_frames * P.W * P.H;
  const size_t b_mbs = n_mbs * sizeof(dryv_mb_desc), b_co = n_mbs * DRYV_COEFFS_PER_MB * sizeof(int16_t);
  const size_t b_yuv = n_mbs * 384;
  if ((st = ensure(ctx, &ctx->d_mbs, &ctx->cap_mbs, b_mbs)) != DRYV_OK) return st;
  if ((st = ensure(ctx, &ctx->d_coeffs, &ctx->cap_coeffs, b_co)) != DRYV_OK) return st;
  if ((st = ensure(ctx, &ctx->d_yuv, &ctx->cap_yuv, b_yuv)) != DRYV_OK) return st;
  hipError_t e;
  if ((e = hipMemcpyAsync(ctx->d_mbs, mbs, b_mbs, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
    return fail(ctx, e, "H2D macroblock records");
  if ((e = hipMemcpyAsync(ctx->d_coeffs, coeffs, b_co, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
    return fail(ctx, e, "H2D coefficients");
  if ((st = launch(ctx, P, ctx->d_mbs, ctx->d_coeffs, ctx->d_yuv)) != DRYV_OK) return st;
  ctx->pending_yuv_bytes = b_yuv;
  ctx->pending_fp = *fp;
  ctx->pending_frames = n_frames;
  ctx->in_flight = true;
  ctx->in_flight_host = true;
  return DRYV_OK;
}

int dryv_recon_wait(dryv_recon_ctx* ctx, uint8_t* yuv_out, size_t yuv_out_bytes) {
  if (!ctx || !yuv_out) return DRYV_E_INVALID;
  if (!ctx->in_flight || !ctx->in_flight_host) return DRYV_E_STATE;
  if (yuv_out_bytes < ctx->pending_yuv_bytes) return DRYV_E_INVALID;
  (void)hipSetDevice(ctx->device);
  int st = finish(ctx);
  hipError_t e = hipSuccess;
  if (st == DRYV_OK) {
    e = hipMemcpyAsync(yuv_out, ctx->d_yuv, ctx->pending_yuv_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  ctx->in_flight = false;
  ctx->in_flight_host = false;
  if (st != DRYV_OK) return st;
  if (e != hipSuccess) return fail(ctx, e, "D2H planes");
  return (*ctx->h_status & 1u) ? DRYV_E_UNSUPPORTED : DRYV_OK;
}


/* ---- output stage ----------------------------------------------------------------------------------- */
size_t dryv_recon_output_bytes(const dryv_frame_params* fp, const dryv_output_desc* od) {
  dryv::PackGeo G;
  return dryv::pack_geometry(fp, od, 1, &G) == DRYV_OK ? G.dst_frame_bytes : 0;
}

int dryv_recon_pack_device(dryv_recon_ctx* ctx, const dryv_frame_params* fp, uint32_t n_frames, const void* d_yuv,
                           const dryv_output_desc* od, void* d_out) {
  if (!ctx || !d_yuv || !d_out || n_frames == 0) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;  // (a flagged batch is re-run at sync: its planes are final only after that)
  dryv::PackGeo G;
  const int st = dryv::pack_geometry(fp, od, n_frames, &G);
  if (st != DRYV_OK) return st;
  (void)hipSetDevice(ctx->device);
  const hipError_t e = dryv::pack_launch(G, d_yuv, d_out, ctx->num_cus, ctx->stream);
  return e == hipSuccess ? DRYV_OK : fail(ctx, e, "pack kernel launch");
}

int dryv_recon_wait_packed(dryv_recon_ctx* ctx, const dryv_output_desc* od, uint8_t* out, size_t out_bytes) {
  return dryv_recon_wait_filtered(ctx, nullptr, od, out, out_bytes);
}

/* ---- in-loop deblocking filter ---------------------------------------------------------------------- */
int dryv_recon_deblock_device(dryv_recon_ctx* ctx, const dryv_frame_params* fp, const dryv_deblock_params* dp, uint32_t n_frames,
                              const void* d_mbs, void* d_yuv) {
  if (!ctx || !d_mbs || !d_yuv) return DRYV_E_INVALID;
  if (((uintptr_t)d_mbs | (uintptr_t)d_yuv) & 15u) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;  // (the pictures are final only after the batch's sync: see dryv_recon_pack_device)
  dryv::deblock::DParams P;
  int skip = 0;
  int st = dryv::deblock::build_dparams(fp, dp, n_frames, &P, &skip);
  if (st != DRYV_OK) return st;
  if (skip) return DRYV_OK;  // disable_deblocking_filter_idc = 1
  (void)hipSetDevice(ctx->device);
  if ((st = ensure(ctx, &ctx->d_dbwork, &ctx->cap_dbwork, dryv::deblock::workspace_bytes(P))) != DRYV_OK) return st;
  hipError_t e = hipMemsetAsync(ctx->d_dbwork, 0, dryv::deblock::reset_bytes(P), ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "hipMemsetAsync(deblock workspace)");
  const long long tasks = 2ll * P.n_frames * ((P.H + 3) / 4);  // a luma and a chroma task per band
  const int wpb = dryv::deblock_waves_per_block();
  long long grid = ctx->grid_override > 0 ? ctx->grid_override : (long long)ctx->num_cus * dryv::deblock_blocks_per_cu();
  grid = std::max(1ll, std::min(grid, (tasks + wpb - 1) / wpb));
  if ((e = hipEventRecord(ctx->ev_start, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipEventRecord");
  if ((e = dryv::deblock_launch(P, d_mbs, d_yuv, ctx->d_status, ctx->d_dbwork, (int)grid, ctx->stream)) != hipSuccess)
    return fail(ctx, e, "deblock_kernel launch");
  if ((e = hipEventRecord(ctx->ev_stop, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipEventRecord");
  if ((e = hipMemcpyAsync(ctx->h_status, ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
    return fail(ctx, e, "hipMemcpyAsync(status)");
  ctx->timed = true;
  ctx->deblock_pending = true;
  return DRYV_OK;
}

/* dryv_recon_wait with the stages behind reconstruction: deblocking (dp, may be NULL) and cropping / packing (od, may be NULL) */
int dryv_recon_wait_filtered(dryv_recon_ctx* ctx, const dryv_deblock_params* dp, const dryv_output_desc* od, uint8_t* out,
                             size_t out_bytes) {
  if (!ctx || !out) return DRYV_E_INVALID;
  if (!ctx->in_flight || !ctx->in_flight_host) return DRYV_E_STATE;
  static const dryv_output_desc plain = {DRYV_OUT_I420, {0, 0, 0}, 0, 0, 0, 0};
  dryv::PackGeo G;
  int st = dryv::pack_geometry(&ctx->pending_fp, od ? od : &plain, ctx->pending_frames, &G);
  if (st != DRYV_OK) return st;  // (the batch stays in flight: dryv_recon_wait can still fetch it)
  dryv::deblock::DParams DP;
  int skip = 1;
  if (dp && (st = dryv::deblock::build_dparams(&ctx->pending_fp, dp, ctx->pending_frames, &DP, &skip)) != DRYV_OK) return st;
  const size_t need = G.dst_frame_bytes * ctx->pending_frames;
  if (out_bytes < need) return DRYV_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (od && (st = ensure(ctx, &ctx->d_pack, &ctx->cap_pack, need)) != DRYV_OK) return st;
  st = finish(ctx);
  const bool unsupported = st == DRYV_OK && (*ctx->h_status & 1u);
  ctx->in_flight = false;
  ctx->in_flight_host = false;
  if (st != DRYV_OK) return st;
  if (dp && !skip) {
    const dryv_frame_params fp = ctx->pending_fp;
    if ((st = dryv_recon_deblock_device(ctx, &fp, dp, ctx->pending_frames, ctx->d_mbs, ctx->d_yuv)) != DRYV_OK) return st;
    if ((st = dryv_recon_sync(ctx)) != DRYV_OK) return st;
  }
  hipError_t e = hipSuccess;
  const void* src = ctx->d_yuv;
  if (od) {
    e = dryv::pack_launch(G, ctx->d_yuv, ctx->d_pack, ctx->num_cus, ctx->stream);
    src = ctx->d_pack;
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, src, need, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "filter / pack / D2H");
  return unsupported ? DRYV_E_UNSUPPORTED : DRYV_OK;
}

/* ---- pinned host memory + pipelined host-buffer path ---------------------------------------------- */
void* dryv_recon_alloc_host(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void dryv_recon_free_host(void* p) {
  if (p) (void)hipHostFree(p);
}

int dryv_recon_submit_host(dryv_recon_ctx* ctx, const dryv_frame_params* fp, uint32_t n_frames, const dryv_mb_desc* mbs,
                           const int16_t* coeffs, uint8_t* yuv_out, size_t yuv_out_bytes) {
  if (!ctx || !mbs || !coeffs || !yuv_out || n_frames == 0) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;
  KParams P;
  int st = build_params(fp, n_frames, &P);
  if (st != DRYV_OK) return st;
  (void)hipSetDevice(ctx->device);
  const size_t per = (size_t)P.W * P.H, n_mbs = per * n_frames;
  if (yuv_out_bytes < n_mbs * 384) return DRYV_E_INVALID;
  if ((st = ensure(ctx, &ctx->d_mbs, &ctx->cap_mbs, n_mbs * sizeof(dryv_mb_desc))) != DRYV_OK) return st;
  if ((st = ensure(ctx, &ctx->d_coeffs, &ctx->cap_coeffs, n_mbs * 768)) != DRYV_OK) return st;
  if ((st = ensure(ctx, &ctx->d_yuv, &ctx->cap_yuv, n_mbs * 384)) != DRYV_OK) return st;
  hipError_t e;
  if (!ctx->s_in && ((e = hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking)) != hipSuccess ||
                     (e = hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking)) != hipSuccess))
    return fail(ctx, e, "hipStreamCreate");
  // Chunks of frames: copy-in of chunk k+1, reconstruction of chunk k and copy-out of chunk k-1 overlap on three
  // streams. A chunk's kernel cannot be shorter than one frame's dependency chain (about a millisecond), so chunks
  // are sized for a copy-in of a few milliseconds: ~128 MB of coefficients.
  uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)128 << 20) / (per * 768));
  if (const char* sv = getenv("DRYV_RECON_CHUNK_FRAMES")) chunk = (uint32_t)std::max(1, atoi(sv));
  const uint32_t n_chunks = (n_frames + chunk - 1) / chunk;
  while (ctx->ev_in.size() < n_chunks) {
    hipEvent_t a, b;
    if ((e = hipEventCreateWithFlags(&a, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&b, hipEventDisableTiming)) != hipSuccess)
      return fail(ctx, e, "hipEventCreate");
    ctx->ev_in.push_back(a);
    ctx->ev_k.push_back(b);
  }
  if ((e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipMemsetAsync(status)");
  st = ensure(ctx, &ctx->d_work, &ctx->cap_work, workspace_bytes(P));
  if (st != DRYV_OK) return st;
  if ((e = hipEventRecord(ctx->ev_start, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipEventRecord");
  for (uint32_t k = 0; k < n_chunks; k++) {
    const uint32_t f0 = k * chunk, nf = std::min(chunk, n_frames - f0);
    const size_t m0 = (size_t)f0 * per, nm = (size_t)nf * per;
    if ((e = hipMemcpyAsync((char*)ctx->d_mbs + m0 * 16, mbs + m0, nm * 16, hipMemcpyHostToDevice, ctx->s_in)) != hipSuccess ||
        (e = hipMemcpyAsync((char*)ctx->d_coeffs + m0 * 768, coeffs + m0 * 384, nm * 768, hipMemcpyHostToDevice, ctx->s_in)) != hipSuccess ||
        (e = hipEventRecord(ctx->ev_in[k], ctx->s_in)) != hipSuccess ||
        (e = hipStreamWaitEvent(ctx->stream, ctx->ev_in[k], 0)) != hipSuccess)
      return fail(ctx, e, "chunk copy-in");
    KParams Pk = P;
    Pk.n_frames = (int)nf;
    st = launch_chunk(ctx, Pk, (char*)ctx->d_mbs + m0 * 16, (char*)ctx->d_coeffs + m0 * 768, (char*)ctx->d_yuv + m0 * 384);
    if (st != DRYV_OK) return st;
    if ((e = hipEventRecord(ctx->ev_k[k], ctx->stream)) != hipSuccess ||
        (e = hipStreamWaitEvent(ctx->s_out, ctx->ev_k[k], 0)) != hipSuccess ||
        (e = hipMemcpyAsync(yuv_out + m0 * 384, (char*)ctx->d_yuv + m0 * 384, nm * 384, hipMemcpyDeviceToHost, ctx->s_out)) != hipSuccess)
      return fail(ctx, e, "chunk copy-out");
  }
  if ((e = hipEventRecord(ctx->ev_stop, ctx->stream)) != hipSuccess) return fail(ctx, e, "hipEventRecord");
  if ((e = hipMemcpyAsync(ctx->h_status, ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
    return fail(ctx, e, "hipMemcpyAsync(status)");
  ctx->timed = true;
  ctx->last_P = P;
  ctx->piped = true;
  ctx->piped_mbs = mbs;
  ctx->piped_coeffs = coeffs;
  ctx->piped_out = yuv_out;
  ctx->in_flight = true;
  ctx->in_flight_host = false;
  return DRYV_OK;
}

int dryv_recon_submit_device(dryv_recon_ctx* ctx, const dryv_frame_params* fp, uint32_t n_frames, const void* d_mbs,
                             const void* d_coeffs, void* d_yuv_out) {
  if (!ctx || !d_mbs || !d_coeffs || !d_yuv_out || n_frames == 0) return DRYV_E_INVALID;
  // 16-byte DMA of the coefficients, dword stores of the planes
  if (((uintptr_t)d_mbs | (uintptr_t)d_coeffs | (uintptr_t)d_yuv_out) & 15u) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;  // one batch at a time: the workspace belongs to the batch in flight
  KParams P;
  int st = build_params(fp, n_frames, &P);
  if (st != DRYV_OK) return st;
  (void)hipSetDevice(ctx->device);
  if ((st = launch(ctx, P, d_mbs, d_coeffs, d_yuv_out)) != DRYV_OK) return st;
  ctx->in_flight = true;
  ctx->in_flight_host = false;
  return DRYV_OK;
}

int dryv_recon_submit_device_queued(dryv_recon_ctx* ctx, const dryv_frame_params* fp, uint32_t n_frames, const void* d_mbs,
                                    const void* d_coeffs, void* d_yuv_out) {
  if (!ctx || !d_mbs || !d_coeffs || !d_yuv_out || n_frames == 0) return DRYV_E_INVALID;
  if (((uintptr_t)d_mbs | (uintptr_t)d_coeffs | (uintptr_t)d_yuv_out) & 15u) return DRYV_E_INVALID;
  // behind batches of its own kind only: anything else in flight owns the context until it has been waited for
  if (ctx->in_flight && ctx->queued.empty()) return DRYV_E_STATE;
  KParams P;
  int st = build_params(fp, n_frames, &P);
  if (st != DRYV_OK) return st;
  (void)hipSetDevice(ctx->device);
  const int lane = ctx->queue_lanes > 1 ? (int)(ctx->queued.size() % (size_t)ctx->queue_lanes) : 0;
  const bool first = ctx->queued.empty();
  if (ctx->queue_lanes > 1 && first) {
    // the status word is cleared once, on the context's stream, and the other lanes start behind that point: whatever the
    // caller has put on the context's stream so far (the queue's inputs) is in front of every lane's launches
    hipError_t e = hipMemsetAsync(ctx->d_status, 0, 32, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_queue, ctx->stream);
    for (dryv_recon_ctx::Lane& l : ctx->lanes)
      if (e == hipSuccess) e = hipStreamWaitEvent(l.stream, ctx->ev_queue, 0);
    if (e != hipSuccess) return fail(ctx, e, "queue lanes");
    // as with one lane, the queue's first batch sizes the workspaces: every lane's, now, while none of them is running
    for (int k = 1; k < ctx->queue_lanes; k++) {
      LaneScope scope(ctx, k);
      if ((st = ensure(ctx, &ctx->d_work, &ctx->cap_work, workspace_bytes(P))) != DRYV_OK) return st;
    }
  }
  {
    LaneScope scope(ctx, lane);
    // a lane's workspace is shared by the launches queued on it (each behind its predecessor); it cannot be re-allocated
    // while one of them may still be running
    if (ctx->in_flight && workspace_bytes(P) > ctx->cap_work) return DRYV_E_STATE;
    ctx->lane_launch = ctx->queue_lanes > 1;
    // (one lane: the status word is cleared in front of the first batch and accumulates over the queue)
    st = launch(ctx, P, d_mbs, d_coeffs, d_yuv_out, first && ctx->queue_lanes == 1);
    ctx->lane_launch = false;
    if (st != DRYV_OK) return st;
  }
  ctx->queued.push_back(dryv_recon_ctx::Queued{P, d_mbs, d_coeffs, d_yuv_out});
  ctx->in_flight = true;
  ctx->in_flight_host = false;
  return DRYV_OK;
}

int dryv_recon_wide_rerun_stats(dryv_recon_ctx* ctx, int* events, int* batches) {
  if (!ctx) return DRYV_E_INVALID;
  if (events) *events = ctx->wide_reruns;
  if (batches) *batches = ctx->wide_rerun_batches;
  return DRYV_OK;
}

int dryv_recon_set_queue_lanes(dryv_recon_ctx* ctx, int lanes) {
  if (!ctx || lanes < 1 || lanes > 4) return DRYV_E_INVALID;
  if (ctx->in_flight) return DRYV_E_STATE;
  (void)hipSetDevice(ctx->device);
  hipError_t e;
  if (!ctx->ev_queue && (e = hipEventCreateWithFlags(&ctx->ev_queue, hipEventDisableTiming)) != hipSuccess) return fail(ctx, e, "hipEventCreate");
  while ((int)ctx->lanes.size() < lanes - 1) {
    dryv_recon_ctx::Lane l;
    if ((e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking)) != hipSuccess) return fail(ctx, e, "hipStreamCreate(queue lane)");
    ctx->lanes.push_back(l);
  }
  ctx->queue_lanes = lanes;   // (lanes beyond it keep their stream and workspace: unused until asked for again)
  return DRYV_OK;
}

int dryv_recon_kernel_ms_stats(dryv_recon_ctx* ctx, uint32_t n_last, float* avg_ms, float* min_ms, float* max_ms) {
  if (!ctx || n_last == 0) return DRYV_E_INVALID;
  if (!ctx->timed || ctx->in_flight) return DRYV_E_STATE;   // after wait / sync
  const unsigned n = std::min(std::min((unsigned)n_last, ctx->ev_n), dryv_recon_ctx::kEvRing);
  if (n == 0) return DRYV_E_STATE;
  double sum = 0;
  float mn = 0, mx = 0;
  for (unsigned k = 0; k < n; k++) {
    const unsigned idx = (ctx->ev_n - 1 - k) % dryv_recon_ctx::kEvRing;
    float ms = 0;
    hipError_t e = hipEventSynchronize(ctx->ev_ring[idx][1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev_ring[idx][0], ctx->ev_ring[idx][1]);
    if (e != hipSuccess) return fail(ctx, e, "hipEventElapsedTime");
    sum += ms;
    mn = k == 0 ? ms : std::min(mn, ms);
    mx = k == 0 ? ms : std::max(mx, ms);
  }
  if (avg_ms) *avg_ms = (float)(sum / n);
  if (min_ms) *min_ms = mn;
  if (max_ms) *max_ms = mx;
  return DRYV_OK;
}

int dryv_recon_sync(dryv_recon_ctx* ctx) {
  if (!ctx) return DRYV_E_INVALID;
  (void)hipSetDevice(ctx->device);
  if (!ctx->in_flight) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, e, "hipStreamSynchronize");
    if (ctx->deblock_pending) {
      ctx->deblock_pending = false;
      if (*ctx->h_status & 4u) {
        char msg[160];
        snprintf(msg, sizeof msg, "deblock kernel: band task %u gave up waiting at step %u for the band above (saw %u macroblocks)",
                 ctx->h_status[1], ctx->h_status[2] >> 16, ctx->h_status[3]);
        ctx->last_error = msg;
        return DRYV_E_DEVICE;
      }
    }
    return DRYV_OK;
  }
  const int st = finish(ctx);
  ctx->queued.clear();
  if (!ctx->in_flight_host) ctx->in_flight = false;
  if (st != DRYV_OK) return st;
  return (*ctx->h_status & 1u) ? DRYV_E_UNSUPPORTED : DRYV_OK;
}

int dryv_recon_last_kernel_ms(dryv_recon_ctx* ctx, float* ms) {
  if (!ctx || !ms) return DRYV_E_INVALID;
  if (!ctx->timed) return DRYV_E_STATE;
  hipError_t e = hipEventSynchronize(ctx->ev_stop);
  if (e != hipSuccess) return fail(ctx, e, "hipEventSynchronize");
  e = hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop);
  if (e != hipSuccess) return fail(ctx, e, "hipEventElapsedTime");
  return DRYV_OK;
}

void* dryv_recon_stream(dryv_recon_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

const char* dryv_recon_strerror(int status) {
  switch (status) {
    case DRYV_OK: return "ok";
    case DRYV_E_INVALID: return "invalid argument";
    case DRYV_E_UNSUPPORTED: return "outside the supported domain (intra 4:2:0 8-bit, mb_kind 0..2, qp 0..51)";
    case DRYV_E_DEVICE: return "HIP runtime error";
    case DRYV_E_NOMEM: return "out of memory";
    case DRYV_E_STATE: return "call out of order";
    case DRYV_E_NODEVICE: return "no HIP device (this library has no CPU path)";
    default: return "unknown status";
  }
}

const char* dryv_recon_last_device_error(dryv_recon_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

#if defined(DRYV_BAND_TIMELINE)
/* diagnostic build only: per band task of the last launch 4 x u64 (claim, BACK's first step, BACK's last step: 100 MHz
   ticks; wave index) */
int dryv_recon_debug_band_timeline(dryv_recon_ctx* ctx, int n_tasks, unsigned long long* out) {
  if (!ctx || !out || n_tasks > 65536 || !ctx->d_work) return DRYV_E_INVALID;
  hipError_t e = hipMemcpy(out, (unsigned char*)ctx->d_work + dryv::band_profile_offset(ctx->last_P) + (size_t)65536 * (16 * 8 + 32),
                           (size_t)n_tasks * 32, hipMemcpyDeviceToHost);
  return e == hipSuccess ? DRYV_OK : DRYV_E_DEVICE;
}
#endif
#if defined(DRYV_BAND_PROFILE) || defined(DRYV_BAND_TRACE)
/* diagnostic build only: reads n_waves x 8 trace words and `n_prog` (mode-record) progress words while the kernel may still be running
   (own stream) */
int dryv_recon_debug_band_trace(dryv_recon_ctx* ctx, int n_waves, unsigned* out, int n_prog, unsigned* prog_out) {
  if (!ctx || !out || n_waves > 65536 || !ctx->d_work) return DRYV_E_INVALID;
  static hipStream_t s2 = nullptr;
  if (!s2 && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return DRYV_E_DEVICE;
  hipError_t e = hipMemcpyAsync(out, (unsigned char*)ctx->d_work + dryv::band_profile_offset(ctx->last_P) + (size_t)65536 * 16 * 8,
                                (size_t)n_waves * 32, hipMemcpyDeviceToHost, s2);
  if (e == hipSuccess && n_prog > 0) e = hipMemcpyAsync(prog_out, (unsigned char*)ctx->d_work + 256, (size_t)n_prog * 4, hipMemcpyDeviceToHost, s2);  /* the bands' mode-record words */
  if (e == hipSuccess) e = hipStreamSynchronize(s2);
  return e == hipSuccess ? DRYV_OK : DRYV_E_DEVICE;
}
/* diagnostic build only: copies the per-wave phase cycle sums of the last band-kernel launch (n_waves x 16 u64) */
int dryv_recon_debug_band_phases(dryv_recon_ctx* ctx, int n_waves, unsigned long long* out) {
  if (!ctx || !out || n_waves > 65536 || !ctx->d_work) return DRYV_E_INVALID;
  hipError_t e = hipMemcpy(out, (unsigned char*)ctx->d_work + dryv::band_profile_offset(ctx->last_P), (size_t)n_waves * 16 * 8,
                           hipMemcpyDeviceToHost);
  return e == hipSuccess ? DRYV_OK : DRYV_E_DEVICE;
}
#endif

/* math.rs:109-117 */
int64_t dryv_math_clamp(int64_t value, int64_t min, int64_t max) {
  if (value < min) return min;
  if (value > max) return max;
  return value;
}

/* math.rs:119-125 */
int64_t dryv_math_inverse_raster_scan(int64_t a, int64_t b, int64_t c, int64_t d, int64_t e) {
  if (e == 0) return (a % (d / b)) * b;
  return (a / (d / b)) * c;
}

}  // extern "C"
