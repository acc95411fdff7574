/*
 * dryv_recon.h — C ABI of the MI355X macroblock-reconstruction backend for dryv's AVC decode path.
 *
 * This is the drop-in boundary for the reference's reconstruction layer
 * (/root/reference/src/video/frame/{mod,transform,pred4x4,pred8x8,pred16x16,trans_chroma}.rs and
 * src/math.rs:109-125). The reference has no FFI of its own; the seam is the three places a `Frame`
 * is used:
 *
 *   Frame::new(&slice)               src/video/decoder.rs:124   -> dryv_recon_create + frame params
 *   frame.decode(slice)   (per MB)   src/video/cabac/mod.rs:208 -> host appends one dryv_mb_desc +
 *                                                                   384 coefficients to the batch
 *   (after slice.data() returns)     src/video/decoder.rs:125   -> dryv_recon_submit
 *   frame.write_to_yuv_file(path)    src/video/decoder.rs:142   -> dryv_recon_wait (same byte order)
 *
 * CABAC parsing, NAL/atom demux and src/byte stay on the host and are untouched. All arithmetic on
 * this path is integer; results are bit-exact with the reference's CPU path (see oracle/).
 *
 * Plain C, fixed-width types, caller-owned buffers, status codes (never aborts / throws across the
 * boundary: the reference's todo!()/panic!() domain is reported as DRYV_E_UNSUPPORTED).
 * A context is NOT thread-safe (the reference is single-threaded); use one context per GPU.
 */
#ifndef DRYV_RECON_H
#define DRYV_RECON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRYV_RECON_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------------- */
enum {
  DRYV_OK = 0,
  DRYV_E_INVALID = -1,     /* NULL pointer, zero-sized picture, buffer too small ...               */
  DRYV_E_UNSUPPORTED = -2, /* outside the reference's implemented domain (frame/mod.rs:86,88,
                              trans_chroma.rs:16,20: PCM, inter, 4:0:0/4:2:2/4:4:4, bit depth != 8),
                              or a macroblock record with mb_kind > 2 / qp > 51 / mode out of range */
  DRYV_E_DEVICE = -3,      /* HIP runtime error (see dryv_recon_last_device_error)                 */
  DRYV_E_NOMEM = -4,       /* host or device allocation failed                                     */
  DRYV_E_STATE = -5,       /* wait without submit, submit while a batch is in flight               */
  DRYV_E_NODEVICE = -6     /* no gfx950 device / HIP runtime unavailable                           */
};

/* ---- per-frame parameter block ------------------------------------------------------------ */
/* The slice/PPS/SPS-derived fields the reconstruction path reads ("frame-level
 * inputs"): slice/mod.rs:125-138, slice/header.rs:125-141,168-183,309-313, atom/avcc/pps.rs:22,24,65.
 * One block describes every frame of a submit. 496 bytes; also the payload rank 0 broadcasts when
 * frames are sharded over GPUs. */
typedef struct dryv_frame_params {
  uint16_t pic_width_in_mbs;               /* PicWidthInMbs   (1..1024)                           */
  uint16_t pic_height_in_mbs;              /* PicHeightInMbs, frame_mbs_only                      */
  uint8_t chroma_array_type;               /* must be 1 (4:2:0)                                   */
  uint8_t bit_depth_y;                     /* must be 8                                           */
  uint8_t bit_depth_c;                     /* must be 8                                           */
  int8_t chroma_qp_index_offset;           /* pps.chroma_qp_index_offset (Cb), -12..12            */
  int8_t second_chroma_qp_index_offset;    /* Cr; host passes the first one when the PPS has no
                                              extra_rbsp_data (transform.rs:198-203)              */
  uint8_t constrained_intra_pred_flag;     /* only affects inter neighbours: no effect for I-only */
  uint8_t transform_8x8_mode_flag;         /* informational: whether mb_kind 1 may occur          */
  uint8_t reserved[5];                     /* zero                                                */
  uint8_t scaling_list4x4[6][16];          /* zig-zag order, flat 16 by default (header.rs:330)   */
  uint8_t scaling_list8x8[6][64];          /* zig-zag order; the path only reads list 0 of each   */
} dryv_frame_params;

/* ---- per-macroblock record ---------------------------------------------------------------- */
/* The fields of `Macroblock` (slice/macroblock.rs:21-129) that Frame::decode reads, 16 bytes, in
 * macroblock-address (raster) order, frames concatenated. */
typedef struct dryv_mb_desc {
  uint8_t mb_kind;                /* 0 Intra4x4, 1 Intra8x8, 2 Intra16x16 (frame/mod.rs:73-84)     */
  uint8_t i16_pred_mode;          /* Intra16x16PredMode 0 V,1 H,2 DC,3 Plane (macroblock.rs:584)   */
  uint8_t intra_chroma_pred_mode; /* 0 DC, 1 Horizontal, 2 Vertical, 3 Plane (trans_chroma.rs:166) */
  uint8_t qp;                     /* QPY == QP'Y for 8-bit video, 0..51 (cabac/mod.rs:186-191)     */
  uint16_t prev_flags;            /* bit b = prev_intra{4x4,8x8}_pred_mode_flag[b]                 */
  uint8_t rem_modes[8];           /* rem_intra{4x4,8x8}_pred_mode[i] = (rem_modes[i>>1] >> 4*(i&1)) & 7 */
  uint16_t nz_mask;               /* reserved for coded-block hints; ignored (write 0xFFFF)        */
} dryv_mb_desc;

/* ---- coefficients --------------------------------------------------------------------------
 * int16_t[384] per macroblock, the reference's list order (macroblock.rs:100-119, zig-zag lists,
 * zero when coded_block_flag == 0):
 *   mb_kind 0: block_luma_4x4[0][blk 0..15][16]
 *   mb_kind 1: block_luma_8x8[0][blk8 0..3][64]
 *   mb_kind 2: block_luma_dc[0][16], then block_luma_ac[0][blk 0..15][15]
 *   then block_chroma_dc[0][0..3], block_chroma_ac[0][blk 0..3][15]   (Cb: 4 + 60)
 *   then block_chroma_dc[1][0..3], block_chroma_ac[1][blk 0..3][15]   (Cr: 4 + 60)
 */
#define DRYV_COEFFS_PER_MB 384

/* ---- output --------------------------------------------------------------------------------
 * Per frame exactly Frame::write_to_yuv_file's byte order (frame/mod.rs:48-70): Y plane
 * (16*H rows of 16*W bytes, row-major), then Cb (8*H rows of 8*W), then Cr; full coded size, no
 * cropping; frames concatenated. */
size_t dryv_recon_frame_bytes(const dryv_frame_params *fp); /* 384 * W * H, 0 if fp invalid */

/* Validates a parameter block / batch size without touching a device: DRYV_OK, DRYV_E_UNSUPPORTED (outside the
 * reference's domain: see above) or DRYV_E_INVALID (zero-sized picture, width > 1024 macroblocks, a frame whose
 * coefficients exceed 4 GB, a batch of 2^31 macroblocks or more, n_frames == 0). submit applies the same checks. */
int dryv_recon_check_params(const dryv_frame_params *fp, uint32_t n_frames);

typedef struct dryv_recon_ctx dryv_recon_ctx;

/* Opens HIP device `device_ordinal`, creates the stream, events and status words. Fails with
 * DRYV_E_NODEVICE when no GPU is visible: there is no CPU fallback in this library. */
int dryv_recon_create(dryv_recon_ctx **out, int device_ordinal);
void dryv_recon_destroy(dryv_recon_ctx *ctx);

/* Host-buffer path (what the Rust shim calls). Copies params/records/coefficients to the device
 * and launches reconstruction asynchronously; inputs must stay valid until dryv_recon_wait returns.
 * n_mbs total = n_frames * W * H. */
int dryv_recon_submit(dryv_recon_ctx *ctx, const dryv_frame_params *fp, uint32_t n_frames,
                      const dryv_mb_desc *mbs, const int16_t *coeffs);
/* Blocks until the batch is done, copies n_frames * dryv_recon_frame_bytes() bytes into yuv_out.
 * Returns DRYV_E_UNSUPPORTED if any macroblock record was outside the supported domain (the
 * affected macroblocks are then left zero-filled, like an undecoded Frame). */
int dryv_recon_wait(dryv_recon_ctx *ctx, uint8_t *yuv_out, size_t yuv_out_bytes);

/* Pipelined host-buffer path (what a dryv process that decodes many pictures should call): the batch is cut into
 * chunks of frames, and copy-in of chunk k+1, reconstruction of chunk k and copy-out of chunk k-1 overlap on three
 * streams; the planes land directly in yuv_out (n_frames * dryv_recon_frame_bytes() bytes). Asynchronous: pair with
 * dryv_recon_sync; all four buffers must stay valid and untouched until it returns. The copies only run at PCIe speed,
 * and only overlap, from page-locked memory: allocate the buffers with dryv_recon_alloc_host (or hipHostRegister
 * them); pageable buffers work but are staged by the driver. */
int dryv_recon_submit_host(dryv_recon_ctx *ctx, const dryv_frame_params *fp, uint32_t n_frames,
                           const dryv_mb_desc *mbs, const int16_t *coeffs, uint8_t *yuv_out, size_t yuv_out_bytes);
void *dryv_recon_alloc_host(size_t bytes); /* page-locked host memory (hipHostMalloc); NULL on failure */
void dryv_recon_free_host(void *p);

/* Device-resident path: all three buffers already live in this device's HBM (device pointers).
 * Nothing is copied; the planes are written straight into d_yuv_out. Asynchronous on the
 * context's stream; pair with dryv_recon_sync. */
int dryv_recon_submit_device(dryv_recon_ctx *ctx, const dryv_frame_params *fp, uint32_t n_frames,
                             const void *d_mbs, const void *d_coeffs, void *d_yuv_out);
int dryv_recon_sync(dryv_recon_ctx *ctx); /* waits, then reports the batch's status word */

/* The same, queued: may be called again before dryv_recon_sync, any number of times, each call with buffers of its own;
 * the batches run back to back on the context's stream with no host round trip between them (a decoder that fills
 * batch k+1 while batch k reconstructs: the reference's per-picture loop, video/decoder.rs:124-143, made asynchronous).
 * dryv_recon_sync then waits for all of them and reports the OR of their status words; every queued batch's inputs
 * must stay valid until it returns. Only behind queued batches: DRYV_E_STATE if anything else is in flight, or if this
 * batch would need a larger workspace than the queue is running on (sync first). */
int dryv_recon_submit_device_queued(dryv_recon_ctx *ctx, const dryv_frame_params *fp, uint32_t n_frames,
                                    const void *d_mbs, const void *d_coeffs, void *d_yuv_out);

/* Queue lanes (1 .. 4, default 1; not while anything is in flight). With n > 1 the batches of
 * dryv_recon_submit_device_queued rotate over n streams of the context, each with a workspace of its own, and every launch
 * takes half of the resident grid: two launches then run side by side, half a launch apart, and the ramp and the drain of
 * one -- a sixth of a 300-picture launch's time, DESIGN.md section 4.4 -- run beside the steady state of the other. Three
 * lanes: 1.05 -> 0.90 ms per 300 x 1080p batch. Results and status words are the same. What changes for the caller: the
 * queue's inputs must be complete, or enqueued on dryv_recon_stream(), BEFORE the first queued submit (the other lanes
 * start behind that point, not behind later work on that stream), and batches may finish out of order -- outputs are valid
 * after dryv_recon_sync, as before. As with one lane, a queue's first batch sizes the workspaces (every lane's). A launch's
 * own duration (dryv_recon_kernel_ms_stats) is that of half the chip. */
int dryv_recon_set_queue_lanes(dryv_recon_ctx *ctx, int lanes);

/* ---- output stage: cropping and NV12 packing on the device ------------------------------------------
 * The reference parses frame_crop_*_offset (sps.rs:252-267) but writes the full coded planes (frame/mod.rs:48-70;
 * README.md:13 unchecked), which stays this library's default output. A caller that wants display-size pictures, or
 * NV12, describes the output here; only those bytes are then produced and copied. Crop values are luma samples (twice
 * the SPS's frame_crop_*_offset for 4:2:0 frame pictures) and must be even. */
enum { DRYV_OUT_I420 = 0, /* Y, Cb, Cr planes (write_to_yuv_file order) */ DRYV_OUT_NV12 = 1 /* Y, interleaved CbCr */ };
typedef struct {
  uint8_t format; /* DRYV_OUT_* */
  uint8_t reserved[3];
  uint16_t crop_left, crop_right, crop_top, crop_bottom;
} dryv_output_desc;
/* Bytes of one output picture; 0 if the description is invalid for these parameters. */
size_t dryv_recon_output_bytes(const dryv_frame_params *fp, const dryv_output_desc *od);
/* Packs n_frames full planar pictures at d_yuv (as written by dryv_recon_submit_device, after dryv_recon_sync) into
 * d_out (n_frames * dryv_recon_output_bytes()). Asynchronous on the context's stream; pair with dryv_recon_sync.
 * DRYV_E_STATE while a batch is in flight. */
int dryv_recon_pack_device(dryv_recon_ctx *ctx, const dryv_frame_params *fp, uint32_t n_frames, const void *d_yuv,
                           const dryv_output_desc *od, void *d_out);
/* dryv_recon_wait with an output description: blocks until the batch submitted with dryv_recon_submit is done, packs
 * it on the device and copies n_frames * dryv_recon_output_bytes() bytes into out. */
int dryv_recon_wait_packed(dryv_recon_ctx *ctx, const dryv_output_desc *od, uint8_t *out, size_t out_bytes);

/* ---- in-loop deblocking filter ---------------------------------------------------------------------------
 * ITU-T H.264 clause 8.7 for this library's domain (frame macroblocks, 4:2:0, 8 bit, one slice per picture, all macroblocks
 * intra). dryv parses the syntax elements below (slice/header.rs:609-640) and does not filter (README.md:15 unchecked): there
 * is no reference behaviour to be equal to, and reconstruction's default output stays unfiltered, as dryv's is. */
typedef struct {
  uint8_t disable_deblocking_filter_idc; /* 0 filter every edge, 1 none, 2 = 0 for one slice per picture */
  int8_t slice_alpha_c0_offset_div2;     /* -6..6 */
  int8_t slice_beta_offset_div2;         /* -6..6 */
  uint8_t reserved;
} dryv_deblock_params;
/* Filters n_frames reconstructed pictures at d_yuv in place; d_mbs: the batch's records in device memory (qp and kind are
 * read). Asynchronous on the context's stream; pair with dryv_recon_sync. DRYV_E_STATE while a batch is in flight. */
int dryv_recon_deblock_device(dryv_recon_ctx *ctx, const dryv_frame_params *fp, const dryv_deblock_params *dp,
                              uint32_t n_frames, const void *d_mbs, void *d_yuv);

/* dryv_recon_wait with the stages behind reconstruction: blocks until the batch submitted with dryv_recon_submit is done,
 * then deblocks it (dp; NULL: not), crops / packs it (od; NULL: the full planar pictures) and copies the result into out. */
int dryv_recon_wait_filtered(dryv_recon_ctx *ctx, const dryv_deblock_params *dp, const dryv_output_desc *od, uint8_t *out,
                             size_t out_bytes);

/* Device time of the most recent reconstruction kernel launch, from a pair of HIP events on the context's own stream that
 * take the dispatch's own start and end (hipExtLaunchKernelGGL: what rocprofv3 --kernel-trace reports for the kernel).
 * Valid after wait/sync. */
int dryv_recon_last_kernel_ms(dryv_recon_ctx *ctx, float *ms);
/* Average / minimum / maximum device time of the n_last most recent reconstruction kernel launches (at most 64 are
 * remembered; every launch has an event pair of its own, so queued launches are timed one by one). Any of the
 * three pointers may be NULL. Valid after wait/sync. */
int dryv_recon_kernel_ms_stats(dryv_recon_ctx *ctx, uint32_t n_last, float *avg_ms, float *min_ms, float *max_ms);
/* Diagnostics of the 64-bit fallback: how many times a sync / wait found a batch flagged by the fast kernel build (a
 * block beyond its 32-bit arithmetic: no conformant stream has one), and how many batches were launched again with the
 * wide build because of it. A queue of batches is re-run from the first flagged batch on, not from its head. Either
 * pointer may be NULL. */
int dryv_recon_wide_rerun_stats(dryv_recon_ctx *ctx, int *events, int *batches);

/* Raw HIP stream handle (hipStream_t) the context launches on, for callers that want to order
 * their own work or record their own events against it. */
void *dryv_recon_stream(dryv_recon_ctx *ctx);

const char *dryv_recon_strerror(int status);
const char *dryv_recon_last_device_error(dryv_recon_ctx *ctx);
int dryv_recon_abi_version(void);

/* math.rs:109-125 — the two helpers the reference's reconstruction code calls everywhere. Inside
 * the kernels they are inlined; they are exported so a host shim can keep calling them. */
int64_t dryv_math_clamp(int64_t value, int64_t min, int64_t max);
int64_t dryv_math_inverse_raster_scan(int64_t a, int64_t b, int64_t c, int64_t d, int64_t e);

#ifdef __cplusplus
}
#endif
#endif /* DRYV_RECON_H */
