// output_pack.h — geometry of the output stage (output_pack.hip) and its host-side validation, shared with the CPU tests.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"

namespace dryv {

struct PackGeo {
  unsigned sw, sh;                  // source planes: coded luma width / height (16 * macroblocks)
  unsigned ow, oh;                  // output luma width / height
  unsigned crop_left, crop_top;     // luma samples, even
  unsigned nv12;                    // 0: I420 (Y, Cb, Cr planes), 1: NV12 (Y, interleaved CbCr)
  unsigned n_frames;
  size_t src_frame_bytes, dst_frame_bytes;
};

// DRYV_OK and *G filled, DRYV_E_UNSUPPORTED for a picture format outside the library's domain, DRYV_E_INVALID for an odd
// or oversized crop or an unknown output format.
inline int pack_geometry(const dryv_frame_params* fp, const dryv_output_desc* od, uint32_t n_frames, PackGeo* G) {
  if (!fp || !od || !G) return DRYV_E_INVALID;
  if (fp->pic_width_in_mbs == 0 || fp->pic_height_in_mbs == 0) return DRYV_E_INVALID;
  if (fp->chroma_array_type != 1 || fp->bit_depth_y != 8 || fp->bit_depth_c != 8) return DRYV_E_UNSUPPORTED;
  if (od->format != DRYV_OUT_I420 && od->format != DRYV_OUT_NV12) return DRYV_E_INVALID;
  const unsigned sw = 16u * fp->pic_width_in_mbs, sh = 16u * fp->pic_height_in_mbs;
  // frame_crop_*_offset counts CropUnitX = CropUnitY = 2 luma samples for 4:2:0 frame pictures (7.4.2.1.1): even numbers here
  if ((od->crop_left | od->crop_right | od->crop_top | od->crop_bottom) & 1u) return DRYV_E_INVALID;
  if ((unsigned)od->crop_left + od->crop_right >= sw || (unsigned)od->crop_top + od->crop_bottom >= sh) return DRYV_E_INVALID;
  G->sw = sw; G->sh = sh;
  G->ow = sw - od->crop_left - od->crop_right;
  G->oh = sh - od->crop_top - od->crop_bottom;
  G->crop_left = od->crop_left; G->crop_top = od->crop_top;
  G->nv12 = od->format == DRYV_OUT_NV12;
  G->n_frames = n_frames;
  G->src_frame_bytes = (size_t)sw * sh * 3 / 2;
  G->dst_frame_bytes = (size_t)G->ow * G->oh * 3 / 2;
  return DRYV_OK;
}

// output_pack.hip
#ifdef __HIPCC__
hipError_t pack_launch(const PackGeo& G, const void* d_src, void* d_dst, int num_cus, hipStream_t stream);
#endif

}  // namespace dryv
