"""Host-side mirror of the reference's reconstruction interface, over the C ABI.

The reference exposes reconstruction as three inherent methods on `Frame`
(/root/reference/src/video/frame/mod.rs:16-90):

    Frame::new(&slice)              -> Frame.new(frame_params)
    frame.decode(&mut slice)        -> frame.decode(mb, coeffs)      once per macroblock, in mbaddr order
    frame.write_to_yuv_file(path)   -> frame.write_to_yuv_file(path) same byte order (mod.rs:48-70)

`decode` is called from inside the CABAC macroblock loop (cabac/mod.rs:208). Nothing the parser
does later depends on reconstructed samples, so here `decode` only appends the macroblock's record
and coefficients to the frame's batch; the first call that needs pixels (`planes()`,
`write_to_yuv_file`) submits the whole frame to the GPU in one call. All computation happens in
libdryv_recon.so (HIP); this module moves bytes and mirrors names and error behaviour.
"""
import ctypes as C

import numpy as np

from . import abi


class ReconError(RuntimeError):
    """A non-zero status from the C ABI. Mirrors the reference's todo!()/panic!() domain
    (frame/mod.rs:86,88) as an exception instead of an abort."""

    def __init__(self, status, detail=""):
        self.status = status
        msg = "dryv_recon: %s (%d)" % (abi.strerror(status), status)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


def _check(status, ctx=None):
    if status != abi.DRYV_OK:
        detail = ""
        if ctx is not None and status == abi.DRYV_E_DEVICE:
            detail = abi.load_library().dryv_recon_last_device_error(ctx).decode()
        raise ReconError(status, detail)


class ReconContext:
    """One dryv_recon_ctx: a device, its stream and staging buffers. Not thread-safe."""

    def __init__(self, device_ordinal=0):
        self._lib = abi.load_library()
        h = C.c_void_p()
        _check(self._lib.dryv_recon_create(C.byref(h), int(device_ordinal)))
        self._h = h
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dryv_recon_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host-buffer path -------------------------------------------------------------------
    def submit(self, fp, n_frames, mbs, coeffs):
        """mbs: array of abi.MB_DESC_DTYPE (n_frames*W*H), coeffs: int16 (n_mbs, 384)."""
        mbs = np.ascontiguousarray(mbs, dtype=abi.MB_DESC_DTYPE)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
        n_mbs = int(n_frames) * fp.pic_width_in_mbs * fp.pic_height_in_mbs
        if mbs.size != n_mbs or coeffs.size != n_mbs * abi.COEFFS_PER_MB:
            raise ReconError(abi.DRYV_E_INVALID, "batch size does not match n_frames * W * H")
        self._keep = (mbs, coeffs, fp)  # inputs must stay valid until wait()
        self._pending = n_mbs * 384
        _check(self._lib.dryv_recon_submit(self._h, C.byref(fp), int(n_frames), mbs.ctypes.data,
                                           coeffs.ctypes.data), self._h)

    def wait(self, allow_unsupported=False):
        out = np.empty(self._pending, dtype=np.uint8)
        st = self._lib.dryv_recon_wait(self._h, out.ctypes.data, out.size)
        self._keep = None
        if not (allow_unsupported and st == abi.DRYV_E_UNSUPPORTED):
            _check(st, self._h)
        self.last_status = st
        return out

    def wait_packed(self, od, allow_unsupported=False):
        """dryv_recon_wait_packed: the batch submitted with submit(), cropped / packed on the device as `od`
        (abi.make_output_desc) says; only those bytes are copied back."""
        mbs, coeffs, fp = self._keep
        per = self._lib.dryv_recon_output_bytes(C.byref(fp), C.byref(od))
        if per == 0:
            raise ReconError(abi.DRYV_E_INVALID, "invalid output description")
        n_frames = mbs.size // (fp.pic_width_in_mbs * fp.pic_height_in_mbs)
        out = np.empty(per * n_frames, dtype=np.uint8)
        st = self._lib.dryv_recon_wait_packed(self._h, C.byref(od), out.ctypes.data, out.size)
        self._keep = None
        if not (allow_unsupported and st == abi.DRYV_E_UNSUPPORTED):
            _check(st, self._h)
        self.last_status = st
        return out

    def wait_filtered(self, dp=None, od=None):
        """dryv_recon_wait_filtered: the batch submitted with submit(), deblocked (dp) and / or cropped / packed (od)."""
        mbs, coeffs, fp = self._keep
        n_frames = mbs.size // (fp.pic_width_in_mbs * fp.pic_height_in_mbs)
        per = self._lib.dryv_recon_output_bytes(C.byref(fp), C.byref(od)) if od is not None else self._lib.dryv_recon_frame_bytes(C.byref(fp))
        out = np.empty(per * n_frames, dtype=np.uint8)
        st = self._lib.dryv_recon_wait_filtered(self._h, C.byref(dp) if dp is not None else None,
                                                C.byref(od) if od is not None else None, out.ctypes.data, out.size)
        self._keep = None
        _check(st, self._h)
        self.last_status = st
        return out

    def pack_device(self, fp, n_frames, d_yuv, od, d_out):
        """dryv_recon_pack_device on raw device pointers; completes at sync()."""
        _check(self._lib.dryv_recon_pack_device(self._h, C.byref(fp), int(n_frames), C.c_void_p(d_yuv), C.byref(od),
                                                C.c_void_p(d_out)), self._h)

    def deblock_device(self, fp, dp, n_frames, d_mbs, d_yuv):
        """dryv_recon_deblock_device: H.264 8.7 on reconstructed pictures in device memory, in place; completes at sync()."""
        _check(self._lib.dryv_recon_deblock_device(self._h, C.byref(fp), C.byref(dp), int(n_frames), C.c_void_p(d_mbs),
                                                   C.c_void_p(d_yuv)), self._h)

    def reconstruct(self, fp, n_frames, mbs, coeffs, allow_unsupported=False):
        self.submit(fp, n_frames, mbs, coeffs)
        return self.wait(allow_unsupported=allow_unsupported)

    # ---- device-resident path ---------------------------------------------------------------
    def alloc_host(self, shape, dtype):
        """A page-locked numpy array (dryv_recon_alloc_host): the memory the pipelined host path copies at PCIe speed
        from / to. Free with free_host(array) (or leave it to process exit)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self._lib.dryv_recon_alloc_host(max(n, 1))
        if not p:
            raise ReconError(abi.DRYV_E_NOMEM)
        buf = (C.c_uint8 * n).from_address(p)
        arr = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def free_host(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self._lib.dryv_recon_free_host(p)

    def submit_host(self, fp, n_frames, mbs, coeffs, yuv_out):
        """dryv_recon_submit_host: chunked, three-stream pipelined host path; completes at sync()."""
        mbs = np.ascontiguousarray(mbs)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
        self._keep = (mbs, coeffs, yuv_out)
        _check(self._lib.dryv_recon_submit_host(self._h, C.byref(fp), int(n_frames), mbs.ctypes.data, coeffs.ctypes.data,
                                                yuv_out.ctypes.data, yuv_out.nbytes), self._h)

    def submit_device(self, fp, n_frames, d_mbs, d_coeffs, d_yuv_out):
        """All three are raw device pointers (ints), e.g. torch_tensor.data_ptr()."""
        _check(self._lib.dryv_recon_submit_device(self._h, C.byref(fp), int(n_frames), C.c_void_p(d_mbs),
                                                  C.c_void_p(d_coeffs), C.c_void_p(d_yuv_out)), self._h)

    def submit_device_queued(self, fp, n_frames, d_mbs, d_coeffs, d_yuv_out):
        """dryv_recon_submit_device_queued: like submit_device, but may be called again before sync(); the batches run back
        to back on the stream."""
        _check(self._lib.dryv_recon_submit_device_queued(self._h, C.byref(fp), int(n_frames), C.c_void_p(d_mbs),
                                                         C.c_void_p(d_coeffs), C.c_void_p(d_yuv_out)), self._h)

    def kernel_ms_stats(self, n_last):
        """(average, minimum, maximum) device time in ms of the n_last most recent reconstruction launches (<= 64)."""
        a, lo, hi = C.c_float(), C.c_float(), C.c_float()
        _check(self._lib.dryv_recon_kernel_ms_stats(self._h, int(n_last), C.byref(a), C.byref(lo), C.byref(hi)), self._h)
        return float(a.value), float(lo.value), float(hi.value)

    def set_queue_lanes(self, lanes):
        """dryv_recon_set_queue_lanes: queued device batches rotate over `lanes` streams with half-size grids (see the header)."""
        _check(self._lib.dryv_recon_set_queue_lanes(self._h, int(lanes)), self._h)

    def wide_rerun_stats(self):
        """(events, batches): syncs that found a batch flagged for the 64-bit build, and batches launched again because of it."""
        a, b = C.c_int(), C.c_int()
        _check(self._lib.dryv_recon_wide_rerun_stats(self._h, C.byref(a), C.byref(b)), self._h)
        return int(a.value), int(b.value)

    def sync(self, allow_unsupported=False):
        st = self._lib.dryv_recon_sync(self._h)
        if not (allow_unsupported and st == abi.DRYV_E_UNSUPPORTED):
            _check(st, self._h)
        return st

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(self._lib.dryv_recon_last_kernel_ms(self._h, C.byref(ms)), self._h)
        return float(ms.value)

    @property
    def stream(self):
        return self._lib.dryv_recon_stream(self._h)


class Frame:
    """Mirror of the reference's `Frame` for one picture (one slice NAL, decoder.rs:124)."""

    def __init__(self, fp, ctx=None):
        self.fp = fp
        self.width_l = fp.pic_width_in_mbs * 16   # frame/mod.rs:30-33
        self.height_l = fp.pic_height_in_mbs * 16
        self.width_c = fp.pic_width_in_mbs * 8
        self.height_c = fp.pic_height_in_mbs * 8
        self._n = fp.pic_width_in_mbs * fp.pic_height_in_mbs
        self._mbs = np.zeros(self._n, dtype=abi.MB_DESC_DTYPE)
        self._coeffs = np.zeros((self._n, abi.COEFFS_PER_MB), dtype=np.int16)
        self._next = 0
        self._ctx = ctx
        self._yuv = None

    @classmethod
    def new(cls, fp, ctx=None):
        """Frame::new(&slice) (frame/mod.rs:29-46)."""
        return cls(fp, ctx)

    def decode(self, mb, coeffs):
        """Frame::decode(&mut slice) (frame/mod.rs:72-90) for the macroblock at the next mbaddr.

        mb: mapping / record with the dryv_mb_desc fields; coeffs: 384 int16 in the reference's
        list order (include/dryv_recon.h). I_PCM and inter macroblocks are todo!() in the
        reference; here they raise ReconError(DRYV_E_UNSUPPORTED) before anything is queued."""
        if self._next >= self._n:
            raise ReconError(abi.DRYV_E_INVALID, "more macroblocks than PicSizeInMbs")
        kind = int(mb["mb_kind"])
        if kind not in (0, 1, 2):
            raise ReconError(abi.DRYV_E_UNSUPPORTED, "mb_kind %d (PCM / inter are not implemented)" % kind)
        c = np.asarray(coeffs)
        if c.size != abi.COEFFS_PER_MB:
            raise ReconError(abi.DRYV_E_INVALID, "expected 384 coefficients")
        if c.min(initial=0) < -32768 or c.max(initial=0) > 32767:
            raise ReconError(abi.DRYV_E_UNSUPPORTED, "coefficient outside int16")
        rec = self._mbs[self._next]
        for name in ("mb_kind", "i16_pred_mode", "intra_chroma_pred_mode", "qp", "prev_flags", "nz_mask"):
            rec[name] = mb[name]
        rec["rem_modes"] = mb["rem_modes"]
        self._coeffs[self._next] = c.reshape(-1)
        self._next += 1
        self._yuv = None

    def _flush(self):
        if self._yuv is None:
            if self._next != self._n:
                raise ReconError(abi.DRYV_E_STATE, "frame has %d of %d macroblocks" % (self._next, self._n))
            own = self._ctx is None
            ctx = ReconContext() if own else self._ctx
            try:
                self._yuv = ctx.reconstruct(self.fp, 1, self._mbs, self._coeffs)
            finally:
                if own:
                    ctx.close()
        return self._yuv

    def planes(self):
        """(Y, Cb, Cr) as row-major uint8 arrays [height][width]."""
        yuv = self._flush()
        nl, nc = self.width_l * self.height_l, self.width_c * self.height_c
        return (yuv[:nl].reshape(self.height_l, self.width_l),
                yuv[nl:nl + nc].reshape(self.height_c, self.width_c),
                yuv[nl + nc:].reshape(self.height_c, self.width_c))

    def write_to_yuv_file(self, file_path):
        """Frame::write_to_yuv_file (frame/mod.rs:48-70): Y, then Cb, then Cr, row-major, uncropped."""
        with open(file_path, "wb") as f:
            f.write(self._flush().tobytes())
