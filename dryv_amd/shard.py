"""Frame-level sharding across GPUs.

All-intra frames are independent (QP predictor, CABAC contexts and `Frame` are per slice:
reference slice/mod.rs:153, cabac/mod.rs:72-87, decoder.rs:124), so the path partitions by frame
with no exchange of pixels or coefficients. The only traffic between ranks is control plane:
rank 0 broadcasts the 496-byte dryv_frame_params and the frame table, and at the end every rank
contributes (frames, macroblocks, checksum) to an all-gather. Over RCCL these are latency-bound
messages; xGMI bandwidth never matters here.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import abi


def partition_frames(total_frames, world_size):
    """Contiguous blocks: rank r gets frames [first, first + n). Remainder goes to the low ranks."""
    base, rem = divmod(int(total_frames), int(world_size))
    table, first = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        table.append((first, n))
        first += n
    return table


def pack_control(fp, table):
    """dryv_frame_params + frame table as one uint8 buffer (the broadcast payload)."""
    raw = np.frombuffer(C.string_at(C.addressof(fp), C.sizeof(fp)), dtype=np.uint8)
    tab = np.asarray(table, dtype="<i8").reshape(-1).view(np.uint8)
    return np.concatenate([raw, tab])


def unpack_control(buf, world_size):
    buf = np.asarray(buf, dtype=np.uint8)
    n = C.sizeof(abi.FrameParams)
    fp = abi.FrameParams.from_buffer_copy(buf[:n].tobytes())
    tab = buf[n:n + 16 * world_size].view("<i8").reshape(world_size, 2)
    return fp, [(int(a), int(b)) for a, b in tab]


def broadcast_control(fp, table, device, rank, world_size):
    """Rank 0's parameter block and frame table to every rank. No-op for a single process."""
    if world_size == 1:
        return fp, table
    size = C.sizeof(abi.FrameParams) + 16 * world_size
    if rank == 0:
        t = torch.from_numpy(pack_control(fp, table).copy()).to(device)
    else:
        t = torch.zeros(size, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    return unpack_control(t.cpu().numpy(), world_size)


def gather_reports(frames_done, mbs_done, checksum, device, world_size):
    """All-gather of each rank's (frames, macroblocks, checksum) -> list of tuples on every rank."""
    mine = torch.tensor([int(frames_done), int(mbs_done), int(checksum) & 0x7FFFFFFFFFFFFFFF],
                        dtype=torch.int64, device=device)
    if world_size == 1:
        return [tuple(int(v) for v in mine.cpu())]
    out = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(out, mine)
    return [tuple(int(v) for v in o.cpu()) for o in out]


def plane_checksum(yuv_u8):
    """Cheap order-sensitive checksum of reconstructed planes (device or host uint8 tensor):
    sum(byte * (1 + index mod 251)) in int64."""
    t = yuv_u8.reshape(-1)
    n = t.numel()
    pad = (-n) % 251
    if pad:
        t = torch.cat([t, torch.zeros(pad, dtype=t.dtype, device=t.device)])
    w = torch.arange(1, 252, dtype=torch.int64, device=t.device)
    return int((t.reshape(-1, 251).to(torch.int64) * w).sum().item())
