"""The N>1 control plane on CPU: two gloo ranks exchange the parameter block / frame table and the
completion reports exactly as bench.py does over RCCL. No GPU, no oracle needed."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dryv_amd import abi, shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_frames():
    assert shard.partition_frames(2400, 8) == [(i * 300, 300) for i in range(8)]
    t = shard.partition_frames(10, 4)
    assert t == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert shard.partition_frames(1, 2) == [(0, 1), (1, 0)]


def test_control_roundtrip():
    fp = abi.make_frame_params(120, 68, cqo_cb=-3, cqo_cr=5, transform_8x8=True)
    fp.scaling_list8x8[0][7] = 99
    buf = shard.pack_control(fp, shard.partition_frames(2400, 8))
    assert buf.size == 496 + 8 * 16
    fp2, tab = shard.unpack_control(buf, 8)
    assert bytes(fp2) == bytes(fp) and tab[7] == (2100, 300)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    if rank == 0:
        fp = abi.make_frame_params(6, 4, cqo_cb=2, transform_8x8=True)
        table = shard.partition_frames(5, world)
    else:
        fp, table = abi.make_frame_params(1, 1), None   # garbage until the broadcast arrives
    fp, table = shard.broadcast_control(fp, table, dev, rank, world)
    first, n = table[rank]
    # every rank synthesises only its own frames; together they equal the unsharded batch
    mbs, co = synth.generate(fp, synth.config(i4x4=0.5, i8x8=0.2), 4, first, n)
    fake_planes = torch.from_numpy((co.reshape(-1)[: n * 24 * 384] & 0xFF).astype(np.uint8))
    reports = shard.gather_reports(n, n * 24, shard.plane_checksum(fake_planes), dev, world)
    q.put((rank, fp.pic_width_in_mbs, fp.chroma_qp_index_offset, table, mbs.tobytes(), reports))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_control_plane_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, w0, c0, t0, m0, rep0), (r1, w1, c1, t1, m1, rep1) = res
    assert (w0, c0) == (6, 2) and (w1, c1) == (6, 2)          # rank 1 got rank 0's parameter block
    assert t0 == t1 == [(0, 3), (3, 2)]
    assert rep0 == rep1 and [r[0] for r in rep0] == [3, 2] and [r[1] for r in rep0] == [72, 48]
    fp = abi.make_frame_params(6, 4, cqo_cb=2, transform_8x8=True)
    whole, _ = synth.generate(fp, synth.config(i4x4=0.5, i8x8=0.2), 4, 0, 5)
    assert m0 + m1 == whole.tobytes()                          # shards tile the batch exactly


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    if rank == 0:
        fp = abi.make_frame_params(120, 68)
        table = shard.partition_frames(2400, world)          # BASELINE.json configs[3]: 300 frames per GPU
    else:
        fp, table = abi.make_frame_params(1, 1), None
    fp, table = shard.broadcast_control(fp, table, dev, rank, world)
    first, n = table[rank]
    # one frame of the shard stands in for the planes (the CPU test has no GPU to reconstruct 300)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.7, i8x8=0.0), 2, first, 1)
    planes = torch.from_numpy((co.reshape(-1)[: 8160 * 384] & 0xFF).astype(np.uint8))
    reports = shard.gather_reports(n, n * 8160, shard.plane_checksum(planes), dev, world)
    # the same reductions bench.py uses for the verified flag and the kernel time
    vt = torch.tensor([1.0, 2.0 + 0.01 * rank], dtype=torch.float64)
    vmin = vt.clone()
    dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(vt, op=dist.ReduceOp.MAX)
    q.put((rank, table, reports, float(vmin[0]), float(vt[1]), shard.plane_checksum(planes)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_control_plane_gloo():
    """The control plane of the 8-GPU configuration (2400 frames -> 8 x 300) with eight gloo ranks on the CPU."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want_table = [(i * 300, 300) for i in range(8)]
    sums = [r[5] for r in res]
    for rank, table, reports, vmin, kmax, _ in res:
        assert table == want_table
        assert [r[0] for r in reports] == [300] * 8 and [r[1] for r in reports] == [300 * 8160] * 8
        assert [r[2] for r in reports] == sums            # every rank sees every rank's checksum
        assert vmin == 1.0 and abs(kmax - 2.07) < 1e-9
    assert len(set(sums)) == 8                            # different frames on every rank
