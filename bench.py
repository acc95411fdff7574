#!/usr/bin/env python3
"""bench.py — macroblocks/s of the MI355X reconstruction path on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, launch_ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: every rank reconstructs its
300-frame shard of the 1080p all-intra workload (BASELINE.json configs[1]; N GPUs = configs[3]'s
300 frames per GPU, weak scaling) with inputs already resident in HBM. Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it): the parent starts N fresh child processes,
    one rank per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and
    exits non-zero if any rank fails. It runs before this process has made any GPU call (nothing here imports torch or the
    library), and the children are new processes, never an exec of one that has touched the GPU. All children are polled:
    when one of them fails, the others -- which would sit in the rendezvous or a collective until torch's timeout -- are
    ended and the launcher returns at once."""
    import subprocess
    import tempfile
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        failed = False
        while True:
            rcs = [p.poll() for p in procs]
            if any(rc not in (None, 0) for rc in rcs):
                failed = True
                break
            if all(rc == 0 for rc in rcs):
                break
            time.sleep(0.05)
        if failed:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 5.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        rcs = [p.returncode for p in procs]
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return 1
    return 0


def kernel_source_sha():
    """sha256 (16 hex digits) over the sources of the kernel the bench measures, its tables, its launch geometry and the
    library code that launches it: measured HBM traffic is only quoted for the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("band_kernel.h", "band_diag.h", "recon_band.hip", "wave.h", "kparams.h", "recon_params.h", "band_launch.h", "recon_api.hip"):
        with open(os.path.join(ROOT, "dryv_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


ALG_BYTES_PER_MB = 1168          # 768 B coefficients + 16 B record read, 384 B pixels written
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(fp, mbs, coeffs, n_frames, gpu_out, frame_bytes, sample_frames):
    """Times the oracle (the bit-exact port of the reference's Rust path) on one host core over the
    first `sample_frames` frames of the same workload, and uses the same run to verify the GPU output."""
    import oracle
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    k = min(sample_frames, n_frames)
    t0 = time.perf_counter()
    st, want = oracle.reconstruct(fp, k, mbs[:k * per], coeffs[:k * per])
    dt = time.perf_counter() - t0
    got = gpu_out[:k * frame_bytes].cpu().numpy()
    verified = bool(st == 0 and np.array_equal(got, want))
    out = {"value": k * per / dt, "unit": "macroblocks/s", "cores": 1, "kind": "port",
           "sample": "first %d frames of the workload (%d macroblocks), oracle/dryv_oracle.c -O2, %.1f s"
                     % (k, k * per, dt),
           "gpu_output_verified_bit_exact": verified}
    out["nproc"] = os.cpu_count()
    out["cores_available"] = len(os.sched_getaffinity(0))
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        out["cpu_model"] = None
    # SURVEY.md 8d also asks for "all host cores, one frame per task" (frames are independent; ctypes releases the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = len(os.sched_getaffinity(0))  # every core this process may run on
        if cores > 1:
            def one(f):
                return oracle.reconstruct(fp, 1, mbs[f * per:(f + 1) * per], coeffs[f * per:(f + 1) * per])[0]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                sts = list(ex.map(one, range(k)))
            dt2 = time.perf_counter() - t0
            if all(s == 0 for s in sts):
                out["all_cores"] = {"value": k * per / dt2, "unit": "macroblocks/s", "cores": cores,
                                    "sample": "same frames, one frame per task, %.1f s" % dt2}
    except Exception:
        pass
    return out


def secondary_line(name, device, dev_index, steps, warmup, cpu_frames):
    """BASELINE.json's other single-GPU configuration (configs[2]: 4K, 8x8 transform), measured in the same process after
    the primary workload's timed region: kernel time from the library's HIP events, wall time per step, first and last
    frame against the oracle, and the oracle's rate on a few frames as its CPU baseline."""
    import oracle
    cid, w, h, frames, t8, kw = synth.WORKLOADS[name]
    fp = dryv_amd.make_frame_params(w, h, transform_8x8=t8)
    per = w * h
    mbs, coeffs = synth.generate(fp, synth.config(**kw), cid, 0, frames)
    d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).to(device)
    d_coeffs = torch.from_numpy(coeffs).to(device)
    d_out = torch.zeros(frames * per * 384, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    with dryv_amd.ReconContext(dev_index) as ctx:
        def run(n):
            for _ in range(n):
                ctx.submit_device_queued(fp, frames, d_mbs.data_ptr(), d_coeffs.data_ptr(), d_out.data_ptr())
            ctx.sync()
            return ctx.kernel_ms_stats(n)[0]
        run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms = run(steps)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3 / steps
        # (thirty steps at least: with lanes the first and the last launch of a queue run alone on half the chip)
        piped = pipelined_line(ctx, fp, frames, d_mbs, d_coeffs, d_out, frames * per, max(steps, 30), max(warmup, 6))
    verified = piped["outputs_equal_primary_run"]
    for fidx in sorted({0, frames - 1}):
        st, want = oracle.reconstruct(fp, 1, mbs[fidx * per:(fidx + 1) * per], coeffs[fidx * per:(fidx + 1) * per])
        got = d_out[fidx * per * 384:(fidx + 1) * per * 384].cpu().numpy()
        verified = verified and st == 0 and bool(np.array_equal(got, want))
    k = min(cpu_frames, frames)
    t0 = time.perf_counter()
    st, want = oracle.reconstruct(fp, k, mbs[:k * per], coeffs[:k * per])
    dt = time.perf_counter() - t0
    verified = verified and st == 0 and bool(np.array_equal(d_out[:k * per * 384].cpu().numpy(), want))
    n_mbs = frames * per
    return {"workload": "%s: %dx%d macroblocks x %d frames, resident in HBM" % (name, w, h, frames),
            "value": n_mbs / (wall_ms * 1e-3), "unit": "macroblocks/s", "steps": steps, "warmup": warmup,
            "ms_per_step": wall_ms, "kernel_ms_avg": kernel_ms,
            "frac": n_mbs * ALG_BYTES_PER_MB / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_wall": n_mbs * ALG_BYTES_PER_MB / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "verified_bit_exact": verified, "pipelined": piped,
            "cpu_baseline": {"value": k * per / dt, "unit": "macroblocks/s", "cores": 1, "kind": "port",
                             "sample": "first %d frames (%d macroblocks), oracle/dryv_oracle.c -O2, %.1f s" % (k, k * per, dt)}}


def pipelined_line(ctx, fp, n_frames, d_mbs, d_coeffs, d_out, n_mbs, steps, warmup, lanes=3, world=1, ctrl=None, total_mbs=None):
    """The same workload with the library's queue lanes (dryv_recon_set_queue_lanes): the queued batches rotate over `lanes`
    streams, every launch with half of the resident grid, so that two launches are resident side by side and one's ramp and
    drain run beside the other's steady state. Measured in the same process after the primary timed region, the same K steps
    between two host synchronisations; every lane writes a buffer of its own, each compared with the primary run's output.
    A launch's own duration is then that of half the chip: frac_wall (algorithmic bytes / wall time per step) is the figure."""
    outs = [d_out] + [torch.zeros_like(d_out) for _ in range(lanes - 1)]
    want = shard.plane_checksum(d_out)
    ctx.set_queue_lanes(lanes)

    def run(n):
        done, ksum = 0, 0.0
        while done < n:
            m = min(60, n - done)
            for k in range(m):
                ctx.submit_device_queued(fp, n_frames, d_mbs.data_ptr(), d_coeffs.data_ptr(), outs[(done + k) % lanes].data_ptr())
            ctx.sync()
            ksum += ctx.kernel_ms_stats(m)[0] * m
            done += m
        return ksum / max(n, 1)
    for o in outs[1:]:
        o.zero_()
    run(warmup)
    # (N ranks: every rank runs its own shard at the same time, bracketed and reduced like the primary timed region)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = run(steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_queue_lanes(1)
    same = all(shard.plane_checksum(o) == want for o in outs)
    if world > 1:
        t = torch.tensor([elapsed, 0.0 if same else 1.0], dtype=torch.float64, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, same = float(t[0].item()), bool(t[1].item() == 0.0)
    wall_ms = elapsed * 1e3 / steps
    job_mbs = total_mbs if total_mbs is not None else n_mbs
    return {"what": "the same batch queued over %d lanes of the context (dryv_recon_set_queue_lanes): half-size grids, two launches "
                    "resident side by side" % lanes,
            "lanes": lanes, "launches_in_flight": 2, "steps": steps, "warmup": warmup,
            "ms_per_step": wall_ms, "value": job_mbs / (wall_ms * 1e-3), "unit": "macroblocks/s",
            "frac_wall": n_mbs * ALG_BYTES_PER_MB / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_avg_per_launch_on_half_the_chip": kernel_ms,
            "outputs_equal_primary_run": bool(same)}


def dry_run(args, world, rank):
    """The launcher and the control plane without a GPU (tests/test_bench_launcher.py): every rank joins the process group
    over gloo, receives the parameter block and its shard from rank 0, generates the shard's first frame on the host and
    reports; rank 0 prints one JSON line with value null. Nothing is reconstructed and nothing is measured."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    cid, w, h, frames, t8, kw = synth.WORKLOADS[args.workload]
    per_gpu = args.frames_per_gpu or frames
    fp = dryv_amd.make_frame_params(w, h, transform_8x8=t8)
    table = shard.partition_frames(per_gpu * world, world)
    ctrl = torch.device("cpu")
    fp, table = shard.broadcast_control(fp, table, ctrl, rank, world)
    first, n_frames = table[rank]
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    mbs, coeffs = synth.generate(fp, synth.config(**kw), cid, first, 1)
    checksum = int(np.bitwise_xor.reduce(coeffs.reshape(-1).view(np.uint16).astype(np.uint64))) ^ (first << 32)
    reports = shard.gather_reports(n_frames, n_frames * per, checksum, ctrl, world)
    if rank == 0:
        print(json.dumps({"metric": "macroblocks/s (1080p all-intra)", "value": None, "unit": "macroblocks/s", "n_gpus": world,
                          "dry_run": True, "scaling": "weak",
                          "config": {"workload": args.workload, "frames_per_gpu": n_frames,
                                     "shards": [[int(a), int(b)] for a, b in table],
                                     "backend": "gloo", "n_ranks_reporting": len(reports),
                                     "ranks_reporting": len(reports), "kernel_ms_per_rank": {"min": None, "max": None},
                                     "macroblocks_per_step": sum(r[1] for r in reports)},
                          "cpu_baseline": None}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    if "WORLD_SIZE" not in os.environ and pre.parse_known_args()[0].gpus > 1:
        return launch_ranks(pre.parse_known_args()[0].gpus, sys.argv[1:])   # (before anything below touches the GPU)
    global torch, dist, dryv_amd, shard, synth
    import torch
    import torch.distributed as dist
    import dryv_amd
    from dryv_amd import shard, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2_1080p_intra_4x4", choices=sorted(synth.WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=None)
    ap.add_argument("--cpu-sample-frames", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the per-rank oracle check of the shard's first/last frame")
    ap.add_argument("--preroll-ms", type=float, default=40.0, help="untimed device pre-roll before the warm-up steps")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="wait on the host after every step (round 1/2 behaviour) instead of queueing the steps on the stream")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the `pipelined` object (the same workload over the library's queue lanes) of a single-GPU run")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary line (C3: 4K, 8x8 transform) that the default single-GPU C2 run also measures")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / control-plane rehearsal without a GPU: ranks over gloo, no reconstruction, value null")
    args = ap.parse_args()


    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world   # (under torch.distributed.run the launcher's world size wins)
    if os.environ.get("DRYV_BENCH_FAIL_RANK") == str(rank):
        sys.exit(3)   # (test hook, tests/test_bench_launcher.py: one rank dies before it joins the process group)
    if args.dry_run:
        return dry_run(args, world, rank)
    # One rank per GPU. (DRYV_BENCH_BACKEND=gloo is a rehearsal hook for boxes with fewer GPUs than ranks: the ranks
    # then share devices and the control plane runs over gloo on the host; the product path is RCCL.)
    backend = os.environ.get("DRYV_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ctrl = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    # ---- control plane: rank 0 decides, everyone receives (RCCL broadcast) ----------------------
    cid, w, h, frames, t8, kw = synth.WORKLOADS[args.workload]
    per_gpu = args.frames_per_gpu or frames
    fp = dryv_amd.make_frame_params(w, h, transform_8x8=t8)
    table = shard.partition_frames(per_gpu * world, world)
    fp, table = shard.broadcast_control(fp, table, ctrl, rank, world)
    first, n_frames = table[rank]
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    n_mbs = n_frames * per
    frame_bytes = per * 384

    # ---- synthetic shard, generated on the host and made resident in HBM ------------------------
    mbs, coeffs = synth.generate(fp, synth.config(**kw), cid, first, n_frames)
    d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).to(device)
    d_coeffs = torch.from_numpy(coeffs).to(device)
    d_out = torch.zeros(n_mbs * 384, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()

    ctx = dryv_amd.ReconContext(dev_index)

    def step():
        ctx.submit_device(fp, n_frames, d_mbs.data_ptr(), d_coeffs.data_ptr(), d_out.data_ptr())

    # Device pre-roll (untimed, not part of W): the GPU leaves its idle clocks only after some tens of milliseconds
    # of work; without it the first timed steps of a short run are measured at a lower clock than the rest.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        step()
        ctx.sync()
    def run(n):
        """n passes; returns their kernel times' sum in ms. The passes are queued back to back on the context's stream
        (dryv_recon_submit_device_queued: each one resets the workspace, reconstructs the whole batch and is timed by
        an event pair of its own), with one host wait per 64 of them: the library remembers 64 launch timings."""
        total = 0.0
        if args.sync_each_step:
            for _ in range(n):
                step()
                ctx.sync()
                total += ctx.last_kernel_ms()
            return total
        done = 0
        while done < n:
            m = min(64, n - done)
            for _ in range(m):
                ctx.submit_device_queued(fp, n_frames, d_mbs.data_ptr(), d_coeffs.data_ptr(), d_out.data_ptr())
            ctx.sync()
            total += ctx.kernel_ms_stats(m)[0] * m
            done += m
        return total

    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = [run(args.steps) / max(args.steps, 1)]   # (the average over the timed steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=ctrl)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- every rank checks its own shard against the oracle after the timed region: the first and the last frame
    # of the shard (bit-exact), then the per-rank verdicts, checksums and kernel times are gathered
    verified = True
    if not args.no_verify:
        import oracle
        for fidx in sorted({0, n_frames - 1}):
            st, want = oracle.reconstruct(fp, 1, mbs[fidx * per:(fidx + 1) * per], coeffs[fidx * per:(fidx + 1) * per])
            got = d_out[fidx * frame_bytes:(fidx + 1) * frame_bytes].cpu().numpy()
            verified = verified and st == 0 and bool(np.array_equal(got, want))
    reports = shard.gather_reports(n_frames, n_mbs, shard.plane_checksum(d_out), ctrl, world)
    total_mbs = sum(r[1] for r in reports)
    vt = torch.tensor([1.0 if verified else 0.0, float(np.mean(kernel_ms))], dtype=torch.float64, device=ctrl)
    if world > 1:
        vmin = vt.clone()
        dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(vt, op=dist.ReduceOp.MAX)
        all_verified, kernel_ms_max, kernel_ms_min = bool(vmin[0].item() == 1.0), float(vt[1].item()), float(vmin[1].item())
    else:
        all_verified, kernel_ms_max = verified, float(vt[1].item())
        kernel_ms_min = kernel_ms_max
    if not all_verified:
        sys.exit("bench.py: a rank's shard differs from the oracle")
    kernel_name = "band_kernel"
    # the same K steps over the library's queue lanes, on every rank at the same time (like the secondary line: part of the full
    # default line only -- the profiling scripts, which pass --no-cpu-baseline, time and count the primary launches alone)
    piped = None
    if not args.no_pipelined and not args.sync_each_step and not args.no_cpu_baseline:
        piped = pipelined_line(ctx, fp, n_frames, d_mbs, d_coeffs, d_out, n_mbs, args.steps, min(args.warmup, 12),
                               world=world, ctrl=ctrl, total_mbs=total_mbs)
        if not piped["outputs_equal_primary_run"]:
            sys.exit("bench.py: the queue lanes' output differs from the primary run's")

    if rank == 0:
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = n_mbs * ALG_BYTES_PER_MB / avg_kernel_s / 1e9
        # HBM bytes per launch from the PMC passes of tools/profile_round.sh, quoted only when they were taken on
        # exactly this kernel source (otherwise null: a stale figure is not a measurement of this run)
        traffic, traffic_src, valu = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(args.workload, {})
                if ent.get("kernel_source_sha") == kernel_source_sha():
                    traffic, traffic_src = ent.get("bytes_per_launch"), ent.get("source")
                    if ent.get("valu_per_macroblock"):
                        # the bound that matters for this integer kernel: vector instructions per macroblock
                        # (SQ_INSTS_VALU) x the measured issue cost of a wave-instruction on one of the chip's 1024 SIMDs
                        floor_ms = ent["valu_per_macroblock"] * n_mbs / 1024.0 * ent["ns_per_valu_instruction"] * 1e-6
                        valu = {"per_macroblock": ent["valu_per_macroblock"], "ns_per_instruction_per_simd": ent["ns_per_valu_instruction"],
                                "floor_ms": floor_ms, "kernel_over_floor": avg_kernel_s * 1e3 / floor_ms,
                                "source": "SQ_INSTS_VALU (profiles/) and tools/micro/valu_rate.hip"}
            except Exception:
                traffic = None
        line = {
            "metric": "macroblocks/s (1080p all-intra)" if args.workload.startswith("C2") else
                      "macroblocks/s (4K all-intra)",
            "value": total_mbs * args.steps / elapsed,
            "unit": "macroblocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d macroblocks x %d frames per GPU, resident in HBM; "
                                   "frames sharded contiguously, one rank per GPU, no data-path collective"
                                   % (args.workload, w, h, n_frames),
                       "frames_per_gpu": n_frames, "macroblocks_per_step": total_mbs,
                       "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else "none (one rank)",
                       "n_ranks_reporting": len(reports),
                       "kernel_ms_per_rank": {"min": kernel_ms_min, "max": kernel_ms_max},
                       "steps_queued_on_stream": not args.sync_each_step,
                       "shards_verified_bit_exact": (world if not args.no_verify else 0),
                       "shard_checksums": ["%016x" % (r[2] & 0xFFFFFFFFFFFFFFFF) for r in reports]},
            # frac: the contract's figure (algorithmic bytes of this rank's launch / the kernel's average duration by HIP events);
            # frac_wall: the same bytes / this job's wall time per step (what `value` is made of: it also pays for the
            # workspace reset in front of every launch and for whatever the queue leaves between launches)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_wall": n_mbs * ALG_BYTES_PER_MB / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms_avg": avg_kernel_s * 1e3,
                         "kernel_ms_max_over_ranks": kernel_ms_max, "kernel_source_sha": kernel_source_sha(),
                         "algorithmic_bytes_per_launch": n_mbs * ALG_BYTES_PER_MB, "vector_issue": valu},
        }
        # the CPU path beside the GPU number at every N: rank 0 times the oracle over (a sample of) its own shard
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fp, mbs, coeffs, n_frames, d_out, frame_bytes,
                                                args.cpu_sample_frames)
        if piped is not None:
            line["pipelined"] = piped
        if world == 1 and args.workload.startswith("C2") and not args.no_secondary and not args.no_cpu_baseline:
            ctx.close()
            del d_mbs, d_coeffs, d_out
            line["secondary"] = secondary_line("C3_4k_intra_8x8", device, dev_index, steps=10, warmup=3, cpu_frames=20)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()   # (the other ranks wait here while rank 0 times the CPU baseline)
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
