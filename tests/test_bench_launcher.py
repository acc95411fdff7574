"""bench.py's own multi-rank launcher (`python bench.py --gpus N` typed as is) and its control plane, without a GPU:
`--dry-run` runs the ranks over gloo, broadcasts the parameter block and the frame table, gathers the per-rank reports
and prints rank 0's line; nothing is reconstructed. The RCCL path proper is the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["HIP_VISIBLE_DEVICES"] = "-1"
    return e


@pytest.mark.parametrize("n", [2, 4])
def test_bench_starts_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--frames-per-gpu", "3"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["dry_run"] is True and d["value"] is None
    assert d["config"]["ranks_reporting"] == n and d["config"]["n_ranks_reporting"] == n
    # the keys an N > 1 line carries so that a scaling record shows who took part (values are null without a GPU)
    assert d["config"]["backend"] == "gloo" and set(d["config"]["kernel_ms_per_rank"]) == {"min", "max"}
    assert "cpu_baseline" in d
    assert d["config"]["shards"] == [[3 * k, 3] for k in range(n)]
    assert d["config"]["macroblocks_per_step"] == n * 3 * 120 * 68


def test_launcher_fails_when_a_rank_fails():
    # an unknown workload makes every child exit non-zero before it joins the process group
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--workload", "no_such_workload"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_launcher_ends_the_other_ranks_when_one_fails():
    # only rank 1 dies (before the rendezvous): rank 0 would wait in init_process_group for torch's timeout; the launcher
    # polls all children, ends the survivors and returns at once
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--frames-per-gpu", "2"],
                       env=dict(_env(), DRYV_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60
    assert "rank exit codes" in r.stderr


def test_single_process_under_an_external_launcher_is_not_relaunched():
    # with WORLD_SIZE in the environment (torch.distributed.run, the driver's command) bench.py is a rank, not a launcher
    e = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run", "--frames-per-gpu", "2"], env=e,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
