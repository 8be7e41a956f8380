"""The N > 1 path of bench.py as the driver launches it — ``python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2``
— as a fresh child process on the one GPU of the test box (RT_BENCH_REHEARSAL=1: both ranks use GPU 0 and the collectives
run over gloo on host copies; RCCL refuses two ranks on one device).  What it pins: the launch contract (one JSON line on
stdout from rank 0), the FIXED global problem of BASELINE configs[4] sharded by uid (114,447,177 segments whatever N), the
per-rank load balance, the all-reduce inside the step and the all-gather-v report.  Timings of such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, RT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "segments/s" and d["scaling"] == "strong"
    assert d["config"]["segments_global"] == 114447177 and d["config"]["tracks_global"] == 1043212
    assert d["config"]["failed_tracks"] >= 0  # (tracks on which the reference's own Σℓ check fails on the substitute BWR mesh)
    pr = d["per_rank"]
    assert sum(pr["segments"]) == 114447177 and sum(pr["tracks"]) == 1043212 and len(pr["segments"]) == 2
    assert d["config"]["segments_rank0"] == pr["segments"][0] and d["config"]["segments_rank1"] == pr["segments"][1]
    assert abs(pr["segments"][0] - pr["segments"][1]) < 0.02 * 114447177  # Σℓ-balanced uid ranges: segments ∝ ℓ
    assert "allreduce_ms_exposed" in d and "ms" in d["allgather"], d.get("allgather")
    assert d["allgather"]["bytes_received_per_rank"] == 44.0 * pr["segments"][1]  # rank 0 receives rank 1's shard
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "rehearsal" in d, (d.get("kernel_ms"), d["roofline"], d.get("per_rank"), r.stderr[-1500:])


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2`, exactly as the driver types it for N = 1 with another number: no launcher around it, no WORLD_SIZE.
    bench.py must become the launcher itself — start the ranks with torch.distributed.run as a child process, pass rank 0's one
    JSON line through and leave with the child's exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_REHEARSAL"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["segments_global"] == 114447177 and d["value"] > 0
    assert sum(d["per_rank"]["segments"]) == 114447177 and "rehearsal" in d


def test_bench_two_ranks_extras_watchdog():
    """An extra after the timed region that does not return in time must not cost the run its line: with a 1-s limit the gloo
    all-gather of the 5-GB global list is still running when rank 0 prints the headline line without the extras and every rank
    leaves with exit code 0."""
    env = dict(os.environ)
    env["RT_BENCH_REHEARSAL"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--extras-timeout", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["config"]["segments_global"] == 114447177 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "extras" in d and "allgather" not in d


def test_bench_one_rank_rccl_group_runs_the_multi_gpu_path():
    """`bench.py --force-dist`: the N > 1 code path on a REAL one-rank RCCL communicator (backend "nccl"), as a fresh child process —
    the pipelined side-stream all-reduce of the volumes inside the timed region, the all-gather-v (`SegmentGather`) and the sharded
    sweep's exchange (`--sharded-sweep`) on device tensors.  The two-rank tests above run over gloo (RCCL refuses two ranks on one
    device); this is the one that drives the RCCL backend before the driver's first multi-GPU run does."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--workload", "c5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--sharded-sweep", "--no-concurrent"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["segments_global"] == 114447177 and d["value"] > 0
    # the keys of an N > 1 line
    assert d["per_rank"]["segments"] == [114447177] and d["per_rank"]["tracks"] == [1043212]
    assert d["allreduce_ms_exposed"] is not None and abs(d["allreduce_ms_exposed"]) < 1.0
    assert "error" not in d["allgather"] and d["allgather"]["ms"] >= 0 and d["allgather"]["bytes_received_per_rank"] == 0.0
    ss = d["sharded_sweep"]
    assert "error" not in ss and ss["ms_per_sweep"] > 0 and ss["groups"] == 7, ss
    reg = d["shard_regime"]
    assert reg["march_waves_rank0"] == (1043212 + 63) // 64 and reg["march_rounds_rank0"] > 1
    assert d["config"]["regime"]["two_phase"] is True
