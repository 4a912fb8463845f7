"""CPU-only: `python bench.py --gpus 2` with no torchrun around it must start 2 ranks itself (one per GPU on the GPU box),
give each its RANK / LOCAL_RANK / WORLD_SIZE, form the process group and report the rank count the collective saw.
Here the ranks have no GPU, so --launch-check stops after the process group (gloo); the workload path is the -m gpu tests'."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_spawns_its_own_ranks():
    r, lines = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                    # rank 0 only
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_in_process_group"] == 2 and lines[0]["rank_sum"] == 1
    assert lines[0]["ingest_scatter_ok"] == 2             # both ranks received their own shard from rank 0 (point-to-point)
    assert lines[0]["collectives_ok"] is True            # max / vector / text / summary gathers: every collective of the real run
    assert lines[0]["launched_by"] == "bench.py"


def test_bench_single_rank_is_one_process():
    r, lines = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0 and lines == [{"launch_check": True, "n_gpus": 1, "ranks_in_process_group": 1, "rank_sum": 0, "ingest_scatter_ok": 1, "collectives_ok": True,
                                            "scaling": "weak", "streams_per_rank": [256], "ipc_mode_legacy": "0",
                                            "backend": "gloo", "launched_by": "single process"}], r.stdout + r.stderr[-1000:]


def test_world_size_mismatch_is_an_error():
    """--gpus 4 inside a 2-rank torchrun (the driver's own launch line, wrong N) must fail loudly, not run 2 ranks."""
    from nrsc5_amd import shard
    rc = shard.launch_ranks(os.path.join(ROOT, "bench.py"), ["--gpus", "4", "--launch-check"], 2)
    assert rc != 0


def test_eight_ranks_strong_scaling_split():
    """The first 8-GPU driver run must need no code change: 8 ranks (gloo here), configs[3]'s fixed 2048 streams split into
    contiguous ranges of 256, every rank in the process group, dmabuf IPC mode set for the ranks."""
    r, lines = _run(["--gpus", "8", "--launch-check", "--scaling", "strong", "--total-streams", "2048"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["ranks_in_process_group"] == 8 and lines[0]["rank_sum"] == 28
    assert lines[0]["scaling"] == "strong" and lines[0]["streams_per_rank"] == [256] * 8 and lines[0]["ingest_scatter_ok"] == 8 and lines[0]["collectives_ok"] is True
    assert lines[0]["ipc_mode_legacy"] == "0"


def test_driver_style_torchrun_line_without_ipc_env():
    """The driver starts the ranks with its own torchrun line; if HSA_ENABLE_IPC_MODE_LEGACY is not in that environment the
    ranks must set it themselves before the first HIP call (RCCL needs dmabuf IPC on this driver)."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--standalone", "--local-addr", "127.0.0.1",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=e, capture_output=True, text=True, timeout=600)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    assert lines[0]["ipc_mode_legacy"] == "0" and lines[0]["launched_by"] == "external torchrun" and lines[0]["streams_per_rank"] == [256, 256]


def test_single_rank_forced_process_group():
    """one rank, but through a real process group (gloo here; on the GPU box the same line forms a one-rank RCCL group: tests/test_gpu_rccl_single_rank.py)"""
    r, lines = _run(["--gpus", "1", "--launch-check", "--force-process-group"], env={"MASTER_PORT": str(29600 + os.getpid() % 300)})
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    assert lines[0]["ranks_in_process_group"] == 1 and lines[0]["collectives_ok"] is True and lines[0]["ingest_scatter_ok"] == 1
