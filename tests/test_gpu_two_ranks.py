"""`-m gpu`: bench.py's multi-rank flow on real hardware with the one GPU a test box has -- two ranks started by bench.py's own launcher (torch.distributed.run),
both on GPU 0 (NRSC5_BENCH_SHARE_GPU=1), collectives over gloo on CPU tensors (NRSC5_SHARD_BACKEND=gloo: RCCL cannot put two ranks on one device): barrier + max-over-ranks
timing, the stream partition, the summary gather and EVERY rank's own all-stream comparison with the unmodified reference, gathered into rank 0's line.
The throughput of such a run means nothing; the flow and the verdicts do.

Round 6: the verdict of a rank is read from (1) rank 0's JSON line, which carries every rank's counts AND failure strings (gathered through the process group), and
(2) that rank's own file NRSC5_BENCH_VERDICT_DIR/rank-parity-<rank>.json -- never from the stderr the ranks, the launcher and 16 checker processes share (GPUTEST_r05:
one of two `rank-parity` lines lost there).  The second test breaks one stream of rank 1 on purpose: the run must fail and name it."""
import json
import os
import subprocess
import sys

import pytest

from tests import common
from oracle import ref

pytestmark = pytest.mark.gpu


def _run(tmp_path, streams, extra_env=None):
    env = dict(os.environ, NRSC5_BENCH_SHARE_GPU="1", NRSC5_SHARD_BACKEND="gloo", NRSC5_BENCH_VERDICT_DIR=str(tmp_path))
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--streams", str(streams), "--seconds", "5", "--steps", "2", "--warmup", "1",
                        "--cpu-baseline-seconds", "1", "--parity-processes", "8"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    files = {}
    for rank in (0, 1):
        p = tmp_path / f"rank-parity-{rank}.json"
        if p.exists():
            files[rank] = json.loads(p.read_text())
    return r, (json.loads(lines[-1]) if lines else None), files


def test_gpu_two_ranks_share_one_gpu_and_each_proves_its_own_streams(hip_lib, tmp_path):
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    r, line, files = _run(tmp_path, 24)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["ranks_in_process_group"] == 2 and len(line["per_rank_ms_per_step"]) == 2
    assert line["config"]["total_streams"] == 48
    per_rank = line["parity"]["per_rank_reference_equality"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    for p in per_rank:
        assert p["checker_ran"] and p["streams"] == 24 and p["streams_compared"] == 24 and p["streams_equal"] == 24 and p["parity_failures"] == 0 and p["failures"] == [], p
    assert line["parity_failures"] == []
    # each rank's own file says the same as rank 0's line
    assert sorted(files) == [0, 1]
    for rank in (0, 1):
        assert files[rank] == per_rank[rank]


def test_gpu_two_ranks_a_broken_stream_of_rank_1_fails_the_run(hip_lib, tmp_path):
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    r, line, files = _run(tmp_path, 8, {"NRSC5_BENCH_BREAK_STREAM": "1:3"})
    assert r.returncode != 0
    assert line is not None, r.stderr[-3000:]
    per_rank = line["parity"]["per_rank_reference_equality"]
    assert per_rank[0]["parity_failures"] == 0 and per_rank[0]["streams_equal"] == 8
    assert per_rank[1]["parity_failures"] >= 1 and per_rank[1]["streams_equal"] == 7 and per_rank[1]["failures"], per_rank[1]
    assert any(f.startswith("rank 1:") for f in line["parity_failures"]), line["parity_failures"]
    assert files[1]["parity_failures"] >= 1 and files[1]["failures"] == per_rank[1]["failures"]
