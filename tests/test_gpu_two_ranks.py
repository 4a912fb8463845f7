"""`-m gpu`: bench.py's multi-rank flow on real hardware with the one GPU a test box has -- two ranks started by bench.py's own launcher (torch.distributed.run),
both on GPU 0 (NRSC5_BENCH_SHARE_GPU=1), collectives over gloo on CPU tensors (NRSC5_SHARD_BACKEND=gloo: RCCL cannot put two ranks on one device): barrier + max-over-ranks
timing, the stream partition, the summary gather and -- round 5 -- EVERY rank's own all-stream comparison with the unmodified reference, gathered into rank 0's line.
The throughput of such a run means nothing; the flow and the verdicts do."""
import json
import os
import subprocess
import sys

import pytest

from tests import common
from oracle import ref

pytestmark = pytest.mark.gpu


def test_gpu_two_ranks_share_one_gpu_and_each_proves_its_own_streams(hip_lib):
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    env = dict(os.environ, NRSC5_BENCH_SHARE_GPU="1", NRSC5_SHARD_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--streams", "24", "--seconds", "5", "--steps", "2", "--warmup", "1",
                        "--cpu-baseline-seconds", "1", "--parity-processes", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_in_process_group"] == 2 and len(line["per_rank_ms_per_step"]) == 2
    assert line["config"]["total_streams"] == 48
    per_rank = line["parity"]["per_rank_reference_equality"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    for p in per_rank:
        assert p["checker_ran"] and p["streams"] == 24 and p["streams_compared"] == 24 and p["streams_equal"] == 24 and p["parity_failures"] == 0, p
    assert line["parity_failures"] == []
    assert sum(1 for l in r.stderr.splitlines() if l.startswith("rank-parity ")) == 2
