"""`-m gpu`: the collectives of nrsc5_amd/shard.py on DEVICE tensors through RCCL itself.  A test box has one GPU and RCCL cannot put two ranks on one device, so
this is a ONE-rank `nccl` process group: communicator set-up (dmabuf IPC environment, `dist.barrier(device_ids=...)`), all_reduce MAX / SUM, all_gather of float
vectors, of padded byte strings and of int64 summary rows, the self-path of the ingest scatter and the orderly shutdown -- every call `bench.py --gpus N` makes,
none of which had ever run on RCCL before round 6.  What a single rank cannot show is the exchange between devices; that is the driver's 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu


def test_gpu_one_rank_rccl_group_runs_every_collective(hip_lib):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NRSC5_SHARD_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "1", "--launch-check", "--force-process-group"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl(RCCL)" and line["ranks_in_process_group"] == 1 and line["collectives_ok"] is True and line["ingest_scatter_ok"] == 1, line
    assert line["ipc_mode_legacy"] == "0"
