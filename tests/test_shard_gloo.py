"""CPU-only, world_size 2 over gloo: the stream partition and the summary gather used by
bench.py --gpus N (no data-path collective exists: streams are independent)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    from nrsc5_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = shard.init_from_env(backend="gloo")
    dev = torch.device("cpu")
    mine = shard.stream_range(total, w, r)
    rows = np.array([[s, 10 + s, 2, 2, 30, 28, 1000 + s] for s in mine], dtype=np.int64)
    shard.barrier(dev)
    allrows = shard.gather_summaries(rows, dev)
    t = shard.max_over_ranks(1.0 + r, dev)
    tot = shard.sum_over_ranks([len(mine)], dev)
    # ingest scatter: rank 0 holds every stream's "capture" (row s = s, s + 1, ...), each rank receives its own range
    def rows_for_rank(k):
        rng = shard.stream_range(total, w, k)
        return torch.tensor([[s + c for c in range(5)] for s in rng], dtype=torch.uint8)
    got = shard.scatter_rows(rows_for_rank, len(mine), (5,), torch.uint8, dev)
    assert got.tolist() == [[s + c for c in range(5)] for s in mine], (r, got.tolist())
    # per-rank parity verdicts (bench.py --gpus N): one fixed-length vector per rank, gathered in rank order
    verdicts = shard.gather_vectors([float(r), float(len(mine)), float(len(mine)), float(len(mine)) - r, 0.0], dev)
    assert verdicts.shape == (w, 5) and [int(v[0]) for v in verdicts] == list(range(w)) and int(verdicts[1][3]) == len(shard.stream_range(total, w, 1)) - 1
    # ... and each rank's list of failure strings, verbatim (ragged lengths incl. an empty one; non-ASCII survives)
    import json
    texts = shard.gather_texts(json.dumps([] if r == 0 else [f"fm: stream {mine[0]} differs \u2260 reference" * 3]), dev)
    assert len(texts) == w and json.loads(texts[0]) == [] and json.loads(texts[1])[0].startswith(f"fm: stream {shard.stream_range(total, w, 1)[0]} differs \u2260")
    q.put((r, list(mine), allrows.tolist(), t, float(tot[0])))
    dist.destroy_process_group()


def test_two_rank_partition_and_gather():
    from nrsc5_amd import shard
    total, world = 7, 2
    assert [list(shard.stream_range(total, world, r)) for r in range(world)] == [[0, 1, 2, 3], [4, 5, 6]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, mine, allrows, t, tot in res:
        assert sorted(row[0] for row in allrows) == list(range(total))      # every stream exactly once
        assert all(row[6] == 1000 + row[0] for row in allrows)
        assert t == 2.0 and tot == 7.0
