"""Multi-GPU layout: one process per GPU, streams are independent units, zero exchange on the
data path.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU
tests) is used only for the barrier around the timed region, the max-over-ranks wall time, the
gather of fixed-size per-stream result summaries and -- optionally, before the timed region -- the
ingest scatter for captures that arrive on one rank (scatter_rows)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

SUMMARY_FIELDS = ("stream", "blocks", "p1_frames", "p1_frames_ok", "pids_frames", "fine_blocks", "frame_hash")


def stream_range(total_streams: int, world: int, rank: int) -> range:
    """Contiguous block partition: rank r owns streams [lo, hi)."""
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def init_from_env(backend: str | None = None, expect_world: int | None = None, always: bool = False):
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1 (always=True: also for a single rank --
    a one-rank RCCL group is the only way a 1-GPU box can run this module's collectives on device tensors through RCCL itself:
    tests/test_gpu_rccl_single_rank.py).  expect_world: the rank count the caller was asked for (--gpus N) -- a mismatch with what
    the process group reports is an error, never a silent single-rank run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # dmabuf IPC (this driver supports nothing else): also when an external torchrun started the ranks without it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (world > 1 or always) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # NRSC5_SHARD_BACKEND=gloo: TEST MODE of bench.py (with NRSC5_BENCH_SHARE_GPU=1: N ranks on ONE GPU, collectives on CPU tensors) -- the multi-rank
        # flow incl. the per-rank parity gather on a single-GPU box; RCCL cannot put two ranks on one device
        backend = backend or os.environ.get("NRSC5_SHARD_BACKEND")
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                rank=rank, world_size=world)
    seen = dist.get_world_size() if dist.is_initialized() else 1
    if seen != world or (expect_world is not None and seen != expect_world):
        raise RuntimeError(f"asked for {expect_world} ranks, WORLD_SIZE={world}, process group reports {seen}")
    return rank, world, local


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_ranks(script: str, argv, nproc: int, extra_env: dict | None = None) -> int:
    """Start `nproc` ranks of `script argv` on this node -- one process per GPU -- with torch's own launcher, as the driver does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... script argv`).  The rendezvous is torchrun's standalone
    c10d store on 127.0.0.1: the launcher binds the port itself (no bind-close-reuse window in which a concurrent launch on
    the node could take it).  Returns the launcher's exit code; the ranks inherit stdout / stderr (rank 0 prints the result
    line)."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--standalone", "--local-addr", "127.0.0.1", script] + list(argv)
    return subprocess.call(cmd, env=env)


def shutdown(device=None):
    """Leave the process group in step: a rank that exits while its peers still talk to it aborts them (gloo) or hangs them (RCCL)."""
    if dist.is_initialized():
        barrier(device)
        dist.destroy_process_group()


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device) -> list:
    """one float per rank, in rank order (per-rank pass times, stream counts)"""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_vectors(values, device) -> np.ndarray:
    """one fixed-length float64 vector per rank -> [world, len(values)] in rank order (per-rank parity verdicts)"""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return t.cpu().numpy()[None, :]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def gather_texts(text: str, device) -> list:
    """one UTF-8 string per rank, in rank order (each rank's own list of failures, as JSON): lengths first, then the bytes padded to the
    longest -- so that rank 0's result line carries every rank's verdict verbatim and nobody has to find it on a shared stderr"""
    if not dist.is_initialized():
        return [text]
    raw = np.frombuffer(text.encode("utf-8"), dtype=np.uint8)
    world = dist.get_world_size()
    n = torch.tensor([raw.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = max(1, int(max(s.item() for s in sizes)))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    if raw.size:
        pad[:raw.size] = torch.from_numpy(raw.copy()).to(device)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [bytes(o[:int(s.item())].cpu().numpy()).decode("utf-8") for o, s in zip(out, sizes)]


def sum_over_ranks(values, device) -> np.ndarray:
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_summaries(local: np.ndarray, device) -> np.ndarray:
    """all_gather of int64 [n_local, len(SUMMARY_FIELDS)] rows (padded to the largest shard)."""
    local = np.ascontiguousarray(local, dtype=np.int64).reshape(-1, len(SUMMARY_FIELDS))
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = int(max(s.item() for s in sizes))
    pad = torch.full((mx, len(SUMMARY_FIELDS)), -1, dtype=torch.int64, device=device)
    pad[:local.shape[0]] = torch.from_numpy(local).to(device)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    rows = [o[:int(s.item())].cpu().numpy() for o, s in zip(out, sizes)]
    return np.concatenate(rows, axis=0)


def scatter_shards(shard_for_rank, specs, device, src: int = 0):
    """Ingest scatter: the captures arrive on rank `src`, every rank ends up with its own contiguous range.
    shard_for_rank(r) -> list of tensors (one per spec) on `device` is only called on `src`, one destination at a time (so
    the root never holds more than one foreign shard), and every tensor goes out as ONE point-to-point send -- root to peer
    over that peer's own xGMI link with RCCL, not a ring.  specs = [(shape, dtype), ...] of THIS rank's tensors.
    Returns this rank's list of tensors."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(shard_for_rank(0))
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != src:
        out = [torch.empty(tuple(shape), dtype=dtype, device=device) for shape, dtype in specs]
        for t in out:
            dist.recv(t, src=src)
        return out
    mine = None
    for r in range(world):
        tensors = [t.contiguous() for t in shard_for_rank(r)]
        if r == src:
            mine = tensors
        else:
            for t in tensors:
                dist.send(t, dst=r)
            del tensors
    return mine


def scatter_rows(rows_for_rank, n_local: int, row_shape, dtype, device, src: int = 0) -> torch.Tensor:
    """scatter_shards for a single [n, *row_shape] tensor per rank."""
    return scatter_shards(lambda r: [rows_for_rank(r)], [((n_local,) + tuple(row_shape), dtype)], device, src)[0]
