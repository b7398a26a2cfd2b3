"""Time-step sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path partitions by time step: every step's contours and events depend on that step's field only.
Exactly two exchanges exist: (1) all-reduce(max) of one int32 -- contour selection uses
``contours.exp_lon.max()`` over the WHOLE frame (streamer_index.py:106, overturning_index.py:105,
cutoff_index.py:90); (2) one gather of the event tables to rank 0 (counts first, then the payload),
concatenated in time order.
"""
import numpy as np

EVENT_FIELDS = ("step", "level_idx", "contour", "ind1", "ind2", "box", "orientation", "sums", "ncells")


def shard(nt_total, rank, world):
    """Contiguous block of time steps owned by ``rank``: (first, count)."""
    base, rem = divmod(int(nt_total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def global_max_nux(local_max, dist=None):
    """exchange (1)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(local_max)
    import torch
    t = torch.tensor([int(local_max)], dtype=torch.int32, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def pack_events(ev, step_offset=0, contour_offset=0):
    """One event table -> (n, 18) float64 matrix (all integer columns are exactly representable)."""
    n = ev["step"].size
    m = np.zeros((n, 18), dtype=np.float64)
    m[:, 0] = ev["step"] + step_offset
    m[:, 1] = ev["level_idx"]
    m[:, 2] = ev["contour"] + contour_offset
    m[:, 3] = ev["ind1"]
    m[:, 4] = ev["ind2"]
    m[:, 5:9] = ev["box"]
    m[:, 9] = ev["orientation"]
    m[:, 10:15] = ev["sums"]
    m[:, 15] = ev["ncells"]
    return m


def unpack_events(m):
    m = np.asarray(m, dtype=np.float64).reshape(-1, 18)
    i32 = lambda c: m[:, c].astype(np.int32)
    return dict(step=i32(0), level_idx=i32(1), contour=m[:, 2].astype(np.int64), ind1=i32(3), ind2=i32(4),
                box=m[:, 5:9].astype(np.int32), orientation=i32(9), sums=m[:, 10:15].copy(), ncells=i32(15))


def gather_event_tables(tables, dist=None, dst=0):
    """exchange (2): ``tables`` = list of packed (n_k, 18) matrices (one per index).  Returns on ``dst`` a list
    of per-index matrices concatenated in rank (= time) order, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [np.asarray(t) for t in tables]
    import torch
    world, rank, dev = dist.get_world_size(), dist.get_rank(), _device(dist)
    counts = torch.tensor([t.shape[0] for t in tables], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = torch.stack(all_counts).cpu().numpy()                 # (world, n_tables)
    rows = all_counts.sum(axis=1)
    width = 18
    mx = int(rows.max())
    buf = torch.zeros((mx, width), dtype=torch.float64, device=dev)
    mine = np.concatenate([np.asarray(t, dtype=np.float64).reshape(-1, width) for t in tables], axis=0)
    if mine.shape[0]:
        buf[: mine.shape[0]] = torch.from_numpy(mine).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    res = [[] for _ in tables]
    for r in range(world):
        block = out[r].cpu().numpy()
        o = 0
        for k in range(len(tables)):
            c = int(all_counts[r, k])
            res[k].append(block[o:o + c])
            o += c
    return [np.concatenate(v, axis=0) if v else np.zeros((0, width)) for v in res]
