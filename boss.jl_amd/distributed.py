"""One-process-per-GPU helpers (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The hot path shards embarrassingly (candidates / outputs /
hyper-parameter samples); the exchanges are all tiny and latency-bound:
  candidates  one 16-byte (value, index) all-gather for the arg-max — RCCL has no MAXLOC, so
              every rank reduces the gathered pairs locally;
  outputs     one all-gather of the owners' (mu_i, var_i) rows (2·M·8 bytes per output);
  samples     one all-reduce(sum) of the M partial acquisition sums."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def _dist():
    import sys
    if "torch" not in sys.modules:      # a process that never imported torch cannot have initialised torch.distributed
        return None                     # (and single-process users do not pay the one-second import)
    try:
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() else None
    except Exception:
        return None


def rank_world(group=None) -> Tuple[int, int]:
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(group), d.get_world_size(group)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of n items (first n % world ranks get one extra) —
    the analogue of get_sample_counts (src/utils/sampling.jl:6-13)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device_for(d, group):
    import torch
    backend = d.get_backend(group)
    if backend == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_BUF = {}


def _buffers(d, group, dev, world, n):
    """Per-(group, device, length) exchange buffers, allocated once: the arg-max exchange runs once per
    acquisition pass and must not allocate device tensors every time."""
    key = (id(group), str(dev), world, n)
    b = _BUF.get(key)
    if b is None:
        import torch
        b = (torch.empty(n, dtype=torch.float64, device=dev), torch.empty(world * n, dtype=torch.float64, device=dev))
        _BUF[key] = b
    return b


def argmax_exchange(val: float, idx: int, group=None) -> Tuple[float, int]:
    """Global arg-max of per-rank (value, GLOBAL index) pairs: larger value wins, NaN counts as
    the largest (Julia argmax), ties go to the smaller index — identical on every rank.
    One 16-byte all-gather (all_gather_into_tensor on preallocated buffers), one device->host copy."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return val, idx
    import torch
    dev = _device_for(d, group)
    world = d.get_world_size(group)
    mine, out = _buffers(d, group, dev, world, 2)
    mine.copy_(torch.tensor([val, float(idx)], dtype=torch.float64))
    d.all_gather_into_tensor(out, mine, group=group)
    flat = out.cpu().tolist()
    pairs = [(flat[2 * r], int(flat[2 * r + 1])) for r in range(world)]
    return reduce_pairs(pairs)


def shared_seed(seed, group=None) -> int:
    """The seed every rank must use so that all ranks draw the SAME candidates / prior samples (the exchanged global
    index is applied to local arrays).  An explicit seed is returned as is; with seed=None rank 0 draws one and
    broadcasts it (8 bytes) — independent default_rng(None) streams on the ranks would silently return different
    points on different ranks."""
    if seed is not None:
        return int(seed)
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return int(np.random.SeedSequence().entropy % (1 << 63))
    import torch
    dev = _device_for(d, group)
    t = torch.zeros(1, dtype=torch.int64, device=dev)
    if d.get_rank(group) == 0:
        t[0] = int(np.random.SeedSequence().entropy % (1 << 62))
    d.broadcast(t, src=d.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.cpu()[0])


def broadcast_array(local, shape, src: int, group=None) -> np.ndarray:
    """float64 array of known `shape` from group rank `src` to every rank (one fixed-size broadcast)."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return np.asarray(local, dtype=np.float64).reshape(shape)
    import torch
    dev = _device_for(d, group)
    if d.get_rank(group) == src:
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(shape).copy()).to(dev)
    else:
        t = torch.empty(tuple(shape), dtype=torch.float64, device=dev)
    d.broadcast(t, src=d.get_global_rank(group, src) if group is not None else src, group=group)
    return t.cpu().numpy()


def owner_of_index(i: int, n: int, world: int) -> int:
    """The rank whose shard_range(n, rank, world) contains item i."""
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        if lo <= i < hi:
            return r
    raise IndexError(i)


def reduce_pairs(pairs):
    best_v, best_i = pairs[0]
    for v, i in pairs[1:]:
        vn, bn = v != v, best_v != best_v
        if vn != bn:
            better = vn
        elif not vn and v != best_v:
            better = v > best_v
        else:
            better = i < best_i
        if better:
            best_v, best_i = v, i
    return best_v, best_i


def allgather_concat(local: np.ndarray, group=None) -> np.ndarray:
    """Concatenate per-rank 1-D float64 arrays (ragged shards allowed): one all-gather of the lengths, one
    fixed-size all-gather of the (padded) payloads — tensors only, nothing is pickled."""
    d = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
    if d is None or d.get_world_size(group) == 1:
        return local
    import torch
    dev = _device_for(d, group)
    world = d.get_world_size(group)
    n_mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    n_all = torch.empty(world, dtype=torch.int64, device=dev)
    d.all_gather_into_tensor(n_all, n_mine, group=group)
    counts = [int(v) for v in n_all.cpu().tolist()]
    nmax = max(counts)
    if nmax == 0:
        return np.zeros(0)
    pad = np.zeros(nmax)
    pad[:local.shape[0]] = local
    mine = torch.from_numpy(pad).to(dev)
    out = torch.empty(world * nmax, dtype=torch.float64, device=dev)
    d.all_gather_into_tensor(out, mine, group=group)
    a = out.cpu().numpy()
    return np.concatenate([a[r * nmax:r * nmax + counts[r]] for r in range(world)])


def allreduce_sum(local: np.ndarray, group=None) -> np.ndarray:
    """Element-wise sum over ranks of equally shaped float64 arrays (one all-reduce)."""
    d = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if d is None or d.get_world_size(group) == 1:
        return local
    import torch
    t = torch.from_numpy(local.copy()).to(_device_for(d, group))
    d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def owner_of(i: int, world: int) -> int:
    """Output i (or any independent unit i) lives on rank i mod world."""
    return i % world


def allgather_owned(local_rows: dict, n_units: int, row_shape, group=None) -> np.ndarray:
    """Every rank contributes the float64 rows it owns ({unit index: array of row_shape}, unit i
    owned by rank i mod world); returns the full [n_units, *row_shape] array on every rank with
    ONE fixed-size all-gather (ranks owning fewer units pad)."""
    d = _dist()
    row_shape = tuple(row_shape)
    out = np.zeros((n_units,) + row_shape)
    if d is None or d.get_world_size(group) == 1:
        for i, r in local_rows.items():
            out[i] = r
        return out
    import torch
    rank, world = d.get_rank(group), d.get_world_size(group)
    per = (n_units + world - 1) // world
    buf = np.zeros((per,) + row_shape)
    for i, r in local_rows.items():
        assert owner_of(i, world) == rank
        buf[i // world] = r
    dev = _device_for(d, group)
    mine = torch.from_numpy(buf).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    d.all_gather(gathered, mine, group=group)
    for r, t in enumerate(gathered):
        a = t.cpu().numpy()
        for slot in range(per):
            i = slot * world + r
            if i < n_units:
                out[i] = a[slot]
    return out
