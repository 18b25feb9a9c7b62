"""Sequence sharding over the GPUs of one node.

Every fold is independent (the reference parallelises the same way: one CLI process
per sequence, benchmark_results/bench_fft.py:10-22), so a batch is partitioned by
longest-processing-time-first over ranks and results are gathered on the host.
No data-path collective: the only communication is the final gather of result
records (dot-bracket strings + dcal)."""
import heapq


def lpt_shards(lengths, n_shards, cost=lambda L: L * L + 64 * L):
    """Greedy LPT partition of sequence indices; returns n_shards index lists.
    Cost proxy: L^2 (correlation/scan work) + a per-sequence term."""
    order = sorted(range(len(lengths)), key=lambda i: -cost(lengths[i]))
    heap = [(0, r) for r in range(n_shards)]
    heapq.heapify(heap)
    shards = [[] for _ in range(n_shards)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(i)
        heapq.heappush(heap, (load + cost(lengths[i]), r))
    for s in shards:
        s.sort()
    return shards


def fold_sharded(sequences, fold_fn=None, dist=None, device=None, **fold_kwargs):
    """Fold `sequences` across the ranks of an initialised torch.distributed group.

    Every rank calls this with the same `sequences`; rank r folds shard r with
    `fold_fn(list_of_sequences, **fold_kwargs)` (default: rafft_amd.fold_batch on this
    rank's GPU) and rank 0 returns the results in input order (other ranks: None)."""
    if fold_fn is None:
        from .rafft import fold_batch as fold_fn
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    shards = lpt_shards([len(s) for s in sequences], world)
    mine = shards[rank]
    kw = dict(fold_kwargs)
    if device is not None:
        kw["device"] = device
    local = fold_fn([sequences[i] for i in mine], **kw) if mine else []
    payload = list(zip(mine, local))
    if world == 1:
        gathered = [payload]
    else:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(payload, gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * len(sequences)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out
