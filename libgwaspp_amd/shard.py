"""SNP-range sharding of the scan across ranks and the gather of the per-SNP
records (plumbing over torch.distributed; RCCL on GPUs, gloo in the CPU tests).

Every SNP row is independent (algorithms/maf_func.cpp:304-314 touches only row
i and the shared masks), so the path shards by contiguous SNP ranges with no
data-path collective; the one exchange step is the gather of the 96-byte
records to rank 0."""
from __future__ import annotations


def snp_range(n_snps_total: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of the rank's contiguous shard; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank outside the world")
    base, extra = divmod(n_snps_total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def chunk_bounds(n_rows: int, n_chunks: int) -> list[int]:
    n_chunks = max(1, min(n_chunks, max(1, n_rows)))
    return [(n_rows * c) // n_chunks for c in range(n_chunks + 1)]


def gather_records(dist, local, rank: int, world: int, dst: int = 0, async_op: bool = False, out=None):
    """Gathers equally sized 1-D uint8 record buffers to `dst`.  `out` may be a
    preallocated list of `world` tensors on dst.  Returns (list of per-rank
    tensors on dst | None, work | None)."""
    import torch
    if world == 1:
        return [local], None
    if rank != dst:
        out = None
    elif out is None:
        out = [torch.empty_like(local) for _ in range(world)]
    work = dist.gather(local, out, dst=dst, async_op=async_op)
    return out, work


def gather_ragged(dist, local, rank: int, world: int, sizes: list[int], dst: int = 0):
    """Shards whose sizes differ (n_snps_total % world != 0): pad to the
    largest, gather, trim on dst."""
    import torch
    biggest = max(sizes)
    if local.numel() < biggest:
        local = torch.cat([local, local.new_zeros(biggest - local.numel())])
    out, _ = gather_records(dist, local, rank, world, dst)
    if rank != dst:
        return None
    return [t[:s] for t, s in zip(out, sizes)]


def pair_row_blocks(n_snps: int, rank: int, world: int, block: int = 1024) -> list[tuple[int, int]]:
    """Sharding of the SNP-pair space {(i, j): i < j} for the BOOST pre-screen: the rows i are cut
    into blocks of `block` SNPs, dealt over the ranks in snake order (0..W-1, W-1..0, ...: row
    block k holds about block * (n_snps - k*block) pairs, so the snake balances the triangle to
    well under 1 %); the rank screens
    [a, a+n) x [a, n_snps) for each of its blocks.  Every rank holds the whole table (the reference's
    computeBoost needs every row against every other; 100 k SNPs x 10 k samples is 250 MB).
    Returns [(a_begin, n_a), ...]."""
    if not (0 <= rank < world):
        raise ValueError("rank outside the world")
    out = []
    n_blocks = (n_snps + block - 1) // block
    for k in range(n_blocks):
        rnd, pos = divmod(k, world)
        owner = pos if rnd % 2 == 0 else world - 1 - pos
        if owner == rank:
            a = k * block
            out.append((a, min(block, n_snps - a)))
    return out


def union_boost_lists(raws, capacity: int):
    """The exchange step of the sharded pair screen seen from any rank after the all-gather: `raws` = every rank's hit
    list as bytes (gw_boost_screen_device's layout: 32-byte header {hits, pairs, aux, aux} + capacity records).
    -> (hits of all ranks sorted by (snp_a, snp_b), pairs evaluated by all ranks, needed_capacity, outside_model):
    needed_capacity > capacity means some rank's list overflowed -- every rank sees the same headers, so every rank takes
    the same decision to repeat the step with that much room (hits beyond a full list were counted, not stored);
    outside_model != 0 means a rank's pre-filter saw a pair outside its error model (or lost candidates) and the step is
    to be repeated with every pair in FP64 (algorithms::shardedBoostScreen does the same in C++)."""
    import numpy as np
    from . import capi
    parsed = [capi.parse_boost_list(np.asarray(r, dtype=np.uint8), capacity) for r in raws]
    needed = max([p[1] for p in parsed] + [0])
    outside = sum(p[3] for p in parsed)
    hits = np.concatenate([p[0] for p in parsed]) if parsed else np.zeros(0, dtype=capi.BOOST_HIT_DTYPE)
    return np.sort(hits, order=["snp_a", "snp_b"]), sum(p[2] for p in parsed), needed, outside
