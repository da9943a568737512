"""The N > 1 path on CPU: SNP-range sharding + the gather of per-SNP records,
two gloo ranks.  The GPU kernels cannot run here, so each rank's records come
from the oracle on its shard of the synthetic table (global SNP index =
shard begin + row, as bench.py generates them); rank 0 checks the gathered
result against a single-process scan of the whole range."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = 96


def _init_gloo(dist, rdzv, rank, world):
    """Rendezvous through a file in the test's own tmp_path: no port is picked by anyone, so none can be taken."""
    dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)


def _records(oracle, seed, begin, count, n):
    table, g = oracle.sim_table_stream(seed, begin, count, n)
    cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
    counts, stats = oracle.scan_stream_cc(table, g, cs, ct, nc, nt)
    # the layout bench.py hands to RCCL: counts block then stats block
    return np.concatenate([counts.view(np.uint8).reshape(-1), stats.view(np.uint8).reshape(-1)])


def _worker(rank, world, rdzv, total, n, seed, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from libgwaspp_amd import shard
    from oracle import oracle
    _init_gloo(dist, rdzv, rank, world)
    try:
        sizes = []
        for r in range(world):
            b, e = shard.snp_range(total, r, world)
            sizes.append((e - b) * RECORD)
        begin, end = shard.snp_range(total, rank, world)
        local = torch.from_numpy(_records(oracle, seed, begin, end - begin, n).copy())
        got = shard.gather_ragged(dist, local, rank, world, sizes)
        # the chunked form bench.py uses (equal chunk sizes on every rank)
        even = (total // world) * world
        b2, e2 = shard.snp_range(even, rank, world)
        bounds = shard.chunk_bounds(e2 - b2, 3)
        chunks = []
        for c in range(3):
            rec = torch.from_numpy(_records(oracle, seed, b2 + bounds[c], bounds[c + 1] - bounds[c], n).copy())
            out, work = shard.gather_records(dist, rec, rank, world, async_op=True)
            chunks.append((out, work))
        for _, w in chunks:
            w.wait()
        if rank == 0:
            q.put(([t.numpy() for t in got], [[t.numpy() for t in out] for out, _ in chunks], bounds))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_snp_range_partition():
    from libgwaspp_amd import shard
    for total in (0, 1, 7, 8, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard.snp_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.snp_range(10, 2, 2)
    assert shard.chunk_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert shard.chunk_bounds(2, 4) == [0, 1, 2]


def test_pair_row_blocks_cover_the_triangle_once():
    from libgwaspp_amd import shard
    for n in (1, 7, 1000, 4097, 100_000):
        for world in (1, 2, 3, 8):
            seen = np.zeros(n, dtype=np.int32)
            pairs = []
            for r in range(world):
                blocks = shard.pair_row_blocks(n, r, world, block=512)
                for a, na in blocks:
                    seen[a:a + na] += 1
                pairs.append(sum(na * (n - a) - na * (na + 1) // 2 for a, na in blocks))
            assert np.all(seen == 1)
            assert sum(pairs) == n * (n - 1) // 2
            if n >= 100_000:
                assert max(pairs) / (sum(pairs) / world) < 1.02   # balanced


def test_two_rank_gather_equals_single_process(oracle, tmp_path):
    import torch.multiprocessing as mp
    world, total, n, seed = 2, 37, 3000, 0x5EED
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_worker, args=(r, world, rdzv, total, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, chunked, bounds = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from libgwaspp_amd import shard
    for r in range(world):
        b, e = shard.snp_range(total, r, world)
        assert np.array_equal(got[r], _records(oracle, seed, b, e - b, n))
    even = (total // world) * world
    for c in range(3):
        for r in range(world):
            b, _ = shard.snp_range(even, r, world)
            want = _records(oracle, seed, b + bounds[c], bounds[c + 1] - bounds[c], n)
            assert np.array_equal(chunked[c][r], want)


def _oracle_block_list(oracle, flat, m, blocks, threshold, capacity):
    """What gw_boost_screen_device leaves in HBM for a rank's row blocks, made by the oracle: the hits of
    [a, a + n) x [a, m) for each block appended to ONE fixed-capacity list (header: hits found, pairs evaluated, 0, 0)."""
    from libgwaspp_amd import capi
    iu = np.triu_indices(m, 1)
    mine = np.zeros(len(flat), dtype=bool)
    for a, na in blocks:
        mine |= (iu[0] >= a) & (iu[0] < a + na)
    hit = mine & (flat > threshold)            # NaN > threshold is false, as in the kernels
    recs = np.zeros(int(hit.sum()), dtype=capi.BOOST_HIT_DTYPE)
    recs["snp_a"], recs["snp_b"], recs["measure"] = iu[0][hit], iu[1][hit], flat[hit]
    raw = np.zeros(capi.BOOST_LIST_HEADER_BYTES + 16 * capacity, dtype=np.uint8)
    raw[:32].view(np.uint64)[:] = (len(recs), int(mine.sum()), 0, 0)
    kept = recs[:capacity]                      # a full list counts further hits without storing them
    raw[32:32 + 16 * len(kept)] = kept.view(np.uint8)
    return raw


def _pair_worker(rank, world, rdzv, m, n, seed, threshold, capacities, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from libgwaspp_amd import shard
    from oracle import oracle
    _init_gloo(dist, rdzv, rank, world)
    try:
        # every rank holds the whole table (config 5 replicates it) and screens the row blocks dealt to it
        table, g = oracle.sim_table_stream(seed, 0, m, n)
        cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
        flat = oracle.boost_all_pairs(table, g, cs, ct, nc, nt)
        blocks = shard.pair_row_blocks(m, rank, world, block=16)
        results = []
        caps = list(capacities)                 # what every rank knows about every rank's list (attempt 0: as configured)
        for attempt in range(3):
            raw = _oracle_block_list(oracle, flat, m, blocks, threshold, caps[rank])
            size = 32 + 16 * max(caps)          # the exchange is fixed-size: lists padded to the largest capacity in play
            padded = np.zeros(size, dtype=np.uint8)
            padded[:len(raw)] = raw
            parts = [torch.empty(size, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(padded))
            raws = [p.numpy()[:32 + 16 * c] for p, c in zip(parts, caps)]
            overflow = any(int(r[:8].view(np.uint64)[0]) > c for r, c in zip(raws, caps))
            hits, n_pairs, needed, outside = shard.union_boost_lists(raws, max(caps))
            results.append((attempt, overflow, needed, outside, len(hits), n_pairs))
            if not overflow:
                break
            caps = [needed] * world             # every rank read the same headers: the same decision everywhere
        if rank == 0:
            q.put((results, hits, n_pairs))
        else:
            q.put((results, None, None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_pair_screen_union_equals_single_process(oracle, tmp_path):
    """The config-5 step with two gloo ranks on CPU (SURVEY.md 8e "C5"; the loop sharded is epistasis_func.cpp:393-486): each
    rank screens the row blocks shard.pair_row_blocks deals it -- the oracle stands in for the kernel -- and writes the
    product's device-list format; the lists are all-gathered; the union parsed by capi.parse_boost_list /
    shard.union_boost_lists must be the screen of the whole triangle.  First attempt with a capacity of 4 on rank 1: its
    list overflows, EVERY rank sees that in the gathered headers and repeats with the room the header asks for."""
    import torch.multiprocessing as mp
    world, m, n, seed, threshold = 2, 60, 2000, 0x5EED, 6.0
    table, g = oracle.sim_table_stream(seed, 0, m, n)
    cs, ct, nc, nt = oracle.sim_phenotype_stream(seed, n)
    flat = oracle.boost_all_pairs(table, g, cs, ct, nc, nt)
    iu = np.triu_indices(m, 1)
    want = flat > threshold
    assert want.sum() > 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_pair_worker, args=(r, world, rdzv, m, n, seed, threshold, [4096, 4], q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    logs = [g_[0] for g_ in got]
    assert logs[0] == logs[1]                                   # both ranks took the same decisions from the same headers
    assert logs[0][0][1] is True and logs[0][0][2] > 4          # attempt 0: overflow seen, the header says how much room
    assert logs[0][-1][1] is False and logs[0][-1][3] == 0      # last attempt: everything stored, error model intact
    hits, n_pairs = next((h, p) for _, h, p in got if h is not None)
    assert n_pairs == m * (m - 1) // 2 and len(hits) == int(want.sum())
    assert np.array_equal(hits["snp_a"], iu[0][want]) and np.array_equal(hits["snp_b"], iu[1][want])
    assert np.array_equal(hits["measure"], flat[want])


def test_c_abi_shard_range_equals_the_python_partition():
    """gw_shard_range (what a C++ host uses, host-only) and shard.snp_range cut a job the same way."""
    import libgwaspp_amd as gw
    from libgwaspp_amd import shard
    for total in (0, 1, 7, 8, 1000, 1_000_000, 1_000_003):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert gw.shard_range(total, r, world) == shard.snp_range(total, r, world)
    with pytest.raises(gw.GwError):
        gw.shard_range(10, 2, 2)


def test_c_abi_pair_blocks_equal_the_python_partition():
    """gw_shard_pair_blocks (host-only) deals the row blocks of the pair triangle exactly as shard.pair_row_blocks does,
    adjacent blocks merged; every rank's share covers the triangle once."""
    import libgwaspp_amd as gw
    from libgwaspp_amd import shard
    for n in (0, 1, 7, 1000, 4097, 100_000):
        for world in (1, 2, 3, 8):
            pairs = 0
            for r in range(world):
                merged = []
                for a, na in sorted(shard.pair_row_blocks(n, r, world, block=512)):
                    if merged and merged[-1][0] + merged[-1][1] == a:
                        merged[-1] = (merged[-1][0], merged[-1][1] + na)
                    else:
                        merged.append((a, na))
                got = gw.shard_pair_blocks(n, r, world, 512)
                assert got == merged, (n, world, r)
                pairs += sum(na * (n - a) - na * (na + 1) // 2 for a, na in got)
            assert pairs == n * (n - 1) // 2
    with pytest.raises(gw.GwError):
        gw.shard_pair_blocks(10, 2, 2)


def test_missing_rccl_is_an_error_with_a_message_not_a_crash():
    """librccl is loaded at run time; when it cannot be, gw_comm_new_id / gw_comm_create return GW_E_HIP and
    gw_last_error() names every file tried (round 2 crashed here on a NULL dlerror())."""
    import subprocess
    import sys
    code = ("import libgwaspp_amd as gw\n"
            "for f in (gw.comm_new_id, lambda: gw.GwComm(0, 0, 1, bytes(128))):\n"
            "    try:\n"
            "        f(); raise SystemExit('no error')\n"
            "    except gw.GwError as e:\n"
            "        assert e.code == -2 and 'librccl not loadable' in str(e) and '/nonexistent/librccl.so' in str(e), str(e)\n"
            "print('ok')\n")
    env = dict(os.environ, GW_RCCL_LIBRARY="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
