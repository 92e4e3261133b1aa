"""World-size-2 gloo tests (CPU) of the multi-GPU path: independent images shard over ranks with no
data-path collective; loss records are all-reduced."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from convex_optimization_amd.distributed import allreduce_loss_records, global_stop, shard_range


def test_shard_range_partitions_exactly():
    for n in [0, 1, 7, 8, 1024, 3072, 3073]:
        for world in [1, 2, 3, 4, 8]:
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_items, rank, world)
        iters = 4
        # rank-local "loss records": value encodes (iteration, global image index)
        ring = torch.zeros(iters, hi - lo, 8, dtype=torch.float64)
        for it in range(iters):
            for i in range(lo, hi):
                ring[it, i - lo, :] = 100.0 * it + i
        full = allreduce_loss_records(ring, n_items, lo)
        expect = torch.zeros(iters, n_items, 8, dtype=torch.float64)
        for it in range(iters):
            for i in range(n_items):
                expect[it, i, :] = 100.0 * it + i
        ok = torch.equal(full, expect)
        # stop only when every rank's images converged
        s1 = global_stop(torch.tensor([1] * (hi - lo)))
        s2 = global_stop(torch.tensor([1] * (hi - lo)) if rank == 0 else torch.tensor([1] * max(hi - lo - 1, 0) + [0]))
        out[rank] = (ok, s1, s2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (4, 11), (8, 8)])
def test_world2_gloo_loss_allreduce(world, n_items):
    """Loss records of rank-local images land at their global image index on every rank, and the global stop needs
    every rank's images: two, four and eight ranks (uneven and one-image-per-rank shards)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_items, out), nprocs=world, join=True)
    for r in range(world):
        ok, s1, s2 = out[r]
        assert ok and s1 is True and s2 is False


# ------------------------------------------------------------------ row-slab decomposition of one image
def _code(tile, spec, row, col):
    return complex(1000.0 * tile + 100.0 * spec + col, float(row))


def _slab_worker(rank, world, port, H, C, n_tiles, out):
    from convex_optimization_amd.slab import SlabExchange, SlabLayout, combine_loss_records
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = SlabLayout(H, 2 * C * (n_tiles - 1), C, n_tiles, rank, world)
        x = SlabExchange(lay, torch.complex64, "cpu")
        for tl in range(n_tiles):
            for s in range(2):
                for r in range(lay.rows):
                    for c in range(C):
                        x.spec_rows[tl, s, r, c] = _code(tl, s, lay.row_lo + r, c)
        sent = x.spec_rows.clone()
        x.rows_to_cols()
        ok_cols = True
        for s in range(2):
            for lt in range(lay.local_tiles):
                for h in range(H):
                    for c in range(C):
                        # blocked: [source rank][tile][spectrum][rows][C] as delivered; else rows put in order
                        got = x.spec_cols[h // lay.rows, lt, s, h % lay.rows, c] if x.blocked else x.spec_cols[s, lt, h, c]
                        ok_cols &= complex(got) == _code(lay.tile_lo + lt, s, h, c)
        x.spec_rows.zero_()
        x.cols_to_rows()
        ok_back = torch.equal(x.spec_rows, sent)
        # loss records: l1 and l2^2 add up over the slabs
        local = torch.zeros(3, 8, dtype=torch.float64)
        local[:, 0] = rank + 1.0
        local[:, 1] = 3.0 if rank == 0 else 4.0
        local[:, 4] = 2.0 * (rank + 1.0)
        local[:, 5] = 1.0
        tot = combine_loss_records(local, 100.0)
        ok_loss = bool(torch.allclose(tot[:, 0], torch.full((3,), 3.0, dtype=torch.float64)) and torch.allclose(tot[:, 1], torch.full((3,), 5.0, dtype=torch.float64))
                       and torch.allclose(tot[:, 2], torch.full((3,), 0.25, dtype=torch.float64)) and torch.allclose(tot[:, 5], torch.full((3,), 2.0 ** 0.5, dtype=torch.float64)))
        out[rank] = (bool(ok_cols), bool(ok_back), ok_loss, lay.tile_spans, x.blocked)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H,C,n_tiles", [(8, 4, 5), (6, 2, 2)])
def test_world2_gloo_slab_transposes(H, C, n_tiles):
    """rows -> columns all-to-all puts every rank's rows of a tile at their global row index; the way
    back restores the row-side buffer; uneven tile spans (the tile of column W/2 makes the count odd)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_slab_worker, args=(world, _free_port(), H, C, n_tiles, out), nprocs=world, join=True)
    for r in range(world):
        ok_cols, ok_back, ok_loss, spans, blocked = out[r]
        assert ok_cols and ok_back and ok_loss
        assert blocked == (H // world in (1, 2, 4, 8))   # (8, ...) runs the blocked layout, (6, ...) the repacking one
        assert spans[0][0] == 0 and spans[-1][1] == n_tiles


def _slab_worker_groups(rank, world, port, H, C, n_tiles, groups, out):
    """The exchanges of one image in row slabs with the tile counts of the large configurations, built and checked with
    tensor expressions: the single all-to-all, and the tile groups of the pipelined column stage issued one by one in
    the order SlabExchange.pipelined issues them."""
    from convex_optimization_amd.slab import SlabExchange, SlabLayout
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = SlabLayout(H, 2 * C * (n_tiles - 1), C, n_tiles, rank, world)

        def fill_rows(x):
            tl = torch.arange(n_tiles).view(-1, 1, 1, 1)
            sp = torch.arange(2).view(1, -1, 1, 1)
            rw = (lay.row_lo + torch.arange(lay.rows)).view(1, 1, -1, 1)
            cc = torch.arange(C).view(1, 1, 1, -1)
            x.spec_rows.copy_(torch.complex((1000.0 * tl + 100.0 * sp + cc).float().expand(n_tiles, 2, lay.rows, C),
                                            rw.float().expand(n_tiles, 2, lay.rows, C)))

        def cols_ok(x):
            # blocked layout: [source rank][local tile][spectrum][rows][C]
            src = torch.arange(world).view(-1, 1, 1, 1, 1)
            lt = (lay.tile_lo + torch.arange(lay.local_tiles)).view(1, -1, 1, 1, 1)
            sp = torch.arange(2).view(1, 1, -1, 1, 1)
            rr = torch.arange(lay.rows).view(1, 1, 1, -1, 1)
            cc = torch.arange(C).view(1, 1, 1, 1, -1)
            shape = (world, lay.local_tiles, 2, lay.rows, C)
            want = torch.complex((1000.0 * lt + 100.0 * sp + cc).float().expand(shape), (src * lay.rows + rr).float().expand(shape))
            return bool(torch.equal(x.spec_cols, want))

        x = SlabExchange(lay, torch.complex64, "cpu")
        assert x.blocked
        fill_rows(x)
        sent = x.spec_rows.clone()
        x.rows_to_cols()
        ok1 = cols_ok(x)
        x.spec_rows.zero_()
        x.cols_to_rows()
        ok1 &= bool(torch.equal(x.spec_rows, sent))
        # the pipelined exchange: groups of tiles, the same range rule on the sending and the receiving side
        xg = SlabExchange(lay, torch.complex64, "cpu", exchange_groups=groups)
        fill_rows(xg)
        G = xg.groups
        covered = torch.zeros(lay.local_tiles, dtype=torch.int32)
        for g in range(G):
            a, b = xg.group_range(lay.local_tiles, g)
            covered[a:b] += 1
        ok_cover = bool((covered == 1).all())
        order = list(range(min(2, G)))                       # inbound runs two groups ahead (SlabExchange.pipelined)
        for g in range(G):
            if g + 2 < G:
                order.append(g + 2)
        for g in order:
            xg._a2a_group(g, True)
        ok2 = cols_ok(xg)
        xg.spec_rows.zero_()
        for g in range(G):
            xg._a2a_group(g, False)
        ok2 &= bool(torch.equal(xg.spec_rows, sent))
        out[rank] = (ok1, ok2, ok_cover, G, lay.tile_spans)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,C,n_tiles,groups", [(4, 8, 2, 129, 4), (8, 8, 2, 129, 4), (8, 16, 1, 1025, 8), (4, 4, 1, 1025, 5), (2, 4, 2, 129, 4)])
def test_gloo_slab_exchanges_with_the_tile_counts_of_the_large_images(world, H, C, n_tiles, groups):
    """Four and eight ranks with the spectral tile counts of a 4096-pixel-wide image (129 tiles of 16 columns: an odd
    count from column W/2, so the spans are uneven) and of a 32768-pixel-wide one (1025 tiles); the plain exchange and
    the tile groups of the pipelined one, in both directions."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_slab_worker_groups, args=(world, _free_port(), H, C, n_tiles, groups, out), nprocs=world, join=True)
    sizes = []
    for r in range(world):
        ok1, ok2, ok_cover, G, spans = out[r]
        assert ok1 and ok2 and ok_cover, (r, ok1, ok2, ok_cover)
        assert G == groups
        sizes = [hi - lo for lo, hi in spans]
        assert spans[0][0] == 0 and spans[-1][1] == n_tiles and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(sizes) - min(sizes) == 1          # the odd tile count: uneven spans


def test_slab_layout_rejects_ragged_rows():
    from convex_optimization_amd.slab import SlabLayout
    with pytest.raises(ValueError):
        SlabLayout(10, 16, 4, 3, 0, 4)
