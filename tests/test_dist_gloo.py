"""The N>1 logic on CPU with world_size-2 gloo (torch.distributed): the data-parallel step is
    forward_backward(loss scale = GLOBAL batch) -> all-reduce(SUM) of the flat gradient bucket (+ trailing loss slot) -> identical update
(SURVEY.md §8e).  The compute here is the oracle standing in for a rank's GPU (allowed: tests/ may use the oracle as the
checker); what is under test is the exchange protocol in annonet_amd/dist.py and its invariants:
  * every rank ends the step with identical parameters;
  * the summed bucket equals the gradient of the global-batch loss with per-rank batch-norm statistics (the declared
    BN policy), computed in one process;
  * tile sharding covers every tile exactly once, in row-major order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from annonet_amd import dist as aad
from conftest import random_params
from oracle import oracle as orc
from oracle.oracle import OracleNet

CFG = dict(levels=1, in_ch=3, classes=3, scaler=0.25, min_filters=4)


def make_rank_batch(rank, d):
    rng = np.random.default_rng(100 + rank)
    img = rng.integers(0, 256, (2, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (2, d, d)).astype(np.uint16)
    w = np.stack([orc.set_weights(lab[i], 0.5, 0.5) for i in range(2)])
    return img, lab, w


class OracleTrainer:
    """Duck-types the TrainingNet methods data_parallel_step uses, with the oracle as the compute."""

    def __init__(self):
        self.net = OracleNet(**CFG)
        p, r = random_params(self.net, 5)
        self.net.params[:], self.net.running[:] = p, r
        self.net.set_hyper(lr=0.05)
        self.bucket = torch.zeros(self.net.n_params + 1, dtype=torch.float32)
        self.batch = None

    def forward_backward_device(self, d_images, d_labels, d_weights, n, h, w, loss_scale_n):
        img, lab, wt = self.batch
        loss = self.net.train_step(img, lab, wt, loss_scale_n=loss_scale_n, apply_update=False)
        self.bucket[:-1] = torch.from_numpy(self.net.grads.copy())
        self.bucket[-1] = loss

    def apply_update(self, grad_scale):
        g = self.bucket[:-1].numpy() * grad_scale
        lr, wd, mom = 0.05, 0.0005, 0.9
        mask = np.zeros(self.net.n_params)
        for L in self.net.layers:
            mask[L.w_off:L.w_off + L.k * L.k * L.cin * L.cout] = 1
        v = mom * self.net.momentum - wd * lr * mask * self.net.params - lr * g
        self.net.momentum[:] = v
        self.net.params[:] = self.net.params + v


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = OracleTrainer()
    d = t.net.recommended_input_dim(15)
    t.batch = make_rank_batch(rank, d)
    aad.data_parallel_step(t, t.bucket, 0, 0, 0, 2, d, d, world)
    gathered = [torch.zeros_like(torch.from_numpy(t.net.params.copy())) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(t.net.params.copy()))
    if rank == 0:
        out["params"] = [g.numpy() for g in gathered]
        out["bucket"] = t.bucket.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_data_parallel_step_over_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(2, free_port(), out), nprocs=2, join=True)
    p0, p1 = out["params"]
    np.testing.assert_array_equal(p0, p1)  # identical update on every rank
    # single-process statement of the same step: per-rank BN statistics, loss scale = global batch, gradients summed
    want_g, want_loss = None, 0.0
    d = OracleNet(**CFG).recommended_input_dim(15)
    for rank in range(2):
        net = OracleNet(**CFG)
        p, r = random_params(net, 5)
        net.params[:], net.running[:] = p, r
        img, lab, w = make_rank_batch(rank, d)
        want_loss += net.train_step(img, lab, w, loss_scale_n=4, apply_update=False)
        want_g = net.grads.copy() if want_g is None else want_g + net.grads
    np.testing.assert_allclose(out["bucket"][:-1], want_g, rtol=1e-6, atol=1e-9)
    assert abs(out["bucket"][-1] - want_loss) < 1e-6


def exchange_worker(rank, world, port, out):
    """The exchange step of sharded inference over a real process group: every rank holds planes that are non-zero only
    inside its own tiles (stand-ins for its blended partial sums); after OverlapExchange.run the pixels shared with other
    ranks hold the sum over ranks, every other pixel is untouched."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, K = 301, 230, 3
    tiles = orc.get_tiles(W, H, 96, 112, 19, 19)
    owner = aad.tile_owner(len(tiles), world)
    planes = np.zeros((K, H, W), dtype=np.float32)
    rng = np.random.default_rng(7 + rank)
    for (full, _), r in zip(tiles, owner):
        if r == rank:
            t, b, l, rr = max(full[1], 0), min(full[3], H - 1), max(full[0], 0), min(full[2], W - 1)
            planes[:, t:b + 1, l:rr + 1] += rng.normal(size=(K, b - t + 1, rr - l + 1)).astype(np.float32)
    before = planes.copy()
    tens = torch.from_numpy(planes)
    ex = aad.OverlapExchange(tiles, world, W, H, torch.device("cpu"))
    ex.run(tens)
    gathered = [torch.zeros((K, H, W)) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(before))
    if rank == 0:
        out["after"] = tens.numpy().copy()
        out["before"] = [g.numpy() for g in gathered]
        out["pixels"] = ex.pixels()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlap_exchange_over_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(exchange_worker, args=(2, free_port(), out), nprocs=2, join=True)
    W, H = 301, 230
    tiles = orc.get_tiles(W, H, 96, 112, 19, 19)
    shared = np.zeros((H, W), dtype=bool)
    for (l, t, r, b) in aad.cross_rank_overlaps(tiles, 2, W, H):
        shared[t:b + 1, l:r + 1] = True
    assert out["pixels"] == int(shared.sum()) > 0
    b0, b1 = out["before"]
    want = np.where(shared[None], b0 + b1, b0)      # rank 0's planes: sums where shared, its own values elsewhere
    np.testing.assert_array_equal(out["after"], want)


def gather_worker(rank, world, port, out):
    """The last step of a multi-rank annonet_infer(): every rank holds the right labels where its own tiles cover the image and
    arbitrary values elsewhere; LabelGather.run leaves the ONE complete map on rank 0 (incl. the all-NaN label 65535)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, K = 301, 230, 5
    tiles = orc.get_tiles(W, H, 96, 112, 19, 19)
    truth = np.random.default_rng(99).integers(0, K, size=(H, W)).astype(np.uint16)
    truth[::17, ::13] = 65535
    mine = np.zeros((H, W), dtype=bool)
    for (full, _) in aad.shard_tiles(tiles, rank, world):
        mine[max(full[1], 0):min(full[3], H - 1) + 1, max(full[0], 0):min(full[2], W - 1) + 1] = True
    held = np.where(mine, truth, np.uint16(K - 1 - rank % K))          # wrong on purpose outside this rank's tiles
    gather = aad.LabelGather(tiles, world, rank, W, H, torch.device("cpu"), K)
    whole = gather.run(torch.from_numpy(held.view(np.int16).copy()))
    assert (whole is None) == (rank != 0)
    if rank == 0:
        out["whole"] = whole.numpy().view(np.uint16).copy()
        out["truth"] = truth
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_label_gather_over_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(gather_worker, args=(3, free_port(), out), nprocs=3, join=True)
    np.testing.assert_array_equal(out["whole"], out["truth"])
    with pytest.raises(ValueError):
        aad.LabelGather([], 1, 0, 4, 4, torch.device("cpu"), 300)


def host_map_worker(rank, world, port, out):
    """Per-rank label delivery: every rank writes the cells of ITS tiles into ONE shared-memory host map;
    rank 0 creates the map and publishes its name, the others attach.  The copies here come from host arrays (deliver_from_host: the
    ownership logic); on a GPU the same rectangles travel by anh_labels_rect_to_host (tests/test_gpu_sharded_infer.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, K = 301, 230, 5
    tiles = orc.get_tiles(W, H, 96, 112, 19, 19)
    truth = np.random.default_rng(98).integers(0, K, size=(H, W)).astype(np.uint16)
    mine = np.zeros((H, W), dtype=bool)
    for (full, _) in aad.shard_tiles(tiles, rank, world):
        mine[max(full[1], 0):min(full[3], H - 1) + 1, max(full[0], 0):min(full[2], W - 1) + 1] = True
    held = np.where(mine, truth, np.uint16(60000 + rank))          # wrong on purpose outside this rank's tiles
    names = [None]
    hm = None
    if rank == 0:
        hm = aad.HostLabelMap(tiles, world, rank, W, H, pin=False)
        hm.array[:] = 12345
        names = [hm.name]
    dist.broadcast_object_list(names, src=0)
    if rank != 0:
        hm = aad.HostLabelMap(tiles, world, rank, W, H, name=names[0], pin=False)
    hm.deliver_from_host(held)
    sizes = [None] * world
    dist.all_gather_object(sizes, hm.bytes)
    dist.barrier()
    if rank == 0:
        out["map"] = hm.array.copy()
        out["truth"] = truth
        out["bytes"] = sizes
    dist.barrier()
    hm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_every_rank_delivers_its_own_rows_into_one_shared_host_map(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(host_map_worker, args=(world, free_port(), out), nprocs=world, join=True)
    np.testing.assert_array_equal(out["map"], out["truth"])
    assert sum(out["bytes"]) == 301 * 230 * 2 and all(b > 0 for b in out["bytes"])     # every byte written by exactly one rank
    # the cells partition the image and lie inside their tiles; a tile list that is not a grid is refused
    tiles = orc.get_tiles(301, 230, 96, 112, 19, 19)
    cells = aad.tile_cells(tiles, 301, 230)
    cover = np.zeros((230, 301), dtype=int)
    for (full, _), c in zip(tiles, cells):
        assert full[0] <= c[0] <= c[2] <= full[2] and full[1] <= c[1] <= c[3] <= full[3]
        cover[c[1]:c[3] + 1, c[0]:c[2] + 1] += 1
    assert (cover == 1).all()
    with pytest.raises(ValueError):
        aad.tile_cells([((0, 0, 9, 9), (0, 0, 9, 9)), ((3, 3, 19, 19), (10, 10, 19, 19))], 20, 20)


def test_cross_rank_overlaps_are_exactly_the_pixels_shared_between_ranks():
    """The exchange step of sharded inference moves the plane sums of these pixels and no others."""
    W, H = 301, 230
    tiles = orc.get_tiles(W, H, 96, 112, 19, 19)
    for world in (1, 2, 3, 7):
        owner = aad.tile_owner(len(tiles), world)
        assert owner == sorted(owner) and [owner.count(r) for r in range(world)] == [len(aad.shard_tiles(tiles, r, world)) for r in range(world)]
        ranks_at = [np.zeros((H, W), dtype=bool) for _ in range(world)]
        for (full, _), r in zip(tiles, owner):
            ranks_at[r][max(full[1], 0):min(full[3], H - 1) + 1, max(full[0], 0):min(full[2], W - 1) + 1] = True
        shared = np.sum(ranks_at, axis=0) >= 2
        mask = np.zeros((H, W), dtype=bool)
        for (l, t, r, b) in aad.cross_rank_overlaps(tiles, world, W, H):
            mask[t:b + 1, l:r + 1] = True
        np.testing.assert_array_equal(mask, shared)
        assert (world == 1) == (not mask.any())


def test_tile_sharding_partitions_the_tile_list():
    tiles = orc.get_tiles(4096, 3000, 1024, 1024, 35, 35)
    for world in (1, 2, 3, 8, 40):
        shards = [aad.shard_tiles(tiles, r, world) for r in range(world)]
        assert sum(shards, []) == tiles
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    a = np.arange(12, dtype=np.float32).reshape(3, 2, 2)
    np.testing.assert_array_equal(aad.reduce_shard_planes([a, 2 * a]), 3 * a)
