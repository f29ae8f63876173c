"""Host logic of the single-process multi-device path (anh_set_devices; include/annonet_hip.h): how a mini-batch / a tile list is
split over the replicas and which pixels the replicas exchange.  CPU only — the C ABI functions under test are pure host code;
the Python statements they are compared with are annonet_amd/dist.py's (the multi-process path) and plain arithmetic."""
import numpy as np

import annonet_amd as aa
from annonet_amd import dist as aad


def test_shard_range_is_a_partition_into_contiguous_near_equal_chunks():
    for n in (0, 1, 7, 32, 100, 257):
        for world in (1, 2, 3, 8):
            chunks = [aa.shard_range(n, world, r) for r in range(world)]
            assert chunks[0][0] == 0 and chunks[-1][1] == n
            assert all(chunks[i][1] == chunks[i + 1][0] for i in range(world - 1))      # contiguous, every unit exactly once
            sizes = [hi - lo for lo, hi in chunks]
            assert max(sizes) - min(sizes) <= 1
            assert chunks == [((n * r) // world, (n * (r + 1)) // world) for r in range(world)]   # = dist.shard_tiles' rule


def test_cross_replica_overlaps_equal_the_multi_process_statement():
    for (w, h, tile, ov) in ((300, 230, 96, 19), (1000, 700, 256, 35), (4096, 4096, 1024, 35)):
        tiles = aa.tiling.get_tiles(w, h, aa.tiling.parameters(tile, tile, ov, ov))
        for world in (1, 2, 3, 5, 8):
            got = aa.cross_replica_overlaps(tiles, world, w, h)
            want = aad.cross_rank_overlaps(tiles, world, w, h)
            assert got == [tuple(int(v) for v in r) for r in want]
            if world == 1:
                assert got == []


def test_exchanged_pixels_are_exactly_those_covered_by_tiles_of_two_replicas():
    w, h = 500, 420
    tiles = aa.tiling.get_tiles(w, h, aa.tiling.parameters(128, 128, 21, 21))
    world = 3
    owner = aad.tile_owner(len(tiles), world)
    cover = np.zeros((world, h, w), bool)
    for (full, _), o in zip(tiles, owner):
        l, t, r, b = max(full[0], 0), max(full[1], 0), min(full[2], w - 1), min(full[3], h - 1)
        cover[o, t:b + 1, l:r + 1] = True
    shared = cover.sum(0) >= 2
    mask = np.zeros((h, w), bool)
    for (l, t, r, b) in aa.cross_replica_overlaps(tiles, world, w, h):
        mask[t:b + 1, l:r + 1] = True
    np.testing.assert_array_equal(mask, shared)
