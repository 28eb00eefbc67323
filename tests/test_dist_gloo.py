"""The N>1 path on CPU: world_size-2 gloo processes run BatchIntegrator with a stub backend.

What is checked is everything that is NOT a kernel: the single all-gather of header + list slots with ragged list
lengths (padding, trimming, agreed capacity growth), and that every rank applies the lists of ranks 0..N-1 in rank order with
the exact bytes the producing rank emitted -- the property that makes the replicas identical to
sequential integration."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ufomap_amd.capi import KeysInfo
from ufomap_amd.dist import ENTRY_BYTES, BatchIntegrator


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class StubBackend:
    """Produces a deterministic, rank-dependent ragged update list; records what it is told to apply."""

    def __init__(self, rank, sizes, color=False):
        self.rank, self.sizes, self.step, self.applied, self.color = rank, sizes, 0, [], color

    def scan(self, origin, d_xyz_ptr, n, max_range, depth, discrete, d_rgb_ptr=None):
        n_hit, n_miss = self.sizes[self.step % len(self.sizes)][self.rank]
        rng = np.random.default_rng(1000 * self.step + self.rank)
        # colour maps: a colour section (32 bytes per hit record) follows the records, flagged in `reserved` bit 1
        nbytes = (n_hit + n_miss) * ENTRY_BYTES + (n_hit * 32 if self.color else 0)
        payload = torch.from_numpy(rng.integers(0, 256, nbytes, dtype=np.uint8))
        info = KeysInfo.from_list([n_hit, n_miss, 3 + self.rank, 4, 5, 6, 7, 8 + self.rank, depth, 2 if self.color else 0])
        self.step += 1
        return payload, torch.tensor(info.to_list(), dtype=torch.int32)

    def apply(self, rank, header_row, payload):
        self.applied.append((rank, header_row.tolist(), hashlib.sha256(payload.numpy().tobytes()).hexdigest(), payload.numel()))


def _worker(rank, world, port, sizes, q, color=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = StubBackend(rank, sizes, color)
    bi = BatchIntegrator(backend=be, group=dist.group.WORLD, initial_cap=1024)  # small: the growth path runs too
    for _ in range(len(sizes)):
        bi.integrate(np.zeros(3), 0, 0, 20.0, 0, True, d_rgb_ptr=(1 if color else None))
    q.put((rank, be.applied))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("color", [False, True])
def test_two_rank_exchange_applies_all_lists_in_rank_order(color):
    world = 2
    # per step: [(n_hit, n_miss) for rank 0, for rank 1]; ragged, one empty list, one empty step
    sizes = [[(5, 40), (9, 13)], [(0, 0), (3, 7)], [(128, 1000), (1, 0)], [(0, 0), (0, 0)]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, q, color)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # what each rank should have seen: for every step, rank 0's list then rank 1's list, bytes intact
    expect = []
    for step, per_rank in enumerate(sizes):
        for r, (nh, nm) in enumerate(per_rank):
            rng = np.random.default_rng(1000 * step + r)
            raw = rng.integers(0, 256, (nh + nm) * ENTRY_BYTES + (nh * 32 if color else 0), dtype=np.uint8).tobytes()
            hdr = [nh, nm, 3 + r, 4, 5, 6, 7, 8 + r, 0, 2 if color else 0]
            expect.append((r, hdr, hashlib.sha256(raw).hexdigest(), len(raw)))
    assert got[0] == expect
    assert got[1] == expect
