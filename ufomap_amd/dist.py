"""Batched multi-sensor integration across GPUs (BASELINE config C4).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU
tests).  Per batch every rank ray-casts ITS scan -- the part of the path that never reads the map --
into an *update list* (16-byte records, ``include/ufomap_hip.h``), the lists are exchanged with ONE
padded all-gather (header and list in one fixed-size slot per rank), and every rank applies the lists of
ranks 0..N-1 in rank order to its replica of the map -- with ONE walk of the tree for the whole batch
(``ufomap_map_apply_keys_batch``).  Applying in order reproduces the reference's
sequential integration bit-exactly on every replica; a float all-reduce of log-odds deltas would not
(clamping after every hit phase and every miss phase is not associative).

The collective logic is independent of the device: ``BatchIntegrator.exchange`` works on CPU tensors with gloo
(tests/test_dist_gloo.py) exactly as it does on HBM tensors with RCCL.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .capi import KeysInfo

ENTRY_BYTES = 16


class MapBackend:
    """The real thing: scan_keys / get_keys / apply_keys of the HIP library on one GPU."""

    overlaps = True  # apply_all returns after enqueueing: the integrator must not reuse the lists' memory at once

    def __init__(self, occupancy_map, device):
        self.m = occupancy_map
        self.device = device
        self.m.set_option("async_apply", 1)
        self._buf = torch.empty(0, dtype=torch.uint8, device=device)

    def scan(self, origin, d_xyz_ptr, n, max_range, depth, discrete, d_rgb_ptr=None):
        info = self.m.scan_keys(origin, d_xyz_ptr, n, max_range, depth, discrete, d_rgb_ptr=d_rgb_ptr)
        nbytes = info.list_bytes  # records + colour section (colour maps)
        if self._buf.numel() < nbytes:
            self._buf = torch.empty(max(nbytes, 2 * self._buf.numel()), dtype=torch.uint8, device=self.device)
        if nbytes:
            self.m.get_keys(self._buf.data_ptr(), self._buf.numel() // ENTRY_BYTES, info)  # syncs the map's stream
        header = torch.tensor(info.to_list(), dtype=torch.int32, device=self.device)
        return self._buf[:nbytes], header

    def join(self):
        self.m.insertPointCloudWait()

    def apply(self, rank, header_row, payload):
        info = KeysInfo.from_list(header_row.tolist())
        if info.n_hit + info.n_miss:
            self.m.apply_keys(payload.data_ptr(), info)

    def apply_all(self, headers, lists):
        """All ranks' lists in rank order. Depth-0 scans go through ONE walk of the tree
        (``ufomap_map_apply_keys_batch``); a batch with deeper scans is applied list by list."""
        infos = [KeysInfo.from_list(headers[r].tolist()) for r in range(len(lists))]
        if all(k.depth == 0 for k in infos) and len(infos) <= 128:
            self.m.apply_keys_batch([lists[r].data_ptr() if lists[r].numel() else 0 for r in range(len(lists))], infos)
        else:
            for r in range(len(lists)):
                self.apply(r, headers[r], lists[r])


HDR_BYTES = 64  # KeysInfo as int32[10], padded: travels in front of the list, in the same collective


class BatchIntegrator:
    """``integrate`` = one batch step: scan locally, exchange, apply every rank's list in rank order.

    The exchange is ONE ``all_gather_into_tensor`` per step: every rank contributes a fixed-size slot
    (header + list + padding) of a capacity all ranks agree on; persistent buffers, no per-step
    allocation. If some rank's list does not fit, every rank sees that in the gathered headers, all
    double the capacity to the same value and the exchange is repeated (rare: the capacity only grows)."""

    def __init__(self, occupancy_map=None, group=None, device=None, backend=None, initial_cap=1 << 20):
        self.group = group
        self.backend = backend if backend is not None else MapBackend(occupancy_map, device)
        self._cap = max(int(initial_cap), HDR_BYTES)
        self._send = self._recv = None

    def _buffers(self, dev, world):
        if self._send is None or self._send.numel() != self._cap or self._send.device != dev:
            self._send = torch.zeros(self._cap, dtype=torch.uint8, device=dev)
            # two receive buffers, used alternately: the update of batch i may still read its lists from one
            # while batch i+1 is gathered into the other (MapBackend applies asynchronously)
            self._recv2 = [torch.empty((world, self._cap), dtype=torch.uint8, device=dev) for _ in range(2)]
            self._flip = 0
        return self._send, self._recv2[self._flip]

    def exchange(self, payload, header):
        """Returns (headers [world, WORDS] int32 on CPU, lists: per rank a uint8 view of the gathered buffer)."""
        world = dist.get_world_size(self.group)
        dev = payload.device
        hdr_bytes = header.to(torch.int32).contiguous().view(torch.uint8)
        if self._send is not None:
            self._flip ^= 1
        while True:
            send, recv = self._buffers(dev, world)
            send[: hdr_bytes.numel()] = hdr_bytes.to(dev)
            fits = HDR_BYTES + payload.numel() <= self._cap
            if fits and payload.numel():
                send[HDR_BYTES: HDR_BYTES + payload.numel()] = payload
            dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
            headers = recv[:, : 4 * KeysInfo.WORDS].contiguous().view(torch.int32).view(world, KeysInfo.WORDS).cpu()
            nbytes = (headers[:, 0].to(torch.int64) + headers[:, 1].to(torch.int64)) * ENTRY_BYTES
            nbytes = nbytes + torch.where((headers[:, 9] & 2) != 0, headers[:, 0].to(torch.int64) * 32, torch.zeros_like(nbytes))  # colour sections
            need = HDR_BYTES + int(nbytes.max().item())
            if need <= self._cap:
                return headers, [recv[r, HDR_BYTES: HDR_BYTES + int(nbytes[r])] for r in range(world)]
            while self._cap < need:  # same decision on every rank: all see the same headers
                self._cap *= 2
            if hasattr(self.backend, "join"):
                self.backend.join()  # an update still reading the old receive buffers finishes before they go

    def integrate(self, origin, d_xyz_ptr, n, max_range=-1.0, depth=0, discrete=True, d_rgb_ptr=None):
        payload, header = (self.backend.scan(origin, d_xyz_ptr, n, max_range, depth, discrete, d_rgb_ptr) if d_rgb_ptr
                           else self.backend.scan(origin, d_xyz_ptr, n, max_range, depth, discrete))
        headers, lists = self.exchange(payload, header)
        if payload.is_cuda:
            torch.cuda.current_stream(payload.device).synchronize()  # RCCL ran on torch's stream, apply runs on the map's
        if hasattr(self.backend, "apply_all"):
            self.backend.apply_all(headers, lists)
        else:
            for r in range(len(lists)):
                self.backend.apply(r, headers[r], lists[r])
        return headers


class CBatchIntegrator:
    """The same batch step through the C ABI alone (``ufomap_map_insert_batch``, include/ufomap_hip.h): scan, RCCL
    all-gather and apply are one library call; torch.distributed is only used here to hand the communicator's
    128-byte id from rank 0 to the others (any out-of-band channel would do: this is what a C++ host does with
    its own transport)."""

    def __init__(self, occupancy_map, device_index: int, group=None):
        from .occupancy_map import Comm
        self.m = occupancy_map
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ids = [Comm.unique_id() if rank == 0 else None]
        if world > 1:
            # (src is a GLOBAL rank: the first rank of the group, which need not be global rank 0)
            src = dist.get_global_rank(group, 0) if group is not None and hasattr(dist, "get_global_rank") else 0
            dist.broadcast_object_list(ids, src=src, group=group)
        self.comm = Comm(ids[0], world, rank, device_index)
        self.m.set_option("async_apply", 1)

    def integrate(self, origin, d_xyz_ptr, n, max_range=-1.0, depth=0, discrete=True, d_rgb_ptr=None):
        self.m.insert_batch(self.comm, origin, d_xyz_ptr, n, max_range, depth, discrete, d_rgb_ptr=d_rgb_ptr)

    def close(self):
        self.m.insertPointCloudWait()
        self.comm.close()
