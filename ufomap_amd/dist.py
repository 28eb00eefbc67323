"""Batched multi-sensor integration across GPUs (BASELINE config C4).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU
tests).  Per batch every rank ray-casts ITS scan -- the part of the path that never reads the map --
into an *update list* (16-byte records, ``include/ufomap_hip.h``), the lists are exchanged with ONE
padded all-gather (plus a tiny all-gather of their headers), and every rank applies the lists of
ranks 0..N-1 in rank order to its replica of the map -- with ONE walk of the tree for the whole batch
(``ufomap_map_apply_keys_batch``).  Applying in order reproduces the reference's
sequential integration bit-exactly on every replica; a float all-reduce of log-odds deltas would not
(clamping after every hit phase and every miss phase is not associative).

The collective logic is independent of the device: ``exchange_lists`` works on CPU tensors with gloo
(tests/test_dist_gloo.py) exactly as it does on HBM tensors with RCCL.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .capi import KeysInfo

ENTRY_BYTES = 16


def exchange_lists(payload: torch.Tensor, header: torch.Tensor, group=None):
    """All-gather variable-length update lists.

    payload: uint8 tensor [n_local * 16] (this rank's list, on the group's device);
    header:  int32 tensor [KeysInfo.WORDS] on the same device.
    Returns (headers [world, WORDS] on CPU, lists: list of uint8 tensors, one per rank, trimmed).
    """
    world = dist.get_world_size(group)
    dev = payload.device
    headers = torch.empty((world, header.numel()), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(headers.view(-1), header.contiguous(), group=group)
    headers_cpu = headers.cpu()
    counts = (headers_cpu[:, 0].to(torch.int64) + headers_cpu[:, 1].to(torch.int64)) * ENTRY_BYTES
    max_bytes = int(counts.max().item())
    if max_bytes == 0:
        return headers_cpu, [payload.new_empty(0) for _ in range(world)]
    send = torch.zeros(max_bytes, dtype=torch.uint8, device=dev)  # ring collectives want equal shares: pad
    send[: payload.numel()] = payload
    recv = torch.empty((world, max_bytes), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    return headers_cpu, [recv[r, : int(counts[r])] for r in range(world)]


class MapBackend:
    """The real thing: scan_keys / get_keys / apply_keys of the HIP library on one GPU."""

    def __init__(self, occupancy_map, device):
        self.m = occupancy_map
        self.device = device
        self._buf = torch.empty(0, dtype=torch.uint8, device=device)

    def scan(self, origin, d_xyz_ptr, n, max_range, depth, discrete):
        info = self.m.scan_keys(origin, d_xyz_ptr, n, max_range, depth, discrete)
        nbytes = (info.n_hit + info.n_miss) * ENTRY_BYTES
        if self._buf.numel() < nbytes:
            self._buf = torch.empty(max(nbytes, 2 * self._buf.numel()), dtype=torch.uint8, device=self.device)
        if nbytes:
            self.m.get_keys(self._buf.data_ptr(), self._buf.numel() // ENTRY_BYTES, info)  # syncs the map's stream
        header = torch.tensor(info.to_list(), dtype=torch.int32, device=self.device)
        return self._buf[:nbytes], header

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


class BatchIntegrator:
    """``integrate`` = one batch step: scan locally, exchange, apply every rank's list in rank order."""

    def __init__(self, occupancy_map=None, group=None, device=None, backend=None):
        self.group = group
        self.backend = backend if backend is not None else MapBackend(occupancy_map, device)

    def integrate(self, origin, d_xyz_ptr, n, max_range=-1.0, depth=0, discrete=True):
        payload, header = self.backend.scan(origin, d_xyz_ptr, n, max_range, depth, discrete)
        headers, lists = exchange_lists(payload, header, self.group)
        if payload.is_cuda:
            torch.cuda.current_stream(payload.device).synchronize()  # RCCL ran on torch's stream, apply runs on the map's
        if hasattr(self.backend, "apply_all"):
            self.backend.apply_all(headers, lists)
        else:
            for r in range(len(lists)):
                self.backend.apply(r, headers[r], lists[r])
        return headers
