"""Contig -> GPU assignment for the multi-GPU path.

Counts are sums over contigs (find_motifs_bin.py:1273-1283), so contigs — not bins — are the sharding unit:
longest-processing-time-first on contig length balances the bytes each GPU streams per scoring step; the
contigs of one bin may land on several GPUs and the per-candidate count tables are summed with one RCCL
all-reduce per step.
"""
from __future__ import annotations

import heapq

import numpy as np


def assign_contigs(lengths, world_size: int):
    """list (per rank) of ascending contig-index arrays; deterministic (ties by index)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    out = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(lengths[i]), r))
    return [np.array(sorted(x), dtype=np.int64) for x in out]
