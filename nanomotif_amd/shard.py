"""Bin -> GPU and contig -> GPU assignment for the multi-GPU path.

Bins are independent searches (find_motifs_bin.py:152-171): when the bins can be spread evenly, every GPU takes WHOLE
bins (``assign_bins``) and runs their searches alone — no collective on the data path, the motif rows are gathered at
the end.  Otherwise (few / huge bins, a single isolate genome) contigs are the unit:

Counts are sums over contigs (find_motifs_bin.py:1273-1283), so contigs — not bins — are the sharding unit and a
bin may span GPUs; the per-candidate count tables are summed with one RCCL all-reduce per scoring step.

The assignment is longest-processing-time-first on *pieces*: a bin that is small against a GPU's share is one piece
(all its contigs go to the least loaded GPU); a larger bin is cut into pieces of consecutive contigs no bigger
than that limit (down to single contigs for a metagenome with a few huge bins, or a single isolate genome).  Keeping bins whole where possible means a rank
only compiles and uploads the candidate programs of the bins it holds (the engine skips candidates whose bin
has no contig on the device), so the fixed per-step host cost shrinks with the number of GPUs too.
"""
from __future__ import annotations

import heapq

import numpy as np


def assign_contigs(lengths, world_size: int, bins=None, whole_bin_fraction: float = 0.05):
    """list (per rank) of ascending contig-index arrays; deterministic (ties by index / name).

    bins: optional bin label per contig.  Pieces are at most ``whole_bin_fraction`` of the per-rank target load
    (which bounds the final imbalance to about that fraction)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if bins is None:
        items = [(int(lengths[i]), (i,)) for i in range(n)]
    else:
        target = lengths.sum() / max(world_size, 1)
        groups = {}
        for i, b in enumerate(bins):
            groups.setdefault(b, []).append(i)
        items = []
        for b in sorted(groups, key=str):
            idx = groups[b]
            size = int(lengths[idx].sum())
            limit = whole_bin_fraction * target
            if size <= limit:
                items.append((size, tuple(idx)))
                continue
            piece, psize = [], 0
            for i in sorted(idx, key=lambda i: (-int(lengths[i]), i)):
                if piece and psize + int(lengths[i]) > limit:
                    items.append((psize, tuple(piece)))
                    piece, psize = [], 0
                piece.append(i)
                psize += int(lengths[i])
            if piece:
                items.append((psize, tuple(piece)))
    items.sort(key=lambda it: (-it[0], it[1][0]))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    out = [[] for _ in range(world_size)]
    for size, idx in items:
        load, r = heapq.heappop(heap)
        out[r].extend(idx)
        heapq.heappush(heap, (load + size, r))
    return [np.array(sorted(x), dtype=np.int64) for x in out]


def assign_bins(bin_sizes: dict, world_size: int, tolerance: float = 0.15):
    """Whole bins -> ranks, longest-processing-time-first on the bins' total bp (ties by name).  Returns a list (per
    rank) of bin-name lists in ``bin_sizes`` order, or None when the heaviest rank would exceed the mean load by more
    than ``tolerance`` (then contigs should be the sharding unit, ``assign_contigs``)."""
    if world_size <= 1:
        return [list(bin_sizes)]
    order = sorted(bin_sizes, key=lambda b: (-int(bin_sizes[b]), str(b)))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    owner = {}
    for b in order:
        load, r = heapq.heappop(heap)
        owner[b] = r
        heapq.heappush(heap, (load + int(bin_sizes[b]), r))
    loads = [0] * world_size
    for b, r in owner.items():
        loads[r] += int(bin_sizes[b])
    mean = sum(loads) / world_size
    if mean == 0 or max(loads) > (1.0 + tolerance) * mean:
        return None
    return [[b for b in bin_sizes if owner[b] == r] for r in range(world_size)]
