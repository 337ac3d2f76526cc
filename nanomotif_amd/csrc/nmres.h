// Pinned host buffers of the file parsers, kept between calls (csrc/nmpool.cpp).  Measured on a fresh process (tools/alloc_costs_probe.hip,
// profiles/r6/alloc_costs.txt): hipHostMalloc 0.17 ms per MB, hipHostFree 0.09 ms per MB — the FASTA parser's ring (3 x 32 MB) and the
// pileup parser's two chunks were 27 ms of allocations and 14 ms of frees in every run of the command line; a new stream costs 10 - 15 ms
// (the parsers copy on the ctx's copy stream instead of creating one each).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace nmres {
// a pinned buffer of at least `bytes`: an idle cached one (of at most twice the size, pinned under the caller's current device) or a new hipHostMalloc
hipError_t pinned_take(void **p, size_t bytes);
// back to the cache; what does not fit (more than PINNED_KEEP_BYTES idle in all) is freed.  The caller has waited for every copy that
// used the buffer.  nullptr is ignored.
void pinned_give(void *p);
// frees every idle buffer (nm_block_cache(0, ...) calls it); buffers in use are not touched
void pinned_trim();
constexpr size_t PINNED_KEEP_BYTES = 160u << 20;
}  // namespace nmres
