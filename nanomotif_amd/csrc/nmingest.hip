// libnmscan — raw pileup ingestion: the reference's three pre-filters on the device, classification into the state
// planes, kept-row counts (C ABI: nm_ingest_pileup / nm_ingest_results, include/nmscan.h).
#include "nmscan_internal.h"

using namespace nmdetail;

namespace {

// ------------------------------------------------------------------------------------------------------
// Raw pileup ingestion: the reference's three pre-filters evaluated on the device (dataload.py:191-247, in the order
// of find_motifs_bin.py:399-414), then classification into the state planes and a compact list of the surviving
// confidently methylated rows (the input of window extraction, find_motifs_bin.py:625-661).
// ------------------------------------------------------------------------------------------------------

struct RawRows {
    uint64_t n;
    const uint32_t *contig;      // engine-local contig id, 0xFFFFFFFF = contig not resident (row ignored)
    const uint32_t *position;
    const int8_t *mod;           // 0..NM_MAX_MOD_CODES-1
    const uint8_t *strand;
    const double *frac;
    const int32_t *nvalid;
};

// (1) coverage filter + per (contig, mod code) counts for the frequency filter
__global__ void ingest_count_kernel(RawRows r, uint32_t n_contigs, const uint64_t *__restrict__ contig_len,
                                    int min_cov, double meth_thr, unsigned int *cnt /*[contig][mod][2]*/, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    const uint32_t c = r.contig[i];
    if (c == 0xFFFFFFFFu) return;
    const int m = r.mod[i];
    if (c >= n_contigs || r.position[i] >= contig_len[c] || m < 0 || m >= NM_MAX_MOD_CODES) { atomicOr(err, 1u); return; }
    if (r.nvalid[i] <= min_cov) return;                                     // dataload.py:199: Nvalid_cov > 5
    unsigned int *p = cnt + ((size_t)c * NM_MAX_MOD_CODES + m) * 2;
    atomicAdd(p, 1u);
    if (r.frac[i] > meth_thr) atomicAdd(p + 1, 1u);                         // dataload.py:215: fraction_mod > 0.7
}

// (2) frequency filter verdict per (contig, mod code): n_mod / n > 1e-4 and n_mod > 50 (dataload.py:218-219)
__global__ void ingest_group_kernel(uint32_t n_groups, const unsigned int *__restrict__ cnt, double min_freq,
                                    unsigned int min_mods, uint8_t *ok) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const unsigned int n = cnt[g * 2], nm = cnt[g * 2 + 1];
    ok[g] = n > 0 && ((double)nm / (double)n) > min_freq && nm > min_mods;
}

__device__ __forceinline__ bool ingest_row_alive(const RawRows &r, uint64_t i, int min_cov, const uint8_t *ok,
                                                 uint32_t *c_out, bool *plus_out) {
    const uint32_t c = r.contig[i];
    const int m = r.mod[i];
    if (c == 0xFFFFFFFFu || m < 0 || m >= NM_MAX_MOD_CODES || r.nvalid[i] <= min_cov) return false;
    if (!ok[(size_t)c * NM_MAX_MOD_CODES + m]) return false;
    const uint8_t st = r.strand[i];
    if (st != '+' && st != '-') return false;      // other strand labels form groups of their own and are never scored
    *c_out = c;
    *plus_out = st == '+';
    return true;
}

// (3a) adjacency filter, scatter: per strand the maximal fraction at every position (mod codes mixed, dataload.py:237)
__global__ void ingest_scatter_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                      const uint32_t *__restrict__ contig_chunk, unsigned long long *dense_plus,
                                      unsigned long long *dense_minus) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    uint32_t c;
    bool plus;
    if (!ingest_row_alive(r, i, min_cov, ok, &c, &plus)) return;
    const uint64_t g = (uint64_t)contig_chunk[c] * CHUNK_BP + r.position[i];
    // fractions are >= 0, so their IEEE bit patterns order like the values
    atomicMax((plus ? dense_plus : dense_minus) + g, (unsigned long long)__double_as_longlong(r.frac[i]));
}

// (3b) adjacency verdict + classification + confident-row list.  A row survives iff its fraction equals the maximum
// over positions p-d .. p+d of its contig and strand, or is below the threshold (dataload.py:244).
struct IngestSlots {
    int slot_of_mod[NM_MAX_MOD_CODES];        // -1: mod code not scored
    uint32_t *planes[NM_MAX_MOD_SLOTS][6];    // M U MP UP MM UM per slot
    uint32_t can_l[NM_MAX_MOD_SLOTS];         // canonical base C (1) or A (0)
};

__global__ void ingest_decide_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                     const uint32_t *__restrict__ contig_chunk, const unsigned long long *__restrict__ dense_plus,
                                     const unsigned long long *__restrict__ dense_minus, int adjacency, double meth_thr,
                                     double low, double high, IngestSlots sl, const uint32_t *__restrict__ H,
                                     const uint32_t *__restrict__ L, const uint32_t *__restrict__ V,
                                     unsigned int *kept /*[contig][mod]*/, unsigned long long *n_kept,
                                     unsigned long long *conf_count, uint32_t *conf_contig, uint32_t *conf_pos,
                                     uint8_t *conf_strand, int8_t *conf_mod, uint64_t conf_cap, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    uint32_t c;
    bool plus;
    if (!ingest_row_alive(r, i, min_cov, ok, &c, &plus)) return;
    const double f = r.frac[i];
    const uint64_t g = (uint64_t)contig_chunk[c] * CHUNK_BP + r.position[i];
    if (!(f < meth_thr)) {
        const unsigned long long *d = plus ? dense_plus : dense_minus;
        unsigned long long mx = 0;
        for (int k = -adjacency; k <= adjacency; ++k) mx = max(mx, d[g + k]);   // >= 64 zero positions around every contig
        if (mx != (unsigned long long)__double_as_longlong(f)) return;
    }
    const int m = r.mod[i];
    atomicAdd(kept + (size_t)c * NM_MAX_MOD_CODES + m, 1u);
    atomicAdd(n_kept, 1ull);
    const int slot = sl.slot_of_mod[m];
    if (slot < 0) return;
    const bool meth = f >= high, non = f <= low;
    if (meth) {
        const unsigned long long at = atomicAdd(conf_count, 1ull);
        if (at < conf_cap) {
            conf_contig[at] = c;
            conf_pos[at] = r.position[i];
            conf_strand[at] = plus ? '+' : '-';
            conf_mod[at] = (int8_t)m;
        }
    }
    if (!meth && !non) return;
    const size_t w = g >> 5;
    const uint32_t bit = 1u << (g & 31);
    uint32_t *const *pl = sl.planes[slot];
    uint32_t *gen = plus ? (meth ? pl[2] : pl[3]) : (meth ? pl[4] : pl[5]);
    const uint32_t old = atomicOr(gen + w, bit);
    if (old & bit) atomicOr(err, 4u);
    const uint32_t h = (H[w] & bit) != 0, l = (L[w] & bit) != 0, v = (V[w] & bit) != 0;
    const uint32_t want_h = plus ? 0u : 1u;
    if (v && h == want_h && l == sl.can_l[slot]) atomicOr((meth ? pl[0] : pl[1]) + w, bit);
}

}  // namespace

extern "C" {

int nm_ingest_pileup(nm_ctx *c, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_code,
                     const uint8_t *strand, const double *fraction_mod, const int32_t *nvalid_cov,
                     const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8], double low, double high,
                     int rows_on_device, uint64_t *n_kept, uint64_t *n_confident) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (!slot_of_mod || !canonical_of_mod || !n_kept || !n_confident) return fail(NM_EINVAL, "NULL argument");
    if (n_rows && (!contig_id || !position || !mod_code || !strand || !fraction_mod || !nvalid_cov)) return fail(NM_EINVAL, "NULL column");
    if (!(high > low)) return fail(NM_EINVAL, "high threshold must exceed low");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t words = plane_words(c);
    IngestSlots sl{};
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m) {
        sl.slot_of_mod[m] = slot_of_mod[m];
        if (slot_of_mod[m] < 0) continue;
        if (slot_of_mod[m] >= NM_MAX_MOD_SLOTS) return fail(NM_EINVAL, "slot %d >= %d", slot_of_mod[m], NM_MAX_MOD_SLOTS);
        if (canonical_of_mod[m] != 'A' && canonical_of_mod[m] != 'C') return fail(NM_EINVAL, "canonical base must be 'A' or 'C'");
        ModSlot &ms = c->slots[slot_of_mod[m]];
        drop_slot_ranks(ms);
        for (auto &p : ms.planes) {
            if (!p) HIP_TRY(hipMalloc(&p, words * 4));
            HIP_TRY(hipMemsetAsync(p, 0, words * 4, c->stream));
        }
        ms.present = true;
        ms.canonical = canonical_of_mod[m];
        ms.low = low;
        ms.high = high;
        ms.n_rows = 0;
        for (int k = 0; k < 6; ++k) sl.planes[slot_of_mod[m]][k] = ms.planes[k];
        sl.can_l[slot_of_mod[m]] = canonical_of_mod[m] == 'C' ? 1u : 0u;
    }
    // device copies of the raw columns
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *p : owned) (void)hipFree(p); };
    RawRows r{};
    r.n = n_rows;
    if (rows_on_device) {
        r.contig = contig_id; r.position = position; r.mod = mod_code; r.strand = strand; r.frac = fraction_mod; r.nvalid = nvalid_cov;
    } else {
        const void *src[6] = {contig_id, position, mod_code, strand, fraction_mod, nvalid_cov};
        const size_t esz[6] = {4, 4, 1, 1, 8, 4};
        void *dst[6];
        for (int k = 0; k < 6; ++k) {
            dst[k] = nullptr;
            if (hipMalloc(&dst[k], std::max<size_t>(n_rows * esz[k], 16)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory for the raw pileup"); }
            owned.push_back(dst[k]);
            if (n_rows && hipMemcpyAsync(dst[k], src[k], n_rows * esz[k], hipMemcpyHostToDevice, c->stream) != hipSuccess) { cleanup(); return fail(NM_EHIP, "H2D copy of the raw pileup failed"); }
        }
        r.contig = (const uint32_t *)dst[0]; r.position = (const uint32_t *)dst[1]; r.mod = (const int8_t *)dst[2];
        r.strand = (const uint8_t *)dst[3]; r.frac = (const double *)dst[4]; r.nvalid = (const int32_t *)dst[5];
    }
    const size_t n_groups = (size_t)c->n_contigs * NM_MAX_MOD_CODES;
    const uint64_t npos = (uint64_t)c->n_chunks * CHUNK_BP;
    unsigned int *d_cnt = nullptr, *d_kept = nullptr;
    uint8_t *d_ok = nullptr;
    unsigned long long *d_dense = nullptr, *d_scalars = nullptr;
    uint32_t *d_cc = nullptr, *d_cp = nullptr;
    uint8_t *d_cs = nullptr;
    int8_t *d_cm = nullptr;
    const uint64_t conf_cap = std::max<uint64_t>(n_rows, 1);
#define ING_ALLOC(ptr, bytes) do { void *q_ = nullptr; if (hipMalloc(&q_, (bytes)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory in nm_ingest_pileup (%zu bytes)", (size_t)(bytes)); } owned.push_back(q_); ptr = (decltype(ptr))q_; } while (0)
    ING_ALLOC(d_cnt, n_groups * 2 * 4);
    ING_ALLOC(d_kept, n_groups * 4);
    ING_ALLOC(d_ok, n_groups);
    ING_ALLOC(d_dense, npos * 8 * 2);
    ING_ALLOC(d_scalars, 16);
    ING_ALLOC(d_cc, conf_cap * 4);
    ING_ALLOC(d_cp, conf_cap * 4);
    ING_ALLOC(d_cs, conf_cap);
    ING_ALLOC(d_cm, conf_cap);
#undef ING_ALLOC
    hipError_t e = hipSuccess;
    e = hipMemsetAsync(d_cnt, 0, n_groups * 2 * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_kept, 0, n_groups * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_dense, 0, npos * 8 * 2, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_scalars, 0, 16, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream);
    if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "memset failed: %s", hipGetErrorString(e)); }
    if (n_rows) {
        const dim3 grid((unsigned)((n_rows + 255) / 256)), blk(256);
        hipLaunchKernelGGL(ingest_count_kernel, grid, blk, 0, c->stream, r, c->n_contigs, c->d_contig_len, 5, 0.7, d_cnt, c->d_err);
        hipLaunchKernelGGL(ingest_group_kernel, dim3((unsigned)((n_groups + 255) / 256)), blk, 0, c->stream, (uint32_t)n_groups, d_cnt, 0.0001, 50u, d_ok);
        hipLaunchKernelGGL(ingest_scatter_kernel, grid, blk, 0, c->stream, r, 5, d_ok, c->d_contig_chunk, d_dense, d_dense + npos);
        hipLaunchKernelGGL(ingest_decide_kernel, grid, blk, 0, c->stream, r, 5, d_ok, c->d_contig_chunk, d_dense, d_dense + npos, 8, 0.7,
                           low, high, sl, c->dH, c->dL, c->dV, d_kept, d_scalars, d_scalars + 1, d_cc, d_cp, d_cs, d_cm, conf_cap, c->d_err);
        e = hipGetLastError();
        if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "ingest launch failed: %s", hipGetErrorString(e)); }
    }
    unsigned long long scal[2] = {0, 0};
    unsigned int err = 0;
    e = hipMemcpyAsync(scal, d_scalars, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "ingest failed: %s", hipGetErrorString(e)); }
    const uint64_t nconf = scal[1];
    drop_ingest_rows(c);
    c->ing_kept.resize(n_groups);
    if (nconf) {        // exact-size copies; the n_rows-sized scratch goes away with cleanup()
        e = hipMalloc(&c->d_ing_contig, nconf * 4);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_pos, nconf * 4);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_strand, nconf);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_mod, nconf);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_contig, d_cc, nconf * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_pos, d_cp, nconf * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_strand, d_cs, nconf, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_mod, d_cm, nconf, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { cleanup(); drop_ingest_rows(c); return fail(NM_EHIP, "keeping the confident rows failed: %s", hipGetErrorString(e)); }
        c->ing_nconf = nconf;
    }
    (void)hipMemcpy(c->ing_kept.data(), d_kept, n_groups * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (err & 1u) return fail(NM_EINVAL, "pileup row with contig_id / position / mod code outside the uploaded assembly");
    if (err & 4u) return fail(NM_EINVAL, "duplicate (contig, position, strand) rows within one modification type: the reference's np.isin(assume_unique=True) requires unique positions (find_motifs_bin.py:1258)");
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m)
        if (slot_of_mod[m] >= 0) c->slots[slot_of_mod[m]].n_rows = scal[0];
    *n_kept = scal[0];
    *n_confident = nconf;
    return NM_OK;
}

int nm_ingest_results(nm_ctx *c, uint32_t *conf_contig, uint32_t *conf_position, uint8_t *conf_strand, int8_t *conf_mod,
                      uint64_t capacity, uint32_t *kept_per_contig_mod /*[n_contigs][8]*/) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    const size_t n = c->ing_nconf;
    if (conf_contig || conf_position || conf_strand || conf_mod) {          // all NULL: only the kept table is wanted
        if (capacity < n) return fail(NM_ERANGE, "capacity %llu < %zu confident rows", (unsigned long long)capacity, n);
        if (n) {
            if (!conf_contig || !conf_position || !conf_strand || !conf_mod) return fail(NM_EINVAL, "NULL argument");
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipMemcpy(conf_contig, c->d_ing_contig, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_position, c->d_ing_pos, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_strand, c->d_ing_strand, n, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_mod, c->d_ing_mod, n, hipMemcpyDeviceToHost));
        }
    }
    if (kept_per_contig_mod) memcpy(kept_per_contig_mod, c->ing_kept.data(), c->ing_kept.size() * 4);
    return NM_OK;
}

}  // extern "C"
