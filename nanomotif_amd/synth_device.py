"""Device-side twin of ``synth.py``: the same counter-based hash evaluated with torch on the GPU, so the
1 Gbp bench inputs are produced in HBM in seconds instead of minutes on the host.

Bit-identical to ``SynthMetagenome.contig_codes`` / ``contig_pileup`` (tests/test_gpu_synth.py checks every
byte and every row on a small metagenome), which lets the bench regenerate any subset of contigs on the
host for the CPU baseline and for parity checks.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth

M32 = 0xFFFFFFFF


def _mix32(x):
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M32        # int64 product may wrap; the low 32 bits are still exact
    x = x ^ (x >> 16)
    return x


def _stream(key, pos, stream_id: int):
    salt = (stream_id * 0x632BE5AB + 0x9E3779B9) & M32
    return _mix32((_mix32(pos ^ key) + salt) & M32)


def _popcount32(x):
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F
    return ((x * 0x01010101) & M32) >> 24


def _match_starts(bit, seg_id, masks):
    """bool over all positions s of the concatenated bin: masks match at s and s..s+len-1 lie in one contig."""
    n = len(masks)
    total = bit.numel()
    ok = torch.zeros(total, dtype=torch.bool, device=bit.device)
    if total < n:
        return ok
    span = total - n + 1
    acc = seg_id[:span] == seg_id[n - 1:n - 1 + span]
    for j, m in enumerate(masks.tolist()):
        if m == 15:
            continue
        acc &= (bit[j:j + span] & m) != 0
    ok[:span] = acc
    return ok


_FRAC_TABLE = {}


def _fraction_table(device):
    key = str(device)
    if key not in _FRAC_TABLE:
        _FRAC_TABLE[key] = torch.from_numpy(synth.pct_to_fraction(np.arange(10001))).to(device)
    return _FRAC_TABLE[key]


class DeviceBin:
    """Generated data of one bin: ASCII codes of its contigs (concatenated) and pileup columns per mod type."""

    def __init__(self, contigs, ascii_cat, starts, pileups):
        self.contigs, self.ascii_cat, self.starts, self.pileups = contigs, ascii_cat, starts, pileups


def generate_bin(mg: synth.SynthMetagenome, bin_name: str, device, min_cov: int = 5, only=None, keep_nvalid=True,
                 contigs=None) -> DeviceBin:
    """Contigs of ``bin_name`` (restricted to the set ``only`` when given — a rank's shard).  Pileup rows are
    filtered with ``Nvalid_cov > min_cov`` (dataload.py:199), pass min_cov=-1 to keep every row.  ``contig_id``
    in the returned columns is the GLOBAL contig index of ``mg``."""
    assert mg.spec.n_fraction == 0, "device generator does not plant N runs"
    if contigs is None:
        contigs = [i for i, b in enumerate(mg.bin_names) if b == bin_name and (only is None or i in only)]
    if not contigs:
        return DeviceBin([], torch.zeros(0, dtype=torch.uint8, device=device), torch.zeros(0, dtype=torch.int64, device=device),
                         {mt: None for mt in mg.spec.mod_types})
    lens = torch.tensor([int(mg.lengths[i]) for i in contigs], dtype=torch.int64, device=device)
    keys = torch.tensor([synth.contig_key(mg.spec.seed, i) for i in contigs], dtype=torch.int64, device=device)
    gid = torch.tensor(contigs, dtype=torch.int64, device=device)
    starts = torch.cumsum(lens, 0) - lens
    total = int(lens.sum())
    seg = torch.repeat_interleave(torch.arange(len(contigs), device=device), lens)
    pos = torch.arange(total, device=device, dtype=torch.int64) - starts[seg]
    key = keys[seg]

    h = _stream(key, pos, 0)
    gc_thr = int(round(mg.bin_gc[bin_name] * 65536))
    is_gc = (h & 0xFFFF) < gc_thr
    second = (h >> 16) & 1
    codes = torch.where(is_gc, 1 + second, 3 * second)                       # A0 C1 G2 T3
    ascii_lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    ascii_cat = ascii_lut[codes]
    bit = (1 << codes).to(torch.uint8)

    pileups = {}
    for mt in mg.spec.mod_types:
        can = "ACGT".index(synth.MOD_CANONICAL[mt])
        comp = 3 - can
        stream_base = 16 * (1 + ["a", "m", "21839"].index(mt))
        planted_plus = torch.zeros(total, dtype=torch.bool, device=device)
        planted_minus = torch.zeros(total, dtype=torch.bool, device=device)
        for iupac, mpos, mmt in mg.bin_motifs[bin_name]:
            if mmt != mt:
                continue
            fm = synth.motif_masks(iupac)
            n = len(fm)
            st = _match_starts(bit, seg, fm)
            planted_plus[mpos:] |= st[:total - mpos]
            rm = synth.revcomp_masks(fm)
            st = _match_starts(bit, seg, rm)
            off = n - 1 - mpos
            planted_minus[off:] |= st[:total - off]
        cols = []
        for strand_char, base_code, planted in ((ord("+"), can, planted_plus), (ord("-"), comp, planted_minus)):
            idx = torch.nonzero(codes == base_code).squeeze(1)
            p, k = pos[idx], key[idx]
            sid = stream_base + (0 if strand_char == ord("+") else 4)
            h1, h2, h3 = _stream(k, p, sid + 1), _stream(k, p, sid + 2), _stream(k, p, sid + 3)
            v = h1 & 0xFFFF
            meth = 10000 - ((v * v) >> 21)
            bg = (v * v) >> 22
            is_site = planted[idx] & ((h1 >> 16) < int(mg.spec.methylated_fraction * 65536))
            pct = torch.where(is_site, meth, bg)
            fp = (h2 & 0xFFFF) < 328
            pct = torch.where(fp & ~is_site, 6500 + ((h2 >> 16) % 3500), pct)
            probe = ((h2 >> 6) & 0x3FF) == 5
            probe_vals = torch.tensor([3000, 7000, 2999, 7001, 3001, 6999, 0, 10000], dtype=torch.int64, device=device)
            pct = torch.where(probe, probe_vals[(h2 >> 20) & 7], pct)
            cov = _popcount32(h3) + _popcount32(h2 & 0x0FFFFFFF)
            low = (h3 & 0xFFF) == 1
            cov = torch.where(low, (h3 >> 12) & 7, cov)
            keep = cov > min_cov
            cols.append((gid[seg[idx]][keep], p[keep], torch.full((int(keep.sum()),), strand_char, dtype=torch.uint8, device=device),
                         pct[keep], cov[keep]))
        cid = torch.cat([c[0] for c in cols]).to(torch.int32)
        pp = torch.cat([c[1] for c in cols]).to(torch.int32)
        st = torch.cat([c[2] for c in cols])
        pct = torch.cat([c[3] for c in cols])
        # float64 division on the device is not guaranteed to round like numpy's: look the fraction up in a
        # table computed on the host with synth.pct_to_fraction (10 001 possible values)
        frac = _fraction_table(device)[pct]
        pileups[mt] = dict(contig_id=cid.contiguous(), position=pp.contiguous(), strand=st.contiguous(),
                           fraction_mod=frac.contiguous())
        if keep_nvalid:
            pileups[mt]["nvalid"] = torch.cat([c[4] for c in cols])
    return DeviceBin(contigs, ascii_cat, starts, pileups)


def load_engine_from_device(engine, mg: synth.SynthMetagenome, device, low=0.3, high=0.7, min_cov=5, contigs=None,
                            progress=None):
    """Generate the metagenome (or the shard ``contigs`` of it: global contig indices, ascending) on ``device``
    and hand it to ``engine`` through the device-pointer entry points (nm_upload_contigs_device /
    nm_upload_pileup_device).  Engine contig k = contigs[k]; bin ids are global.  Returns rows per mod type."""
    import ctypes as C
    from . import _lib
    mine = list(range(len(mg.names))) if contigs is None else sorted(int(i) for i in contigs)
    local_of = {g: k for k, g in enumerate(mine)}
    lengths = np.asarray([int(mg.lengths[i]) for i in mine], dtype=np.uint64)
    offsets = np.zeros(len(lengths) + 1, dtype=np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    ascii_all = torch.empty(int(offsets[-1]), dtype=torch.uint8, device=device)
    bins = sorted(set(mg.bin_names))
    local_lut = torch.full((len(mg.names),), -1, dtype=torch.int32, device=device)
    local_lut[torch.tensor(mine, dtype=torch.int64, device=device)] = torch.arange(len(mine), dtype=torch.int32, device=device)
    staged = []
    by_bin = {b: [] for b in bins}
    for i in mine:
        by_bin[mg.bin_names[i]].append(i)
    for bi, b in enumerate(bins):
        db = generate_bin(mg, b, device, min_cov=min_cov, keep_nvalid=False, contigs=by_bin[b])
        st = db.starts.tolist()
        for k, i in enumerate(db.contigs):
            L = int(mg.lengths[i])
            o = int(offsets[local_of[i]])
            ascii_all[o:o + L] = db.ascii_cat[st[k]:st[k] + L]
        if db.contigs:
            for mt in mg.spec.mod_types:
                db.pileups[mt]["contig_id"] = local_lut[db.pileups[mt]["contig_id"].to(torch.int64)].contiguous()
            staged.append(db.pileups)
        if progress and (bi + 1) % 100 == 0:
            progress(f"generated {bi + 1}/{len(bins)} bins")
    torch.cuda.synchronize(device)
    engine.upload_assembly_device([mg.names[i] for i in mine], lengths, [mg.bin_names[i] for i in mine], ascii_all.data_ptr(),
                                  bin_names=bins)
    engine.slot_of_mod = {mt: k for k, mt in enumerate(mg.spec.mod_types)}
    rows = {mt: 0 for mt in mg.spec.mod_types}
    for mt in mg.spec.mod_types:
        first = True
        for pile in staged:
            p = pile[mt]
            n = int(p["position"].numel())
            _lib.check(engine.lib.nm_upload_pileup_device(
                engine.ctx, engine.slot_of_mod[mt], ord(synth.MOD_CANONICAL[mt]), float(low), float(high), n,
                C.c_void_p(p["contig_id"].data_ptr()), C.c_void_p(p["position"].data_ptr()),
                C.c_void_p(p["strand"].data_ptr()), C.c_void_p(p["fraction_mod"].data_ptr()), 0 if first else 1))
            rows[mt] += n
            first = False
    torch.cuda.synchronize(device)
    del ascii_all, staged
    return rows
