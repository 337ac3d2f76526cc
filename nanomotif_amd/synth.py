"""Seeded synthetic metagenome + modkit-style pileup generator.

Every value is a pure function of ``(seed, contig index, position, stream)`` through a 32-bit
integer hash (counter-based, no sequential RNG state), so

* any subset of contigs can be regenerated bit-identically anywhere (host numpy here, the
  torch/device variant in ``synth_device.py`` for the 1 Gbp bench inputs), and
* percent-modified values are integer hundredths of a percent, so the text ``.bed`` form and the
  binary SoA form agree exactly (``fraction_mod = hundredths / 100 / 100`` is evaluated the same
  way the reference's loader does it: ``float(col11_text) / 100``, dataload.py:84-85).

The layout follows SURVEY.md §8(d): iid bases with a per-bin GC content, one pileup row per
canonical base on ``+`` and per complement base on ``-``, planted REBASE-style motifs per bin.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# base codes used throughout the package: A=0 C=1 G=2 T=3, 4 = anything else
ASCII_OF_CODE = np.frombuffer(b"ACGTN", dtype=np.uint8)
CODE_OF_ASCII = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    CODE_OF_ASCII[_c] = _i
    CODE_OF_ASCII[ord(chr(_c).lower())] = _i

IUPAC_MASK = {  # bit0=A bit1=C bit2=G bit3=T
    "A": 1, "C": 2, "G": 4, "T": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
    "B": 14, "D": 13, "H": 11, "V": 7, "N": 15,
}
COMPLEMENT_CODE = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
MOD_CANONICAL = {"a": "A", "m": "C", "21839": "C"}

# (IUPAC motif, mod_position, mod_type) — REBASE-style list used for planting
MOTIF_LIBRARY = [
    ("GATC", 1, "a"), ("GAATTC", 2, "a"), ("CTGCAG", 4, "a"), ("GANTC", 1, "a"),
    ("ACCCA", 4, "a"), ("CCAAAT", 4, "a"), ("GRNGAAGY", 5, "a"), ("GCACNNNNNNGTT", 2, "a"),
    ("AACNNNNNNGTGC", 1, "a"), ("TTCGAA", 5, "a"), ("GTAC", 2, "a"), ("CAGAG", 3, "a"),
    ("GGTGA", 4, "a"), ("CACNNNNNTGG", 1, "a"),
    ("CCWGG", 1, "m"), ("GCGC", 1, "m"), ("CCGG", 1, "m"), ("GGCC", 2, "m"),
    ("RGCY", 2, "m"), ("GCNGC", 1, "m"), ("ACGT", 1, "m"), ("CTCGAG", 2, "m"),
    ("GGNCC", 3, "m"), ("CGCG", 0, "m"),
]

_U32 = np.uint32


def mix32(x):
    """lowbias32 integer hash on uint32 arrays (wraps mod 2**32)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> _U32(16)
    x *= _U32(0x7FEB352D)
    x ^= x >> _U32(15)
    x *= _U32(0x846CA68B)
    x ^= x >> _U32(16)
    return x


def contig_key(seed: int, contig_index: int) -> int:
    k = (seed * 0x9E3779B1 + contig_index * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF
    return int(mix32(np.array([k], dtype=np.uint32))[0])


def stream(key: int, pos, stream_id: int):
    """uint32 hash value for every position of one contig on one stream."""
    pos = np.asarray(pos, dtype=np.uint32)
    salt = _U32((stream_id * 0x632BE5AB + 0x9E3779B9) & 0xFFFFFFFF)
    return mix32(mix32(pos ^ _U32(key)) + salt)


def motif_masks(iupac: str) -> np.ndarray:
    return np.array([IUPAC_MASK[c] for c in iupac], dtype=np.uint8)


def revcomp_masks(masks: np.ndarray) -> np.ndarray:
    """Complement every 4-bit set (A<->T, C<->G) and reverse."""
    m = masks.astype(np.uint8)
    comp = ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3)
    return comp[::-1].copy()


def match_starts(codes: np.ndarray, masks: np.ndarray) -> np.ndarray:
    """Boolean array over start offsets: masks match codes[s:s+len] (non-ACGT matches only N=15)."""
    n = len(masks)
    L = len(codes)
    if L < n:
        return np.zeros(0, dtype=bool)
    ok = np.ones(L - n + 1, dtype=bool)
    bit = np.where(codes < 4, np.left_shift(1, codes.astype(np.uint8) & 3), 0).astype(np.uint8)
    for j, m in enumerate(masks):
        if m == 15:
            continue
        ok &= (bit[j:j + L - n + 1] & m) != 0
    return ok


@dataclass
class SynthSpec:
    n_contigs: int = 1
    total_bp: int = 200_000
    n_bins: int = 1
    mod_types: tuple = ("a",)
    seed: int = 1
    motifs_per_bin: tuple = (1, 4)       # inclusive range, drawn per bin by hash
    fixed_motifs: tuple | None = None    # [(iupac, pos, modtype), ...] applied to every bin
    lognormal_sigma: float = 0.8
    min_contig_bp: int = 2_000
    n_fraction: float = 0.0              # fraction of positions turned into 'N' runs (0 = none)
    methylated_fraction: float = 0.97    # P(site of a planted motif is methylated)


@dataclass
class SynthMetagenome:
    spec: SynthSpec
    names: list
    bin_names: list                      # per contig
    lengths: np.ndarray
    bin_gc: dict
    bin_motifs: dict                     # bin -> [(iupac, pos, modtype)]
    _seq_cache: dict = field(default_factory=dict)
    key_index: dict = field(default_factory=dict)   # contig index -> index whose hash key it shares (a second placement of a contig)

    # ---------------------------------------------------------------- sequences
    def contig_codes(self, i: int) -> np.ndarray:
        if i in self._seq_cache:
            return self._seq_cache[i]
        L = int(self.lengths[i])
        key = contig_key(self.spec.seed, i)
        pos = np.arange(L, dtype=np.uint32)
        h = stream(key, pos, 0)
        gc_thr = _U32(int(round(self.bin_gc[self.bin_names[i]] * 65536)))
        is_gc = (h & _U32(0xFFFF)) < gc_thr
        second = ((h >> _U32(16)) & _U32(1)).astype(np.uint8)
        # AT pair: A(0)/T(3); GC pair: C(1)/G(2)
        codes = np.where(is_gc, _U32(1) + second, _U32(3) * second).astype(np.uint8)
        if self.spec.n_fraction > 0:
            # N runs of 1..16 bp starting where a sparse hash fires
            hn = stream(key, pos, 7)
            start_thr = _U32(int(self.spec.n_fraction / 8.5 * 2**32))
            starts = np.flatnonzero(hn < start_thr)
            for s in starts:
                run = 1 + int(hn[s] & _U32(15))
                codes[s:s + run] = 4
        if len(self._seq_cache) < 64:
            self._seq_cache[i] = codes
        return codes

    def contig_ascii(self, i: int) -> np.ndarray:
        return ASCII_OF_CODE[self.contig_codes(i)]

    def contig_str(self, i: int) -> str:
        return self.contig_ascii(i).tobytes().decode("ascii")

    # ---------------------------------------------------------------- pileup
    def contig_pileup(self, i: int, mod_type: str):
        """SoA rows of one contig / one mod type, sorted by (position) with '+' and '-' interleaved.

        Returns dict(position int64, strand uint8 ('+'/'-' ASCII), pct_hundredths int32, nvalid int32).
        """
        codes = self.contig_codes(i)
        L = len(codes)
        key = contig_key(self.spec.seed, self.key_index.get(i, i))
        can = "ACGT".index(MOD_CANONICAL[mod_type])
        comp = 3 - can
        stream_base = 16 * (1 + ["a", "m", "21839"].index(mod_type))

        planted_plus = np.zeros(L, dtype=bool)
        planted_minus = np.zeros(L, dtype=bool)
        for iupac, mpos, mt in self.bin_motifs[self.bin_names[i]]:
            if mt != mod_type:
                continue
            fm = motif_masks(iupac)
            st = np.flatnonzero(match_starts(codes, fm))
            planted_plus[st + mpos] = True
            rm = revcomp_masks(fm)
            st = np.flatnonzero(match_starts(codes, rm))
            planted_minus[st + (len(fm) - 1 - mpos)] = True

        out = {}
        for strand_char, base_code, planted in ((ord("+"), can, planted_plus), (ord("-"), comp, planted_minus)):
            pos = np.flatnonzero(codes == base_code).astype(np.uint32)
            sid = stream_base + (0 if strand_char == ord("+") else 4)
            h1 = stream(key, pos, sid + 1)
            h2 = stream(key, pos, sid + 2)
            v = (h1 & _U32(0xFFFF)).astype(np.uint64)
            # methylated sites: skewed to 100 %, floor 79.52 %
            meth = (10000 - ((v * v) >> np.uint64(21))).astype(np.int32)
            # background: skewed to 0 %, ceiling 10.23 %
            bg = ((v * v) >> np.uint64(22)).astype(np.int32)
            is_site = planted[pos] & ((h1 >> _U32(16)) < _U32(int(self.spec.methylated_fraction * 65536)))
            pct = np.where(is_site, meth, bg)
            # 0.5 % false positives 65.00 .. 99.99 %
            fp = (h2 & _U32(0xFFFF)) < _U32(328)
            pct = np.where(fp & ~is_site, 6500 + ((h2 >> _U32(16)) % _U32(3500)).astype(np.int32), pct)
            # 1/1024: exact threshold probes (inclusive >= high / <= low compares)
            probe = ((h2 >> _U32(6)) & _U32(0x3FF)) == _U32(5)
            probe_vals = np.array([3000, 7000, 2999, 7001, 3001, 6999, 0, 10000], dtype=np.int32)
            pct = np.where(probe, probe_vals[(h2 >> _U32(20)) & _U32(7)], pct).astype(np.int32)
            # coverage: Binomial(60, .5)-like, with 1/4096 low-coverage rows (0..7)
            h3 = stream(key, pos, sid + 3)
            cov = _popcount32(h3) + _popcount32(h2 & _U32(0x0FFFFFFF))
            low = (h3 & _U32(0xFFF)) == _U32(1)
            cov = np.where(low, (h3 >> _U32(12)) & _U32(7), cov).astype(np.int32)
            out[strand_char] = (pos.astype(np.int64), pct, cov)

        pos = np.concatenate([out[ord("+")][0], out[ord("-")][0]])
        strand = np.concatenate([np.full(len(out[ord("+")][0]), ord("+"), np.uint8),
                                 np.full(len(out[ord("-")][0]), ord("-"), np.uint8)])
        pct = np.concatenate([out[ord("+")][1], out[ord("-")][1]])
        cov = np.concatenate([out[ord("+")][2], out[ord("-")][2]])
        order = np.argsort(pos, kind="stable")
        return dict(position=pos[order], strand=strand[order], pct_hundredths=pct[order], nvalid=cov[order])

    def pileup_columns(self, mod_type: str, contigs=None):
        """Concatenated SoA columns (contig_id, position, strand, fraction_mod, nvalid) for a mod type."""
        idx = range(len(self.names)) if contigs is None else contigs
        cid, pos, strand, frac, cov = [], [], [], [], []
        for i in idx:
            p = self.contig_pileup(i, mod_type)
            cid.append(np.full(len(p["position"]), i, dtype=np.uint32))
            pos.append(p["position"])
            strand.append(p["strand"])
            frac.append(pct_to_fraction(p["pct_hundredths"]))
            cov.append(p["nvalid"])
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        return dict(contig_id=cat(cid, np.uint32), position=cat(pos, np.int64), strand=cat(strand, np.uint8),
                    fraction_mod=cat(frac, np.float64), nvalid=cat(cov, np.int32))

    # ---------------------------------------------------------------- text forms
    def write_fasta(self, path, width=80):
        with open(path, "w") as f:
            for i, name in enumerate(self.names):
                s = self.contig_str(i)
                f.write(f">{name}\n")
                for k in range(0, len(s), width):
                    f.write(s[k:k + width] + "\n")

    def write_contig_bin(self, path):
        with open(path, "w") as f:
            for name, b in zip(self.names, self.bin_names):
                f.write(f"{name}\t{b}\n")

    def write_bed(self, path):
        """modkit bedMethyl, 18 tab-separated columns, no header (dataload.py:15-34)."""
        with open(path, "w") as f:
            for i, name in enumerate(self.names):
                rows = []
                for mt in self.spec.mod_types:
                    p = self.contig_pileup(i, mt)
                    for pos, st, pct, cov in zip(p["position"].tolist(), p["strand"].tolist(),
                                                 p["pct_hundredths"].tolist(), p["nvalid"].tolist()):
                        nmod = int(round(cov * pct / 10000))
                        rows.append((pos, mt, chr(st), cov, pct, nmod))
                rows.sort(key=lambda r: (r[0], r[1]))
                for pos, mt, st, cov, pct, nmod in rows:
                    f.write(f"{name}\t{pos}\t{pos + 1}\t{mt}\t{cov}\t{st}\t{pos}\t{pos + 1}\t255,0,0\t"
                            f"{cov}\t{pct // 100}.{pct % 100:02d}\t{nmod}\t{cov - nmod}\t0\t0\t0\t0\t0\n")


def pct_to_fraction(pct_hundredths) -> np.ndarray:
    """fraction_mod exactly as the reference loader computes it from the text column.

    The bed column holds ``"%d.%02d" % divmod(h, 100)``; parsing that decimal gives the double
    nearest to h/100, which equals ``h / 100.0`` (correctly rounded division of two exactly
    representable integers); the loader then divides by 100 (dataload.py:85).
    """
    return (np.asarray(pct_hundredths, dtype=np.float64) / 100.0) / 100.0


def _popcount32(x):
    x = np.asarray(x, dtype=np.uint32)
    x = x - ((x >> _U32(1)) & _U32(0x55555555))
    x = (x & _U32(0x33333333)) + ((x >> _U32(2)) & _U32(0x33333333))
    x = (x + (x >> _U32(4))) & _U32(0x0F0F0F0F)
    return ((x * _U32(0x01010101)) >> _U32(24)).astype(np.int32)


def make_metagenome(spec: SynthSpec) -> SynthMetagenome:
    n = spec.n_contigs
    # contig lengths: log-normal from the hash (Box-Muller on two uniforms), rescaled to total_bp
    idx = np.arange(n, dtype=np.uint32)
    k = _U32((spec.seed * 0x2545F491 + 99) & 0xFFFFFFFF)
    u1 = (mix32(idx * _U32(2) + k).astype(np.float64) + 1.0) / 4294967297.0
    u2 = (mix32(idx * _U32(2) + _U32(1) + k).astype(np.float64) + 1.0) / 4294967297.0
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    w = np.exp(spec.lognormal_sigma * z) if n > 1 else np.ones(1)
    lengths = np.maximum((w / w.sum() * spec.total_bp).astype(np.int64), min(spec.min_contig_bp, spec.total_bp))
    lengths[np.argmax(lengths)] += spec.total_bp - int(lengths.sum())  # exact total
    assert lengths.min() > 0 and int(lengths.sum()) == spec.total_bp

    width = max(4, len(str(n)))
    names = [f"contig_{i:0{width}d}" for i in range(n)]
    bw = max(3, len(str(spec.n_bins)))
    # deal contigs to bins round-robin over the length-sorted order => balanced bin sizes
    order = np.argsort(-lengths, kind="stable")
    bin_names = [None] * n
    for r, i in enumerate(order.tolist()):
        bin_names[i] = f"bin_{r % spec.n_bins:0{bw}d}"
    bins = sorted(set(bin_names))
    hb = mix32(np.arange(len(bins), dtype=np.uint32) + _U32((spec.seed * 7919 + 13) & 0xFFFFFFFF))
    bin_gc = {b: 0.30 + 0.40 * float(hb[j]) / 4294967296.0 for j, b in enumerate(bins)}
    bin_motifs = {}
    for j, b in enumerate(bins):
        if spec.fixed_motifs is not None:
            bin_motifs[b] = [m for m in spec.fixed_motifs if m[2] in spec.mod_types]
            continue
        lo, hi = spec.motifs_per_bin
        hm = mix32(np.arange(8, dtype=np.uint32) + _U32((j * 8 + spec.seed * 104729) & 0xFFFFFFFF))
        count = lo + int(hm[0] % _U32(hi - lo + 1))
        avail = [m for m in MOTIF_LIBRARY if m[2] in spec.mod_types]
        chosen, t = [], 1
        while len(chosen) < min(count, len(avail)):
            cand = avail[((int(hm[t % 8]) + t * 2654435761) & 0xFFFFFFFF) % len(avail)]
            if cand not in chosen:
                chosen.append(cand)
            t += 1
        bin_motifs[b] = chosen
    return SynthMetagenome(spec, names, bin_names, lengths, bin_gc, bin_motifs)


def from_sequences(names, sequences, bin_names, motifs, seed: int = 1, mod_types=("a",), methylated_fraction: float = 0.97) -> SynthMetagenome:
    """A synthetic PILEUP on GIVEN contigs (e.g. the reference's packaged geobacillus plasmids, whose own pileup is not
    distributable — SURVEY §8(d) cfg 1): the sequences are taken as they are, the rows and the planted motifs
    ``[(iupac, pos, mod_type), ...]`` come from the same counter-based hash as in make_metagenome."""
    seqs = [s if isinstance(s, str) else bytes(s).decode("ascii") for s in sequences]
    lengths = np.array([len(s) for s in seqs], dtype=np.int64)
    spec = SynthSpec(n_contigs=len(names), total_bp=int(lengths.sum()), n_bins=len(set(bin_names)), mod_types=tuple(mod_types), seed=seed,
                     fixed_motifs=tuple(tuple(m) for m in motifs), methylated_fraction=methylated_fraction)
    bins = sorted(set(bin_names))
    mg = SynthMetagenome(spec, list(names), list(bin_names), lengths, {b: 0.5 for b in bins},
                         {b: [tuple(m) for m in motifs if m[2] in mod_types] for b in bins})
    for i, s in enumerate(seqs):                       # contig_codes() serves these instead of generating
        mg._seq_cache[i] = CODE_OF_ASCII[np.frombuffer(s.upper().encode("ascii"), dtype=np.uint8)]
    return mg


# named configurations of BASELINE.json ("configs")
def config(name: str) -> SynthSpec:
    if name == "cfg2":   # single 5 Mbp contig, 6mA
        return SynthSpec(n_contigs=1, total_bp=5_000_000, n_bins=1, mod_types=("a",), seed=1,
                         fixed_motifs=(("GATC", 1, "a"), ("GCACNNNNNNGTT", 2, "a"), ("AACNNNNNNGTGC", 1, "a")))
    if name == "cfg3":   # 100 Mbp, 1000 contigs, 50 bins, 6mA + 5mC
        return SynthSpec(n_contigs=1000, total_bp=100_000_000, n_bins=50, mod_types=("a", "m"), seed=1)
    if name in ("cfg4", "cfg5"):  # 1 Gbp, 10 000 contigs, 500 bins
        return SynthSpec(n_contigs=10_000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1)
    raise KeyError(name)


def random_candidates(n: int, seed: int = 2, mod_types=("a", "m")):
    """cfg 5 candidate table: stripped length U{4..15}, 3..8 specified positions,
    alphabet weights ACGT .70 / two-fold .20 / three-fold .05 / N .05, canonical base at a uniformly
    chosen mod_position.  Returns list of (regex-style motif string, mod_position, mod_type)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    single = ["A", "C", "G", "T"]
    two = ["[AG]", "[CT]", "[CG]", "[AT]", "[GT]", "[AC]"]
    three = ["[CGT]", "[AGT]", "[ACT]", "[ACG]"]
    out = []
    for i in range(n):
        mt = mod_types[i % len(mod_types)]
        length = int(rng.integers(4, 16))
        n_spec = int(min(rng.integers(3, 9), length))
        modpos = int(rng.integers(0, length))
        others = [p for p in range(length) if p != modpos]
        # first and last position must be specified so the motif is already stripped
        must = {0, length - 1} - {modpos}
        rest = [p for p in others if p not in must]
        k = max(0, n_spec - 1 - len(must))
        chosen = set(must) | set(rng.choice(rest, size=min(k, len(rest)), replace=False).tolist() if rest and k else [])
        chars = ["."] * length
        chars[modpos] = MOD_CANONICAL[mt]
        for p in sorted(chosen):
            r = rng.random()
            if r < 0.70:
                chars[p] = single[int(rng.integers(4))]
            elif r < 0.90:
                chars[p] = two[int(rng.integers(6))]
            elif r < 0.95:
                chars[p] = three[int(rng.integers(4))]
            else:
                chars[p] = "." if p not in (0, length - 1) else single[int(rng.integers(4))]
        out.append(("".join(chars), modpos, mt))
    return out
