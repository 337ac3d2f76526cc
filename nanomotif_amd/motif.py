"""Host-side motif type of the MI355X build — mirrors the interface of the reference's ``Motif``
(nanomotif/motif.py:18-359) and the IUPAC helpers (seq.py:539-601, utils.py:12-34).

Unlike the reference (string surgery everywhere) a motif is parsed ONCE into a tuple of 4-bit base
sets (bit0 A, bit1 C, bit2 G, bit3 T; ``ANY`` = 15 for ``.``), which is also the form the HIP engine
consumes (include/nmscan.h: cand_masks).  All relations are evaluated on those sets; the string form
is kept only because it is the identity of a motif in the reference's graph / TSV output.
"""
from __future__ import annotations

import itertools
import re
from functools import lru_cache, reduce

import numpy as np

A, C, G, T, ANY = 1, 2, 4, 8, 15
_BIT = {"A": A, "C": C, "G": G, "T": T}
_LETTER = {A: "A", C: "C", G: "G", T: "T"}
#                       A  T  G  C   — the reference's PSSM / one-hot row order (constants.py:1)
ONE_HOT_ORDER = (A, T, G, C)
BASES = ["A", "T", "G", "C"]
MOD_TYPE_TO_CANONICAL = {"m": "C", "a": "A", "21839": "C"}
_IUPAC_OF_SET = {A: "A", C: "C", G: "G", T: "T", A | G: "R", C | T: "Y", C | G: "S", A | T: "W", G | T: "K",
                 A | C: "M", C | G | T: "B", A | G | T: "D", A | C | T: "H", A | C | G: "V", ANY: "N"}
_SET_OF_IUPAC = {v: k for k, v in _IUPAC_OF_SET.items()}
_COMP_CHAR = str.maketrans("ATGCNRYSWKMBDHV.[]", "TACGNYRSWMKVHDB.][")


def complement_set(m: int) -> int:
    return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3)


def set_to_token(m: int) -> str:
    """Regex-style token for a base set: literal, ``.`` for all four, sorted ``[..]`` otherwise."""
    if m == ANY:
        return "."
    if m in _LETTER:
        return _LETTER[m]
    return "[" + "".join(ch for ch in "ACGT" if _BIT[ch] & m) + "]"


_SET_OF_BYTE = bytes(ANY if chr(i) in ".N" else _BIT.get(chr(i), 0) for i in range(256))
_TOKEN = re.compile(r"\[[^\]]*\]|.", re.S)
_SET_OF_TOKEN = {".": ANY, "N": ANY, "A": A, "C": C, "G": G, "T": T}


def _set_of_token(tok: str) -> int:
    m = _SET_OF_TOKEN.get(tok)
    if m is None:
        if tok == "[":
            raise ValueError("Unmatched bracket")
        m = 0
        if tok[0] == "[":
            for b in tok[1:-1]:
                m |= _BIT.get(b, 0)
        _SET_OF_TOKEN[tok] = m                  # other single characters: the empty set
    return m


@lru_cache(maxsize=1 << 16)
def _parse(s: str):
    """-> (tokens, sets).  ``N`` inside a regex-style motif is kept as its own token (the reference
    special-cases it in get_parent_scores / isolated-base checks) with set ANY.  Tokenised by one regular expression:
    a search creates thousands of 41-character strings, a character loop in the interpreter was half of the cold
    post-processing time."""
    if "[" not in s:                            # one character per token: both tuples come straight out of C code
        return tuple(s), tuple(s.encode("latin-1", "replace").translate(_SET_OF_BYTE))
    toks = tuple(_TOKEN.findall(s))
    return toks, tuple([_set_of_token(t) for t in toks])


def regex_to_iupac(regex_str: str) -> str:
    toks, sets = _parse(regex_str)
    return "".join("N" if t == "." else (_IUPAC_OF_SET.get(m, "") if t.startswith("[") else t)
                   for t, m in zip(toks, sets))


def iupac_to_regex(iupac: str) -> str:
    return "".join(set_to_token(_SET_OF_IUPAC[ch]) for ch in iupac)


def reverse_compliment(seq: str) -> str:
    return seq.translate(_COMP_CHAR)[::-1]


def motif_type(iupac: str) -> str:
    """utils.py:26-34."""
    runs = re.findall(r"N{2,}", iupac)
    if len(runs) >= 2:
        return "ambiguous"
    if any(len(r) >= 3 for r in runs):
        return "bipartite"
    return "palindrome" if reverse_compliment(iupac) == iupac else "non-palindrome"


def _tok_subset(ta: str, ma: int, tb: str, mb: int) -> bool:
    """``set(ta) <= set(tb)`` on the reference's character sets (brackets are characters there)."""
    if ta == ".":
        return tb == "."
    if tb == "." or ta == "N" or tb == "N":
        return ta == tb
    return (ma & ~mb) == 0 and not (ta.startswith("[") and not tb.startswith("["))


class Motif(str):
    """``str`` subclass + ``mod_position`` (index into the bracket-aware split) — motif.py:18-36."""

    def __new__(cls, motif_string, *a, **k):
        return str.__new__(cls, motif_string)

    def __init__(self, _, mod_position):
        self.mod_position = mod_position
        self.string = str.__str__(self)
        self.tokens, self.sets = _parse(self.string)

    def __eq__(self, other):
        return isinstance(other, Motif) and self.mod_position == other.mod_position and self.string == other.string

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.string, self.mod_position))

    def __repr__(self):
        return f"Motif({self.string!r}, pos={self.mod_position})"

    def __reduce__(self):
        return (Motif, (self.string, self.mod_position))

    # ---- structure
    @classmethod
    def from_sets(cls, sets, mod_position):
        return cls("".join(set_to_token(m) for m in sets), mod_position)

    def split(self):
        return list(self.tokens)

    def length(self):
        return len(self.tokens)

    def trimmed_length(self):
        return len(self.tokens) - self.string.count(".")

    def strip(self, character="."):
        return self.string.strip(character)

    def _dot_bounds(self):
        n = len(self.tokens)
        lo = 0
        while lo < n and self.tokens[lo] == ".":
            lo += 1
        hi = n
        while hi > lo and self.tokens[hi - 1] == ".":
            hi -= 1
        return lo, hi

    def new_stripped_motif(self, character="."):
        cached = self.__dict__.get("_stripped")            # (motifs are immutable; post-processing asks thousands of times)
        if cached is not None:
            return cached
        lo, hi = self._dot_bounds()
        if lo == len(self.tokens):
            out = self
        elif lo == 0 and hi == len(self.tokens):
            out = Motif(self.string, self.mod_position)
            out.__dict__["_stripped"] = out
        else:
            out = Motif("".join(self.tokens[lo:hi]), self.mod_position - lo)
            out.__dict__["_stripped"] = out
        self.__dict__["_stripped"] = out
        return out

    def stripped_sets(self):
        """(sets of the stripped motif as uint8 array, stripped mod_position) — the engine's candidate form."""
        lo, hi = self._dot_bounds()
        if lo == len(self.tokens):       # all dots: nothing to strip (motif.py:218-219); the engine rejects it
            return np.array(self.sets, dtype=np.uint8), self.mod_position
        return np.array(self.sets[lo:hi], dtype=np.uint8), self.mod_position - lo

    def reverse_compliment(self):
        return Motif(reverse_compliment(self.string), len(self.tokens) - self.mod_position - 1)

    def one_hot(self):
        arr = np.zeros((len(self.sets), 4), dtype=int)
        for i, (t, m) in enumerate(zip(self.tokens, self.sets)):
            for k, b in enumerate(ONE_HOT_ORDER):
                arr[i, k] = 1 if m & b else 0
        return arr

    def iupac(self):
        return regex_to_iupac(self.string)

    def from_iupac(self):
        return Motif(iupac_to_regex(self.string), self.mod_position)

    def identical(self, other):
        return self == other

    # ---- relations (motif.py:57-158)
    def sub_motif_of(self, other):
        if self.string == other.string:
            return False
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        if len(a.tokens) < len(b.tokens):
            return False
        off = b.mod_position - a.mod_position
        if off > 0:
            return False
        for i in range(len(a.tokens)):
            k = i + off
            if k < 0:
                continue
            if k >= len(b.tokens):
                return True
            if b.tokens[k] != "." and not _tok_subset(a.tokens[i], a.sets[i], b.tokens[k], b.sets[k]):
                return False
        return True

    def sub_motif_of_any(self, others):
        return any(self.sub_motif_of(o) for o in others)

    def sub_string_of(self, other):
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        if a.string == b.string:
            return False
        na, nb = len(a.tokens), len(b.tokens)
        for shift in range(na - nb + 1):
            if all(j + shift >= na or b.tokens[j] == "."
                   or _tok_subset(a.tokens[j + shift], a.sets[j + shift], b.tokens[j], b.sets[j]) for j in range(nb)):
                return True
        return False

    def distance(self, other):
        s0, s1 = -self.mod_position, -other.mod_position
        e0, e1 = len(self.tokens) - self.mod_position, len(other.tokens) - other.mod_position
        d = 0
        for i in range(min(s0, s1), max(e0, e1)):
            # the reference's elif ladder (motif.py:140-157): left overhang first, then right overhang
            if i < s0:
                d += other.tokens[i - s1] != "."
            elif i < s1:
                d += self.tokens[i - s0] != "."
            elif i >= e0:
                d += other.tokens[i - s1] != "."
            elif i >= e1:
                d += self.tokens[i - s0] != "."
            else:
                ta, tb = self.tokens[i - s0], other.tokens[i - s1]
                same = (ta == tb) or (ta not in ".N" and tb not in ".N" and self.sets[i - s0] == other.sets[i - s1]
                                      and ta.startswith("[") == tb.startswith("["))
                d += not same
        return int(d)

    def _isolated(self, k):
        toks = self.tokens
        n = len(toks)
        cnt = 0
        for p, t in enumerate(toks):
            if t == ".":
                continue
            nb = toks[max(p - k, 0):p] + toks[p + 1:min(p + k + 1, n - 1)]
            if nb and all(x == "." for x in nb):
                cnt += 1
            if nb and all(x == "N" for x in nb):
                cnt += 1
        return cnt

    def have_isolated_bases(self, isolation_size=2):
        return self._isolated(isolation_size) > 0

    def count_isolated_bases(self, isolation_size=2):
        return self._isolated(isolation_size)

    # ---- merging (motif.py:268-352)
    @staticmethod
    def merge_bases(b1, b2):
        (t1,), (m1,) = _parse(b1)
        (t2,), (m2,) = _parse(b2)
        if t1 == "." or t2 == ".":
            return "."
        m = m1 | m2
        return set_to_token(m)

    @staticmethod
    def _merge_sets(ta, ma, tb, mb):
        return ANY if (ta == "." or tb == ".") else (ma | mb)

    def merge(self, other):
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        ia, ib = 0, 0
        off = a.mod_position - b.mod_position
        if off > 0:
            ia = off
        elif off < 0:
            ib = -off
        n = min(len(a.tokens) - ia, len(b.tokens) - ib)
        sets = [self._merge_sets(a.tokens[ia + k], a.sets[ia + k], b.tokens[ib + k], b.sets[ib + k]) for k in range(max(n, 0))]
        return Motif.from_sets(sets, min(a.mod_position, b.mod_position)).new_stripped_motif(".")

    def merge_no_strip(self, other):
        ia, ib = 0, 0
        off = self.mod_position - other.mod_position
        if off > 0:
            ia = off
        elif off < 0:
            ib = -off
        n = min(len(self.tokens) - ia, len(other.tokens) - ib)
        sets = [self._merge_sets(self.tokens[ia + k], self.sets[ia + k], other.tokens[ib + k], other.sets[ib + k])
                for k in range(max(n, 0))]
        s = "." * abs(off) + "".join(set_to_token(m) for m in sets)
        ra, rb = len(self.tokens) - self.mod_position, len(other.tokens) - other.mod_position
        s += "." * abs(ra - rb)
        return Motif(s, min(self.mod_position, other.mod_position))

    def explode_motif(self):
        opts = [[_LETTER[b] for b in (A, C, G, T) if m & b] if t.startswith("[") else [t]
                for t, m in zip(self.tokens, self.sets)]
        return [Motif("".join(c), self.mod_position) for c in itertools.product(*opts)]


def align_motifs(motifs):
    """motif.py:362-387."""
    if not motifs:
        return []
    mx = max(m.mod_position for m in motifs)
    left = [Motif("." * (mx - m.mod_position) + m.string, mx) for m in motifs]
    width = max(m.length() for m in left)
    return [Motif(m.string + "." * (width - m.length()), mx) for m in left]


def explode_with_mask(motif, mask):
    """motif.py:389-416."""
    opts = []
    for i in mask:
        t, m = motif.tokens[i], motif.sets[i]
        opts.append(["A", "C", "G", "T"] if t == "." else ([ch for ch in t[1:-1]] if t.startswith("[") else [t]))
    out = set()
    base = list(motif.tokens)
    for combo in itertools.product(*opts):
        for k, p in enumerate(mask):
            base[p] = combo[k]
        out.add(Motif("".join(base), motif.mod_position))
    return out


def merge_and_find_new_variants(motifs):
    """motif.py:484-519."""
    if not motifs:
        return None, set(), set()
    motifs = align_motifs(motifs)
    width = motifs[0].length()
    mask = [i for i in range(width) if any(m.tokens[i] != "." for m in motifs)]
    pre = set()
    for m in motifs:
        pre |= explode_with_mask(m, mask)
    merged = reduce(lambda a, b: a.merge_no_strip(b), motifs)
    new = explode_with_mask(merged, mask) - pre
    return (merged.new_stripped_motif(), {v.new_stripped_motif() for v in pre},
            {v.new_stripped_motif() for v in new})


def _maximal_cliques(nodes, adj):
    out = []

    def expand(r, p, x):
        if not p and not x:
            out.append(sorted(r, key=lambda m: (m.string, m.mod_position)))
            return
        pivot = max(p | x, key=lambda u: len(adj[u] & p))
        for v in sorted(p - adj[pivot], key=lambda m: (m.string, m.mod_position)):
            expand(r | {v}, p & adj[v], x & adj[v])
            p = p - {v}
            x = x | {v}

    expand(set(), set(nodes), set())
    return out


def merge_motifs(motifs, connectivity_dist=2, min_length=4):
    """motif.py:522-560 -> list of [merged, cluster, pre_variants, new_variants] (clique order deterministic)."""
    keep = []
    for m in motifs:
        if m.trimmed_length() > min_length and m not in keep:
            keep.append(m)
    adj = {m: set() for m in keep}
    for i, a in enumerate(keep):
        for b in keep[i + 1:]:
            if a.distance(b) <= connectivity_dist:
                adj[a].add(b)
                adj[b].add(a)
    res = []
    for cluster in _maximal_cliques(keep, adj):
        if len(cluster) == 1:
            continue
        merged, pre, new = merge_and_find_new_variants(cluster)
        if merged is None or merged.trimmed_length() < min_length:
            continue
        res.append([merged, cluster, pre, new])
    return res
