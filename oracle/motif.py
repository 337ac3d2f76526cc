"""Oracle restatement of the motif data type (reference: nanomotif/motif.py:18-560, seq.py:539-601,
utils.py:12-34, constants.py).  Test infrastructure only — see oracle/__init__.py."""
from __future__ import annotations

import itertools
import re
from functools import reduce

import numpy as np

BASES = ["A", "T", "G", "C"]                       # constants.py:1 — PSSM / one-hot row order
ONE_HOT = {"A": (1, 0, 0, 0), "T": (0, 1, 0, 0), "G": (0, 0, 1, 0), "C": (0, 0, 0, 1),
           "N": (1, 1, 1, 1), ".": (1, 1, 1, 1)}   # constants.py:21-28
COMPLEMENT = {"A": "T", "T": "A", "G": "C", "C": "G", "N": "N", "R": "Y", "Y": "R", "S": "S", "W": "W",
              "K": "M", "M": "K", "B": "V", "D": "H", "H": "D", "V": "B", ".": ".", "[": "]", "]": "["}
MOD_TYPE_TO_CANONICAL = {"m": "C", "a": "A", "21839": "C"}  # constants.py:29-33

_SET_TO_IUPAC = {"A": "A", "T": "T", "C": "C", "G": "G", "AG": "R", "CT": "Y", "CG": "S", "AT": "W", "GT": "K",
                 "AC": "M", "CGT": "B", "AGT": "D", "ACT": "H", "ACG": "V"}
_IUPAC_TO_SET = {v: k for k, v in _SET_TO_IUPAC.items()}
_IUPAC_TO_SET["N"] = "."


def regex_to_iupac(regex_str: str) -> str:
    """seq.py:539-569 — ``[..]`` -> IUPAC letter (sorted contents, unknown -> ''), ``.`` -> ``N``."""
    def repl(m):
        return _SET_TO_IUPAC.get("".join(sorted(m.group(1))), "")
    return re.sub(r"\[([A|T|C|G]+)\]", repl, regex_str).replace(".", "N")


def iupac_to_regex(iupac: str) -> str:
    """seq.py:571-601."""
    out = []
    for ch in iupac:
        s = _IUPAC_TO_SET[ch]
        out.append(s if len(s) == 1 else "[" + s + "]")
    return "".join(out)


def reverse_compliment(seq: str) -> str:
    """seq.py:645-647."""
    return "".join(COMPLEMENT[c] for c in reversed(seq))


def motif_type(iupac: str) -> str:
    """utils.py:12-34: >=2 runs of N{2,} -> ambiguous; any N{3,} -> bipartite; palindrome test last."""
    if len(re.findall(r"(N){2,}", iupac)) >= 2:
        return "ambiguous"
    if re.search(r"(N){3,}", iupac):
        return "bipartite"
    if reverse_compliment(iupac) == iupac:
        return "palindrome"
    return "non-palindrome"


class Motif(str):
    """``str`` subclass carrying ``mod_position`` (index into the bracket-aware split), motif.py:18-36."""

    def __new__(cls, motif_string, *a, **k):
        return str.__new__(cls, motif_string)

    def __init__(self, _, mod_position):
        self.mod_position = mod_position
        self.string = str.__str__(self)

    def __eq__(self, other):
        return isinstance(other, Motif) and self.string == other.string and self.mod_position == other.mod_position

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.string, self.mod_position))

    def __repr__(self):
        return f"Motif({self.string!r}, pos={self.mod_position})"

    # -- structure ---------------------------------------------------------------------------
    def split(self) -> list:
        """motif.py:226-245."""
        s, out, i = self.string, [], 0
        while i < len(s):
            if s[i] == "[":
                j = s.find("]", i)
                if j < 0:
                    raise ValueError("Unmatched bracket")
                out.append(s[i:j + 1])
                i = j + 1
            else:
                out.append(s[i])
                i += 1
        return out

    def length(self) -> int:
        return len(self.split())

    def trimmed_length(self) -> int:
        return len(self.split()) - self.string.count(".")   # motif.py:201-205

    def strip(self, character="."):
        return self.string.lstrip(character).rstrip(character)

    def new_stripped_motif(self, character="."):
        """motif.py:213-224."""
        m = re.search("[^.]", self.string)
        if m is None:
            return self
        return Motif(self.strip(character), self.mod_position - m.start())

    def reverse_compliment(self):
        """motif.py:260-266 — per CHARACTER reversal (brackets swap, so they stay balanced)."""
        return Motif(reverse_compliment(self.string), self.length() - self.mod_position - 1)

    def one_hot(self) -> np.ndarray:
        """motif.py:247-258 — rows = split positions, columns A,T,G,C; bracket chars are skipped."""
        sp = self.split()
        arr = np.zeros((len(sp), 4), dtype=int)
        for i, tok in enumerate(sp):
            for ch in tok:
                if ch in ONE_HOT:
                    arr[i, :] += ONE_HOT[ch]
        return arr

    def iupac(self):
        return regex_to_iupac(self.string)

    def from_iupac(self):
        return Motif(iupac_to_regex(self.string), self.mod_position)

    def identical(self, other):
        return self.string == other.string and self.mod_position == other.mod_position

    # -- relations ---------------------------------------------------------------------------
    def sub_motif_of(self, other) -> bool:
        """motif.py:57-87."""
        if self.string == other.string:
            return False
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        if a.length() < b.length():
            return False
        asp, bsp = a.split(), b.split()
        off = b.mod_position - a.mod_position
        if off > 0:
            return False
        for i, tok in enumerate(asp):
            k = i + off
            if k < 0:
                continue
            if k >= len(bsp):
                return True
            if not set(tok) <= set(bsp[k]) and bsp[k] != ".":
                return False
        return True

    def sub_motif_of_any(self, others) -> bool:
        return any(self.sub_motif_of(o) for o in others)

    def sub_string_of(self, other) -> bool:
        """motif.py:98-125."""
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        if a.string == b.string:
            return False
        asp, bsp = a.split(), b.split()
        for shift in range(len(asp) - len(bsp) + 1):
            ok = True
            for j, tok in enumerate(bsp):
                if j + shift >= len(asp):
                    continue
                if not set(asp[j + shift]) <= set(tok) and tok != ".":
                    ok = False
            if ok:
                return True
        return False

    def distance(self, other) -> int:
        """motif.py:127-158 — mismatching aligned positions + specified overhang positions."""
        a, b = self.split(), other.split()
        s0, s1 = -self.mod_position, -other.mod_position
        e0, e1 = len(a) - self.mod_position, len(b) - other.mod_position
        d = 0
        for i in range(min(s0, s1), max(e0, e1)):
            if i < s0:
                d += b[i - s1] != "."
            elif i < s1:
                d += a[i - s0] != "."
            elif i >= e0:
                d += b[i - s1] != "."
            elif i >= e1:
                d += a[i - s0] != "."
            elif set(a[i - s0]) != set(b[i - s1]):
                d += 1
        return int(d)

    def _isolated(self, isolation_size):
        sp = self.split()
        n = 0
        for p, tok in enumerate(sp):
            if tok == ".":
                continue
            lo = max(p - isolation_size, 0)
            hi = min(p + isolation_size + 1, len(sp) - 1)     # motif.py:170 — note the len-1 cap
            nb = set(sp[lo:p] + sp[p + 1:hi])
            if nb == {"."}:
                n += 1
            if nb == {"N"}:
                n += 1
        return n

    def have_isolated_bases(self, isolation_size=2) -> bool:
        return self._isolated(isolation_size) > 0              # motif.py:160-176

    def count_isolated_bases(self, isolation_size=2) -> int:
        return self._isolated(isolation_size)                  # motif.py:178-194

    # -- merging -----------------------------------------------------------------------------
    @staticmethod
    def merge_bases(b1, b2):
        """motif.py:335-352."""
        canon = {"A", "C", "G", "T"}
        s = (set(b1[1:-1]) if "[" in b1 else set(b1)) | (set(b2[1:-1]) if "[" in b2 else set(b2))
        if s == canon or "." in s:
            return "."
        if len(s & canon) > 1:
            return "[" + "".join(sorted(s & canon)) + "]"
        return "".join(s)

    def merge(self, other):
        """motif.py:268-292."""
        a, b = self.new_stripped_motif(), other.new_stripped_motif()
        asp, bsp = a.split(), b.split()
        off = a.mod_position - b.mod_position
        if off > 0:
            asp = asp[off:]
        elif off < 0:
            bsp = bsp[-off:]
        n = min(len(asp), len(bsp))
        s = "".join(self.merge_bases(x, y) for x, y in zip(asp[:n], bsp[:n]))
        return Motif(s, min(a.mod_position, b.mod_position)).new_stripped_motif(".")

    def merge_no_strip(self, other):
        """motif.py:294-333."""
        asp, bsp = self.split(), other.split()
        la, lb = len(asp), len(bsp)
        off = self.mod_position - other.mod_position
        if off > 0:
            asp = asp[off:]
        elif off < 0:
            bsp = bsp[-off:]
        n = min(len(asp), len(bsp))
        s = "".join(self.merge_bases(x, y) for x, y in zip(asp[:n], bsp[:n]))
        pos = min(self.mod_position, other.mod_position)
        if self.mod_position != other.mod_position:
            s = "." * abs(self.mod_position - other.mod_position) + s
        ra, rb = la - self.mod_position, lb - other.mod_position
        if ra != rb:
            s = s + "." * abs(ra - rb)
        return Motif(s, pos)

    def explode_motif(self):
        parts = [list(t[1:-1]) if t.startswith("[") else [t] for t in self.split()]
        return [Motif("".join(c), self.mod_position) for c in itertools.product(*parts)]


def align_motifs(motifs):
    """motif.py:362-387 — left-pad to a common mod_position, right-pad to a common length."""
    if not motifs:
        return []
    mx = max(m.mod_position for m in motifs)
    left = [Motif("." * (mx - m.mod_position) + m.string, mx) for m in motifs]
    L = max(m.length() for m in left)
    return [Motif(m.string + "." * (L - m.length()), m.mod_position) for m in left]


def explode_with_mask(motif, mask):
    """motif.py:389-416 — expand only the masked positions ('.' -> ACGT, brackets -> members)."""
    seg = motif.split()
    opts = []
    for i in mask:
        t = seg[i]
        opts.append(["A", "C", "G", "T"] if t == "." else (list(t[1:-1]) if t.startswith("[") else [t]))
    out = set()
    for combo in itertools.product(*opts):
        s = list(seg)
        for k, p in enumerate(mask):
            s[p] = combo[k]
        out.add(Motif("".join(s), motif.mod_position))
    return out


def merge_and_find_new_variants(motifs):
    """motif.py:484-519."""
    if not motifs:
        return None, set(), set()
    motifs = align_motifs(motifs)
    segs = [m.split() for m in motifs]
    mask = [i for i in range(len(segs[0])) if any(s[i] != "." for s in segs)]
    pre = set()
    for m in motifs:
        pre |= explode_with_mask(m, mask)
    merged = reduce(lambda a, b: a.merge_no_strip(b), motifs)
    new = explode_with_mask(merged, mask) - pre
    return (merged.new_stripped_motif(), {v.new_stripped_motif() for v in pre},
            {v.new_stripped_motif() for v in new})


def maximal_cliques(nodes, adj):
    """All maximal cliques (Bron–Kerbosch with pivot); replaces networkx.find_cliques (motif.py:481-482)."""
    out = []

    def bk(r, p, x):
        if not p and not x:
            out.append(list(r))
            return
        pivot = max(p | x, key=lambda u: len(adj[u] & p))
        for v in list(p - adj[pivot]):
            bk(r | {v}, p & adj[v], x & adj[v])
            p = p - {v}
            x = x | {v}

    bk(set(), set(nodes), set())
    return out


def merge_motifs(motifs, connectivity_dist=2, min_length=4):
    """motif.py:522-560 — returns list of [merged, cluster, pre_variants, new_variants]."""
    motifs = [m for m in motifs if m.trimmed_length() > min_length]
    nodes = []
    for m in motifs:
        if m not in nodes:
            nodes.append(m)
    adj = {m: set() for m in nodes}
    for a in nodes:
        for b in nodes:
            if not a.identical(b) and a.distance(b) <= connectivity_dist:
                adj[a].add(b)
                adj[b].add(a)
    res = []
    for cluster in maximal_cliques(nodes, adj):
        if len(cluster) == 1:
            continue
        merged, pre, new = merge_and_find_new_variants(list(cluster))
        if merged is None or merged.trimmed_length() < min_length:
            continue
        res.append([merged, list(cluster), pre, new])
    return res
