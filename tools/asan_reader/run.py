"""Writes good and damaged pileup / FASTA files and runs the host readers over them under AddressSanitizer + UBSan
(tools/asan_reader/driver.cpp linked with nanomotif_amd/csrc/nmbed.cpp).  CPU only.  Exit code 0: no sanitizer report,
every file loaded or refused with a message."""
import gzip
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from helpers import write_bgzf_tabix
from nanomotif_amd import synth

tmp = tempfile.mkdtemp(prefix="nm_asan_")
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=5, total_bp=200_000, n_bins=2, mod_types=("a", "m"), seed=3, min_contig_bp=20_000))
bed = os.path.join(tmp, "p.bed")
mg.write_bed(bed)
mg.write_fasta(os.path.join(tmp, "a.fasta"))
text = open(bed, "rb").read()
rng = np.random.default_rng(5)
cases = [("bed", bed), ("t1", bed), ("counts", bed)]


def put(name, data):
    p = os.path.join(tmp, name)
    open(p, "wb").write(data)
    return p


# text damage: truncated mid-line, no final newline, empty, only newlines, huge field, NUL bytes, missing columns, CRLF
lines = text.split(b"\n")
cases += [("bed", put("cut.bed", text[:len(text) // 2 + 7])), ("bed", put("nonl.bed", text.rstrip(b"\n"))), ("bed", put("empty.bed", b"")),
          ("bed", put("nl.bed", b"\n\n\n")), ("bed", put("long.bed", lines[0][:10] + b"9" * 5000 + lines[0][10:] + b"\n" + text[:5000])),
          ("bed", put("nul.bed", text[:3000] + b"\0\0\0" + text[3000:9000])), ("bed", put("cols.bed", b"\n".join(l[:l.rfind(b"\t", 0, 40)] for l in lines[:50]) + b"\n")),
          ("bed", put("crlf.bed", text[:20000].replace(b"\n", b"\r\n"))), ("counts", put("cols12.bed", b"\n".join(b"\t".join(l.split(b"\t")[:12]) for l in lines[:50]) + b"\n"))]
for k in range(6):                                             # random byte damage in the text
    d = bytearray(text[:60_000])
    for at in rng.integers(0, len(d), 40):
        d[at] = int(rng.integers(0, 256))
    cases.append(("bed", put(f"noise{k}.bed", bytes(d))))
# gzip / bgzf
with gzip.open(os.path.join(tmp, "p.bed.gz"), "wb") as g:
    g.write(text)
cases.append(("bed", os.path.join(tmp, "p.bed.gz")))
gz_raw = open(os.path.join(tmp, "p.bed.gz"), "rb").read()
cases += [("bed", put("cutgz.bed.gz", gz_raw[:len(gz_raw) // 2])), ("bed", put("tailgz.bed.gz", gz_raw + b"garbage"))]
for bs, lvl in ((0xFF00, 6), (3000, 6), (20000, 0)):
    p = os.path.join(tmp, f"b{bs}_{lvl}.bed.gz")
    write_bgzf_tabix(text, p, block_size=bs, level=lvl)
    cases.append(("bed", p))
    cases.append(("bed", p, p + ".tbi", ",".join(mg.names[1::2] + ["nope"])))
    raw = bytearray(open(p, "rb").read())
    tbi = open(p + ".tbi", "rb").read()
    for k in range(8):                                         # damaged blocks: payload, headers (BSIZE, XLEN), trailers
        d = bytearray(raw)
        for at in rng.integers(0, len(d), 1 + k):
            d[at] ^= 1 << int(rng.integers(0, 8))
        q = put(f"b{bs}_{lvl}_dmg{k}.bed.gz", bytes(d))
        open(q + ".tbi", "wb").write(tbi)
        cases.append(("bed", q))
        cases.append(("bed", q, q + ".tbi", ",".join(mg.names[:3])))
    cases.append(("bed", put(f"b{bs}_{lvl}_cut.bed.gz", bytes(raw[:len(raw) * 2 // 3]))))
    # a trailer that claims more text than a BGZF block may hold (64 KiB): refused, whole file and tabix subset
    import struct
    d, off, heads = bytearray(raw), 0, []
    while off < len(d):
        heads.append(off)
        off += struct.unpack_from("<H", d, off + 16)[0] + 1
    struct.pack_into("<I", d, (heads[len(heads) // 2 + 1] if len(heads) > 2 else len(d)) - 4, 70_000)
    q = put(f"b{bs}_{lvl}_isize.bed.gz", bytes(d))
    open(q + ".tbi", "wb").write(tbi)
    cases.append(("bed", q))
    cases.append(("bed", q, q + ".tbi", ",".join(mg.names)))
    # damaged indexes: truncated, random bytes (re-compressed so that they inflate), not an index
    import zlib
    idx_raw = gzip.decompress(tbi)
    for k in range(6):
        d = bytearray(idx_raw)
        for at in rng.integers(0, len(d), 1 + 2 * k):
            d[at] = int(rng.integers(0, 256))
        q = os.path.join(tmp, f"b{bs}_{lvl}_idx{k}.tbi")
        with gzip.open(q, "wb") as g:
            g.write(bytes(d[:len(d) - (k % 3) * 11]))
        cases.append(("bed", p, q, ",".join(mg.names)))
    cases.append(("bed", p, bed, mg.names[0]))
# FASTA damage
fa = open(os.path.join(tmp, "a.fasta"), "rb").read()
cases += [("fasta", os.path.join(tmp, "a.fasta")), ("fasta", put("cut.fasta", fa[:len(fa) // 3])), ("fasta", put("nohdr.fasta", fa[fa.find(b"\n") + 1:])),
          ("fasta", put("empty.fasta", b"")), ("fasta", put("hdrs.fasta", b">a\n>b\n>c\n")), ("fasta", put("crlf.fasta", fa[:30000].replace(b"\n", b"\r\n")))]
with gzip.open(os.path.join(tmp, "a.fasta.gz"), "wb") as g:
    g.write(fa)
cases.append(("fasta", os.path.join(tmp, "a.fasta.gz")))
listing = os.path.join(tmp, "cases.txt")
open(listing, "w").write("".join("\t".join(c) + "\n" for c in cases))
exe = os.path.join(tmp, "driver")
subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                os.path.join(ROOT, "tools/asan_reader/driver.cpp"), os.path.join(ROOT, "nanomotif_amd/csrc/nmbed.cpp"), "-lz", "-o", exe], check=True)
r = subprocess.run([exe, listing], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
out = r.stdout.splitlines()
print("\n".join(l.replace(tmp + "/", "") for l in out[:12]), "\n...")
print(f"{len(cases)} cases: {sum('loaded' in l or 'records' in l for l in out)} loaded, {sum('refused' in l for l in out)} refused; exit code {r.returncode}")
if r.returncode or r.stderr.strip():
    print(r.stderr[-4000:])
    sys.exit(1)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
