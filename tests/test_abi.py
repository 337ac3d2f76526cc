"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/nmscan.h declares
(no compute calls — there is no GPU in the build container)."""
import ctypes
import os
import re

from nanomotif_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nmscan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[a-z_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols():
    path = build.build()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 12 and set(syms) == set(_lib.SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in nmscan.h but not exported"
    lib.nm_abi_version.restype = ctypes.c_int
    assert lib.nm_abi_version() == 1


def test_bad_arguments_fail_loudly_without_gpu():
    lib = _lib.load()
    assert lib.nm_ctx_create(0, None) == -1          # NM_EINVAL, no HIP call made
    assert b"NULL" in lib.nm_last_error()
    assert lib.nm_score_batch(None, 0, None, None, None, None, None, None, None) == -1


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it;
    nothing under nanomotif_amd/ (the product) may import it, and the product has no CPU fallback for the engine."""
    import ast
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "nanomotif_amd", "**", "*.py"), recursive=True):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{path} imports the oracle"
    src = open(os.path.join(root, "nanomotif_amd", "_lib.py")).read()
    assert "no CPU fallback" in src and "raise NmScanError" in src


def test_console_entry_point_is_the_references():
    """setup.py:48-52 of the reference installs `nanomotif = nanomotif.main:main`; pyproject.toml installs the same command
    on this package, and the target parses the reference's motif_discovery flags (argparser.py:101-136)."""
    import importlib
    import tomli
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        scripts = tomli.load(f)["project"]["scripts"]
    assert list(scripts) == ["nanomotif"]
    mod, fn = scripts["nanomotif"].split(":")
    main = getattr(importlib.import_module(mod), fn)
    assert callable(main)
    from nanomotif_amd.argparser import create_parser
    a = create_parser().parse_args(["motif_discovery", "a.fasta", "p.bed", "-c", "cb.tsv", "--out", "o", "-t", "3", "--seed", "7",
                                    "--search_frame_size", "40", "--min_motif_score", "0.2", "--minimum_kl_divergence", "0.05"])
    assert (a.command, a.assembly, a.pileup, a.contig_bin, a.out, a.threads, a.seed) == ("motif_discovery", "a.fasta", "p.bed", "cb.tsv", "o", 3, 7)


def test_status_codes_of_the_python_host_are_the_headers():
    """The nm_status values the Python host compares with (``_lib.NM_E*``; ``NmScanError.code``) are the header's."""
    import re
    from nanomotif_amd import _lib
    text = open(os.path.join(ROOT, "include", "nmscan.h")).read()
    enum = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\b(NM_(?:OK|E[A-Z]+))\s*=\s*(-?\d+)", text))
    assert enum["NM_OK"] == 0 and len(set(enum.values())) == len(enum) >= 9
    for name in ("NM_EINDEX", "NM_EDECLINED", "NM_ESEQUENCE"):
        assert getattr(_lib, name) == enum[name], name
    e = _lib.NmScanError("x")
    assert e.code is None
