"""Command line of the MI355X build: the ``motif_discovery`` sub-command with the reference's arguments
(nanomotif/argparser.py:13-136), plus ``--device`` for the GPU to use.  The reference's other sub-commands
(detect_contamination, include_contigs, MTase-linker) are out of scope."""
import argparse

__version__ = "1.1.2+mi355x.r1"


def create_parser():
    formatter = lambda prog: argparse.HelpFormatter(prog, max_help_position=28)
    parser = argparse.ArgumentParser(prog="nanomotif", description="Motif identification (MI355X-native motif_discovery)",
                                     formatter_class=formatter)
    parser.add_argument("--version", action="version", version="%(prog)s {}".format(__version__))
    sub = parser.add_subparsers(help="-- Command descriptions --", dest="command", title="commands",
                                metavar="{motif_discovery, check_installation}")
    p = sub.add_parser("motif_discovery", help="Finds motifs directly on bin level in provided assembly", add_help=False)
    p.add_argument("assembly", type=str, help="path to the assembly file.")
    p.add_argument("pileup", type=str, help="path to the modkit pileup file.")
    gm = p.add_argument_group("contig bin arguments, use one of:")
    g = gm.add_mutually_exclusive_group(required=True)
    g.add_argument("-c", "--contig_bin", type=str, help="TSV file specifying which bin contigs belong.")
    g.add_argument("-f", "--files", nargs="+", help="List of bin FASTA files with contig names as headers.")
    g.add_argument("-d", "--directory", help="Directory containing bin FASTA files with contig names as headers.")
    gm.add_argument("--extension", type=str, default=".fasta",
                    help="File extension of the bin FASTA files if using -d (DIRECTORY) argument. Default is '.fasta'.")
    o = p.add_argument_group("Options")
    o.add_argument("--out", type=str, help="path to the output folder", default="nanomotif")
    o.add_argument("--methylation_threshold_low", type=float, default=0.30,
                   help="A position is considered non-methylated if fraction of methylation is below this threshold. Default: %(default)s")
    o.add_argument("--methylation_threshold_high", type=float, default=0.70,
                   help="A position is considered methylated if fraction of methylated reads is above this threshold. Default: %(default)s")
    o.add_argument("--search_frame_size", type=int, default=40,
                   help="length of the sequnces sampled around confident methylation sites. Default: %(default)s")
    o.add_argument("--minimum_kl_divergence", type=float, default=0.05,
                   help="Minimum KL-divergence for a position to considered for expansion in  motif search. Default: %(default)s")
    o.add_argument("--min_motif_score", type=float, default=1.5,
                   help="Minimum score for a motif to be kept after identification. Default: %(default)s")
    o.add_argument("--threshold_valid_coverage", type=int, default=5,
                   help="Minimum valid base coverage (Nvalid_cov) for a position to be considered. Default: %(default)s")
    o.add_argument("--min_motifs_bin", type=int, default=50,
                   help="Minimum number of motif observations in a bin. Default: %(default)s")
    o.add_argument("--device", type=int, default=None, help="GPU to use (default: LOCAL_RANK or 0).")
    o.add_argument("--shard", choices=["auto", "bins", "contigs"], default="auto",
                   help="Multi-GPU runs: give every GPU whole bins (independent searches, no collective) or shard the "
                        "contigs of every bin over the GPUs (count tables all-reduced per round). Default: bins when "
                        "they balance within 15%%, else contigs.")
    gen = p.add_argument_group("general arguments")
    gen.add_argument("-t", "--threads", type=int, default=1, help="Accepted for compatibility; the GPU engine does not use worker processes.")
    gen.add_argument("-v", "--verbose", action="store_true", help="Increase output verbosity. (set logger to debug level)")
    gen.add_argument("--seed", type=int, default=1, help="Seed for random number generator. Default: %(default)s")
    gen.add_argument("-h", "--help", action="help", help="show this help message and exit")
    sub.add_parser("check_installation", help="Run motif_discovery on a small synthetic data set", add_help=True)
    return parser
