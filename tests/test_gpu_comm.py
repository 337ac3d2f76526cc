"""nm_comm_* / nm_allreduce_counts (include/nmscan.h): the RCCL exchange step of the C ABI.  The GPU test box has ONE
device, so the collective itself runs with a world of one rank (the sum over one rank is the identity, all the
stream / event ordering is exercised); two ranks on one device must come back as a clear error, not a hang."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_orders_allreduce_after_scoring():
    import torch
    from nanomotif_amd import synth
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=4, total_bp=400_000, n_bins=2, mod_types=("a",), seed=7))
    eng = ScanEngine(0)
    with pytest.raises(NmScanError, match="nm_comm_init"):
        eng.allreduce_host(np.arange(4, dtype=np.int64))
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(4)], mg.bin_names)
    for i in range(4):
        p = mg.contig_pileup(i, "a")
        eng.upload_pileup("a", np.full(len(p["position"]), i, np.uint32), p["position"], p["strand"],
                          synth.pct_to_fraction(p["pct_hundredths"]), append=i > 0)
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    with pytest.raises(NmScanError, match="already has a communicator"):
        eng.comm_init(0, 1, uid)
    cands = [(Motif(s, p), "a", b) for b in sorted(set(mg.bin_names)) for s, p in (("GATC", 1), ("A", 0), ("CA.T", 1))]
    want = eng.score(cands)
    batch = eng.make_batch(cands)
    tables = [torch.zeros((len(cands), 2), dtype=torch.int64, device="cuda:0") for _ in range(2)]
    for k in range(6):                                   # the bench's double-buffered step
        i = k & 1
        eng.comm_wait(i)
        eng.score_into_device(batch, tables[i].data_ptr())
        eng.allreduce_counts_device(tables[i].data_ptr(), tables[i].numel(), i)
    eng.comm_sync()
    torch.cuda.synchronize()
    assert np.array_equal(tables[0].cpu().numpy(), want) and np.array_equal(tables[1].cpu().numpy(), want)
    a = np.arange(10, dtype=np.int64).reshape(5, 2)
    assert np.array_equal(eng.allreduce_host(a), a)
    b = np.arange(6, dtype=np.int32)
    r = eng.allreduce_host(b)
    assert r.dtype == np.int32 and np.array_equal(r, b)
    eng.close()


def test_native_search_reduces_through_the_c_abi_with_batches_in_flight():
    """nm_search_run with a reduce callback that lands in nm_allreduce_counts_host (what a contig-sharded run with the
    native communicator does every round), RCCL world of one: a lock-step round has its window batch AND its scoring batch
    open (nm_*_begin) when the window counts are reduced, so the reduction must not take the staging pair the open
    scoring batch's counts sit in (round-3 advisor finding).  Same motif rows as the run without a collective."""
    import torch
    from nanomotif_amd import e2e_synth, postprocess, synth
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.find_motifs_bin import use_native_allreduce
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=40, total_bp=12_000_000, n_bins=8, mod_types=("a", "m"), seed=5))
    eng = ScanEngine(0)
    rows0, t0 = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    eng = ScanEngine(0)
    eng.comm_init(0, 1, eng.comm_unique_id())
    calls = []
    inner = eng.allreduce_host
    eng.allreduce_host = lambda a: (calls.append(a.size), inner(a))[1]
    use_native_allreduce(eng)
    try:
        rows1, t1 = e2e_synth.run(mg, eng, torch.device("cuda:0"), use_dist=True)
    finally:
        use_native_allreduce(None)
        eng.close()
    assert len(rows0) > 8 and t0["search_iterations"] > 20        # (rounds = scoring batches: few, since children are scored speculatively)
    assert postprocess.format_bin_motifs(rows1) == postprocess.format_bin_motifs(rows0)
    # the sharded run scores every child in a round of its own (no speculation where the windows are spread over ranks): more scoring
    # batches than the run above, and a reduction of the window AND of the count table in every one
    assert t1["speculation_hits"] == 0 and t0["speculation_hits"] > 0 and t1["rounds"] > t0["rounds"]
    assert len(calls) > 2 * t1["rounds"] - 10


_TWO_RANKS = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd._lib import NmScanError
dist.init_process_group("gloo")
rank = dist.get_rank()
eng = ScanEngine(0)                                     # both ranks on device 0
uid = [eng.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
try:
    eng.comm_init(rank, 2, uid[0])
    print("RANK", rank, "INIT-OK")
except NmScanError as e:
    print("RANK", rank, "REFUSED", e)
eng.close()
"""


def test_two_ranks_on_one_device_are_refused_not_hung(tmp_path):
    script = tmp_path / "two.py"
    script.write_text(_TWO_RANKS)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.fail("nm_comm_init with two ranks on one device hung instead of failing")
    out = r.stdout + r.stderr
    assert "REFUSED" in out and "INIT-OK" not in out, out[-2000:]
    assert "CommInitRank" in out


_RING = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from nanomotif_amd import synth
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import Motif
from nanomotif_amd.shard import assign_contigs
dist.init_process_group("gloo")                        # rendezvous + the 128-byte id only; the tables travel over RCCL
rank, world = dist.get_rank(), dist.get_world_size()
dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=24, total_bp=3_000_000, n_bins=4, mod_types=("a", "m"), seed=11))
bins = sorted(set(mg.bin_names))

def load(eng, idx):
    eng.upload_assembly([mg.names[i] for i in idx], [mg.contig_ascii(i) for i in idx], [mg.bin_names[i] for i in idx], bin_names=bins)
    for mt in ("a", "m"):
        first = True
        for k, i in enumerate(idx):
            p = mg.contig_pileup(i, mt)
            eng.upload_pileup(mt, np.full(len(p["position"]), k, np.uint32), p["position"], p["strand"],
                              synth.pct_to_fraction(p["pct_hundredths"]), append=not first)
            first = False
        if first:
            eng.upload_pileup(mt, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros(0))

raw = synth.random_candidates(400, seed=2, mod_types=("a", "m"))
cands = [(Motif(s, p), mt, bins[k % len(bins)]) for k, (s, p, mt) in enumerate(raw)]
whole = ScanEngine(dev)
load(whole, list(range(len(mg.names))))
want = whole.score(cands)                              # the unsharded table
whole.close()
mine = sorted(int(i) for i in assign_contigs(mg.lengths, world, bins=mg.bin_names)[rank])
eng = ScanEngine(dev)
load(eng, mine)
uid = [eng.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
eng.comm_init(rank, world, uid[0])
info = eng.comm_info()
assert info["world"] == world and info["rank"] == rank, info
batch = eng.make_batch(cands)
tables = [torch.zeros((len(cands), 2), dtype=torch.int64, device=f"cuda:{dev}") for _ in range(4)]
eng.set_score_lanes(2)
for k in range(13):                                    # the bench's N > 1 headline path: two lanes, four-table ring
    i = k % 4
    eng.comm_wait(i)
    eng.score_into_device(batch, tables[i].data_ptr())
    eng.allreduce_counts_device(tables[i].data_ptr(), tables[i].numel(), i)
eng.comm_sync()
eng.sync()
torch.cuda.synchronize()
ok = all(np.array_equal(t.cpu().numpy(), want) for t in tables)
local = eng.score(cands)
host_sum = eng.allreduce_host(local)                   # the host-table form the lock-step search uses
ok = ok and np.array_equal(host_sum, want) and (world == 1 or not np.array_equal(local, want))
print("RANK", rank, "RING-OK" if ok else "RING-MISMATCH", info, flush=True)
eng.close()
dist.barrier()
"""


def _run_ring(tmp_path, world):
    script = tmp_path / "ring.py"
    script.write_text(_RING)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("RING-OK") == world and "RING-MISMATCH" not in out, out[-3000:]


def test_ring_script_with_one_rank(tmp_path):
    """The script of the multi-GPU test below with a world of ONE (what the one-GPU pool can run): keeps it from rotting."""
    _run_ring(tmp_path, 1)


def test_two_lanes_four_table_ring_over_real_rccl(tmp_path):
    """Two ranks on two GPUs: nm_comm_init + nm_allreduce_counts_async with two scoring lanes over the four-table ring
    (the N > 1 headline path of bench.py); every table equals the unsharded one.  Needs >= 2 GPUs: skipped on the
    one-GPU pool, live on a multi-GPU node."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least 2 GPUs (RCCL refuses two ranks on one device)")
    _run_ring(tmp_path, min(n, 4))
