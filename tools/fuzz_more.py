"""Extended seeded fuzz on the GPU box (not part of the suite): heavy batches and light batches (siblings, shared parents,
pruning rounds, sets, wide, general, one slot) over many seeds, also with the segment split forced."""
import os
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as t
bad = 0
for seed in range(10, 50):
    try:
        t.test_fuzz_counts_match_oracle(seed)
    except AssertionError as e:
        bad += 1; print("FAIL heavy", seed, str(e)[:300])
styles = ["literal", "sets", "wide", "general"]
for seed in range(10, 90):
    style = styles[seed % 4]
    try:
        t.test_fuzz_light_batches_common_factoring_and_siblings(seed if seed % 7 else 5, style)   # seed 5: single slot
    except AssertionError as e:
        bad += 1; print("FAIL light", seed, style, str(e)[:300])
print("fuzz done, failures:", bad, "NM_SPLIT", os.environ.get("NM_SPLIT"), "NM_FINE", os.environ.get("NM_FINE"))
