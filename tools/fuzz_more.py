import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as t
bad = 0
for seed in range(10, 70):
    try:
        t.test_fuzz_counts_match_oracle(seed)
    except AssertionError as e:
        bad += 1; print("FAIL", seed, str(e)[:300])
print("fuzz seeds 10..69 done, failures:", bad)
