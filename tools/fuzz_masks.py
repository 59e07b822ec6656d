"""HIP engine vs oracle on random TAS cycles with node-feasibility masks, beyond the seeds the test suite pins:
   python tools/fuzz_masks.py <first seed> <count>      (first-pass cycles; every 4th seed also as a second-pass population)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kqo
import tests.test_tas_cycle_engine as T

first, count = int(sys.argv[1]), int(sys.argv[2])
kqo.build()
bad = n = nsp = 0
for seed in range(first, first + count):
    try:
        T._random_masked(kqo, T._hip, seed); n += 1
        if seed % 4 == 0:
            T._second_pass_masked(kqo, T._hip, seed); nsp += 1
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
print(f"masked cycles on the HIP engine vs the oracle: seeds {first}..{first + count - 1}: {n} first-pass + {nsp} second-pass populations, {bad} mismatches")
