"""tests/fuzz_seed_counter.txt += 1: the next run of tests/test_gpu_fuzz.py draws other streams / sessions.  Commit the file with the run's record."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fuzz_seed_counter.txt")
n = int(open(p).read().split()[0]) + 1
open(p, "w").write(f"{n}\n")
print(n)
