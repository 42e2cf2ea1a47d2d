#!/usr/bin/env python
"""Run the row-owner weight-gradient tests (tests/test_updat16_rows_gpu.py: configs[2] itself with a gated call, the sixteen random shapes, the
fp32 path) under many values of BSMM_TEST_SEED on one GPU lease: the per-block criterion of tests/_parity.py must hold for
EVERY seed, not for the one the suite ships with.  Usage (on the GPU box): python scripts/gpu_seed_sweep.py [seeds=20] > gpurun_out/seed_sweep.txt"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["tests/test_updat16_rows_gpu.py::test_row_owner_updat_at_configs2",
         "tests/test_updat16_rows_gpu.py::test_row_owner_updat_random_shapes",
         "tests/test_updat16_rows_gpu.py::test_fp32_updat_through_the_row_owner_kernel"]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    bad = 0
    for seed in range(seeds):
        env = dict(os.environ, BSMM_TEST_SEED=str(seed))
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + TESTS, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-1]
        print("seed %2d  rc %d  %5.1f s  %s" % (seed, r.returncode, time.time() - t0, tail), flush=True)
        if r.returncode:
            bad += 1
            print(r.stdout[-3000:], flush=True)
    print("seeds %d  failed %d" % (seeds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
