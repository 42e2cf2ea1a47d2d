"""Experiment builds of libbsmm_hip.so with ablation switches, in parallel: python scripts/build_variants.py NAME=-DFLAG[,-DFLAG] ..."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blocksparse_amd import build
out_dir = os.path.join(os.path.dirname(build.OUT), "variants")
os.makedirs(out_dir, exist_ok=True)
jobs = []
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    jobs.append((os.path.join(out_dir, "libbsmm_%s.so" % name), [f for f in flags.split(",") if f]))
with ThreadPoolExecutor(max_workers=6) as ex:
    for o in ex.map(lambda j: build.build_variant(*j), jobs):
        print("built", o)
