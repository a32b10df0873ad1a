"""Runs <tree>/tests/tools/prof_fit.py (the copy of a given commit, built under tools/_bisect/<hash>) and reports which
libgpimhip.so the process really mapped.    python tools/r3_bisect_run.py <tree> N T M kernel"""
import os, runpy, sys
tree = os.path.abspath(sys.argv[1])
script = os.path.join(tree, "tests", "tools", "prof_fit.py")
sys.argv = [script] + sys.argv[2:]
try:
    runpy.run_path(script, run_name="__main__")
finally:
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libgpimhip" in l})
    print("loaded:", libs, flush=True)
