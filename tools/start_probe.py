"""Where a cold `python -m nanomotif_amd` process spends its time before the first useful byte: interpreter, imports, library
load, HIP runtime start, context creation, and the teardown at exit."""
import os, subprocess, sys, time
code = r'''
import time, sys
t0 = time.perf_counter()
import numpy
t1 = time.perf_counter()
from nanomotif_amd import main as m
t2 = time.perf_counter()
from nanomotif_amd import _lib
_lib._cli_process = lambda: True          # like `python -m nanomotif_amd`: no torch in the process
lib = _lib.load()
t3 = time.perf_counter()
from nanomotif_amd.engine import ScanEngine
eng = ScanEngine(0)
t4 = time.perf_counter()
eng.close()
t5 = time.perf_counter()
print("numpy %.3f  package %.3f  dlopen %.3f  ctx_create %.3f  close %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4), flush=True)
import os
print("end_of_script %.6f" % time.time(), flush=True)
'''
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=os.getcwd()))
    t1 = time.time()
    end = float(r.stdout.split("end_of_script")[1]) if "end_of_script" in r.stdout else t1
    print(r.stdout.splitlines()[0] if r.stdout else r.stderr[-500:], " | process wall %.3f, exit after script end %.3f" % (t1 - t0, t1 - end), flush=True)
t0 = time.time(); subprocess.run([sys.executable, "-c", "pass"]); print("bare interpreter %.3f" % (time.time() - t0))
