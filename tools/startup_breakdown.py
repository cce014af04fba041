#!/usr/bin/env python3
"""Where the second of process start-up goes on a GPU box: interpreter + numpy, `import torch`, the first CUDA (HIP)
call, loading libscoary_hip.so and creating the handle, the warm-up pass (code-object load of the library).
    python tools/startup_breakdown.py        (run it twice: the first run of a fresh box pages the image in)"""
import os
import sys
import time

t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
t1 = time.perf_counter()
import torch  # noqa: E402
t2 = time.perf_counter()
torch.cuda.init()
torch.zeros(1, device="cuda:0")
torch.cuda.synchronize()
t3 = time.perf_counter()
from scoary_amd import methods as m  # noqa: E402
from scoary_amd.engine import AssociationEngine  # noqa: E402
t4 = time.perf_counter()
eng = AssociationEngine()
t5 = time.perf_counter()
m._warm_up(eng)
t6 = time.perf_counter()
m._warm_up(eng)
t7 = time.perf_counter()
print("numpy %.3f | import torch %.3f | first HIP call (context, allocator) %.3f | import scoary_amd %.3f | "
      "library + handle %.3f | warm-up pass %.3f (a second one: %.3f) | total %.3f s"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t7 - t0))
