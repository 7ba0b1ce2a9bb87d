"""Where the 16 us of the one-wave potrf64 go: dependent launches of the kernel with one piece removed at a time
(mnk_debug_potrf64).  usage: python tools/potrf_ablation.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

ctx = mj.HipContext(0)
names = {0: "full kernel", 1: "no 16x16 inverses", 2: "no dvec/dinv/info stores", 3: "only the critical update MFMA",
         4: "no per-lane selection of the 4x4 factor", 5: "load + store only (launch + memory floor)", 6: "no Dout stores"}
for v in (0, 1, 2, 3, 4, 6, 5, 0):
    ms = C.c_double(0.0)
    L.check(mj.lib().mnk_debug_potrf64(ctx.handle, v, 2000, C.byref(ms)), "mnk_debug_potrf64")
    print(f"variant {v} ({names[v]}): {1e3 * ms.value:.2f} us per launch")
