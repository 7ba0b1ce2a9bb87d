"""Hardware counters of the task-DAG schedule AS IT RUNS (two persistent kernels side by side), through rocprofiler-sdk's
device counting service (tools/devcount/mnk_devcount.cpp; `rocprofv3 --pmc` serializes dispatches and cannot run it).
One counter set per process; the region counted = K whole `factorize!` calls of the bench's C3 system (sparse source:
scatter, pivot chain + bulk kernel, inverses, inertia words), device-wide.

usage (on a GPU box):  ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libmnk_devcount.so python tools/devcount_dag.py <set> [K] [case]
       set = mfma | fetch | write | <comma separated counter names>
prints one JSON line: per-call sums of every counter (summed over all dimension instances) and the event-timed ms per call."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import opf_shaped  # noqa: E402

SETS = {
    "mfma": "SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE",
    "fetch": "TCC_EA0_RDREQ,TCC_EA0_RDREQ_32B,TCC_BUBBLE",
    "write": "TCC_EA0_WRREQ,TCC_EA0_WRREQ_64B",
}
which = sys.argv[1] if len(sys.argv) > 1 else "mfma"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
case = sys.argv[3] if len(sys.argv) > 3 else "case1354pegase"
names = SETS.get(which, which)
lib_path = os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":")[0]
if not lib_path:
    raise SystemExit("set ROCP_TOOL_LIBRARIES=<path>/libmnk_devcount.so before starting python")
dc = C.CDLL(lib_path)
dc.mnk_devcount_error.restype = C.c_char_p

dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
ctx = mj.HipContext(0, stream=st.cuda_stream)
P = opf_shaped(case, du=1e-8)
k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
dj, dh = torch.from_numpy(P.jac).to(dev), torch.from_numpy(P.hess).to(dev)
dp, dd = torch.from_numpy(P.pr_diag).to(dev), torch.from_numpy(P.du_diag).to(dev)
k.compress_jacobian(dj); k.compress_hessian(dh); k.build_kkt(dp, dd)
ls = k.linear_solver
for _ in range(3):
    ls.factorize_async()
assert ls.inertia() == (P.n, 0, 0)
torch.cuda.synchronize()
if not dc.mnk_devcount_available():
    raise SystemExit("device counting service not initialized: " + (dc.mnk_devcount_error() or b"").decode())
rc = dc.mnk_devcount_start(names.encode())
if rc:
    raise SystemExit(f"mnk_devcount_start({names}) failed: " + (dc.mnk_devcount_error() or b"").decode())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    ls.factorize_async()
e1.record()
torch.cuda.synchronize()
cap = 32
nbuf = C.create_string_buffer(64 * cap)
vals = (C.c_double * cap)()
n = dc.mnk_devcount_sample(nbuf, vals, cap)
dc.mnk_devcount_stop()
if n < 0:
    raise SystemExit("sample failed: " + (dc.mnk_devcount_error() or b"").decode())
out = {"counters_per_call": {nbuf.raw[64 * i:64 * i + 64].split(b"\0")[0].decode(): vals[i] / K for i in range(n)},
       "calls": K, "ms_per_call": e0.elapsed_time(e1) / K, "N": P.n, "case": case,
       "schedule_panel_algo": ls.get_stat("panel_algo"), "pp_fallbacks": ls.get_stat("pp_fallbacks"),
       "inertia_ok": ls.inertia() == (P.n, 0, 0),
       "method": "rocprofiler-sdk device counting service (agent-wide sampling, no dispatch serialization), sums over all "
                 "dimension instances between start and sample"}
print(json.dumps(out))
