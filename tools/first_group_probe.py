"""Which activity of a SECOND host thread makes the first task-DAG factorization of a process run into its bounded wait?
Thread 0: one solver, R factorizations.  Thread 1: the activity named on the command line, started together with thread 0.
usage: python tools/first_group_probe.py N activity   (malloc | copy | h2d | h2d_once | h2d_pinned | d2h | d2h_pinned | solver | sync | none)"""
import os
import sys
import threading
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import madnlp_jl_amd as mj  # noqa: E402

N, act = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(N)
Rm = torch.randn(N, 48, dtype=torch.float64, device=dev, generator=g)
A = Rm @ Rm.T
A.diagonal().add_(float(N))
torch.cuda.synchronize()
go = threading.Event()
out = {}


def t0():
    st = torch.cuda.Stream(dev)
    c = mj.HipContext(0, stream=st.cuda_stream)
    with torch.cuda.stream(st):
        M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
        go.set()
        sites = []
        for rep in range(4):
            M.factorize()
            sites.append((M.get_stat("pp_fallbacks"), int(M.get_stat("timeout_site"))))
        out["sites"] = sites
        M.close()
    c.close()


def t1():
    go.wait()
    st = torch.cuda.Stream(dev)
    t_end = time.perf_counter() + 0.15
    with torch.cuda.stream(st):
        if act == "solver":
            c = mj.HipContext(0, stream=st.cuda_stream)
            M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
            M.transfer_only() if hasattr(M, "transfer_only") else None
            time.sleep(0.2)
            M.close(); c.close()
            return
        keep = []
        while time.perf_counter() < t_end:
            if act == "malloc":
                keep.append(torch.empty(1 << 27, dtype=torch.float64, device=dev))   # 1 GB each
            elif act == "copy":
                B = torch.empty_like(A); B.copy_(A)
            elif act == "h2d":
                keep.append(torch.from_numpy(np.ones(1 << 16)).to(dev))
            elif act == "h2d_pinned":
                if not keep:
                    keep.append(torch.ones(1 << 16, dtype=torch.float64).pin_memory())
                    keep.append(torch.empty(1 << 16, dtype=torch.float64, device=dev))
                keep[1].copy_(keep[0], non_blocking=True); st.synchronize()
            elif act == "d2h":
                if not keep:
                    keep.append(torch.ones(1 << 16, dtype=torch.float64, device=dev))
                _ = keep[0].cpu()
            elif act == "d2h_pinned":
                if not keep:
                    keep.append(torch.ones(1 << 16, dtype=torch.float64, device=dev))
                    keep.append(torch.empty(1 << 16, dtype=torch.float64).pin_memory())
                keep[1].copy_(keep[0], non_blocking=True); st.synchronize()
            elif act == "h2d_once":
                if not keep:
                    keep.append(torch.from_numpy(np.ones(1 << 16)).to(dev))
                time.sleep(0.005)
            elif act in ("d2h_raw", "h2d_raw"):   # hipMemcpy itself on a plain numpy buffer (no staging by torch)
                import ctypes
                if not keep:
                    keep.append(ctypes.CDLL("libamdhip64.so"))
                    keep.append(torch.ones(1 << 16, dtype=torch.float64, device=dev))
                    keep.append(np.ones(1 << 16))
                hip, dten, harr = keep
                if act == "d2h_raw":
                    hip.hipMemcpy(ctypes.c_void_p(harr.ctypes.data), ctypes.c_void_p(dten.data_ptr()), ctypes.c_size_t(harr.nbytes), 2)
                else:
                    hip.hipMemcpy(ctypes.c_void_p(dten.data_ptr()), ctypes.c_void_p(harr.ctypes.data), ctypes.c_size_t(harr.nbytes), 1)
            elif act == "sync":
                st.synchronize()
            else:
                time.sleep(0.01)
        st.synchronize()


th = [threading.Thread(target=t0), threading.Thread(target=t1)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"N={N} second thread: {act}: (fallbacks, timeout site) after each factorization {out['sites']}")
