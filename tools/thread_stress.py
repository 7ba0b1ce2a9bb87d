"""Stress of the device arbiter: T host threads, each with its own context / stream / solver, factorize and solve R times
concurrently (dense SPD-like LDL' matrix of order N on the device).  Prints one line: fall-backs and the wait that expired.
usage: python tools/thread_stress.py N T R"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import madnlp_jl_amd as mj  # noqa: E402

N, T, R = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(N)
Rm = torch.randn(N, 48, dtype=torch.float64, device=dev, generator=g)
A = Rm @ Rm.T
A.diagonal().add_(float(N))
n1 = 2 * N // 3
A[n1:, n1:].neg_()
if "--warm-torch" in sys.argv:   # torch's lazy initialization (first GEMV / randn) before the solver threads start: see INTEGRATION.md section 0
    _b = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    _ = (A @ _b).abs().max().item()
torch.cuda.synchronize()
res = []
first_done = threading.Event()


def worker(i):
    st = torch.cuda.Stream(dev)
    c = mj.HipContext(0, stream=st.cuda_stream)
    fb, sites, bad = 0, [], 0
    with torch.cuda.stream(st):
        M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
        gg = torch.Generator(device=dev).manual_seed(100 + i)
        for rep in range(R):
            if "--first-alone" in sys.argv and rep == 0:   # (diagnostic: thread 0's first factorization has the device to itself)
                if i > 0:
                    first_done.wait()
            M.factorize()
            if "--first-alone" in sys.argv and rep == 0 and i == 0:
                st.synchronize()
                first_done.set()
            f = M.get_stat("pp_fallbacks")
            if f != fb:
                sites.append((rep, int(M.get_stat("timeout_site"))))
                fb = f
            b = torch.randn(N, dtype=torch.float64, device=dev, generator=gg)
            if "--host-rhs" in sys.argv:   # (a HOST vector: uploaded and fetched back by the library, pageable memory)
                xh = M.solve_linear_system(b.cpu().numpy().copy())
                x = torch.from_numpy(xh).to(dev)
            else:
                x = b.clone()
                st.synchronize()
                M.solve_linear_system(x)
                M.check_solve()
            if ((A @ x - b).abs().max() / (A.abs().sum(dim=1).max() * x.abs().max())).item() > 1e-13:
                bad += 1
        M.close()
    c.close()
    res.append((i, fb, sites, bad))


t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"N={N} T={T} R={R} opts={os.environ.get('MNK_OPTIONS', '')!r}: {time.perf_counter() - t0:.1f} s, "
      f"fallbacks {sum(r[1] for r in res)}, wrong solves {sum(r[3] for r in res)}, sites {[r[2] for r in res if r[2]]}")
