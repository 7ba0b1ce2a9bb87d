"""Does a long-lived GPU process beside it make tools/thread_stress.py lose a factorization to its bounded waits?  The parent
creates K idle contexts (each with its streams) on the device, then runs the stress as a child process.
usage: python tools/parent_child_probe.py [K] [child args ...]   (environment is inherited: e.g. MNK_PANEL_CUS=0)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import madnlp_jl_amd as mj  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
child = sys.argv[2:] or ["6400", "3", "3"]
dev = torch.device("cuda", 0)
keep = []
for i in range(K):
    st = torch.cuda.Stream(dev)
    c = mj.HipContext(0, stream=st.cuda_stream)
    A = torch.eye(2048, dtype=torch.float64, device=dev) * 3.0
    M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    M.factorize()
    keep.append((st, c, M))
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(3):
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "thread_stress.py")] + child, capture_output=True, text=True, timeout=300)
    print(f"parent with {K} idle contexts (MNK_PANEL_CUS={os.environ.get('MNK_PANEL_CUS', '')!r}) | child:", (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
