"""Slot utilisation of the persistent bulk kernel (option dag_trace): per workgroup {lifetime, ticks waited, tasks, ticks in
the finalization}.  usage: python tools/dag_util.py [N] [LDL|CHOLESKY]"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11192
alg = sys.argv[2] if len(sys.argv) > 2 else "LDL"
chunk = int(os.environ.get("DAG_CHUNK", "64"))   # (the library's defaults; other values are set on the solver below)
band = int(os.environ.get("DAG_BAND", "16"))
taper0 = int(os.environ.get("DAG_TAPER0", "2"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(N)
    R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5, single_rows=0))
    ls.set_option("dag_min_rows", 0)
    ls.set_option("dag_chunk", chunk)
    ls.set_option("dag_band", band)
    ls.set_option("dag_taper0", taper0)
    ls.factorize(); s.synchronize()
    ls.set_option("dag_fill", 0)   # (the trace is indexed by task: the list without the zero-fill tasks is the one tools/dag_tasks.py mirrors)
    ls.set_option("dag_trace", 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); ls.factorize(); e1.record(s); s.synchronize()
    print(f"factorize (traced) {e0.elapsed_time(e1):.3f} ms")
v = C.c_double(0.0)
nt = int(os.environ.get("DAG_NTASKS", "0"))
tr = np.zeros(4_000_000, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, tr.ctypes.data, tr.size), "trace")
# find ntasks: wgstat sits behind tasks*8 + 16384; scan for the layout by the option-independent marker: tasks have tr[0] > 0
nz = np.nonzero(tr)[0]
last = nz.max()
# wgstat occupies the last 1024*8 words of the buffer the library allocated: ntasks*8 + 16384 + 8192 = buffer size is unknown here;
# recover ntasks from the first zero run of 16384+ words after the task region
tasks8 = tr[: (last // 8 + 1) * 8]
# simple approach: search the offset `off` (multiple of 8) such that tr[off:off+8*720] looks like wgstat (w[3] small counts)
best = None
for off in range(((last - 8 * 1024) // 8) * 8, last, 8):
    w = tr[off: off + 8]
    if 0 < w[3] < 100000 and w[1] > w[0] > 0:
        best = off
        break
assert best is not None
# walk back to the first wgstat entry
off = best
while off - 8 >= 0 and 0 < tr[off - 8 + 3] < 100000 and tr[off - 8 + 1] > tr[off - 8] > 0:
    off -= 8
W = []
o = off
while o + 8 <= tr.size and tr[o + 1] > tr[o] > 0:
    W.append(tr[o:o + 8].astype(np.float64)); o += 8
W = np.array(W)
life = (W[:, 1] - W[:, 0]) / 100.0
wait = W[:, 2] / 100.0
fin = W[:, 4] / 100.0
print(f"{len(W)} workgroups: lifetime mean {life.mean():.0f} us (min {life.min():.0f}, max {life.max():.0f}); waited mean {wait.mean():.0f} us "
      f"({100*wait.sum()/life.sum():.1f} % of the slot-time); finalization {fin.mean():.0f} us ({100*fin.sum()/life.sum():.1f} %); tasks/WG {W[:,3].mean():.1f}")
span = (W[:, 1].max() - W[:, 0].min()) / 100.0
print(f"kernel span {span:.0f} us; slot-time not waiting/finalizing: {100*(life.sum()-wait.sum()-fin.sum())/(len(W)*span):.1f} % of slots x span")

# ---- waits by task class (the task list rebuilt as dag_build_tasks does)
Np = (N + 127) // 128 * 128
ntile = Np // 128
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dag_tasks import dag_tasks  # noqa: E402
ts = dag_tasks(ntile, chunk, band // 2, taper0=taper0)
nt = len(ts)
bulk = tr[: nt * 8].reshape(nt, 8).astype(np.float64)
t0 = bulk[:, 0][bulk[:, 0] > 0].min()
cls = np.array([t[1] for t in ts])
ready = np.array([t[0] for t in ts])
# (tile-closing tasks reuse words 6 / 7 for the finalization's stage stamps: their gate wait is grab -> front ready)
gatew = np.where(cls == 1, np.where(bulk[:, 2] > 0, bulk[:, 2] - bulk[:, 0], 0.0), bulk[:, 7]) / 100.0
diagw = np.where(cls == 1, (bulk[:, 4] - bulk[:, 3]) / 100.0, 0.0)
dur = (bulk[:, 5] - bulk[:, 0]) / 100.0
for c, name in ((0, "band-acc"), (1, "tile-closing"), (2, "body chunk")):
    m = cls == c
    print(f"{name:13s}: {m.sum():6d} tasks, mean duration {dur[m].mean():7.1f} us, gate wait {gatew[m].mean():6.1f} us, diag wait {diagw[m].mean():6.1f} us; "
          f"total slot-time {dur[m].sum()/1e3:8.1f} ms, waits {100*(gatew[m].sum()+diagw[m].sum())/dur[m].sum():.1f} %")
# over time: per 500 us bin, slot-time spent waiting by the tasks that END in the bin
grab = (bulk[:, 0] - t0) / 100.0
end = (bulk[:, 5] - t0) / 100.0
edges = np.arange(0, end.max() + 500, 500)
print("bin start us | tasks ended | mean dur | gate wait % | diag wait % | chain position (ready index) of the tasks")
for a, b in zip(edges[:-1], edges[1:]):
    m = (end >= a) & (end < b)
    if m.sum() == 0:
        continue
    print(f"{a:8.0f} | {m.sum():5d} | {dur[m].mean():7.1f} | {100*gatew[m].sum()/dur[m].sum():5.1f} | {100*diagw[m].sum()/dur[m].sum():5.1f} | {ready[m].min()}..{ready[m].max()}")

# ---- anatomy of the tile-closing tasks late in the factorization (the chain-bound part)
m = (cls == 1) & (ready >= ntile // 2) & (bulk[:, 2] > 0)
mclose = m.copy()
if m.sum():
    b = bulk[m]
    print(f"tile-closing tasks with chain position >= {ntile//2}: {m.sum()}; mean us: grab->front ready {(b[:,2]-b[:,0]).mean()/100:.1f}, "
          f"front ready->K-loop done {(b[:,1]-b[:,2]).mean()/100:.1f}, K-loop done->tile applied (chunk order + epilogue) {(b[:,3]-b[:,1]).mean()/100:.1f}, diagonal wait {(b[:,4]-b[:,3]).mean()/100:.1f}, "
          f"finalize + publish {(b[:,5]-b[:,4]).mean()/100:.1f}")

# ---- anatomy of the body chunks while the machine is saturated (1 .. 6.5 ms): time per k-step and the fixed cost per task
klen = np.array([t[7] - t[6] for t in ts])
sat = (cls == 2) & (grab > 1000) & (end < 6500) & (bulk[:, 1] > 0)
print("body chunks ending in 1..6.5 ms, by length: tasks | grab->K-loop done | ->tile applied (order wait + epilogue) | ->published | us per k-step of the K-loop")
for n in sorted(set(klen[sat])):
    m = sat & (klen == n)
    b = bulk[m]
    kl = (b[:, 1] - b[:, 0]).mean() / 100
    print(f"  {n:2d} k-steps: {m.sum():5d} | {kl:7.1f} | {(b[:,3]-b[:,1]).mean()/100:6.1f} | {(b[:,5]-b[:,3]).mean()/100:5.1f} | {kl/n:5.1f}")
b = bulk[sat]
print(f"  all: K-loop {((b[:,1]-b[:,0]).sum())/1e5:.0f} ms, epilogue {((b[:,3]-b[:,1]).sum())/1e5:.0f} ms, publish {((b[:,5]-b[:,3]).sum())/1e5:.0f} ms of slot-time; "
      f"k-steps {klen[sat].sum()} -> {(b[:,5]-b[:,0]).sum()/100/klen[sat].sum():.1f} us of slot-time per k-step (41.2 = MFMA-bound at three workgroups per CU)")

# ---- stages of the finalization of those tile-closing tasks (six 16-bit stage times packed into words 6 and 7)
m = mclose
if m.sum():
    raw = tr[: nt * 8].reshape(nt, 8)
    w6, w7 = raw[m][:, 6], raw[m][:, 7]
    st = [((w6 >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(np.float64) / 100 for i in range(4)] + \
         [((w7 >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(np.float64) / 100 for i in range(2)]
    names = ["diagonal block a staged", "substitution a + stores", "L(b, a) staged", "update", "diagonal block b staged", "substitution b + stores"]
    prev = 0.0
    print("finalization stages (us, mean): " + ", ".join(f"{n} {s_.mean() - prev_:.1f}" for n, s_, prev_ in zip(names, st, [0.0] + [x.mean() for x in st[:-1]])),
          f"; end of task {((bulk[m][:,5]-bulk[m][:,4])/100).mean() - st[5].mean():.1f} after the last stage")
