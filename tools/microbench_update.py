"""Time the (b)-shaped trailing update under the schedules of the factorization (diagnostics)."""
import ctypes as C
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    A = torch.randn(K, M, dtype=torch.float64, device="cuda")  # column-major M x K, lda = M
    Cm = torch.zeros(M, M, dtype=torch.float64, device="cuda")
    C2 = torch.zeros(M, M, dtype=torch.float64, device="cuda")
    s.synchronize()
    names = ["static/ctx", "queue/ctx", "static/su", "queue/su", "queue/su+sp", "static/su + half-size static/sp",
             "static/ctx chunks of 512 tiles", "static/ctx chunks of 1024 tiles", "static/ctx chunks of 256 tiles",
             "static/ctx, all tiles read the same A/B blocks", "static/su, all tiles read the same A/B blocks",
             "static/su, k-loop without barriers (wrong results)", "static/su, k-loop without global loads / LDS stores",
             "static/su, register staging instead of LDS-DMA", "static/ctx, register staging instead of LDS-DMA",
             "static/su, BK=16 and 2 workgroups per CU", "static/ctx, BK=16 and 2 workgroups per CU"]
    ntiles = (M // 128) * (M // 128 + 1) // 2
    for v in [0, 2, 0, 1, 2, 3, 4, 9, 10, 11, 12, 13, 15]:
        for it in range(2):
            ms = C.c_double()
            L.check(L.lib().mnk_debug_update(ctx.handle, v, M, K, A.data_ptr(), M, Cm.data_ptr(), C2.data_ptr(), M, reps,
                                             C.byref(ms)), "mnk_debug_update")
        t = ms.value / reps
        print(f"{names[v]:36s} {t*1e3:9.1f} us/update  {ntiles*128*128*K*2/t/1e9:7.2f} TFLOP/s (lower tiles)")

    # LDS-DMA staging (default) against register staging: one update from C = 0
    for v, Cx in ((2, Cm), (15, C2)):
        Cx.zero_()
        s.synchronize()
        ms = C.c_double()
        L.check(L.lib().mnk_debug_update(ctx.handle, v, M, K, A.data_ptr(), M, Cx.data_ptr(), Cx.data_ptr(), M, 1,
                                         C.byref(ms)), "mnk_debug_update")
    s.synchronize()
    print("LDS-DMA staging == register staging (lower triangle):", bool(torch.equal(torch.triu(Cm), torch.triu(C2))),
          float((torch.triu(Cm) - torch.triu(C2)).abs().max()))
    # shader clock under load: a fixed dependent FMA chain timed with the constant-rate timer
    for v, name in ((20, "probe alone"), (20, "probe alone"), (21, "probe beside update on su"), (21, "probe beside update on su"),
                    (20, "probe alone")):
        ms = C.c_double()
        L.check(L.lib().mnk_debug_update(ctx.handle, v, M, K, A.data_ptr(), M, Cm.data_ptr(), C2.data_ptr(), M, 4,
                                         C.byref(ms)), "mnk_debug_update")
        print(f"{name:30s} {ms.value*10/1e3:9.1f} us for 192000 dependent fp64 FMAs -> {ms.value*10/192000*1e0:.3f} ns each")
