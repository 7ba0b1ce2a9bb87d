"""One interior-point run on a two-stage QP through `SchurComplementKKTSystem`, with the scenario blocks assembled on the device
(round 6: mnk_schur_assemble) and with the host assembly of rounds 4-5 (numpy slicing + upload of every dense block): wall time of
the run, of its build_kkt! calls and of the assembly part of them.
usage: python tools/bench_schur_kkt.py [ns nv nd nc nc_eq density_v density_d]   -> two JSON lines
(default rows are sparse: the reference's pair lists -- and the library's source lists -- grow with the square of a row's length;
with the generator's dense design part, 410 entries per row, they would hold 1.4e9 pairs at ns = 128)"""
import json
import os
import sys
import time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import random_twostage_qp  # noqa: E402
from madnlp_jl_amd.schur_kkt import SchurComplementKKTSystem  # noqa: E402

ns, nv, nd, nc, nc_eq = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (128, 384, 256, 128, 64)
dv, dd = (float(a) for a in sys.argv[6:8]) if len(sys.argv) > 7 else (0.03, 0.05)
nlp = random_twostage_qp(ns=ns, nv=nv, nd=nd, nc=nc, nc_eq=nc_eq, seed=3, density_v=dv, density_d=dd)
ctx = mj.HipContext(0)
for device_assembly in (True, False):
    t_build, t_asm = [], []

    def make(info):
        k = SchurComplementKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_eq"],
                                     info["ind_lb"], info["ind_ub"], ctx=ctx, **nlp.schur_opts())
        k.device_assembly = device_assembly
        inner_build, inner_asm = k.build_kkt, (k.stage.assemble if device_assembly else k.assemble_blocks)

        def timed_build():
            ctx.synchronize(); t0 = time.perf_counter(); inner_build(); ctx.synchronize(); t_build.append(time.perf_counter() - t0)

        def timed_asm(*a):
            ctx.synchronize(); t0 = time.perf_counter(); r = inner_asm(*a); ctx.synchronize(); t_asm.append(time.perf_counter() - t0)
            return r
        k.build_kkt = timed_build
        if device_assembly:
            k.stage.assemble = timed_asm
        else:
            k.assemble_blocks = timed_asm
        return k
    t0 = time.perf_counter()
    s = MadNLPSolver(nlp, make, IPMOptions(), sparse=True)
    setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.solve()
    wall = time.perf_counter() - t0
    print(json.dumps({"problem": f"random_twostage_qp(ns={ns}, nv={nv}, nd={nd}, nc={nc}, nc_eq={nc_eq}, density_v={dv}, density_d={dd}): n={nlp.n}, m={nlp.m}, "
                                 f"blk={s.kkt.blk}, nnz(J)={len(nlp.jac_I)}",
                      "assembly": "device (mnk_schur_assemble)" if device_assembly else "host (numpy slicing + upload of the dense blocks; rounds 4-5)",
                      "status": s.status, "iterations": s.cnt.k, "objective": s.obj_val, "setup_s": setup, "solve_wall_s": wall,
                      "build_kkt_calls": len(t_build), "build_kkt_ms_mean": 1e3 * float(np.mean(t_build)),
                      "assembly_ms_mean": 1e3 * float(np.mean(t_asm)), "assembly_ms_min": 1e3 * float(np.min(t_asm))}), flush=True)
    s.kkt.close()
