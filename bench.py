#!/usr/bin/env python
"""bench.py -- MadNLP KKT hot path on MI355X: IP iterations/s + ms per factorize!/solve!.

One "step" = the per-iteration hot path of one interior-point iteration on one batch of
synthetic input, with every input already resident in HBM when the timed region starts:

    compress_jacobian! + compress_hessian! + build_kkt! + factorize! (n_f = 1)
    + n_s = 2 x solve_linear_system!

on the OPF-shaped sparse-condensed KKT system of BASELINE.json configs[2]
(case1354pegase-shaped, N = 11192; the metric is quoted on "condensed KKT n~1e4"), with the
reference's default algorithm (BUNCHKAUFMAN, served by the static-pivot LDL^T).  Timing
follows the reference's `timing_linear_solver` protocol (src/utils.jl:185-197): warm-up,
then the mean over the timed repetitions.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Multi-GPU = independent NLP instances, one per GPU/process (weak scaling); the only
collectives are a barrier and the aggregation of the timings (RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "IP iterations/sec + ms per factorize!/solve!, condensed KKT n≈1e4"
PEAK_FP64_TFLOPS = 78.6   # MI355X fp64 matrix/vector peak (datasheet)
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--case", default="case1354pegase")
    ap.add_argument("--algorithm", default="BUNCHKAUFMAN", choices=["BUNCHKAUFMAN", "CHOLESKY", "LDL"])
    ap.add_argument("--nsolve", type=int, default=2)
    ap.add_argument("--outer-block", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1,
                    help="independent NLP instances per GPU, each on its own context/stream (BASELINE config 5 "
                         "uses 16 per GPU); a step advances every instance by one iteration")
    ap.add_argument("--concurrency", type=int, default=4,
                    help="contexts (stream sets) the batch is spread over; more than ~4 oversubscribes the "
                         "hardware queues (measured: 16 contexts run 2.6x slower than 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="test-only: exercise the multi-process harness on CPU (gloo) without any kernel")
    return ap.parse_args()


# ---------------------------------------------------------------------------- distributed harness
def dist_setup(ngpus, backend):
    """One process per GPU; returns (rank, world, local_rank, dist-or-None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and ngpus == 1:
        return 0, 1, 0, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != ngpus:
        raise SystemExit(f"--gpus {ngpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {ngpus}")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, dist


def timed_region(step_fn, steps, warmup, sync_fn, dist, device_tensor_fn):
    """Contract: W untimed steps, barrier + device sync, K timed steps, device sync + barrier,
    MAX over ranks."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = device_tensor_fn([elapsed])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
    return elapsed


def gather_stats(vec, dist, world, device_tensor_fn):
    """The single timing all-gather: every rank contributes its per-phase averages."""
    if dist is None:
        return [list(vec)]
    t = device_tensor_fn(list(vec))
    out = [t.clone() for _ in range(world)]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


# ---------------------------------------------------------------------------- CPU baseline leg
def cpu_baseline(P, algorithm, nsolve):
    """The oracle (CPU restatement: numpy assembly + OpenBLAS LAPACK dsytrf/dsytrs through
    scipy, the routine family MadNLP's LapackCPUSolver calls) timed on this host's cores for
    ONE iteration of the same hot path (bounded sample)."""
    from oracle import kernels as ok
    from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, LapackCPUSolver
    from oracle.sparse_condensed import SparseCondensedKKTSystem as OracleSC
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    alg = CHOLESKY if algorithm == "CHOLESKY" else BUNCHKAUFMAN
    k = OracleSC(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                 lambda A: LapackCPUSolver(A, alg))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    ok.set_aug_diagonal(k)
    b = np.random.default_rng(0).standard_normal(P.n)
    t = {}
    t0 = time.perf_counter(); k.compress_jacobian(); k.compress_hessian(); t["compress"] = time.perf_counter() - t0
    t0 = time.perf_counter(); k.build_kkt(); t["build"] = time.perf_counter() - t0
    t0 = time.perf_counter(); k.linear_solver.factorize(); t["factorize"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(nsolve):
        k.linear_solver.solve_linear_system(b.copy())
    t["solve"] = (time.perf_counter() - t0) / max(nsolve, 1)
    total = t["compress"] + t["build"] + t["factorize"] + nsolve * t["solve"]
    return {
        "value": 1.0 / total, "unit": "it/s", "cores": int(cores), "kind": "port",
        "sample": f"1 iteration of the same hot path ({P.name}-shaped, N={P.n}): numpy assembly + "
                  f"scipy/OpenBLAS {'dpotrf/dpotrs' if alg == CHOLESKY else 'dsytrf/dsytrs'}, {cores} BLAS threads",
        "ms_per_factorize": 1e3 * t["factorize"], "ms_per_solve": 1e3 * t["solve"],
        "ms_build": 1e3 * (t["build"] + t["compress"]),
    }


def pmc_traffic(N, args):
    """HBM bytes per factorize! call from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, profiles/r01_pmc_traffic_factorize_N11192.md); counters cannot be
    collected inside the timed run, so this is the value measured for the same shape, or null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if N == 11192 and os.path.exists(path):
        try:
            return json.load(open(path))["traffic_bytes"]
        except Exception:
            return None
    return None


# ---------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.cpu_dry_run:
        return dry_run(args)

    import torch
    rank, world, local, dist = dist_setup(args.gpus, "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the KKT hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.problems import OPF_CASES, opf_shaped

    dev = torch.device("cuda", local)
    # a non-default stream: its handle is non-NULL, the library enqueues on it and torch's
    # events (HIP events on that same stream) bracket the kernels
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    ctx = mj.HipContext(local, stream=tstream.cuda_stream)
    # independent instances: seed = base + rank * batch + b; instance 0 uses the bench stream
    base_seed = OPF_CASES[args.case][0] + rank * args.batch
    insts = []
    ctxs = [(ctx, tstream)]
    for bidx in range(args.batch):
        slot = bidx % max(1, args.concurrency)
        if slot >= len(ctxs):
            st = torch.cuda.Stream(dev)
            ctxs.append((mj.HipContext(local, stream=st.cuda_stream), st))
        ictx, istream = ctxs[slot]
        Pb = opf_shaped(args.case, seed=base_seed + bidx, du=1e-8)
        kb = mj.SparseCondensedKKTSystem(
            Pb.n, Pb.m, Pb.jac_I, Pb.jac_J, Pb.hess_I, Pb.hess_J, Pb.ind_ineq, Pb.ind_lb, Pb.ind_ub, ctx=ictx,
            opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=args.algorithm, outer_block=args.outer_block))
        dev_in = dict(jac=torch.from_numpy(Pb.jac).to(dev), hess=torch.from_numpy(Pb.hess).to(dev),
                      pr=torch.from_numpy(Pb.pr_diag).to(dev), du=torch.from_numpy(Pb.du_diag).to(dev),
                      rhs=torch.from_numpy(np.random.default_rng(base_seed + bidx).standard_normal(Pb.n)).to(dev))
        dev_in["x"] = torch.empty_like(dev_in["rhs"])
        insts.append((Pb, kb, istream, dev_in))
    P, kkt = insts[0][0], insts[0][1]
    ls = kkt.linear_solver
    d_jac, d_hess, d_pr, d_du, d_rhs, d_x = (insts[0][3][k] for k in ("jac", "hess", "pr", "du", "rhs", "x"))

    def step_one(kb, st, din):
        with torch.cuda.stream(st):
            kb.compress_jacobian(din["jac"])
            kb.compress_hessian(din["hess"])
            kb.build_kkt(din["pr"], din["du"])
            kb.linear_solver.factorize_async()
            for _ in range(args.nsolve):
                din["x"].copy_(din["rhs"])
                kb.linear_solver.solve_linear_system(din["x"])

    def step():
        for (_, kb, st, din) in insts:
            step_one(kb, st, din)

    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    dt = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)  # noqa: E731

    # check the factorization once (inertia must be correct: otherwise the numbers are void)
    step()
    sync()
    for (_, kb, _, _) in insts:
        inertia = kb.linear_solver.inertia()
        if not kb.is_inertia_correct(*inertia):
            raise SystemExit(f"unexpected inertia {inertia} on the benchmark system")

    elapsed = timed_region(step, args.steps, args.warmup, sync, dist, dt)

    # per-phase breakdown, the reference's `timing_linear_solver` protocol (src/utils.jl:185-197):
    # each phase alone between device synchronizations, HIP events on the launch stream.
    def phase_ms(fn, reps):
        out = []
        for _ in range(reps):
            sync()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            sync()
            out.append(a.elapsed_time(b))
        return float(np.mean(out))

    def assemble():
        kkt.compress_jacobian(d_jac)
        kkt.compress_hessian(d_hess)
        kkt.build_kkt(d_pr, d_du)

    def one_solve():
        d_x.copy_(d_rhs)
        ls.solve_linear_system(d_x)

    reps = max(3, min(10, args.steps))
    ms = {"assemble": phase_ms(assemble, reps), "factorize": phase_ms(ls.factorize_async, reps),
          "solve": phase_ms(one_solve, reps)}
    allms = gather_stats([ms["assemble"], ms["factorize"], ms["solve"]], dist, world, dt)

    out = None
    if rank == 0:
        N = P.n
        flops = N ** 3 / 3.0
        fact_ms = float(np.mean([a[1] for a in allms]))
        ach = flops / (fact_ms * 1e-3) / 1e12
        out = {
            "metric": METRIC, "value": world * args.batch * args.steps / elapsed, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.case}-shaped sparse-condensed KKT (SURVEY 8d C3), N={N}, m={P.m}, "
                                   f"nnz(K)={kkt.nnz_aug}, one instance per GPU; step = compress_J+compress_H+"
                                   f"build_kkt+factorize (n_f=1) + {args.nsolve} solve_linear_system",
                       "algorithm": f"{args.algorithm} (device: {'Cholesky' if args.algorithm == 'CHOLESKY' else 'static-pivot LDL^T'})",
                       "outer_block": args.outer_block, "batch_per_gpu": args.batch,
                       "parallelism": f"{world} GPU(s) x {args.batch} independent instance(s)"},
            "ms_per_factorize": fact_ms,
            "ms_per_solve": float(np.mean([a[2] for a in allms])),
            "ms_assemble": float(np.mean([a[0] for a in allms])),
            "per_rank_ms": allms,
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP64_TFLOPS, "traffic": pmc_traffic(N, args),
                         "kernel": "factorize! (densify + blocked LDL^T/Cholesky; N^3/3 flop per call, "
                                   "HIP-event timed on the launch stream)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(P, args.algorithm, args.nsolve)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def dry_run(args):
    """CPU/gloo exercise of the launch contract (no kernels, not a measurement)."""
    import torch
    rank, world, local, dist = dist_setup(args.gpus, "gloo")
    dt = lambda v: torch.tensor(v, dtype=torch.float64)  # noqa: E731
    step = lambda: time.sleep(0.002 * (1 + rank))  # noqa: E731  (rank-dependent: MAX must pick the slowest)
    elapsed = timed_region(step, args.steps, args.warmup, lambda: None, dist, dt)
    allms = gather_stats([1.0 + rank, 2.0 + rank, 3.0 + rank], dist, world, dt)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": world * args.steps / elapsed, "unit": "it/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "dry-run (no kernels, not a measurement)", "per_rank_ms": allms,
                          "config": {"workload": "cpu dry run"}}))


if __name__ == "__main__":
    main()
