#!/usr/bin/env python
"""bench.py -- MadNLP KKT hot path on MI355X: IP iterations/s + ms per factorize!/solve!.

One "step" = the per-iteration hot path of one interior-point iteration on one batch of
synthetic input, with every input already resident in HBM when the timed region starts:

    compress_jacobian! + compress_hessian! + build_kkt! + factorize! (n_f = 1) + inertia fetch
    + n_s = 2 x solve_linear_system!

on the OPF-shaped sparse-condensed KKT system of BASELINE.json configs[2]
(case1354pegase-shaped, N = 11192; the metric is quoted on "condensed KKT n~1e4"), with the
reference's default algorithm (BUNCHKAUFMAN, served by the static-pivot LDL^T).  Timing
follows the reference's `timing_linear_solver` protocol (src/utils.jl:185-197): warm-up,
then the mean over the timed repetitions.

Launch:  python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: the script
         starts its own N ranks under torch.distributed.run, rendezvous on 127.0.0.1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
N = 1 runs BASELINE config C3 (one instance).  N > 1 runs BASELINE config C5's shape: independent
case1354pegase-shaped scenarios, 16 per GPU (seeds 1354 + i), one process per GPU (weak scaling);
the only collectives are a barrier and the aggregation of the timings (RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def _cpu_quota():
    """CPUs this process may actually use: min(visible cpus, cgroup v2 cpu.max quota).  The GPU boxes of the pool show
    256 logical cpus but carry `cpu.max = 1600000 100000` (16 CPUs): BLAS / OpenMP pools sized after the visible count
    spin on 100+ threads, the cgroup gets throttled and EVERY thread of the process -- the one driving the GPU included --
    freezes for 40-80 ms at a time (measured: GPU events 11.9 ms, host wait up to 88 ms; cpu.stat nr_throttled rising)."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


CPU_CAP = _cpu_quota()
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(CPU_CAP))  # must be set before numpy / torch create their pools

import numpy as np  # noqa: E402

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
    ap.add_argument("--outer-block", type=int, default=0, help="outer panel width of the factorization; 0 = the library's default (by size)")
    ap.add_argument("--batch", type=int, default=None,
                    help="independent NLP instances per GPU, each on its own context/stream; a step advances "
                         "every instance by one iteration.  Default: 1 at --gpus 1 (BASELINE config C3, the "
                         "configuration the metric is quoted on), 16 at --gpus N > 1 (BASELINE config C5: "
                         "128 scenarios sharded 16 per GPU, seeds 1354 + i)")
    ap.add_argument("--concurrency", type=int, default=1,
                    help="contexts (stream sets) the batch is spread over.  Default 1.  The persistent operations of all "
                         "contexts of a process take turns on the device (the arbiter of csrc/common.h) and stay on the "
                         "task-DAG schedule: measured per GPU at batch 16 with the batch API, 107-109 it/s with 1 context and "
                         "with 4 (profiles/r04_config_C5_*.json)")
    ap.add_argument("--no-batch-api", action="store_true",
                    help="batch > 1: enqueue the factorizations one by one instead of through mnk_factorize_batch_begin/_end "
                         "(one merged persistent launch for all instances of a step: they fill each other's chain-bound ends)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="keep roofline.traffic from the committed counter record instead of measuring it in this run (two child processes, ~15 s)")
    ap.add_argument("--no-c5-shape", action="store_true",
                    help="skip the supplementary `c5_shape_per_gpu` record (16 instances per step on this GPU, ~15 s)")
    ap.add_argument("--no-c4", action="store_true", help="skip the supplementary config-C4 record (case9241pegase shape, ~15 s)")
    ap.add_argument("--no-ipm-loop", action="store_true",
                    help="skip the supplementary end-to-end IPM run (device-resident vectors) reported as `end_to_end_ipm`")
    ap.add_argument("--cpu-baseline-budget", type=float, default=100.0,
                    help="seconds of host time the cpu_baseline leg may spend (it drops the slowest legs first)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="test-only: exercise the multi-process harness on CPU (gloo) without any kernel")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 1 if args.gpus == 1 else 16
    return args


# ---------------------------------------------------------------------------- distributed harness
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU
    under torch.distributed.run, rendezvous on 127.0.0.1) and hand back their exit code.  Under a
    launcher (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


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
    trace = os.environ.get("MNK_BENCH_STEP_TIMES")  # diagnostics: host time stamps per step (steps end with a sync)
    stamps = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
        if trace:
            stamps.append(time.perf_counter())
    sync_fn()
    elapsed = time.perf_counter() - t0
    if trace:
        prev = [t0] + stamps[:-1]
        print("step times (ms):", " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(prev, stamps)), file=sys.stderr)
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


# ---------------------------------------------------------------------------- supplementary end-to-end IPM run
def ipm_loop(args, ctx, model="acopf"):
    """Full interior-point run with every vector AND the model's callbacks in HBM (`DeviceMadNLPSolver`): iterations,
    factorizations, back-solves and wall clock of the second (warm) of two runs.  model = "acopf": the polar AC-OPF NLP on
    the synthetic grid of `args.case` (callbacks: csrc/opf_eval.hip; nonconvex, so inertia corrections refactorize);
    "qp": the convex QP with the same sparsity (callbacks: SpMV on the KKT handle's compressed matrices)."""
    import torch
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm import IPMOptions
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    from madnlp_jl_amd.problems import ACOPFModel, SparseQPModel
    nlp = ACOPFModel(args.case) if model == "acopf" else SparseQPModel(args.case)
    what = ("polar AC-OPF NLP on the synthetic grid" if model == "acopf" else "convex QP with the sparsity")

    def factory(info):
        return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J,
                                           info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                           opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=args.algorithm,
                                                                                  outer_block=args.outer_block),
                                           device_kkt_ops=True)
    rec = None
    for _ in range(2):
        o = IPMOptions(tol=1e-6)
        o.relax_equality, o.dual_initialization = True, "zero"
        s = DeviceMadNLPSolver(nlp, factory, o)
        s.initialize()
        s._upload()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        rec = {"problem": f"{what} of {args.case} (n={nlp.n}, m={nlp.m}), vectors and callbacks device-resident",
               "status": s.status, "objective": s.obj_val, "iterations": s.cnt.k, "factorizations": s.cnt.factorization_cnt,
               "backsolves": s.cnt.backsolve_cnt, "wall_s": wall, "ms_per_iteration": 1e3 * wall / max(1, s.cnt.k),
               "it_per_s": s.cnt.k / wall}
        try:   # (how many of the factorizations were trials the static-pivot tier rejected at their first non-positive pivot)
            rec["early_rejected_trials"] = int(s.kkt.linear_solver.get_stat("early_rejects") +
                                               (s.kkt.spare_solver.get_stat("early_rejects") if getattr(s.kkt, "spare_solver", None) else 0))
        except Exception:
            pass
        # speculative first corrections (ipm_dev.inertia_correction): trials factorized ahead of the verdict on the unperturbed
        # matrix, in one merged launch with it; `wasted`: the unperturbed matrix was accepted after all
        # leading-block probes (ipm_dev.inertia_correction): unperturbed matrices rejected by a factorization of their leading principal
        # block alone / probes that passed (the full factorization followed)
        rec["probe_hits"], rec["probe_misses"] = int(s.probe_hits), int(s.probe_misses)
        rec["speculative_factorizations"] = int(s.speculative_factorizations)
        rec["speculative_wasted"] = int(s.speculative_wasted)
        s.cb.close()
        s.K.close()
        s.kkt.close()
    if model == "acopf":
        # checker: the same NLP solved by the host driver on the oracle back-end (numpy assembly + LAPACK Bunch-Kaufman), a
        # committed fixture (tests/golden/make_acopf_case1354_golden.py; tests/test_acopf.py asserts the agreement)
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "acopf_case1354_oracle.json")))
            if gold["case"] == args.case:
                rec["oracle_golden"] = {"fixture": "tests/golden/acopf_case1354_oracle.json", "status": gold["status"],
                                        "iterations": gold["iterations"], "factorizations": gold["factorizations"],
                                        "objective": gold["objective"],
                                        "objective_rel_diff": abs(rec["objective"] - gold["objective"]) / abs(gold["objective"]),
                                        "oracle_host_wall_s": gold["host_wall_s"]}
        except Exception:
            pass
    return rec


# ---------------------------------------------------------------------------- CPU baseline leg
def _host_threads():
    """(physical cores, logical cpus) of this host."""
    logical = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
    except Exception:
        phys = max(1, logical // 2)
    return int(phys), int(logical)


def cpu_baseline(P, algorithm, nsolve, budget_s=100.0):
    """The oracle (CPU restatement: numpy assembly + CSC->dense `transfer_matrix!` + OpenBLAS LAPACK
    dsytrf/dsytrs | dpotrf/dpotrs through scipy -- the routine family MadNLP's LapackCPUSolver calls,
    reference src/LinearSolvers/lapack.jl:145-172) timed on this host's cores on a BOUNDED sample of the
    same workload: one iteration of the hot path per (algorithm, BLAS-thread) setting, after a warm-up
    factorization at N = 2048 that spins the BLAS thread pool up (the reference protocol is warm-up +
    mean of 10 trials, src/utils.jl:185-197; ten 1-thread factorizations at N = 11192 would take
    minutes, so each setting is timed once and the sample says so).

    Thread settings: 1 BLAS thread = MadNLP's default `blas_num_threads = 1`
    (reference src/IPM/options.jl:127, applied at src/IPM/IPM.jl:143), and the better of
    {physical cores / 2, physical cores}.  `value`/`cores` quote the FASTEST setting of the bench's
    algorithm; `reference_default` is the 1-thread run."""
    from threadpoolctl import threadpool_limits
    from oracle import kernels as ok
    from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, LapackCPUSolver
    from oracle.sparse_condensed import SparseCondensedKKTSystem as OracleSC
    t_start = time.perf_counter()
    phys, logical = _host_threads()
    main_alg = CHOLESKY if algorithm == "CHOLESKY" else BUNCHKAUFMAN
    other_alg = BUNCHKAUFMAN if main_alg == CHOLESKY else CHOLESKY
    k = OracleSC(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                 lambda A: LapackCPUSolver(A, main_alg))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    ok.set_aug_diagonal(k)
    b = np.random.default_rng(0).standard_normal(P.n)
    # assembly (thread-independent numpy): warm-up once, then the mean of 3
    k.compress_jacobian(); k.compress_hessian(); k.build_kkt()
    t0 = time.perf_counter()
    for _ in range(3):
        k.compress_jacobian(); k.compress_hessian(); k.build_kkt()
    ms_build = 1e3 * (time.perf_counter() - t0) / 3
    # warm-up matrix (SPD, N = 2048)
    rng = np.random.default_rng(1)
    Wm = rng.standard_normal((2048, 256))
    Wm = np.asfortranarray(Wm @ Wm.T + 2048 * np.eye(2048))

    def one(alg, threads, samples=1):
        tfs, tss = [], []
        with threadpool_limits(limits=threads, user_api="blas"):
            LapackCPUSolver(Wm, alg).factorize()  # warm-up
            for _ in range(samples):
                ls = LapackCPUSolver(k.aug_com, alg)
                t0 = time.perf_counter(); ls.factorize(); tfs.append(time.perf_counter() - t0)
                ls.solve_linear_system(b.copy())
                t0 = time.perf_counter()
                for _ in range(nsolve):
                    ls.solve_linear_system(b.copy())
                tss.append((time.perf_counter() - t0) / max(nsolve, 1))
        tf, ts = float(np.mean(tfs)), float(np.mean(tss))
        total = 1e-3 * ms_build + tf + nsolve * ts
        return {"algorithm": alg, "lapack": "dpotrf/dpotrs" if alg == CHOLESKY else "dsytrf/dsytrs",
                "blas_threads": int(threads), "samples": samples, "ms_per_factorize": 1e3 * tf, "ms_per_solve": 1e3 * ts,
                "ms_per_factorize_samples": [1e3 * v for v in tfs],
                "it_per_s": 1.0 / total, "gflops_factorize": P.n ** 3 / 3.0 / tf / 1e9}

    # Most informative legs first.  Thread counts never exceed what the process may use (CPU_CAP = min(cores, cgroup
    # quota): on the pool's boxes 16 of the 256 visible cpus -- round 2's first protocol ran 64 / 128 BLAS threads there
    # and measured the cgroup throttle, not OpenBLAS: 12.6 s for dsytrf).
    cap = max(1, min(phys, CPU_CAP))
    few = max(1, min(16, cap))
    plan = [(main_alg, 1), (main_alg, few), (main_alg, cap), (other_alg, cap), (other_alg, 1),
            (main_alg, max(1, cap // 2)), (other_alg, max(1, cap // 2)), (other_alg, few)]
    runs, skipped, seen = [], [], set()
    for alg, thr in plan:
        if (alg, thr) in seen:
            continue
        seen.add((alg, thr))
        # predicted cost of this leg: the slowest run of the same algorithm so far (1-thread legs: 40 s if unknown)
        prev = [r["ms_per_factorize"] for r in runs if r["algorithm"] == alg]
        est = 1e-3 * max(prev) if prev else (8.0 if thr == 1 else 4.0)   # (measured on the pool's hosts: 5.3 s at 1 thread, 1.6 s at 16)
        if runs and time.perf_counter() - t_start + est > budget_s:
            skipped.append({"algorithm": alg, "blas_threads": int(thr), "reason": "cpu-baseline time budget"})
            continue
        # the bench's algorithm at the reference default (1 BLAS thread) and at the fastest candidate settings: mean of three
        # samples (the reference's protocol is a mean over trials); the other legs once
        n_samp = 3 if (alg == main_alg and len(runs) < 3 and time.perf_counter() - t_start + 3 * est <= budget_s) else 1
        runs.append(one(alg, thr, n_samp))
    mine = [r for r in runs if r["algorithm"] == main_alg]
    best = max(mine, key=lambda r: r["it_per_s"])
    ref_default = next((r for r in mine if r["blas_threads"] == 1), None)
    return {
        "value": best["it_per_s"], "unit": "it/s", "cores": best["blas_threads"], "kind": "port",
        "sample": f"iterations of the same hot path per setting ({P.name}-shaped, N={P.n}; numpy assembly + "
                  f"CSC->dense copy + scipy/OpenBLAS {best['lapack']}): mean of {best.get('samples', 1)} sample(s) for the quoted "
                  f"setting (`runs[].samples` per setting), each setting after a warm-up factorization at "
                  f"N=2048; value = fastest BLAS-thread setting of {main_alg}; host has {phys} physical cores / "
                  f"{logical} logical cpus, of which the process may use {CPU_CAP} (cgroup cpu.max)",
        "ms_per_factorize": best["ms_per_factorize"], "ms_per_solve": best["ms_per_solve"], "ms_build": ms_build,
        "host_physical_cores": phys, "host_logical_cpus": logical, "host_cpu_quota": CPU_CAP,
        "reference_default": ref_default,   # blas_num_threads = 1 (reference src/IPM/options.jl:127)
        "runs": runs, "skipped": skipped,
        "seconds_spent": time.perf_counter() - t_start,
    }


def pmc_traffic(N, args, schedule):
    """Bytes per factorize! call at the L2 -> fabric interface (Infinity Cache + HBM) from the committed counter passes of
    this same system (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes: counters cannot be collected inside
    the timed run).  Returns (bytes or None, the profile the number came from).  A profile only describes the schedule it
    was collected on (its "panel_algo"; the round-2 / round-3 files: 4; round 4: 5, the schedule the bench runs)."""
    # r04: the task-DAG schedule itself, through rocprofiler-sdk's device counting service (tools/devcount_dag.py: agent-wide
    # sampling without dispatch serialization -- `rocprofv3 --pmc` cannot run the two persistent kernels side by side)
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if N == 11192 and args.batch == 1 and os.path.exists(path):
            try:
                rec = json.load(open(path))
                if int(rec.get("panel_algo", 4)) == int(schedule):
                    return rec["traffic_bytes"], "profiles/" + name
            except Exception:
                pass
    return None, None


def live_traffic(N, schedule, calls=8, timeout_s=120):
    """Bytes per factorize! call at the L2 -> fabric interface, MEASURED in this bench run (the replayed record
    goes stale silently with the next tiling change): two short passes of tools/devcount_dag.py in child processes -- rocprofiler-sdk's
    device counting service (agent-wide sampling, no dispatch serialization: the schedule's two persistent kernels run side by side
    as in the timed region), one counter set per pass as MI355X_MICROARCH.md prescribes -- on the same C3 system, `calls` calls each.
    FETCH_SIZE / WRITE_SIZE by their gfx950 definitions, reads x2-corrected (128-B requests tallied at 64 B).  Returns None when the
    counting service is not to be had (no tool library, a pass fails or runs another schedule): the caller falls back to the record."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "devcount", "libmnk_devcount.so")
    if N != 11192 or not os.path.exists(tool):
        return None
    env = dict(os.environ, ROCP_TOOL_LIBRARIES=tool)
    got = {}
    try:
        for which in ("fetch", "write"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "devcount_dag.py"), which, str(calls)], env=env,
                               capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                return None
            rec = json.loads(line[-1])
            if int(rec["schedule_panel_algo"]) != int(schedule) or not rec["inertia_ok"] or rec["pp_fallbacks"] != 0:
                return None
            got[which] = rec
        cf, cw = got["fetch"]["counters_per_call"], got["write"]["counters_per_call"]
        rd, rd32, bub = cf["TCC_EA0_RDREQ"], cf.get("TCC_EA0_RDREQ_32B", 0.0), cf.get("TCC_BUBBLE", 0.0)
        fetch_b = bub * 128 + (rd - bub - rd32) * 64 + rd32 * 32
        wr, wr64 = cw["TCC_EA0_WRREQ"], cw.get("TCC_EA0_WRREQ_64B", 0.0)
        write_b = wr64 * 64 + (wr - wr64) * 32
        return {"traffic": 2.0 * fetch_b + write_b, "read_bytes_x2_corrected": 2.0 * fetch_b, "write_bytes": write_b,
                "calls_per_pass": calls, "ms_per_call_under_counting": 0.5 * (got["fetch"]["ms_per_call"] + got["write"]["ms_per_call"])}
    except Exception:
        return None


def config_c4(ctx, torch, mj):
    """Supplementary record, OUTSIDE the timed region: BASELINE config C4 (case9241pegase-shaped sparse-condensed KKT,
    N = 85 568, 58.7 GB factor) -- the configuration north_star's >= 5x and >= 70 % targets name.  One warm and one timed
    `factorize!`, two timed `solve!`, HIP events on the launch stream; the CPU side is the labelled N^3 extrapolation of
    profiles/r02_config_C4_cpu_extrapolation.json (a 58.7 GB dsytrf does not finish inside a bench run)."""
    from madnlp_jl_amd.problems import opf_shaped
    free, _ = torch.cuda.mem_get_info()
    if free < 75e9:
        return {"skipped": f"free HBM {free/1e9:.0f} GB < 75 GB"}
    t0 = time.perf_counter()
    P = opf_shaped("case9241pegase", du=1e-8)
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                    opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    setup_s = time.perf_counter() - t0
    dev = torch.device("cuda", torch.cuda.current_device())
    dj, dh = torch.from_numpy(P.jac).to(dev), torch.from_numpy(P.hess).to(dev)
    dp, dd = torch.from_numpy(P.pr_diag).to(dev), torch.from_numpy(P.du_diag).to(dev)

    def ev_ms(fn):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    def assemble():
        k.compress_jacobian(dj)
        k.compress_hessian(dh)
        k.build_kkt(dp, dd)
    assemble()
    k.linear_solver.factorize_async()          # warm (first-touch of the 58.7 GB factor)
    ms_a = ev_ms(assemble)
    ms_f = ev_ms(k.linear_solver.factorize_async)
    inertia = k.linear_solver.inertia()
    x = torch.randn(P.n, dtype=torch.float64, device=dev)
    ms_s = float(np.mean([ev_ms(lambda: k.linear_solver.solve_linear_system(x)) for _ in range(2)]))
    N = P.n
    rec = {"workload": f"BASELINE config C4: case9241pegase-shaped sparse-condensed KKT, N={N}, nnz(K)={k.nnz_aug}; one warm + one "
                       f"timed factorize!, two timed solve!",
           "ms_assemble": ms_a, "ms_per_factorize": ms_f, "ms_per_solve": ms_s, "inertia_ok": bool(k.is_inertia_correct(*inertia)),
           "it_per_s_nf1_ns2": 1e3 / (ms_a + ms_f + 2 * ms_s),
           "roofline": {"bound": "mfma", "achieved": N ** 3 / 3.0 / (ms_f * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                        "frac": N ** 3 / 3.0 / (ms_f * 1e-3) / 1e12 / PEAK_FP64_TFLOPS, "traffic": None},
           "schedule_panel_algo": k.linear_solver.get_stat("panel_algo"), "setup_s": setup_s}
    try:   # bytes per factorize! at the L2 -> fabric interface: the committed counter passes of this system on this schedule
        tr = json.load(open(os.path.join(ROOT, "profiles", "r06_config_C4_pmc_traffic.json")))
        if int(rec["schedule_panel_algo"]) == 4:
            rec["roofline"].update(traffic=tr["traffic_bytes"], traffic_source="profiles/r06_config_C4_pmc_traffic.json",
                                   traffic_over_algorithmic=tr["traffic_bytes"] / (8.0 * N * N))
    except Exception:
        pass
    try:
        ext = json.load(open(os.path.join(ROOT, "profiles", "r02_config_C4_cpu_extrapolation.json")))
        best = min((r for r in ext["rows"] if r["routine"] == "dsytrf" and r["N"] == max(q["N"] for q in ext["rows"])),
                   key=lambda r: r["seconds"])
        cpu_s = best["seconds"] * (N / best["N"]) ** 3
        rec["cpu_baseline_extrapolated"] = {
            "kind": "port", "labelled": f"N^3 extrapolation of scipy/OpenBLAS dsytrf at N={best['N']}, {best['threads']} threads "
                                        f"({best['seconds']:.2f} s measured, profiles/r02_config_C4_cpu_extrapolation.json)",
            "s_per_factorize": cpu_s, "gpu_over_cpu_factorize": cpu_s / (ms_f * 1e-3)}
    except Exception:
        pass
    k.close()
    return rec


def shader_clock(local=0):
    """Shader (sclk) and memory clock of the GPU right now, MHz -- read before and after the timed region so that a box that
    runs the same build 2 % slower shows up in the line (9.52 ms on the driver's box against 9.28-9.35 on four
    others).  sysfs first (pp_dpm_sclk: the level marked '*'), `rocm-smi --showclocks` as the fall-back; None if neither
    answers (the bench never fails on this)."""
    import glob
    import re
    import subprocess
    rec = {}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if cards:
            path = cards[min(local, len(cards) - 1)]
            for key, fn in (("sclk_mhz", path), ("mclk_mhz", path.replace("pp_dpm_sclk", "pp_dpm_mclk"))):
                cur = [ln for ln in open(fn).read().splitlines() if ln.strip().endswith("*")]
                if cur:
                    rec[key] = float(re.search(r"(\d+)\s*Mhz", cur[0], re.I).group(1))
            if rec:
                rec["source"] = "sysfs pp_dpm_sclk"
                return rec
    except Exception:
        pass
    try:
        out = subprocess.run(["rocm-smi", "-d", str(local), "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        for key, tag in (("sclk_mhz", "sclk"), ("mclk_mhz", "mclk")):
            m = re.search(tag + r" clock level:?\s*\d*:?\s*\((\d+)Mhz\)", out, re.I)
            if m:
                rec[key] = float(m.group(1))
        if rec:
            rec["source"] = "rocm-smi --showclocks"
            return rec
    except Exception:
        pass
    return None


# ---------------------------------------------------------------------------- main
def main():
    args = parse_args()
    rc = maybe_self_spawn(args)
    if rc is not None:
        raise SystemExit(rc)
    if args.cpu_dry_run:
        return dry_run(args)

    import torch
    rank, world, local, dist = dist_setup(args.gpus, "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the KKT hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.problems import OPF_CASES, opf_shaped
    # GPU phase: no host thread pool may spin beside the thread that drives the GPU (see _cpu_quota)
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        host_pool_limit = threadpool_limits(limits=1)
    except Exception:
        host_pool_limit = None

    dev = torch.device("cuda", local)
    # a non-default stream: its handle is non-NULL, the library enqueues on it and torch's
    # events (HIP events on that same stream) bracket the kernels
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    ctx = mj.HipContext(local, stream=tstream.cuda_stream)
    def make_workload(batch, concurrency):
        """`batch` independent instances on this GPU and the step that advances every one of them by one iteration."""
        # independent instances: seed = base + rank * batch + b; instance 0 uses the bench stream
        base_seed = OPF_CASES[args.case][0] + rank * batch
        insts = []
        ctxs = [(ctx, tstream)]
        for bidx in range(batch):
            slot = bidx % max(1, concurrency)
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

        # One IPM iteration of the hot path.  The inertia is FETCHED inside the step (device sync + D2H of
        # three counters): the IPM cannot decide between "solve" and "regularize + refactorize" without it
        # (reference src/IPM/solver.jl:611-670), so a real iteration pays that synchronization.
        def step_front(kb, st, din):
            with torch.cuda.stream(st):
                kb.compress_jacobian(din["jac"])
                kb.compress_hessian(din["hess"])
                kb.build_kkt(din["pr"], din["du"])
                kb.linear_solver.factorize_async()

        def step_back(kb, st, din):
            with torch.cuda.stream(st):
                inertia = kb.linear_solver.inertia()
                if not kb.is_inertia_correct(*inertia):
                    raise SystemExit(f"unexpected inertia {inertia} on the benchmark system")
                for _ in range(args.nsolve):
                    din["x"].copy_(din["rhs"])
                    kb.linear_solver.solve_linear_system(din["x"])

        use_batch_api = batch > 1 and not args.no_batch_api
        sbatches = []
        if use_batch_api:
            # one ScenarioBatch per context / stream (array entry points: one library call per phase and context instead of
            # four to six per instance)
            for (cctx, cst) in ctxs:
                mine = [it for it in insts if it[2] is cst]
                if mine:
                    sb = mj.ScenarioBatch([it[1] for it in mine])
                    sb.bind([it[3]["jac"] for it in mine], [it[3]["hess"] for it in mine], [it[3]["pr"] for it in mine],
                            [it[3]["du"] for it in mine])
                    sbatches.append((sb, cst, mine, [it[3]["x"] for it in mine], [it[3]["rhs"] for it in mine]))

        def step():
            # batch: every instance's assembly + factorization is enqueued before the first inertia fetch
            # blocks the host, so the contexts keep the chip busy while the host waits
            if use_batch_api:
                with mj.factorize_batch():      # (all contexts' instances in ONE batch)
                    for (sb, cst, mine, xs, rhss) in sbatches:
                        sb.step()
            else:
                for (_, kb, st, din) in insts:
                    step_front(kb, st, din)
            if use_batch_api:
                # every instance's inertia, then solve k of all instances together (mnk_solve_batch_begin / _end: up to four
                # independent systems per launch), k = 1 .. nsolve -- the same work per instance as step_back
                for (sb, cst, mine, xs, rhss) in sbatches:
                    for it, inertia in zip(mine, sb.inertia()):
                        if not it[1].is_inertia_correct(*inertia):
                            raise SystemExit(f"unexpected inertia {inertia} on the benchmark system")
                for _ in range(args.nsolve):
                    with mj.solve_batch():      # (all contexts' instances in ONE batch)
                        for (sb, cst, mine, xs, rhss) in sbatches:
                            with torch.cuda.stream(cst):
                                torch._foreach_copy_(xs, rhss)
                            sb.solve(xs)
            else:
                for (_, kb, st, din) in insts:
                    step_back(kb, st, din)

        def close():
            for (_, kb, _, _) in insts:
                kb.close()
            for (cctx, _) in ctxs[1:]:
                cctx.close()
        return {"insts": insts, "step": step, "use_batch_api": use_batch_api, "close": close, "P": P, "kkt": kkt, "ls": ls,
                "dev_in": insts[0][3]}

    wl = make_workload(args.batch, args.concurrency)
    insts, step, use_batch_api, P, kkt, ls = wl["insts"], wl["step"], wl["use_batch_api"], wl["P"], wl["kkt"], wl["ls"]
    d_jac, d_hess, d_pr, d_du, d_rhs, d_x = (wl["dev_in"][k] for k in ("jac", "hess", "pr", "du", "rhs", "x"))

    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    dt = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)  # noqa: E731

    def c5_shape_solo(step16, nb, steps=3):
        """BASELINE config C5's per-GPU share on THIS GPU alone: `nb` independent instances per step through the batch API,
        one warm-up + `steps` timed steps (device synchronization on both sides)."""
        step16()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step16()
        sync()
        el = time.perf_counter() - t0
        return {"value": nb * steps / el, "unit": "it/s per GPU", "instances_per_step": nb, "steps": steps, "ms_per_step": 1e3 * el / steps,
                "ms_per_instance": 1e3 * el / steps / nb}

    # N > 1: the denominator of the scaling curve is measured in the same run -- rank 0's GPU alone on its own 16 instances while the
    # other ranks wait at a barrier (`--gpus 1` is config C3, one instance; its value is NOT the per-GPU share of C5)
    solo = None
    if dist is not None:
        for _ in range(args.warmup):
            step()
        sync()
        dist.barrier()
        if rank == 0:
            solo = c5_shape_solo(step, args.batch)
        dist.barrier()

    def clocks_now():
        """sysfs / rocm-smi reading (the state between two steps: a box that runs 2.4 GHz under load may show its sleep state) and the
        clock measured under fp64 MFMA load by the library (s_memtime against the 100 MHz counter, the second of two 2.5 ms launches)."""
        rec = shader_clock(local) or {}
        try:
            import ctypes
            from madnlp_jl_amd import _lib as L
            mhz = ctypes.c_double(0.0)
            if L.lib().mnk_debug_shader_clock(ctx.handle, ctypes.byref(mhz)) == 0:
                rec["sclk_under_mfma_load_mhz"] = mhz.value
        except Exception:
            pass
        return rec or None

    clk0 = clocks_now() if rank == 0 else None
    # (the inertia of every factorization is checked inside the step: wrong inertia voids the numbers)
    elapsed = timed_region(step, args.steps, args.warmup, sync, dist, dt)
    clk1 = clocks_now() if rank == 0 else None

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
        return out

    def assemble():
        kkt.compress_jacobian(d_jac)
        kkt.compress_hessian(d_hess)
        kkt.build_kkt(d_pr, d_du)

    def one_solve():
        d_x.copy_(d_rhs)
        ls.solve_linear_system(d_x)

    reps = max(3, min(10, args.steps))
    samples = {"assemble": phase_ms(assemble, reps), "factorize": phase_ms(ls.factorize_async, reps), "solve": phase_ms(one_solve, reps)}
    ms = {k: float(np.mean(v)) for k, v in samples.items()}
    allms = gather_stats([ms["assemble"], ms["factorize"], ms["solve"]], dist, world, dt)

    out = None
    if rank == 0:
        N = P.n
        flops = N ** 3 / 3.0
        fact_ms = float(np.mean([a[1] for a in allms]))
        ach = flops / (fact_ms * 1e-3) / 1e12
        schedule = ls.get_stat("panel_algo")
        # > 0: a device-side hand-off timed out and schedule 1 ran instead; summed over every instance of this rank
        fallbacks = float(sum(it[1].linear_solver.get_stat("pp_fallbacks") for it in insts))
        traffic, traffic_src = pmc_traffic(N, args, schedule)
        out = {
            "metric": METRIC, "value": world * args.batch * args.steps / elapsed, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE config C3: {args.case}-shaped sparse-condensed KKT, one instance"
                                    if world * args.batch == 1 else
                                    f"BASELINE config C5 shape: {world * args.batch} independent {args.case}-shaped "
                                    f"scenarios sharded {args.batch} per GPU across {world} GPU(s), seeds "
                                    f"{OPF_CASES[args.case][0]}+i, {min(args.concurrency, args.batch)} concurrent "
                                    f"contexts per GPU") +
                                   f"; N={N}, m={P.m}, nnz(K)={kkt.nnz_aug}; step = compress_J+compress_H+"
                                   f"build_kkt+factorize (n_f=1) + inertia fetch + {args.nsolve} solve_linear_system"
                                   f" per instance",
                       "algorithm": f"{args.algorithm} (device: {'Cholesky' if args.algorithm == 'CHOLESKY' else 'static-pivot LDL^T'})",
                       "outer_block": args.outer_block, "batch_per_gpu": args.batch,
                       "batch_api": ("mnk_factorize_batch_begin/_end: one merged persistent launch per step" if use_batch_api
                                     else ("off" if args.batch > 1 else "n/a (one instance)")),
                       "parallelism": f"{world} GPU(s) x {args.batch} independent instance(s)"},
            "ms_per_factorize": fact_ms,
            # rank 0's own repetitions of the phase (the mean above is what `roofline` is computed from)
            "ms_per_factorize_min": float(np.min(samples["factorize"])), "ms_per_factorize_median": float(np.median(samples["factorize"])),
            "ms_per_factorize_samples": [float(v) for v in samples["factorize"]],
            "ms_per_solve_min": float(np.min(samples["solve"])), "ms_per_solve_median": float(np.median(samples["solve"])),
            "clocks": {"before_timed_region": clk0, "after_timed_region": clk1},
            "ms_per_solve": float(np.mean([a[2] for a in allms])),
            "ms_assemble": float(np.mean([a[0] for a in allms])),
            "per_rank_ms": allms,
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "factorize! (densify + blocked LDL^T/Cholesky; N^3/3 flop per call, "
                                   "HIP-event timed on the launch stream)",
                         "schedule_panel_algo": schedule, "pp_fallbacks": fallbacks,
                         # host milliseconds lost to expired device-side waits in this process (0 on a healthy run)
                         "stall_ms_process": float(ls.get_stat("stall_ms_process"))},
        }
        if world == 1 and args.batch == 1 and not args.no_live_traffic:
            # (child processes on the same GPU, after the timed region and the phase timings; ~15 s)
            lt = live_traffic(N, schedule)
            if lt is not None:
                out["roofline"].update(traffic=lt["traffic"], traffic_source="measured in this run: tools/devcount_dag.py (rocprofiler-sdk device "
                                       f"counting service), FETCH and WRITE passes of {lt['calls_per_pass']} factorize! calls each",
                                       traffic_read_bytes=lt["read_bytes_x2_corrected"], traffic_write_bytes=lt["write_bytes"],
                                       traffic_over_algorithmic=lt["traffic"] / (8.0 * N * N),
                                       ms_per_call_under_counting=lt["ms_per_call_under_counting"])
        if world > 1 and solo is not None:
            # the scaling curve's own denominator: rank 0's GPU alone on the same per-GPU share (measured above, outside the timed region)
            out["per_gpu_value"] = out["value"] / world
            out["c5_shape_per_gpu"] = dict(solo, workload=f"rank 0 alone: {args.batch} independent {args.case}-shaped instances per step, batch API")
            out["efficiency_vs_c5_shape_per_gpu"] = out["per_gpu_value"] / solo["value"]
        if world == 1 and args.batch == 1 and not args.no_c5_shape:
            # Supplementary, OUTSIDE the timed region: BASELINE config C5's per-GPU share (16 independent instances per step through
            # the batch API) on this GPU -- the number a `--gpus N` line's value / N is to be compared with
            try:
                wl16 = make_workload(16, 1)
                rec16 = c5_shape_solo(wl16["step"], 16)
                f16 = phase_ms(lambda: [wl16["step"]()], 1)[0]   # (one more step between events: the device's view of it)
                rec16["ms_per_step_device"] = f16
                rec16["workload"] = (f"BASELINE config C5 shape on one GPU: 16 independent {args.case}-shaped instances per step (seeds "
                                     f"{OPF_CASES[args.case][0]}+i), mnk_factorize_batch_begin/_end, step = assembly + factorize + inertia + "
                                     f"{args.nsolve} solves per instance")
                # fp64-peak fraction of the step's factorizations if the whole step were factorization (a LOWER bound of theirs)
                rec16["frac_lower_bound"] = 16 * flops / (rec16["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS
                rec16["pp_fallbacks"] = float(sum(it[1].linear_solver.get_stat("pp_fallbacks") for it in wl16["insts"]))
                wl16["close"]()
                out["c5_shape_per_gpu"] = rec16
            except Exception as e:  # never let the supplement break the bench line
                out["c5_shape_per_gpu"] = {"error": repr(e)[:300]}
        if not args.no_ipm_loop and world == 1 and args.batch == 1:
            # Supplementary, OUTSIDE the timed region: complete IPM runs with device-resident vectors and callbacks
            # (madnlp_jl_amd.ipm_dev) -- the AC-OPF NLP of the same grid, and the convex QP of the same shape: real inertia
            # corrections, refinement and line search around the same hot path; `value` above stays the contract's
            # synthetic step.
            try:
                out["end_to_end_ipm"] = ipm_loop(args, ctx, "acopf")
                out["end_to_end_ipm_qp"] = ipm_loop(args, ctx, "qp")
            except Exception as e:  # never let the supplement break the bench line
                out["end_to_end_ipm"] = {"error": repr(e)[:300]}
        if not args.no_c4 and world == 1 and args.batch == 1:
            try:
                for (_, kb, _, _) in insts:   # the C3 instances' 1 GB factors are not needed any more
                    kb.close()
                out["config_C4"] = config_c4(ctx, torch, mj)
            except Exception as e:  # never let the supplement break the bench line
                out["config_C4"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:
            if host_pool_limit is not None:
                host_pool_limit.restore_original_limits()  # the CPU leg sets its own BLAS thread counts
            cb = cpu_baseline(P, args.algorithm, args.nsolve, args.cpu_baseline_budget)
            out["cpu_baseline"] = cb
            # GPU it/s over the CPU port's it/s (NOT `vs_baseline`: BASELINE.md holds no published number)
            out["vs_cpu_baseline"] = {
                "best_threads": out["value"] / cb["value"],
                "reference_default_1_thread": (out["value"] / cb["reference_default"]["it_per_s"]
                                               if cb.get("reference_default") else None),
                "note": "GPU: static-pivot LDL^T (no 2x2 pivots); CPU: pivoted dsytrf -- same inputs, same inertia"}
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
    solo = None
    if dist is not None:   # (the same solo phase as the GPU path: rank 0 alone between two barriers)
        dist.barrier()
        if rank == 0:
            t0 = time.perf_counter()
            for _ in range(3):
                step()
            solo = 3 / (time.perf_counter() - t0)
        dist.barrier()
    elapsed = timed_region(step, args.steps, args.warmup, lambda: None, dist, dt)
    allms = gather_stats([1.0 + rank, 2.0 + rank, 3.0 + rank], dist, world, dt)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": METRIC, "value": world * args.steps / elapsed, "unit": "it/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "dry-run (no kernels, not a measurement)", "per_rank_ms": allms,
               "ms_per_factorize": 1.0, "ms_per_factorize_min": 1.0, "ms_per_factorize_median": 1.0,
               "clocks": {"before_timed_region": shader_clock(0), "after_timed_region": shader_clock(0)},
               "config": {"workload": "cpu dry run", "batch_per_gpu": args.batch}}
        if world > 1 and solo is not None:
            out["per_gpu_value"] = out["value"] / world
            out["c5_shape_per_gpu"] = {"value": solo, "unit": "it/s per GPU", "instances_per_step": args.batch, "steps": 3}
            out["efficiency_vs_c5_shape_per_gpu"] = out["per_gpu_value"] / solo
        print(json.dumps(out))


if __name__ == "__main__":
    main()
