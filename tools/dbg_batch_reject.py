"""Hunt: factorization batches in which some instances are rejected early (round 5).  Alternates rounds with / without indefinite
instances and reports every anomaly (a good instance with a wrong inertia, a spurious rejection, factor bits that differ from its
lone factorization).  EARLY=0: the same with early rejection off.  usage: [EARLY=0] [ROUNDS=12] python tools/dbg_batch_reject.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import madnlp_jl_amd as mj
from madnlp_jl_amd.problems import opf_shaped
from tests.test_hip_c5 import _front
early = os.environ.get("EARLY", "1") != "0"
rounds = int(os.environ.get("ROUNDS", "12"))
dev = torch.device("cuda", 0); st = torch.cuda.Stream(dev); ctx = mj.HipContext(0, stream=st.cuda_stream)
bad = set(int(x) for x in os.environ.get("BAD", "1,4").split(","))
insts = []
for i in range(6):
    P = opf_shaped("case1354pegase", seed=4000 + i, du=1e-8, **(dict(indefinite=True, sigma_s_decades=2.0) if i in bad else {}))
    kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                     opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=getattr(mj, os.environ.get("ALG", "BUNCHKAUFMAN"))), early_reject=early)
    mk = lambda pr, P=P: dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev), pr=torch.from_numpy(pr).to(dev), du=torch.from_numpy(P.du_diag).to(dev))
    insts.append(dict(n=P.n, kh=kh, good=mk(P.pr_diag + 100.0 if i in bad else P.pr_diag), bad=mk(P.pr_diag) if i in bad else None))
torch.cuda.synchronize()
ref = []
for it in insts:
    _front(it["kh"], st, it["good"]); assert it["kh"].linear_solver.inertia() == (it["n"], 0, 0)
    Lf, D = it["kh"].linear_solver.get_factor_device(); ref.append((torch.tril(Lf).clone(), D.clone()))
anomalies = 0
for rnd in range(rounds):
    mode = "withbad" if rnd % 2 == 0 else "allgood"
    with mj.factorize_batch():
        for it in insts:
            _front(it["kh"], st, it["bad"] if (mode == "withbad" and it["bad"] is not None) else it["good"])
    for i, it in enumerate(insts):
        M = it["kh"].linear_solver
        with torch.cuda.stream(st):
            ine = M.inertia()
        if os.environ.get("MNK_DBG_NO_REDO") and not (mode == "withbad" and i in bad) and ine != (it["n"], 0, 0):
            col = int(M.get_stat("early_reject_col"))
            Lf, D = M.get_factor_device()
            Lt = torch.tril(Lf)
            c0 = max(0, col - 63)      # columns < c0 are final L of the partial factorization
            dL = (Lt[:, :c0] - ref[i][0][:, :c0]).abs()
            badcols = (dL > 0).any(dim=0).nonzero().flatten()
            print(rnd, mode, i, "SPURIOUS rejection at column", col, "inertia", ine, "| columns < stop that differ from the lone factor:", int(badcols.numel()),
                  "first", int(badcols[0]) if badcols.numel() else None)
            if badcols.numel():
                c = int(badcols[0]); rows = (dL[:, c] > 0).nonzero().flatten()
                print("     first differing column", c, "(tile col", c // 128, "): rows", int(rows[0]), "..", int(rows[-1]), "count", int(rows.numel()), "tile rows", sorted(set((rows // 128).tolist()))[:20],
                      "max diff", float(dL[:, c].max()))
                tc = c // 128
                blk = dL[:, tc * 128:(tc + 1) * 128]
                trs = sorted(set(((blk > 0).any(dim=1).nonzero().flatten() // 128).tolist()))
                print("     tile rows differing in that tile column:", trs[:40])
            anomalies += 1
            continue
        if mode == "withbad" and i in bad:
            if it["kh"].is_inertia_correct(*ine):
                anomalies += 1; print(rnd, mode, i, "an indefinite instance was ACCEPTED", ine)
            continue
        red0 = M.get_stat("early_reject_redone")
        Lf, D = M.get_factor_device()
        Lt = torch.tril(Lf)
        dL = (Lt - ref[i][0]).abs(); nd = int((dL > 0).sum())
        if ine != (it["n"], 0, 0) or nd or not torch.equal(D, ref[i][1]) or M.get_stat("early_reject_redone") != red0 or M.get_stat("pp_fallbacks") != 0:
            anomalies += 1
            n = it["n"]; nt = (n + 127) // 128
            pad = torch.zeros(nt * 128, nt * 128, dtype=torch.bool, device=dL.device); pad[:n, :n] = dL > 0
            tiles = pad.view(nt, 128, nt, 128).permute(0, 2, 1, 3).reshape(nt, nt, -1).sum(-1).cpu().numpy()
            rows, cols = np.nonzero(tiles)
            print(rnd, mode, i, "inertia", ine, "differing entries", nd, "max", float(dL.max()), "redone now", M.get_stat("early_reject_redone") - red0,
                  "rejects", M.get_stat("early_rejects"), "fallbacks", M.get_stat("pp_fallbacks"), "site", M.get_stat("timeout_site"),
                  "| differing tiles", len(rows), "first (row, col)", (int(rows[0]), int(cols[0])) if len(rows) else None,
                  "tile cols", sorted(set(cols.tolist()))[:12], "tile rows", sorted(set(rows.tolist()))[:12])
            if 0 < len(rows) <= 8:
                for (tr, tc) in zip(rows.tolist(), cols.tolist()):
                    got = Lt[tr * 128:(tr + 1) * 128, tc * 128:(tc + 1) * 128]
                    dd = dL[tr * 128:(tr + 1) * 128, tc * 128:(tc + 1) * 128] > 0
                    rr = dd.any(dim=1).nonzero().flatten().tolist(); cc = dd.any(dim=0).nonzero().flatten().tolist()
                    close = [float((got - ref[k][0][tr * 128:(tr + 1) * 128, tc * 128:(tc + 1) * 128]).abs().max()) for k in range(len(ref))]
                    print(f"     tile ({tr},{tc}): rows {rr[0]}..{rr[-1]} ({len(rr)}) cols {cc[0]}..{cc[-1]} ({len(cc)}) of the tile differ; max|got - ref_k| for k=0..5: "
                          + " ".join(f"{c:.2e}" for c in close) + f"; nan {int(torch.isnan(got).sum())} zeros {int((got == 0).sum())} of {got.numel()}")
print(f"EARLY={int(early)}: {rounds} rounds, {anomalies} anomalies")
