"""Dense S stage of the Schur-complement KKT system (reference src/KKT/Schur/schur.jl:927-1058).

CPU: the oracle restatement is pinned to the block-arrow identity it must satisfy (solving through S equals a dense
solve of the assembled two-stage KKT matrix, the property the reference's own Schur tests rely on:
`test/madnlp_schur.jl` compares SchurComplementKKTSystem with the monolithic SparseKKTSystem solution), and the
scenario sharding with its two collectives is exercised with gloo, world size 2.
GPU: the HIP stage (through the C ABI) against the oracle on the same seeded blocks."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.schur import SchurDenseStage as OracleStage  # noqa: E402


def two_stage_blocks(ns, nv, nc, nd, seed=0, delta=1e-8):
    """Scenario blocks A_k = [[H_k + Sigma_k, J_k'], [J_k, -delta I]] (symmetric indefinite, quasi-definite in this
    order), coupling C_dk = [H_dk, J_dk'] (nd x blk), design block S0 = H_dd + Sigma_d (SPD)."""
    rng = np.random.default_rng(seed)
    blk = nv + nc
    A, Cs = [], []
    for _ in range(ns):
        R = rng.standard_normal((nv, nv))
        H = R @ R.T / nv + np.diag(10.0 ** rng.uniform(-3, 3, nv))
        J = rng.standard_normal((nc, nv))
        Ak = np.zeros((blk, blk))
        Ak[:nv, :nv] = H
        Ak[nv:, :nv] = J
        Ak[:nv, nv:] = J.T
        Ak[nv:, nv:] = -delta * np.eye(nc)
        A.append(np.asfortranarray(Ak))
        Cs.append(np.asfortranarray(rng.standard_normal((nd, blk)) * 0.3))
    # S0 chosen so that the Schur complement S = S0 - sum_k C_dk A_k^-1 C_dk' equals an SPD matrix G by construction
    # (random couplings to the multiplier part of an indefinite A_k would otherwise make S indefinite)
    R = rng.standard_normal((nd, nd))
    G = R @ R.T / nd + np.diag(10.0 ** rng.uniform(0, 2, nd))
    S0 = G + sum(Cs[k] @ np.linalg.solve(A[k], Cs[k].T) for k in range(ns))
    S0 = np.asfortranarray((S0 + S0.T) / 2)
    return A, Cs, S0, blk


def assemble(A, Cs, S0):
    ns, blk, nd = len(A), A[0].shape[0], S0.shape[0]
    N = ns * blk + nd
    K = np.zeros((N, N))
    for k in range(ns):
        K[k * blk:(k + 1) * blk, k * blk:(k + 1) * blk] = A[k]
        K[ns * blk:, k * blk:(k + 1) * blk] = Cs[k]
        K[k * blk:(k + 1) * blk, ns * blk:] = Cs[k].T
    K[ns * blk:, ns * blk:] = S0
    return K


@pytest.mark.parametrize("ns,nv,nc,nd", [(1, 5, 2, 3), (4, 12, 5, 7), (3, 40, 24, 30)])
def test_oracle_schur_stage_equals_block_arrow_solve(ns, nv, nc, nd):
    A, Cs, S0, blk = two_stage_blocks(ns, nv, nc, nd, seed=ns + nd)
    st = OracleStage(A, Cs, S0)
    S = st.build_local()
    # S is the Schur complement of the assembled matrix
    K = assemble(A, Cs, S0)
    n1 = ns * blk
    S_ref = K[n1:, n1:] - K[n1:, :n1] @ np.linalg.solve(K[:n1, :n1], K[:n1, n1:])
    np.testing.assert_allclose(S, S_ref, rtol=0, atol=1e-9 * np.abs(S_ref).max())
    assert st.factorize(S) == (nd, 0, 0)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(n1 + nd)
    rk = b[:n1].reshape(ns, blk).copy()
    rd = b[n1:].copy()
    rd += st.forward(rk)
    st.solve_s(rd)
    st.backward(rk, rd)
    x = np.concatenate([rk.ravel(), rd])
    assert np.abs(K @ x - b).max() <= 1e-8 * (np.abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    # sharding: the ranks' contributions add up to S (S0 lives on rank 0)
    parts = []
    for rank in range(2):
        own = list(range(rank, ns, 2))
        loc = OracleStage([A[k] for k in own], [Cs[k] for k in own], S0)
        parts.append(loc.build_local(with_s0=(rank == 0)))
    np.testing.assert_allclose(parts[0] + parts[1], S, rtol=0, atol=1e-12 * np.abs(S).max())


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from oracle.schur import SchurDenseStage
from tests.test_schur import two_stage_blocks, assemble
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
ns, nv, nc, nd = 5, 10, 4, 6
A, Cs, S0, blk = two_stage_blocks(ns, nv, nc, nd, seed=3)
own = list(range(rank, ns, world))               # madnlp_jl_amd.schur.shard
st = SchurDenseStage([A[k] for k in own], [Cs[k] for k in own], S0)
S = torch.from_numpy(st.build_local(with_s0=(rank == 0)).ravel(order="F").copy())
dist.all_reduce(S)                               # collective 1: nd^2 doubles
S = S.numpy().reshape((nd, nd), order="F")
assert st.factorize(S) == (nd, 0, 0)
rng = np.random.default_rng(7)
b = rng.standard_normal(ns * blk + nd)
rk = np.stack([b[k * blk:(k + 1) * blk] for k in own]).copy()
contrib = torch.from_numpy(st.forward(rk))
dist.all_reduce(contrib)                         # collective 2: nd doubles
rd = b[ns * blk:] + contrib.numpy()
st.solve_s(rd)
st.backward(rk, rd)
K = assemble(A, Cs, S0)
xfull = np.linalg.solve(K, b)
for i, k in enumerate(own):
    assert np.abs(rk[i] - xfull[k * blk:(k + 1) * blk]).max() <= 1e-7 * np.abs(xfull).max()
assert np.abs(rd - xfull[ns * blk:]).max() <= 1e-7 * np.abs(xfull).max()
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_schur_sharding_two_ranks_gloo(tmp_path):
    """Scenarios sharded round-robin over 2 ranks, S0 on rank 0: one all-reduce of S, one of the design right-hand
    side; every rank ends with its scenarios' part of the monolithic solution."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert res.stdout.count("RANK_OK") == 2


# --------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("ns,nv,nc,nd", [(1, 5, 2, 3), (4, 40, 24, 30), (3, 150, 50, 100), (6, 300, 84, 64)])
def test_hip_schur_stage_matches_oracle(ns, nv, nc, nd):
    """S within 1e-11 |S| of the oracle's (two different factorizations of the indefinite blocks), inertia equal,
    solution of the assembled block-arrow system with backward error <= 1e-11, and two handles holding the even / odd
    scenarios (= two ranks) add up to the single-handle S."""
    torch = pytest.importorskip("torch")
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.schur import SchurDenseStage, shard
    A, Cs, S0, blk = two_stage_blocks(ns, nv, nc, nd, seed=ns * 7 + nd)
    ctx = mj.HipContext(0)
    so = OracleStage(A, Cs, S0)
    S_o = so.build_local()
    sh = SchurDenseStage(A, Cs, S0, nd, blk, ctx=ctx)
    S_h = sh.build_kkt().cpu().numpy().reshape((nd, nd), order="F")
    assert np.abs(S_h - S_o).max() <= 1e-11 * np.abs(S_o).max()
    for k in range(ns):
        assert sh.scenario_inertia(k) == (nv, 0, nc)
    sh.factorize_kkt()
    assert sh.inertia() == so.factorize(S_o) == (nd, 0, 0) and sh.is_inertia_correct(*sh.inertia())
    rng = np.random.default_rng(5)
    n1 = ns * blk
    b = rng.standard_normal(n1 + nd)
    rk = torch.from_numpy(b[:n1].reshape(ns, blk).copy()).cuda()
    rd = torch.from_numpy(b[n1:].copy()).cuda()
    sh.solve(rk, rd)
    x = np.concatenate([rk.cpu().numpy().ravel(), rd.cpu().numpy()])
    K = assemble(A, Cs, S0)
    res = np.abs(K @ x - b).max() / (np.abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-11, res
    # oracle solve agrees (conditioning-limited forward agreement)
    rko, rdo = b[:n1].reshape(ns, blk).copy(), b[n1:].copy()
    rdo += so.forward(rko); so.solve_s(rdo); so.backward(rko, rdo)
    xo = np.concatenate([rko.ravel(), rdo])
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    # two "ranks" on one GPU
    parts = []
    for rank in range(2):
        own = shard(ns, rank, 2)
        loc = SchurDenseStage([A[k] for k in own], [Cs[k] for k in own], S0 if rank == 0 else None, nd, blk, ctx=ctx)
        parts.append(loc.build_kkt().cpu().numpy())
        loc.close()
    assert np.abs((parts[0] + parts[1]).reshape((nd, nd), order="F") - S_h).max() <= 1e-12 * np.abs(S_h).max()
    sh.close()
    ctx.close()
