"""world_size-2/3 CPU tests (gloo) of the sharded path's host logic: the row partition and
halo plan of libstan_host.so drive a numpy restatement of cg.hip's distributed iteration
(halo exchange of p, all-reduce of p.Ap and of {r.r, merit}, refresh every 10th iteration,
all-gather of the result), with the oracle's matrix as the operator.  The result must equal
the oracle's single-process CG.  (The GPU kernels are compared with the same plan in
tests/test_gpu_parity.py.)"""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unreduced(A_red, red):
    """K in the layout libstan_hip.so stores: all DOFs, identity on the fixed ones."""
    ndof = red.shape[0]
    free = np.nonzero(red != -1)[0]
    P = sp.csr_matrix((np.ones(free.shape[0]), (free, np.arange(free.shape[0]))),
                      shape=(ndof, A_red.shape[0]))
    fixed = (red == -1).astype(float)
    return (P @ A_red @ P.T + sp.diags(fixed)).tocsr()


def _worker(rank, world, port, n, eps, out_dir, loop="classic"):
    import torch.distributed as dist
    from oracle import pyoracle as O
    from stan_amd import host, problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    job = problem.cube_job(n, jitter=0.05)
    rc, A = O.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    Kfull = _unreduced(A.to_scipy_full(), job.red)
    plan = host.partition_plan(job.node_index, job.conn, world, rank)
    r0, r1 = plan["row_starts"][rank], plan["row_starts"][rank + 1]
    nloc, halo = r1 - r0, plan["halo_glob"]
    # local operator: owned rows, columns = [owned | halo] exactly as the device numbers them
    colmap = -np.ones(Kfull.shape[0] // 3, np.int64)
    colmap[r0:r1] = np.arange(nloc)
    colmap[halo] = nloc + np.arange(halo.shape[0])
    rows = Kfull[3 * r0:3 * r1].tocoo()
    assert (colmap[rows.col // 3] >= 0).all(), "a coupled column is neither owned nor in the halo"
    Aloc = sp.csr_matrix((rows.data, (rows.row, 3 * colmap[rows.col // 3] + rows.col % 3)),
                         shape=(3 * nloc, 3 * (nloc + halo.shape[0])))

    def exchange(v):  # v: [3*(nloc+nhalo)], fills the halo part
        reqs, bufs = [], []
        for i, q in enumerate(plan["nbr"]):
            srows = plan["send_rows"][plan["send_off"][i]:plan["send_off"][i + 1]]
            sb = torch.from_numpy(np.ascontiguousarray(v[:3 * nloc].reshape(-1, 3)[srows]))
            rb = torch.zeros((plan["recv_off"][i + 1] - plan["recv_off"][i], 3), dtype=torch.float64)
            reqs += [dist.isend(sb, int(q)), dist.irecv(rb, int(q))]
            bufs.append((i, rb, sb))
        for r in reqs:
            r.wait()
        for i, rb, _ in bufs:
            v[3 * (nloc + plan["recv_off"][i]):3 * (nloc + plan["recv_off"][i + 1])] = rb.numpy().ravel()

    ncoll = [0]

    def allsum(*vals):
        ncoll[0] += 1
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    # --- cg.hip's iteration, distributed ---
    dloc = Kfull.diagonal()[3 * r0:3 * r1]
    s = np.ones(3 * (nloc + halo.shape[0]))
    s[:3 * nloc] = np.where(dloc > 0, 1 / np.sqrt(np.where(dloc > 0, dloc, 1)), 1.0)
    exchange(s)
    Ah = sp.diags(s[:3 * nloc]) @ Aloc @ sp.diags(s)
    Ffull = np.zeros(job.n_dof)
    free = job.red != -1
    Ffull[free] = job.F
    bh = s[:3 * nloc] * Ffull[3 * r0:3 * r1]
    x = np.zeros(3 * (nloc + halo.shape[0]))
    p = np.zeros_like(x)
    r = bh.copy()
    p[:3 * nloc] = r
    (b2,) = allsum(float(bh @ bh))
    bnorm, rho, prevmf = np.sqrt(b2), b2, 0.0
    its, term = 0, 0
    if loop == "single":
        # cg.hip's single-reduction loop (k_vec_sr + k_spmv<.., 2, ..>, STAN_OPT_CG_SINGLE_REDUCE):
        # ONE all-reduce of {r.r, r.Ar, merit} per iteration; iteration k first takes the decisions
        # of iteration k-1
        ncoll[0] = 0
        rg = np.zeros_like(x); rg[:3 * nloc] = r                 # r is the gathered vector here
        sv = np.zeros(3 * nloc); p[:] = 0.0
        exchange(rg)
        w = Ah @ rg
        gamma, delta, merit = allsum(float(r @ r), float(r @ w), 0.0)
        gprev = aprev = 0.0
        xprev = x.copy()
        k = 0
        while term == 0 and bnorm > 0:
            k += 1
            if k > 1:
                if np.sqrt(gamma) <= eps * bnorm:
                    term, its = 1, k - 1
                    break
                if merit >= prevmf:
                    term, its, x = 7, k - 1, xprev
                    break
            beta = 0.0 if k == 1 else gamma / gprev
            pap = delta if k == 1 else delta - beta * gamma / aprev
            alpha = gamma / pap
            gprev, aprev, prevmf = gamma, alpha, merit
            p[:3 * nloc] = rg[:3 * nloc] + beta * p[:3 * nloc]
            sv = w + beta * sv
            xprev = x.copy()
            x[:3 * nloc] = x[:3 * nloc] + alpha * p[:3 * nloc]
            if k % 10 == 0:
                exchange(x)
                mv = Ah @ x
                rg[:3 * nloc] = bh - mv
                mloc = float((mv - 2 * bh) @ x[:3 * nloc])
            else:
                rg[:3 * nloc] = rg[:3 * nloc] - alpha * sv
                mloc = float(-(rg[:3 * nloc] + bh) @ x[:3 * nloc])
            exchange(rg)
            w = Ah @ rg
            gamma, delta, merit = allsum(float(rg[:3 * nloc] @ rg[:3 * nloc]), float(rg[:3 * nloc] @ w), mloc)
        assert ncoll[0] == k, "one all-reduce per iteration (plus the initial one)"
    while loop == "classic" and term == 0 and bnorm > 0:
        its += 1
        exchange(p)
        v = Ah @ p
        (vmv,) = allsum(float(p[:3 * nloc] @ v))
        alpha = rho / vmv
        cx = x.copy()
        cx[:3 * nloc] = x[:3 * nloc] + alpha * p[:3 * nloc]
        if its % 10 == 0:
            exchange(cx)
            mv = Ah @ cx
            cr = bh - mv
            r2, mf = allsum(float(cr @ cr), float((mv - 2 * bh) @ cx[:3 * nloc]))
        else:
            cr = r - alpha * v
            r2, mf = allsum(float(cr @ cr), float(-(cr + bh) @ cx[:3 * nloc]))
        if np.sqrt(r2) <= eps * bnorm:
            x, term = cx, 1
        elif mf >= prevmf:
            term = 7
        else:
            x, r, prevmf = cx, cr, mf
            p[:3 * nloc] = cr + (r2 / rho) * p[:3 * nloc]
            rho = r2
    # all-gather of the owned rows
    full = torch.zeros(job.n_dof, dtype=torch.float64)
    full[3 * r0:3 * r1] = torch.from_numpy(s[:3 * nloc] * x[:3 * nloc])
    dist.all_reduce(full)
    U = full.numpy()[free]
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), U=U, its=its, term=term)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,loop", [(2, 6, "classic"), (3, 7, "classic"), (2, 6, "single"), (3, 7, "single")])
def test_sharded_cg_matches_single_process_oracle(oracle, built_libs, tmp_path, world, n, loop):
    """loop = "single": the Chronopoulos-Gear form of cg.hip (one all-reduce per iteration) gives the
    oracle's answer with the oracle's iteration count up to rounding."""
    import torch.multiprocessing as mp
    from stan_amd import problem
    # reached before the type-7 rounding floor, so the counts are comparable (the single-reduction
    # recurrences sit on that floor a little earlier: alpha comes from a three-term formula)
    eps = 1e-7 if loop == "classic" else 1e-6
    mp.spawn(_worker, args=(world, _free_port(), n, eps, str(tmp_path), loop), nprocs=world, join=True)
    job = problem.cube_job(n, jitter=0.05)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                            job.mat_E_nu, job.red)
    Uo, rep = oracle.cg(A, job.F, eps)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert int(d["term"]) == rep["terminationtype"]
        assert abs(int(d["its"]) - rep["iterations"]) <= max(3, rep["iterations"] // 20)
        assert np.abs(d["U"] - Uo).max() <= 1e-5 * np.abs(Uo).max()  # two eps=1e-7 solves
        if r:
            assert np.array_equal(d["U"], np.load(os.path.join(str(tmp_path), "r0.npz"))["U"])


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_plan_is_symmetric_and_complete(built_libs, nranks):
    """What r sends to q is exactly what q expects from r, in the same (global) order."""
    from stan_amd import host, problem
    job = problem.cube_job(9, jitter=0.0)
    plans = [host.partition_plan(job.node_index, job.conn, nranks, r) for r in range(nranks)]
    rs = plans[0]["row_starts"]
    assert rs[0] == 0 and rs[-1] == job.xyz.shape[0] and all(x % 64 == 0 for x in rs[:-1])
    for r, p in enumerate(plans):
        assert np.array_equal(p["row_starts"], rs)
        assert np.all(np.diff(p["halo_glob"]) > 0)
        for i, q in enumerate(p["nbr"]):
            sent = rs[r] + p["send_rows"][p["send_off"][i]:p["send_off"][i + 1]]
            pq = plans[q]
            j = list(pq["nbr"]).index(r)
            expected = pq["halo_glob"][pq["recv_off"][j]:pq["recv_off"][j + 1]]
            assert np.array_equal(sent, expected)
    # BFS level structure: a range only couples to ranges a few levels away -- the ones next
    # to it once a range is thicker than a BFS level (3 ranks here), a band of +-2 when it is not
    reach = 1 if nranks <= 3 else 2
    assert all(abs(int(q) - r) <= reach for r, p in enumerate(plans) for q in p["nbr"])


def test_fuzz_plans_are_symmetric(built_libs):
    """Halo plans of random shuffled meshes (tests/fuzz.py), 2..7 ranks incl. empty ranks and up
    to four neighbours: what r sends to q is what q expects from r; every row owned once."""
    from stan_amd import host
    from tests import fuzz
    worst_nbrs = 0
    for seed in range(100, 160):
        job = fuzz.random_job(seed)
        if job is None:
            continue
        nr = 2 + seed % 6
        plans = [host.partition_plan(job.node_index, job.conn, nr, r) for r in range(nr)]
        rs = plans[0]["row_starts"]
        assert rs[0] == 0 and rs[-1] == job.xyz.shape[0]
        for r, p in enumerate(plans):
            worst_nbrs = max(worst_nbrs, len(p["nbr"]))
            assert np.all(np.diff(p["halo_glob"]) > 0)
            assert not np.any((p["halo_glob"] >= rs[r]) & (p["halo_glob"] < rs[r + 1]))
            for i, q in enumerate(p["nbr"]):
                sent = rs[r] + p["send_rows"][p["send_off"][i]:p["send_off"][i + 1]]
                pq = plans[q]
                j = list(pq["nbr"]).index(r)
                assert np.array_equal(sent, pq["halo_glob"][pq["recv_off"][j]:pq["recv_off"][j + 1]])
    assert worst_nbrs >= 4
