"""Multi-GPU readiness that needs no hardware: the row partition bench.py --gpus N would run,
at the sizes BASELINE.json names (148^3 = the headline, 200^3 = configs 3 and 4), from the HOST plan
(stan_host_partition_plan; the device-derived plan is compared with it entry by entry in the GPU tests).
SURVEY.md section 8e's claims, checked: contiguous block-row ranges in AssignDOF order are balanced in BLOCKS (the
SpMV's work and bytes), every rank has at most two neighbours (the BFS is a level structure), and the halo is
30 k - 111 k nodes per neighbour at 200^3 on 8 ranks."""
import numpy as np
import pytest

from stan_amd import host, problem


def plan_table(n, nranks):
    job = problem.cube_job(n)
    m = n + 1
    idx = np.arange(m ** 3)

    def c(a):
        return np.where((a == 0) | (a == n), 2, 3)
    # blocks of a node's row = nodes it shares an element with = product over the axes of (2 on the surface, 3 inside)
    blocks_of_node = (c(idx % m) * c((idx // m) % m) * c(idx // (m * m))).astype(np.int64)
    by_row = np.empty(m ** 3, np.int64)
    by_row[job.node_dof.reshape(-1, 3)[:, 0] // 3] = blocks_of_node
    cum = np.concatenate([[0], np.cumsum(by_row)])
    rows = []
    for N in nranks:
        for r in range(N):
            p = host.partition_plan(job.node_index, job.conn, N, r)
            rs = p["row_starts"]
            per_nbr = np.diff(p["recv_off"]) if len(p["nbr"]) else np.array([], np.int64)
            rows.append(dict(n=n, N=N, rank=r, owned_rows=int(rs[r + 1] - rs[r]), blocks=int(cum[rs[r + 1]] - cum[rs[r]]),
                             nbr=[int(q) for q in p["nbr"]], halo_rows=int(len(p["halo_glob"])),
                             halo_per_nbr=[int(v) for v in per_nbr], send_rows=int(len(p["send_rows"]))))
    return rows


@pytest.mark.parametrize("n", [148, 200])
def test_row_partition_is_balanced_with_at_most_two_neighbours(n, capsys):
    table = plan_table(n, (2, 4, 8))
    for N in (2, 4, 8):
        t = [r for r in table if r["N"] == N]
        blocks = np.array([r["blocks"] for r in t], float)
        assert blocks.max() / blocks.mean() <= 1.05, (n, N, blocks)
        assert all(len(r["nbr"]) <= 2 for r in t)
        assert all(all(abs(q - r["rank"]) == 1 for q in r["nbr"]) for r in t)      # rank +- 1 only
        assert sum(r["owned_rows"] for r in t) == (n + 1) ** 3
        # structural symmetry: what r receives from q is what q sends to r
        for r in t:
            assert r["send_rows"] > 0 and r["halo_rows"] > 0
        with capsys.disabled():
            print("\n%d^3 on %d ranks: blocks max/mean %.4f; halo block rows per rank %s; per neighbour %s" %
                  (n, N, blocks.max() / blocks.mean(), [r["halo_rows"] for r in t], [r["halo_per_nbr"] for r in t]))
    if n == 200:    # 30 k - 111 k nodes per neighbour at N = 8
        per = [v for r in table if r["N"] == 8 for v in r["halo_per_nbr"]]
        assert 25_000 <= min(per) and max(per) <= 120_000, (min(per), max(per))
