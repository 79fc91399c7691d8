"""CPU, world_size 2 over gloo: the N>1 plumbing of SURVEY §8(e) — every rank owns a disjoint shard of the edges and
the only exchange is ONE all-reduce of the partial count (gms_amd/dist.py).  On the GPU box the per-rank partial comes
from gmsx_tc_partial(rank, world); here it comes from the oracle's sharded loop (test-only), which exercises exactly
the same reduce / wrap-around / max-time code."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gms_amd import capi, dist
    from oracle.bindings import Oracle
    r, lr, w = dist.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    csr = capi.HostCSR.generate("kronecker", 10, 16)
    off, ng = csr.offsets(), csr.neighbors()
    partial, edges, _ = Oracle().tc_total_sample(off, ng, world, rank, threads=1)  # vertices u = rank (mod world)
    total = dist.allreduce_count(partial)
    n_edges = dist.allreduce_count(edges)
    # sums that wrap mod 2^64 survive the int64 transport (k!*C_k uses the reference's size_t arithmetic)
    wrapped = dist.allreduce_count((1 << 63) + 5 + rank)
    slowest = dist.allreduce_max(1.0 + rank)
    dist.barrier()
    q.put((rank, partial, total, n_edges, wrapped, slowest))


@pytest.mark.timeout(300)
def test_two_rank_sharded_triangle_count_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    partials = [r[1] for r in res]
    assert partials[0] != partials[1] and sum(partials) == 3 * 74720          # disjoint shards, reference golden
    for _, _, total, n_edges, wrapped, slowest in res:
        assert total == 3 * 74720 and total // 3 == 74720
        assert n_edges == 10496
        assert wrapped == ((1 << 63) + 5 + (1 << 63) + 6) % (1 << 64) == 11
        assert slowest == 2.0
