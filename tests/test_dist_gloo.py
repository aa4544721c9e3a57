"""CPU, world_size 2 over gloo: the N>1 bootstrap and halo protocol, planned by libkmcfield on every
rank (host-only communicators), are mutually consistent: what rank r will send to q is exactly, and in
the same order, what q expects to receive from r (the exchange itself is RCCL send/recv on the GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ctypes as C
        import torch
        import kmcfield_amd as km
        import kmcf_oracle as O
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        S = km.solvers
        d = km.structure.load_device_5nm("init")
        NL = d["N_contact"]
        ks = O.KSystem(d["xyz"], d["lattice"], 0, 3.5, NL, NL)
        counts, displs = S.KMC_comm.partition(ks.n, world)
        lib = km.lib.load()
        h = C.c_void_p()
        km.lib.check(lib.kmcf_comm_create(C.byref(h), -1, world, rank), "comm")

        class _C:
            handle = h
        r0, nr = int(displs[rank]), int(counts[rank])
        rp = (ks.row_ptr[r0:r0 + nr + 1] - ks.row_ptr[r0]).astype(np.int32)
        col = ks.col[ks.row_ptr[r0]:ks.row_ptr[r0 + nr]]
        m = S.Distributed_matrix(_C, ks.n, counts, displs, col, rp, None)
        nb = m.neighbours()
        # global ids of what I send to / expect from each peer
        send = {n["rank"]: (n["rows"] + displs[rank]).tolist() for n in nb[1:]}
        recv = {n["rank"]: (n["cols"] + displs[n["rank"]]).tolist() for n in nb[1:]}
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(send=send, recv=recv))
        ok = True
        for peer, lst in send.items():
            ok &= gathered[peer]["recv"].get(rank) == lst
        for peer, lst in recv.items():
            ok &= gathered[peer]["send"].get(rank) == lst
        # the unique-id broadcast path of KMC_comm.connect (bytes tensor over the process group)
        t = torch.arange(256, dtype=torch.int16).to(torch.uint8) if rank == 0 else torch.zeros(256, dtype=torch.uint8)
        dist.broadcast(t, src=0)
        ok &= t.tolist() == [i % 256 for i in range(256)]
        # distributed solution emulation: every rank runs the oracle's P-rank PCG and agrees bit for bit
        A = O.assemble_K(ks, d["element"], O.update_charge(d["element"], np.zeros(d["N"], np.int32),
                         O.neighbor_list(d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2]), d["metals"]),
                         d["metals"], 1.0, 1e-8, 5.0, P=world)
        x, it, rel = O.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], 1e-14 * ks.n,
                                  10000, P=world)
        its = [None] * world
        dist.all_gather_object(its, (it, float(x.sum())))
        ok &= all(i == its[0] for i in its)
        m.close()
        lib.kmcf_comm_destroy(h)
        dist.destroy_process_group()
        q.put((rank, bool(ok), ""))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.timeout(300)
def test_two_rank_halo_protocol_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, ok, msg in res:
        assert ok, "rank %d: %s" % (rank, msg)
