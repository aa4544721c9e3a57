"""One rank of tests/test_gpu_p2p_processes.py: a separate PROCESS (not a thread) that joins a 2..4-rank solver
group on ONE GPU over the peer-to-peer transport alone -- real hipIpc handles, exchanged by the host program through
torch.distributed (gloo), no RCCL (which refuses several ranks on one device).  Runs the K path of the reference's
5 nm device on its row block and stores what the parent compares with the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    import kmcfield_amd as km
    S = km.solvers
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d = km.structure.load_device_5nm("init")
    NL = d["N_contact"]
    n_if = d["N"] - 2 * NL
    comm = S.KMC_comm(n_if, d["N"] + 1, d["N"], d["N"], rank=rank, size=world, device=0)
    comm.connect_p2p(dist)
    assert comm.transport() == "p2p", comm.transport()
    buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                        buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
    st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"], d["nn_dist"],
                                           len(d["metals"]), 0)
    S.sum_and_gather_potential(buf, NL, comm)
    mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
    # the pieces of one distributed iteration on this transport (both ranks share one GPU here: indicative only)
    diag = {}
    for kind, key in ((0, "allreduce3_us"), (1, "halo_exchange_us"), (2, "spmv_kernels_us")):
        mat.comm_bench(kind, 5)
        diag[key] = round(mat.comm_bench(kind, 100) * 1e3 / 100, 2)
    np.save(os.path.join(out_dir, "charge_%d.npy" % rank), buf.site_charge.cpu().numpy())
    np.save(os.path.join(out_dir, "v_%d.npy" % rank), buf.site_potential_boundary.cpu().numpy())
    json.dump(dict(st=st, diag=diag, info=mat.info(), transport=comm.transport()), open(os.path.join(out_dir, "st_%d.json" % rank), "w"))
    buf.freeGPUmemory()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
