"""One rank of tests/test_gpu_p2p_processes.py: a separate PROCESS (not a thread) that joins a 2..4-rank solver
group on ONE GPU over the peer-to-peer transport alone -- real hipIpc handles, exchanged by the host program through
torch.distributed (gloo), no RCCL (which refuses several ranks on one device).  Runs the K path of the reference's
5 nm device on its row block and stores what the parent compares with the oracle; with KMCF_WORKER_T=1 the current solve
(T path) as well."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


Q = 1.60217663e-19
G0 = 2 * 3.8612e-5 * 1e-5


def run_T(km, torch, comm, d, charge):
    """The current solve of the 5 nm device on this rank of `comm` (a smooth band edge stands in for the CB-edge solve, which
    is a one-rank solve as in the reference); the storage of the tunnel block is the environment's (KMCF_SUB_DENSE).
    Returns what the ranks must agree on."""
    S = km.solvers
    NL, xyz, el = d["N_contact"], d["xyz"], d["element"]
    N = len(el)
    na = int(np.isin(el, [0, 1], invert=True).sum())
    comm.counts_T, comm.displs_T = comm.partition(na + 1, comm.size_T)
    buf = S.GPUBuffers(N, el, xyz[:, 0], xyz[:, 1], xyz[:, 2], 52, d["sigma"], d["k"], d["lattice"], d["metals"])
    buf.site_charge.copy_(torch.as_tensor(np.asarray(charge, np.int32)))
    buf.site_CB_edge = torch.as_tensor(Q * d["Vd"] * (0.5 - np.clip(xyz[:, 0] / 52.0, 0, 1)), device="cuda")
    high_G, low_G, loop_G, tol, m_e, V0 = 1e5 * d["high_G"], d["low_G"], 1e7 * d["high_G"], Q * 0.01, 0.85 * 9.11e-31, 1.6
    S.initialize_sparsity_T(buf, 0, d["nn_dist"], NL, NL, 10, comm)
    S.t_assemble(buf, S.current_params(d["Vd"], high_G, low_G, loop_G, G0, tol, m_e, V0))
    info = S.t_info(buf)
    buf.atom_virtual_potentials.zero_()
    buf.site_power.zero_()
    im, st = S.update_power_gpu_sparse_dist(buf, NL, NL, 10, d["Vd"], high_G, low_G, loop_G, G0, tol, d["nn_dist"], m_e, V0,
                                            len(d["metals"]), True, False, 1.0, cg_tolerance=1e-13, cg_max_iterations=20000)
    out = dict(im=im, st=st, info=info, v=buf.atom_virtual_potentials.cpu().numpy().copy(), pw=buf.site_power.cpu().numpy().copy())
    buf.freeGPUmemory()
    return out


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
    charge = buf.site_charge.cpu().numpy().copy()
    buf.freeGPUmemory()
    if os.environ.get("KMCF_WORKER_T") == "1":                      # the current solve too (the tunnel block's storage: KMCF_SUB_DENSE)
        t = run_T(km, torch, comm, d, charge)
        np.save(os.path.join(out_dir, "t_v_%d.npy" % rank), t["v"])
        np.save(os.path.join(out_dir, "t_pw_%d.npy" % rank), t["pw"])
        json.dump(dict(im=t["im"], st=t["st"], info=t["info"]), open(os.path.join(out_dir, "t_st_%d.json" % rank), "w"))
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
