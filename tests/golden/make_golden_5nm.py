#!/usr/bin/env python3
"""Generate tests/golden/device_5nm.npz from the reference's shipped 5 nm example.

Run in the build container only (needs /root/reference, which does not travel to
the GPU box).  The output is DATA: site coordinates, element codes and the
expected-output potential column of the reference's only golden result
(structures/5nm_device/expected_output/Results_5.000000/snapshot_{init,6}.xyz,
written by Device::writeSnapshot, src/Device.cpp:214-232).  No reference source
text is stored.

    python tests/golden/make_golden_5nm.py
"""
import os
import sys

import numpy as np

REF = os.environ.get("KMCF_REFERENCE", "/root/reference")
S5 = os.path.join(REF, "structures", "5nm_device")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "device_5nm.npz")

# ELEMENT enum order, src/utils.h:37-44 ; strings src/utils.cpp:7-29
ELEMENT = {"d": 0, "Od": 1, "V": 2, "O": 3, "Hf": 4, "Ni": 5, "Ti": 6, "Pt": 7, "N": 8}


def read_xyz(path, ncols):
    el, num = [], []
    with open(path) as f:
        n = int(f.readline())
        f.readline()
        for line in f:
            t = line.split()
            if not t:
                continue
            el.append(ELEMENT[t[0]])
            num.append([float(v) for v in t[1:1 + ncols]])
    assert len(el) == n, (len(el), n)
    return np.asarray(el, dtype=np.int8), np.asarray(num, dtype=np.float64)


def main():
    el_file, xyz = read_xyz(os.path.join(S5, "reordered_device_5.xyz"), 3)
    res = os.path.join(S5, "expected_output", "Results_5.000000")
    el_init, cols_init = read_xyz(os.path.join(res, "snapshot_init.xyz"), 5)
    el_s6, cols_s6 = read_xyz(os.path.join(res, "snapshot_6.xyz"), 5)
    # expected_output/output1_0.txt: "KMC time is: ..." after each of the 6 KMC steps of the 5 V bias point
    kmc_times = []
    with open(os.path.join(S5, "expected_output", "output1_0.txt")) as f:
        for line in f:
            if line.startswith("KMC time is:"):
                kmc_times.append(float(line.split(":")[1]))
    assert len(kmc_times) == 6, kmc_times
    assert np.abs(cols_init[:, :3] - xyz).max() < 1e-4
    assert np.abs(cols_s6[:, :3] - xyz).max() < 1e-4
    np.savez_compressed(
        OUT,
        xyz=xyz,                      # full-precision coordinates [Angstrom]
        element_file=el_file,         # elements in reordered_device_5.xyz
        element_init=el_init,         # after makeSubstoichiometric (400 V)
        element_snap6=el_s6,          # after 6 KMC steps
        potential_snap6=cols_s6[:, 3],  # site_potential_charge after sum_and_gather, 6 sig. digits
        power_snap6=cols_s6[:, 4],
        potential_init=cols_init[:, 3],
        kmc_times=np.asarray(kmc_times),   # cumulative KMC time after steps 1..6 (t_switch = 1e-12 s ends the loop)
        t_switch=np.float64(1e-12),
        attempt_frequency=np.float64(10e13),
        background_temp=np.float64(300.0),
        # structures/5nm_device/parameters.txt
        lattice=np.array([108.984220, 51.150000, 51.150000]),
        Vd=np.float64(5.0),
        nn_dist=np.float64(3.5),
        num_atoms_first_layer=np.int32(576),
        num_atoms_contact=np.int32(5760),
        metals=np.array([ELEMENT["Ti"], ELEMENT["N"]], dtype=np.int32),
        sigma=np.float64(3.5e-10),
        epsilon=np.float64(23.0),
        pbc=np.int32(0),
    )
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
