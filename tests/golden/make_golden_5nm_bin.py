#!/usr/bin/env python3
"""Flat binary of the 5 nm fixture for examples/kmc_loop.cpp (a C++ program cannot read the npz):
int32 N | N x 3 float64 coordinates | N x int32 ELEMENT codes after makeSubstoichiometric (snapshot_init.xyz).
Derived from tests/golden/device_5nm.npz (itself data of the reference's structures/5nm_device/, see
make_golden_5nm.py); needs neither the reference nor a GPU.

    python tests/golden/make_golden_5nm_bin.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.load(os.path.join(HERE, "device_5nm.npz"))
    out = os.path.join(HERE, "device_5nm.bin")
    with open(out, "wb") as f:
        np.int32(len(g["element_init"])).tofile(f)
        np.ascontiguousarray(g["xyz"], np.float64).tofile(f)
        g["element_init"].astype(np.int32).tofile(f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
