"""Device structures for tests and the benchmark.

* load_device_5nm(): the reference's shipped 5 nm example, from the committed
  fixture tests/golden/device_5nm.npz (made from structures/5nm_device/ by
  tests/golden/make_golden_5nm.py).
* synth_crossbar_40nm(): SYNTHETIC stand-in for the 40 nm crossbar whose xyz
  files are missing from the reference checkout (.MISSING_LARGE_BLOBS): the 5 nm
  cell tiled 8x8 in y,z (lattice 108.98 x 409.2 x 409.2 A vs the reference's
  108.98 x 409.6 x 409.6, structures/40nm_crossbar/parameters.txt:12) and carved
  into word lines / bit lines so that the interface matrix has about the shape the
  authors benchmarked (1 632 355 rows, 41 208 963 nnz,
  dist_iterative_test/main_test_cg.cpp:197-201).  Site order "bwmin": contacts'
  outer layers first/last, everything else sorted along y (the reference runs the
  40 nm device from crossbar_40_bwmin.xyz).
* xyz / parameters.txt readers for users who have their own structure files
  (format of src/utils.cpp:72-97 and src/input_parser.cpp).
"""
import os

import numpy as np

# ELEMENT enum, src/utils.h:37-44
DEFECT, OXYGEN_DEFECT, VACANCY, O_EL, Hf_EL, Ni_EL, Ti_EL, Pt_EL, N_EL, NULL_ELEMENT = range(10)
ELEMENT_OF = {"d": DEFECT, "Od": OXYGEN_DEFECT, "V": VACANCY, "O": O_EL, "Hf": Hf_EL, "Ni": Ni_EL,
              "Ti": Ti_EL, "Pt": Pt_EL, "N": N_EL}

# KMC layers of the shipped configuration: compile-time globals of src/structure_input.h:7-48
# (type, E_gen_0, E_rec_1, E_diff_2, E_diff_3 [eV], start_x, end_x [A]); rnd_seed_kmc = 1 (:5)
LAYERS = [
    dict(type="contact", E_gen_0=0.0, E_rec_1=0.0, E_diff_2=0.0, E_diff_3=0.76, start_x=-22.0, end_x=0.0),
    dict(type="interface", E_gen_0=3.93, E_rec_1=0.0, E_diff_2=1.09, E_diff_3=0.76, start_x=0.0, end_x=3.0),
    dict(type="oxide", E_gen_0=3.93, E_rec_1=0.0, E_diff_2=1.09, E_diff_3=0.76, start_x=3.0, end_x=48.1431),
    dict(type="interface", E_gen_0=1.66, E_rec_1=0.0, E_diff_2=1.09, E_diff_3=0.76, start_x=48.1431, end_x=52.6431),
    dict(type="contact", E_gen_0=1.73, E_rec_1=0.0, E_diff_2=0.0, E_diff_3=2.8, start_x=52.6431, end_x=90.0),
]
RND_SEED_KMC = 1

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_5NM = os.path.join(_ROOT, "tests", "golden", "device_5nm.npz")


def read_xyz(path):
    """element codes + coordinates of an xyz file (src/utils.cpp:72-97)."""
    el, xyz = [], []
    with open(path) as f:
        n = int(f.readline())
        f.readline()
        for line in f:
            t = line.split()
            if not t:
                continue
            el.append(ELEMENT_OF[t[0]])
            xyz.append([float(t[1]), float(t[2]), float(t[3])])
    assert len(el) == n
    return np.asarray(el, np.int32), np.asarray(xyz, np.float64)


def read_parameters(path):
    """key = value pairs of a parameters.txt ('//' comments; src/input_parser.cpp)."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.split("//")[0].strip()
            if "=" not in line:
                continue
            k, v = line.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def load_csr_binaries(directory, matrix_size, nnz, suffix=""):
    """The authors' benchmark matrix format (dist_iterative_test/utils.cpp:25-56, main_test_cg.cpp:
    140-170): raw host-endian arrays A_data<suffix>.bin (f64), A_row_ptr<suffix>.bin,
    A_col_indices<suffix>.bin (i32), A_rhs<suffix>.bin and optionally solution<suffix>.bin (f64)."""
    def rd(name, dtype, count, required=True):
        path = os.path.join(directory, name + suffix + ".bin")
        if not os.path.exists(path):
            if required:
                raise FileNotFoundError(path)
            return None
        a = np.fromfile(path, dtype=dtype, count=count)
        if len(a) != count:
            raise ValueError("%s: expected %d entries, found %d" % (path, count, len(a)))
        return a
    return dict(data=rd("A_data", np.float64, nnz), row_ptr=rd("A_row_ptr", np.int32, matrix_size + 1),
                col_indices=rd("A_col_indices", np.int32, nnz), rhs=rd("A_rhs", np.float64, matrix_size),
                solution=rd("solution", np.float64, matrix_size, required=False))


def load_device_5nm(state="init"):
    """dict(xyz, element, lattice, N, N_contact, metals, Vd, nn_dist, ...) of the 5 nm example.
    state: 'file' (as in reordered_device_5.xyz), 'init' (after makeSubstoichiometric: 400 V)
    or 'snap6' (after 6 KMC steps)."""
    g = np.load(GOLDEN_5NM)
    el = g["element_" + state].astype(np.int32)
    return dict(xyz=g["xyz"].copy(), element=el, lattice=g["lattice"].copy(), N=len(el),
                N_contact=int(g["num_atoms_first_layer"]), metals=g["metals"].astype(np.int32),
                Vd=float(g["Vd"]), nn_dist=float(g["nn_dist"]), pbc=int(g["pbc"]), sigma=float(g["sigma"]),
                k=8.987552e9 / float(g["epsilon"]), high_G=1.0, low_G=1e-8,   # src/input_parser.cpp:391-394
                potential_snap6=g["potential_snap6"].copy(), element_snap6=g["element_snap6"].astype(np.int32),
                kmc_times=g["kmc_times"].copy(), t_switch=float(g["t_switch"]), freq=float(g["attempt_frequency"]),
                T_bg=float(g["background_temp"]), name="5nm_device")


def _stripes(coord, length, n_lines, fill):
    """mask of `n_lines` equally spaced stripes covering the fraction `fill` of [0, length)."""
    period = length / n_lines
    return (coord % period) < fill * period


def synth_crossbar_40nm(tiles=8, fill=0.52, n_lines=2, vacancy_fraction=0.05, seed=40, carve=True, order="bwmin", filament=None):
    """Synthetic 40 nm crossbar (see module docstring).  Deterministic for a given seed
    (numpy PCG64); with the defaults: ~1.63e6 interface rows, ~25 nnz/row.

    tiles: the 5 nm cell is repeated tiles x tiles in (y, z), period 51.15 A.
    filament: radius in Angstrom (None: none) of a conductive filament at the first crossing of a word line and a bit
      line: every oxygen site inside a cylinder along x becomes a vacancy.  Such vacancies have >= 2 vacancy neighbours,
      so the charge rule leaves them uncharged (src/potential_solver_gpu.cu:12-63) and neighbouring ones are joined by
      high_G (populate_T_dist, src/current_solver_gpu.cu:1139-1243): a path from electrode to electrode, which the
      random 5 % of vacancies alone does not form (its macroscopic current is zero +- the solver's residual).
    carve: bottom electrode (x below the oxide) kept on `n_lines` word lines (stripes in z,
      running along y), top electrode + Ti reservoir kept on `n_lines` bit lines (stripes in
      y), oxide and interstitial sites kept under either set of lines.
    """
    base = load_device_5nm("file")
    xyz0, el0 = base["xyz"], base["element"]
    period = 51.15
    x = xyz0[:, 0]
    x_left_layer, x_right_layer = x.min(), x.max()
    # regions of the 5 nm stack along x (structures/5nm_device): TiN | HfO2 (+d) | Ti reservoir | TiN
    n_c = 5760
    ox_lo = x[:n_c].max() + 1e-6                 # end of the left contact block (file order)
    right_block_lo = x[-n_c:].min() - 1e-6
    parts_xyz, parts_el = [], []
    for ty in range(tiles):
        for tz in range(tiles):
            s = xyz0.copy()
            s[:, 1] += ty * period
            s[:, 2] += tz * period
            parts_xyz.append(s)
            parts_el.append(el0)
    xyz = np.concatenate(parts_xyz)
    el = np.concatenate(parts_el).astype(np.int32)
    L = tiles * period
    if carve:
        xx, yy, zz = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        word = _stripes(zz, L, n_lines, fill)     # bottom electrode lines, along y
        bit = _stripes(yy, L, n_lines, fill)      # top electrode lines, along z
        is_left = xx < ox_lo
        is_right = xx > right_block_lo
        is_res = (el == Ti_EL) & ~is_left & ~is_right
        keep = np.where(is_left, word, np.where(is_right | is_res, bit, word | bit))
        xyz, el = xyz[keep], el[keep]
    # substoichiometric oxide: a fixed fraction of O -> V (Device::makeSubstoichiometric, src/Device.cpp)
    rng = np.random.Generator(np.random.PCG64(seed))
    o_idx = np.flatnonzero(el == O_EL)
    pick = rng.choice(o_idx, size=int(round(vacancy_fraction * len(o_idx))), replace=False)
    el[pick] = VACANCY
    if filament:
        half = 0.5 * fill * L / n_lines            # centre of the first stripe of either set of lines
        inside = (xyz[:, 1] - half) ** 2 + (xyz[:, 2] - half) ** 2 <= float(filament) ** 2
        el[inside & (el == O_EL)] = VACANCY
    # site order: outer contact layers first / last (they are the Dirichlet boundary), rest by `order`
    xx = xyz[:, 0]
    left = np.flatnonzero(np.abs(xx - x_left_layer) < 1e-6)
    right = np.flatnonzero(np.abs(xx - x_right_layer) < 1e-6)
    mid_mask = np.ones(len(el), bool)
    mid_mask[left] = False
    mid_mask[right] = False
    mid = np.flatnonzero(mid_mask)
    if order == "bwmin":
        mid = mid[np.lexsort((xyz[mid, 0], xyz[mid, 2], xyz[mid, 1]))]   # y, then z, then x
        left = left[np.lexsort((xyz[left, 2], xyz[left, 1]))]
        right = right[np.lexsort((xyz[right, 2], xyz[right, 1]))]
    elif order.startswith("brick"):
        # experiment: space-filling "brick" order (brick edge in Angstrom after the colon)
        edge = float(order.split(":")[1]) if ":" in order else 7.7
        b = np.floor(xyz[mid] / edge).astype(np.int64)
        key = (b[:, 1] * 100000 + b[:, 2]) * 100000 + b[:, 0]
        mid = mid[np.lexsort((xyz[mid, 0], xyz[mid, 2], xyz[mid, 1], key))]
        left = left[np.lexsort((xyz[left, 2], xyz[left, 1]))]
        right = right[np.lexsort((xyz[right, 2], xyz[right, 1]))]
    elif order != "original":
        raise ValueError(order)
    assert len(left) == len(right), (len(left), len(right))
    perm = np.concatenate([left, mid, right])
    xyz, el = np.ascontiguousarray(xyz[perm]), np.ascontiguousarray(el[perm])
    return dict(xyz=xyz, element=el, lattice=np.array([108.98, L, L]), N=len(el), N_contact=len(left),
                metals=base["metals"], Vd=15.0,              # structures/40nm_crossbar/parameters.txt:42
                nn_dist=3.5, pbc=0, sigma=3.5e-10, k=base["k"], high_G=1.0, low_G=1e-8,
                name="synthetic_40nm_crossbar(tiles=%d,fill=%.2f,lines=%d,seed=%d,%s%s)"
                     % (tiles, fill, n_lines, seed, order, ",filament=%g" % filament if filament else ""))


def synth_small(tiles=1, seed=1, order="bwmin", fill=1.0):
    """Small synthetic device of the same family (for quick tests): `tiles` x `tiles` cells, uncarved."""
    return synth_crossbar_40nm(tiles=tiles, seed=seed, carve=fill < 1.0, fill=fill, order=order)
