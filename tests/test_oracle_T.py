"""CPU: the T-path oracle (oracle/kmcf_oracle_T.c) -- PARITY UNPINNED by any reference fixture (the reference's
golden run never executes the current solver, src/KMC_comm.h:243).  What is checked instead:

* the assembled neighbour operator equals the reference's *CPU* formulation (src/current_solver.cpp:56-240:
  dense X over N_atom + 2 nodes, diagonal = minus the row sum, last row/column -- the ground node -- cut),
  restated here independently in numpy, on a small device where the GPU kernels' shortcut for the cut node
  ("every connection to the ground atom is high_G", src/current_solver_gpu.cu:1113-1123) holds;
* trap-to-trap / contact-to-contact WKB values equal a direct evaluation of the formula
  (src/initialize_sparsity_T.cu:596-609), the sub-block is symmetric and its rows sum to zero;
* the split-operator PCG reaches the dense direct solution; I_macro equals the injected current evaluated
  from the solution; dissipated power is non-negative and only non-metal atoms receive it;
* structure counts on the reference's 5 nm device (derived numbers, printed for DESIGN.md).
"""
import numpy as np
import pytest

Q = 1.60217663e-19
DEFECT, OD, VAC, O_EL, HF, NI, TI, PT, N_EL = range(9)


def small_device(seed=0, ny=4, nz=4, n_contact_layers=3, n_oxide_layers=7, a=2.5):
    """Cubic lattice (spacing a = 2.5 A, nn_dist 3.5 A: 6 neighbours), x-ordered: left Ti contact, oxide with
    vacancies, right Ti contact; interstitial DEFECT sites interleaved to exercise the atom filter."""
    rng = np.random.default_rng(seed)
    xyz, el = [], []
    nl = 2 * n_contact_layers + n_oxide_layers
    for l in range(nl):
        for iy in range(ny):
            for iz in range(nz):
                xyz.append((l * a, iy * a, iz * a))
                if l < n_contact_layers or l >= n_contact_layers + n_oxide_layers:
                    el.append(TI)
                else:
                    el.append(VAC if rng.random() < 0.25 else (HF if (l + iy + iz) % 2 else O_EL))
        if n_contact_layers <= l < n_contact_layers + n_oxide_layers:
            for iy in range(ny - 1):        # interstitials of this layer
                xyz.append((l * a + a / 2, iy * a + a / 2, a / 2))
                el.append(OD if rng.random() < 0.3 else DEFECT)
    xyz = np.array(xyz)
    el = np.array(el, np.int32)
    charge = np.where(el == VAC, np.where(rng.random(len(el)) < 0.5, 2, 0), 0).astype(np.int32)
    # a smooth conduction-band edge [J] falling along x, constant inside the contacts
    x = xyz[:, 0]
    x0, x1 = (n_contact_layers - 1) * a, (n_contact_layers + n_oxide_layers) * a
    cb = Q * (1.0 - 2.0 * np.clip((x - x0) / (x1 - x0), 0, 1)) + Q * 0.003 * rng.standard_normal(len(x)) * ((x > x0) & (x < x1))
    return dict(xyz=xyz, element=el, charge=charge, cb=cb, n1=ny * nz, layers=n_contact_layers)


PAR = dict(nn_dist=3.5, Vd=2.0, high_G=1e5, low_G=1e-8, loop_G=1e7, tol=Q * 0.01, m_e=0.85 * 9.11e-31, V0=1.6)


def make_T(oracle, d, x_lo, x_hi):
    return oracle.TSystem(d["xyz"], d["element"], d["charge"], d["cb"], np.array([TI, N_EL], np.int32),
                          PAR["nn_dist"], d["n1"], d["n1"], d["layers"], PAR["Vd"], PAR["high_G"], PAR["low_G"],
                          PAR["loop_G"], PAR["tol"], PAR["m_e"], PAR["V0"], x_lo=x_lo, x_hi=x_hi)


def cpu_formulation_X(d, T):
    """src/current_solver.cpp:75-215 without the tunnelling terms: dense X over N_full nodes."""
    Na = T.N_atom
    pos = np.stack([T.ax, T.ay, T.az], 1)
    el, ch = T.atom_element, T.atom_charge
    Nf = Na + 2
    X = np.zeros((Nf, Nf))
    metal = np.isin(el, [TI, N_EL])
    cvac = (el == VAC) & (ch == 0)
    D = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    nb = (D < PAR["nn_dist"]) & ~np.eye(Na, dtype=bool)
    hi = (metal[:, None] & metal[None, :]) | (cvac[:, None] & cvac[None, :])
    X[2:, 2:] = np.where(nb, np.where(hi, -PAR["high_G"], -PAR["low_G"]), 0.0)
    n1 = d["n1"]
    for i in range(Na):
        if i < n1:
            X[1, i + 2] = X[i + 2, 1] = -PAR["high_G"]
        if i > Na - n1:
            X[0, i + 2] = X[i + 2, 0] = -PAR["high_G"]
    X[0, 1] = X[1, 0] = -PAR["loop_G"]
    X[np.diag_indices(Nf)] = -X.sum(1)
    return X


def test_neighbour_operator_equals_cpu_formulation(oracle):
    import scipy.sparse as sp
    d = small_device()
    T = make_T(oracle, d, x_lo=1e9, x_hi=-1e9)          # no metal tunnel points: vacancies only
    assert T.N_atom == int(((d["element"] != DEFECT) & (d["element"] != OD)).sum())
    assert np.array_equal(T.atom_site, np.nonzero((d["element"] != DEFECT) & (d["element"] != OD))[0])
    A = sp.csr_matrix((T.val, T.col, T.row_ptr), shape=(T.Nsub, T.Nsub)).toarray()
    X = cpu_formulation_X(d, T)
    np.testing.assert_allclose(A, X[:T.Nsub, :T.Nsub], rtol=1e-13, atol=0)
    assert np.array_equal(A != 0, X[:T.Nsub, :T.Nsub] != 0) or np.all(A[(A != 0) != (X[:T.Nsub, :T.Nsub] != 0)] == 0)
    np.testing.assert_allclose(T.diag_neigh, np.diag(A), rtol=0, atol=0)
    # columns ascending in every row; symmetric pattern and values
    for i in range(T.Nsub):
        c = T.col[T.row_ptr[i]:T.row_ptr[i + 1]]
        assert np.all(np.diff(c) > 0)
    assert np.array_equal(A, A.T)


def test_tunnel_block(oracle):
    import scipy.sparse as sp
    d = small_device(seed=3)
    a = 2.5
    # contact atoms of the layers next to the oxide become tunnel points through the x window
    T = make_T(oracle, d, x_lo=(d["layers"] - 1) * a - 0.1, x_hi=(d["layers"] + 7) * a + 0.1)
    assert T.n_t > 0 and len(T.sub_col) > T.n_t
    el = T.atom_element[T.tunnel_idx]
    assert np.all((el == VAC) | (el == TI))
    assert (el == TI).any() and 0 not in T.tunnel_idx
    S = sp.csr_matrix((T.sub_val, T.sub_col, T.sub_row_ptr), shape=(T.n_t, T.n_t)).toarray()
    np.testing.assert_allclose(S, S.T, rtol=1e-12)                  # |dE| is symmetric, so is the WKB value
    np.testing.assert_allclose(S.sum(1), 0, atol=1e-12 * np.abs(S).max())
    np.testing.assert_allclose(np.diag(S), T.diag_tunnel, rtol=0, atol=0)
    # direct evaluation of the single-shot branch for vacancy-vacancy pairs
    prefac = -(np.sqrt(2 * PAR["m_e"]) / 1.054571817e-34) * (2.0 / 3.0)
    pos = np.stack([T.ax, T.ay, T.az], 1)[T.tunnel_idx]
    cb = T.atom_CB_edge[T.tunnel_idx]
    n_checked = 0
    for i in range(T.n_t):
        for j in range(T.n_t):
            if i == j or el[i] != VAC or el[j] != VAC:
                continue
            dist = np.linalg.norm(pos[i] - pos[j])
            dE = abs(cb[i] - cb[j])
            if dist > PAR["nn_dist"] and dE > PAR["tol"]:
                E1 = Q * PAR["V0"]
                E2 = E1 - dE
                want = -np.exp(prefac * (dist * 1e-10 / abs(E1 - E2)) * (E1 ** 1.5 - (E2 ** 1.5 if E2 > 0 else 0.0)))
                assert abs(S[i, j] - want) <= 1e-10 * abs(want), (i, j, S[i, j], want)
                n_checked += 1
            else:
                assert S[i, j] == 0
    assert n_checked > 10
    # preconditioner = diagonal of the merged operator
    M = T.merged_csr().toarray()
    np.testing.assert_allclose(np.diag(M), T.diag, rtol=1e-14)
    xv = np.arange(T.Nsub) * 0.01 + 1
    assert np.all(np.abs(T.spmv(xv) - M @ xv) <= 1e-13 * (np.abs(M) @ np.abs(xv)))


def test_split_pcg_imacro_power(oracle):
    d = small_device(seed=5)
    a = 2.5
    T = make_T(oracle, d, x_lo=(d["layers"] - 1) * a - 0.1, x_hi=(d["layers"] + 7) * a + 0.1)
    M = T.merged_csr().toarray()
    direct = np.linalg.solve(M, T.rhs)
    x, it, rel = T.solve(np.zeros(T.Nsub), 1e-13, 20000)
    assert rel <= 1e-13 and it < 20000
    # driver nodes: injection node at ~ +Vd relative to the extraction node (ground = cut atom, potential 0)
    assert abs(direct[1] - direct[0] - PAR["Vd"]) < 0.01 * PAR["Vd"]
    # conductances span 1e7 ... 1e-8 (and WKB values below): at sites coupled only through low_G both the
    # direct and the iterative answer carry O(1e-3) slack, like the K system; the metallic
    # network and the two driver nodes are tight, the true residual is at rounding level
    res = T.rhs - M @ x
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(T.rhs)
    tight = np.r_[0, 1, 2 + np.nonzero(np.isin(T.atom_element[:-1], [TI, N_EL]))[0]]
    np.testing.assert_allclose(x[tight], direct[tight], rtol=0, atol=2e-6 * np.abs(direct).max())
    np.testing.assert_allclose(x, direct, rtol=0, atol=1e-2 * np.abs(direct).max())
    G0 = 2 * 3.8612e-5 * 1e-5
    m = np.zeros(T.N_atom + 2)
    m[:T.Nsub] = x * G0
    n1 = d["n1"]
    inj = PAR["high_G"] * (m[1] - m[2:2 + n1]).sum()
    assert abs(T.imacro(m) - inj) <= 1e-10 * abs(inj) and inj > 0
    power = np.full(len(d["element"]), -7.0)
    T.power(m.copy(), 1.0, power)
    metal_sites = np.isin(d["element"], [TI, N_EL])
    atoms = T.atom_site[:-1]
    written = np.zeros(len(power), bool)
    written[atoms[~metal_sites[atoms]]] = True
    assert np.all(power[~written] == -7.0)               # metals, interstitials and the cut atom keep their value
    assert np.all(power[written] >= 0) and power[written].max() > 0


def test_5nm_structure_counts(oracle, dev5, ref5):
    d = dev5
    NL = d["N_contact"]
    cb, it, _ = oracle.update_CB_edge(ref5["ks"], d["element"], d["metals"], d["high_G"], d["low_G"], d["Vd"])
    T = oracle.TSystem(d["xyz"], d["element"], ref5["charge"], cb, d["metals"], d["nn_dist"], NL, NL, 10, d["Vd"],
                       1e5 * d["high_G"], d["low_G"], 1e7 * d["high_G"], Q * 0.01, 0.85 * 9.11e-31, 1.6)
    assert T.N_atom == 25681 and T.Nsub == 25682
    assert T.row_ptr[1] == 2 + (NL - 2) and T.row_ptr[2] - T.row_ptr[1] == 2 + NL      # rows 0 and 1
    assert T.n_t == 1913 and (T.atom_element[T.tunnel_idx] == VAC).sum() == 400
    dens = len(T.sub_col) / T.n_t ** 2
    assert 0.3 < dens < 0.4                                  # the authors' 40 nm block: 94 211 070 / 14 854^2 = 0.43
    x, it, rel = T.solve(np.zeros(T.Nsub), 1e-15 * T.N_atom, 2000)
    assert rel <= 1e-15 * T.N_atom and 250 < it < 400
    assert abs(x[1] - d["Vd"]) < 1e-3                        # "non-negligible potential drop" check of :2014-2020
