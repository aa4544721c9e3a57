"""KMC event step (execute_kmc_step_mpi, src/kmc_events.cu:333-563).

CPU: the oracle's MT19937 + libstdc++ uniform_real_distribution restatement against known answers and
against numpy's independent MT19937; the library's std::mt19937-based generator against the oracle's.
GPU: event lists, selected events and event times of the HIP path against the oracle on the reference's
5 nm device at several temperatures (few events per step .. hundreds).

Pinning: the reference's expected_output/output1_0.txt prints the cumulative "KMC time" after each of its six
steps; tests/test_gpu_reference_trajectory.py (GPU) and tests/test_oracle_golden.py::
test_oracle_reproduces_reference_trajectory (CPU) reproduce all six values and the final element state of
snapshot_6.xyz.  (Step 1 of that run executes three events: the first two are vacancy hops with E_A ~ 0 at
~7e14 /s, the third draw reaches 1/freq once those are zeroed out.)  This file checks the event step in
isolation, at temperatures that give from 1-2 to hundreds of events per step."""
import numpy as np
import pytest


def test_mt19937_known_answers(oracle):
    raw = oracle.mt_raw_stream(5489, 10000)
    assert raw[0] == 3499211612 and raw[9999] == 4123659995      # std::mt19937 default seed: ISO C++ [rand.predef]
    raw1 = oracle.mt_raw_stream(1, 4)
    assert raw1.tolist() == [1791095845, 4282876139, 3093770124, 4005303368]
    rs = np.random.RandomState(1)                                 # independent implementation, init_genrand seeding
    want = np.frombuffer(rs.bytes(4 * 2000), dtype="<u4")
    assert np.array_equal(oracle.mt_raw_stream(1, 2000), want.astype(np.uint64))
    u = oracle.mt_uniform_stream(1, 1000)
    x0, x1 = want[0:2000:2].astype(np.float64), want[1:2000:2].astype(np.float64)
    assert np.array_equal(u, (x0 + x1 * 4294967296.0) / 18446744073709551616.0)   # generate_canonical<double,53>
    assert 0.0 <= u.min() and u.max() < 1.0


def test_library_rng_matches_oracle(km, oracle):
    """kmcf_rng = std::mt19937 + std::uniform_real_distribution<double> (src/random_num.h), host only."""
    g = km.solvers.RandomNumberGenerator(km.structure.RND_SEED_KMC)
    got = np.array([g.getRandomNumber() for _ in range(2000)])
    assert np.array_equal(got, oracle.mt_uniform_stream(1, 2000))
    g.setSeed(5)
    assert g.getRandomNumber() == oracle.mt_uniform_stream(5, 1)[0]


def test_site_layers(km, dev5):
    lay = km.solvers.site_layers(dev5["xyz"][:, 0], km.structure.LAYERS)
    assert lay.min() == 0 and lay.max() == 4
    assert np.all(lay[:576] == 0) and np.all(lay[-576:] == 4)
    x = dev5["xyz"][:, 0]
    assert np.all(lay[(x > 3.0) & (x < 48.0)] == 2)


@pytest.fixture(scope="module")
def fields5(oracle, dev5, ref5):
    d = dev5
    NL = d["N_contact"]
    pot = oracle.poisson_gridless(d["xyz"], ref5["charge"], d["sigma"], d["k"])
    pot[NL:NL + ref5["ks"].n] += ref5["x"]
    return pot


@pytest.mark.gpu
@pytest.mark.parametrize("T_bg,fullscan", [(300.0, False), (150.0, False), (77.0, False), (150.0, True), (300.0, "callback"),
                                           (300.0, "narrow"), (300.0, "three_launches")])
def test_kmc_step_matches_oracle(km, oracle, dev5, ref5, fields5, T_bg, fullscan, monkeypatch):
    """Same potentials in, same generator state in: identical event sequence (i, j, type), identical final
    element / charge state, event time to 1e-12.  T_bg scales the rates: 300 K -> hundreds of events per
    step at 5 V, 77 K -> one or two."""
    import torch
    # fullscan: the reference's way of zeroing events (a pass over every slot per event) instead of the
    # neighbour-list shortcut; both must select the same events.  "callback": a caller-supplied uniform source
    # (events then go one per host round trip instead of in pre-drawn batches).
    # "narrow": the batch kernel with a claim range of one tile, so that events take its out-of-range path (sums read
    # again instead of patched); "three_launches": the per-event launches the batch kernel replaced.
    use_callback = fullscan == "callback"
    monkeypatch.delenv("KMCF_EV_TREL", raising=False)
    monkeypatch.delenv("KMCF_EVENTS_PERSISTENT", raising=False)
    if fullscan == "narrow":
        monkeypatch.setenv("KMCF_EV_TREL", "1")
    if fullscan == "three_launches":
        monkeypatch.setenv("KMCF_EVENTS_PERSISTENT", "0")
    fullscan = fullscan is True
    if fullscan:
        monkeypatch.setenv("KMCF_EVENTS_FULLSCAN", "1")
    else:
        monkeypatch.delenv("KMCF_EVENTS_FULLSCAN", raising=False)
    S = km.solvers
    d = dev5
    N = d["N"]
    layers = km.structure.LAYERS
    lay = S.site_layers(d["xyz"][:, 0], layers)
    freq = 10e13                                           # attempt_frequency, structures/5nm_device/parameters.txt
    g_o = oracle.mt_state(1)
    t_o, n_o, log_o, el_o, ch_o = oracle.kmc_step(d["xyz"], ref5["neigh"], lay, T_bg, freq, d["sigma"], d["k"], fields5,
                                                  d["element"], ref5["charge"], layers, g_o, max_events=4096)
    comm = S.KMC_comm(N - 2 * d["N_contact"], N + 1, N, N)
    comm.connect()
    dev = "cuda"
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev)
    el, ch = i32(d["element"]), i32(ref5["charge"])
    rng = S.RandomNumberGenerator(1)
    t, n, log = S.execute_kmc_step_mpi(comm, N, comm.counts_events, comm.displs_events, 52, i32(ref5["neigh"].reshape(-1)),
                                       i32(lay), T_bg, freq, d["sigma"], d["k"], f64(d["xyz"][:, 0]), f64(d["xyz"][:, 1]),
                                       f64(d["xyz"][:, 2]), f64(fields5), el, ch,
                                       rng.getRandomNumber if use_callback else rng, layers, max_events=4096,
                                       return_log=True)
    assert n == n_o and n >= 1
    assert np.array_equal(log, log_o)
    assert np.array_equal(el.cpu().numpy(), el_o) and np.array_equal(ch.cpu().numpy(), ch_o)
    assert t == pytest.approx(t_o, rel=1e-12) and t >= 1 / freq
    # the generator advanced by exactly two draws per event on both sides
    assert rng.getRandomNumber() == oracle.mt_uniform_stream(1, 2 * n + 1)[-1]
    # bookkeeping of the event rules (execute_event, :284-331)
    assert (el_o == 2).sum() - (d["element"] == 2).sum() == (log[:, 2] == 0).sum() - (log[:, 2] == 1).sum()
    comm.close()
