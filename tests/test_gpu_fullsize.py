"""-m gpu: the hot path at BASELINE.json's full C2 size (120 000-point scan vs 1 000 000-point map) -- parity against the
oracle where the oracle is fast enough (it is: one full alignment takes a fraction of a second on a few threads) and
size-independent properties where it is not needed: round trips, self-queries, permutation equivariance, idempotence."""
import numpy as np
import pytest

from mola_lidar_odometry_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    return synth.workload_c2()


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gmap(ctx, c2):
    return capi.Map(ctx, c2.voxel_size, c2.cap).build(c2.map_xyz)


def test_map_round_trip_at_full_size(gmap, c2):
    i = gmap.info()
    assert (i.n_points, i.n_offered) == (len(c2.map_xyz), len(c2.map_xyz))  # the workload's points all survive the cap
    d = gmap.download()
    # a permutation of the input, voxel keys strictly ascending, counts within the cap, in-voxel insertion order
    assert np.array_equal(np.sort(d["src_idx"]), np.arange(len(c2.map_xyz), dtype=np.uint32))
    np.testing.assert_array_equal(d["xyz"], c2.map_xyz[d["src_idx"]])
    k = d["vox_keys"].astype(np.int64)
    packed = ((k[:, 0] + (1 << 20)) << 42) | ((k[:, 1] + (1 << 20)) << 21) | (k[:, 2] + (1 << 20))
    assert np.all(np.diff(packed) > 0) and d["vox_count"].max() <= c2.cap and d["vox_count"].sum() == i.n_points
    first = np.repeat(d["vox_first"], d["vox_count"])
    within = np.arange(i.n_points) - first
    seg_start = within == 0
    assert np.all((np.diff(d["src_idx"].astype(np.int64)) > 0) | seg_start[1:])
    np.testing.assert_array_equal(np.floor(d["xyz"] * np.float32(1.0 / c2.voxel_size)).astype(np.int64), np.repeat(k, d["vox_count"], 0))


def test_every_map_point_finds_itself(ctx, gmap, c2):
    q = capi.Scan(ctx, c2.map_xyz)  # 1 M queries
    r = capi.nn_search_dense(gmap, q, np.eye(4)[:3])
    assert np.all(r["d2"] == 0.0)
    hit = r["global_idx"]
    assert np.all(hit != 0xFFFFFFFF)
    np.testing.assert_array_equal(c2.map_xyz[hit], c2.map_xyz)  # itself, or an exact duplicate stored earlier
    assert np.all(hit <= np.arange(len(hit)))


@pytest.mark.parametrize("match", ["f", "q", "t", "w", "o"])
def test_full_size_alignment_matches_oracle(ctx, gmap, c2, oracle, match, monkeypatch):
    monkeypatch.setenv("MH_MATCH", match)  # f: plan / scan (the default) | q: quad search through the caches | t, w: tiles staged in LDS | o: sorted scan
    om = oracle.Map(c2.voxel_size, c2.cap).insert(c2.map_xyz)
    kw = dict(max_iterations=c2.n_iters, disable_stall_test=True, threshold=c2.threshold, kernel_param=c2.kernel_param)
    scan = capi.Scan(ctx, c2.scan_xyz)
    g = capi.icp_align(gmap, scan, c2.T_guess, capi.ICPParams(**kw), want_pairs=True)
    o = oracle.icp_align(om, c2.scan_xyz, c2.T_guess, oracle.ICPParams(**kw), n_threads=16, want_pairs=True)
    assert [t["n_pairs"] for t in g["trace"]] == [t["n_pairs"] for t in o["trace"]]
    np.testing.assert_array_equal(g["pairs"]["local_idx"], o["pairs"]["local_idx"])
    np.testing.assert_array_equal(g["pairs"]["global_idx"], o["pairs"]["global_idx"])
    np.testing.assert_array_equal(g["pairs"]["d2"], o["pairs"]["d2"])
    np.testing.assert_allclose(g["T"], o["T"], rtol=0, atol=1e-9)       # bar: 1e-4 m / 1e-4 rad
    np.testing.assert_allclose(g["cov"], o["cov"], rtol=1e-6, atol=1e-12)
    assert g["quality"] == o["quality"]
    # (accuracy is not what is tested here: along the street the canyon barely constrains point-to-point ICP; both
    #  implementations end at the same pose, closer to the truth than the guess)
    err = lambda T: np.linalg.norm(np.asarray(T)[[3, 7, 11]] - c2.T_gt[[3, 7, 11]])
    assert err(g["T"]) < err(c2.T_guess)


def test_permuting_the_scan_permutes_the_pairings(ctx, gmap, c2):
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(c2.scan_xyz))
    a = capi.nn_search_dense(gmap, capi.Scan(ctx, c2.scan_xyz), c2.T_guess)
    b = capi.nn_search_dense(gmap, capi.Scan(ctx, c2.scan_xyz[perm]), c2.T_guess)
    np.testing.assert_array_equal(b["global_idx"], a["global_idx"][perm])
    np.testing.assert_array_equal(b["d2"], a["d2"][perm])
    kw = dict(max_iterations=5, disable_stall_test=True, threshold=c2.threshold[:5], kernel_param=c2.kernel_param[:5])
    ra = capi.icp_align(gmap, capi.Scan(ctx, c2.scan_xyz), c2.T_guess, capi.ICPParams(**kw))
    rb = capi.icp_align(gmap, capi.Scan(ctx, c2.scan_xyz[perm]), c2.T_guess, capi.ICPParams(**kw))
    assert ra["n_final_pairs"] == rb["n_final_pairs"]
    np.testing.assert_allclose(ra["T"], rb["T"], rtol=0, atol=1e-11)  # only the summation order differs


def test_aligning_map_points_to_their_map_is_a_fixed_point(ctx, gmap, c2):
    sub = c2.map_xyz[:: len(c2.map_xyz) // 120000][:120000]
    r = capi.icp_align(gmap, capi.Scan(ctx, sub), np.eye(4)[:3],
                       capi.ICPParams(max_iterations=10, threshold=c2.threshold[:10], kernel_param=c2.kernel_param[:10]))
    np.testing.assert_array_equal(r["T"], np.eye(4)[:3].reshape(12))       # zero residuals: the solve returns delta = 0
    assert r["n_final_pairs"] == len(sub) and r["quality"] == 1.0
    assert capi.TERM_NAMES[r["termination_reason"]] == "Stalled" and r["n_iterations"] == 0


@pytest.mark.parametrize("match", ["f", "q", "t", "w", "o"])
def test_lockstep_batch_equals_single_alignments(gmap, c2, match, monkeypatch):
    """mh_icp_align_batch runs large-layer jobs in lock step (one launch per kernel over all jobs, blockIdx.y = job):
    ragged scan sizes, different guesses, a prior on one job, a stall-terminated run where the jobs finish at different
    iterations -- every result is bitwise the single alignment's, and so is the per-stream fallback's."""
    monkeypatch.setenv("MH_MATCH", match)
    sizes = [len(c2.scan_xyz), 50000, 77777, 40001]
    ctxs = [capi.Context(0) for _ in sizes]
    scans = [capi.Scan(c, c2.scan_xyz[:n]) for c, n in zip(ctxs, sizes)]
    rng = np.random.default_rng(5)
    guesses = []
    for _ in sizes:
        g = c2.guess_ypr.copy()
        g[:3] += rng.normal(0, 0.05, 3)
        guesses.append(synth.pose_from_ypr(g))
    prior = (c2.T_guess, np.diag([4.0, 4.0, 4.0, 100.0, 100.0, 100.0]))
    thr, kp = synth.threshold_schedule(c2.sigma, 60)
    for kw, priors in ((dict(max_iterations=6, disable_stall_test=True, threshold=c2.threshold[:6], kernel_param=c2.kernel_param[:6],
                             poll_every=6), None),
                       (dict(max_iterations=60, threshold=thr, kernel_param=kp, poll_every=4), [None, prior, None, None])):
        p = capi.ICPParams(**kw)
        singles = [capi.icp_align(gmap, s, g, p, prior=(priors[i] if priors else None), want_trace=False, want_pairs=True)
                   for i, (s, g) in enumerate(zip(scans, guesses))]
        block = np.zeros(sum(capi.pairs_block_bytes(n) for n in sizes), np.uint8)  # Results::finalPairings of all jobs
        batch = capi.icp_align_batch([gmap] * len(scans), scans, guesses, p, priors=priors, pairs_block=block)
        for a, pr in zip(singles, capi.unpack_pairs_block(block, sizes, batch)):
            for k in ("local_idx", "global_idx", "global_xyz", "d2"):
                assert np.array_equal(pr[k], a["pairs"][k])
        monkeypatch.setenv("MH_NO_LOCKSTEP", "1")
        streams = capi.icp_align_batch([gmap] * len(scans), scans, guesses, p, priors=priors)
        monkeypatch.delenv("MH_NO_LOCKSTEP")
        monkeypatch.setenv("MH_LOCKSTEP_GROUPS", "2")  # (two groups of two jobs)
        grouped = capi.icp_align_batch([gmap] * len(scans), scans, guesses, p, priors=priors)
        monkeypatch.delenv("MH_LOCKSTEP_GROUPS")
        for a, b, c, d in zip(singles, batch, streams, grouped):
            for r in (b, c, d):
                assert (r["n_iterations"], r["termination_reason"], r["n_final_pairs"]) == (
                    a["n_iterations"], a["termination_reason"], a["n_final_pairs"])
                assert np.array_equal(r["T"], a["T"]) and np.array_equal(r["cov"], a["cov"]) and r["quality"] == a["quality"]
    for c in ctxs:
        c.close()
