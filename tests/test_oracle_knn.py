"""Matcher_Points_DistanceThreshold with pairingsPerPoint > 1 (rgbd.yaml:135-141) on nn_multiple_search ("same scan keeping k
best sorted"): the C oracle's restatement against a brute-force numpy reading of the same rule (all
candidates of the 3x3x3 block in scan order, STABLE sort by the fp32 distance, first k, accepted while below the limit) and
known answers."""
import numpy as np
import pytest

I12 = np.eye(4)[:3]


def _brute(dump, vs, q, k, lim):
    """[(global src index, d2)] of one query: the rule, literally."""
    keys = {tuple(kk): i for i, kk in enumerate(dump["vox_keys"].tolist())}
    c = np.floor(q.astype(np.float32) * np.float32(1.0 / vs)).astype(int)
    cand = []
    for ix in (-1, 0, 1):
        for iy in (-1, 0, 1):
            for iz in (-1, 0, 1):
                v = keys.get((c[0] + ix, c[1] + iy, c[2] + iz))
                if v is None:
                    continue
                f, n = int(dump["vox_first"][v]), int(dump["vox_count"][v])
                cand.extend(range(f, f + n))
    if not cand:
        return []
    P = dump["xyz"][cand]
    d = P - q.astype(np.float32)
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # fp32, this order
    order = np.argsort(d2, kind="stable")[:k]
    out = []
    for o in order:
        if not d2[o] < lim:
            break
        out.append((int(dump["src_idx"][cand[o]]), float(d2[o])))
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 8])
def test_k_best_matcher_is_the_stable_sort_of_the_block(oracle, k):
    rng = np.random.default_rng(7 + k)
    pts = rng.uniform(-6, 6, (6000, 3)).astype(np.float32)
    pts[:1500] = np.round(pts[:1500] * 2) / 2          # a half-metre lattice: many exactly equal distances
    m = oracle.Map(1.0, 12).insert(pts)
    d = m.dump()
    q = rng.uniform(-6.5, 6.5, (400, 3)).astype(np.float32)
    q[:100] = np.round(q[:100] * 4) / 4                # queries on lattice symmetry points: ties between candidates
    thr = 0.8
    r = oracle.match_points_k(m, q, I12, thr, k)
    assert r["potential_pairings"] == len(q) * k
    pos = 0
    for i in range(len(q)):
        ref = _brute(d, 1.0, q[i], k, np.float32(thr * thr))
        got = [(int(g), float(dd)) for g, dd in zip(r["global_idx"][pos:pos + len(ref)], r["d2"][pos:pos + len(ref)])]
        assert list(r["local_idx"][pos:pos + len(ref)]) == [i] * len(ref)
        assert got == ref, i
        pos += len(ref)
    assert pos == len(r["local_idx"])
    if k == 1:
        r1 = oracle.match_points(m, q, I12, thr)
        for key in ("local_idx", "global_idx", "d2"):
            np.testing.assert_array_equal(r[key], r1[key])


def test_k_best_known_answer(oracle):
    pts = np.array([[0.5, 0.5, 0.5], [0.625, 0.5, 0.5], [0.375, 0.5, 0.5], [1.25, 0.5, 0.5], [0.5, 0.5, 0.875]], np.float32)
    m = oracle.Map(1.0, 20).insert(pts)
    q = np.array([[0.5, 0.5, 0.5], [5.0, 5.0, 5.0]], np.float32)
    r = oracle.match_points_k(m, q, I12, 0.45, 4)
    # nearest first; the two points 0.125 away tie exactly: the one earlier in scan order (index 1) stays in front; the point
    # 0.375 away passes 0.45, the one 0.75 away is the 5th nearest; the far query pairs with nothing
    assert r["local_idx"].tolist() == [0, 0, 0, 0]
    assert r["global_idx"].tolist() == [0, 1, 2, 4]
    np.testing.assert_array_equal(r["d2"], np.float32([0.0, 0.015625, 0.015625, 0.140625]))
    r = oracle.match_points_k(m, q, I12, 0.3, 4)   # break at the first failure
    assert r["global_idx"].tolist() == [0, 1, 2]
