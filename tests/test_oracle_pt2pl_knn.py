"""Matcher_Point2Plane on a plain point map -- k nearest neighbours + PCA (pipelines/rgbd.yaml:143-151; SURVEY 8a row a13,
"otherwise KNN + PCA" [U]) -- in the C oracle, pinned by an independent numpy reading of the same sentences (brute force over
the 27-voxel block, numpy's sort and eigh instead of the insertion list and the Jacobi sweeps)."""
import numpy as np
import pytest

from mola_lidar_odometry_amd import synth
from oracle import oracle_c

I12 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)


def numpy_pt2pl_knn(map_dump, voxel_size, q, distance_threshold, plane_eigen_threshold, search_radius, knn, min_pts):
    xyz = map_dump["xyz"]
    vk = np.repeat(map_dump["vox_keys"], map_dump["vox_count"], axis=0)  # voxel of every stored point, dump order = scan order
    inv = np.float32(1.0) / np.float32(voxel_size)
    out = []
    r2 = np.float32(search_radius * search_radius)
    for i, p in enumerate(q):
        c = np.floor(p * inv).astype(np.int64)
        sel = np.nonzero((np.abs(vk - c) <= 1).all(1))[0]
        if len(sel) == 0:
            continue
        d = xyz[sel] - p
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        order = np.argsort(d2, kind="stable")[:knn]
        order = order[d2[order] < r2]
        if len(order) < max(3, min_pts):
            continue
        nb = xyz[sel[order]].astype(np.float64)
        mu = nb.mean(0)
        w, V = np.linalg.eigh(np.cov(nb.T))
        if not (w[2] > 0 and w[0] <= plane_eigen_threshold * w[2]):
            continue
        n = V[:, 0] / np.linalg.norm(V[:, 0])
        n = n * (1.0 if n[np.argmax(np.abs(n))] > 0 else -1.0)
        dist = abs(float(n @ (p.astype(np.float64) - mu)))
        out.append((i, mu, n, dist, w[0] / w[2], float(d2[order][-1])))
    return out


@pytest.mark.parametrize("voxel,knn,min_pts", [(0.5, 10, 6), (1.0, 5, 3), (0.4, 16, 8)])
def test_c_oracle_matches_the_numpy_reading(voxel, knn, min_pts):
    cloud = synth.ndt_cloud(3)
    m = oracle_c.Map(voxel, 20).insert(cloud)
    rng = np.random.default_rng(7)
    q = (cloud[rng.choice(len(cloud), 1500, replace=False)] + rng.normal(0, 0.04, (1500, 3))).astype(np.float32)
    thr, eig_thr, radius = 0.08, 2e-2, 0.6
    r = oracle_c.match_pt2pl_knn(m, q, I12, thr, eig_thr, radius, knn, min_pts)
    ref = numpy_pt2pl_knn(m.dump(), voxel, q, thr, eig_thr, radius, knn, min_pts)
    # decisions that sit on a threshold to within rounding may differ between Jacobi and eigh: compare away from the thresholds
    sure = {i for i, mu, n, dist, ratio, dk in ref if dist <= thr and abs(dist - thr) > 1e-6 and abs(ratio - eig_thr) > 1e-6 * eig_thr}
    maybe = {i for i, mu, n, dist, ratio, dk in ref if dist <= thr + 1e-6}
    got = set(int(i) for i in r["local_idx"])
    assert sure <= got <= maybe and len(sure) > 300
    by_i = {i: (mu, n) for i, mu, n, dist, ratio, dk in ref}
    for k, i in enumerate(r["local_idx"]):
        mu, n = by_i[int(i)]
        assert np.allclose(r["centroid"][k], mu, atol=2e-6)
        assert np.allclose(r["normal"][k], n, atol=2e-5), (r["normal"][k], n)
        assert abs(np.linalg.norm(r["normal"][k]) - 1.0) < 1e-6


def test_too_few_neighbours_a_blob_and_a_far_point_pair_with_nothing():
    rng = np.random.default_rng(1)
    plane = np.stack([rng.uniform(-2, 2, 4000), rng.uniform(-2, 2, 4000), rng.normal(0.0, 0.003, 4000)], 1).astype(np.float32)
    blob = rng.normal([6.0, 0.0, 1.0], 0.15, (1500, 3)).astype(np.float32)
    lonely = np.array([[20.0, 20.0, 5.0], [20.1, 20.0, 5.0]], np.float32)
    m = oracle_c.Map(0.5, 20).insert(np.concatenate([plane, blob, lonely]))
    q = np.array([[0.1, 0.2, 0.02],      # on the plane: pairs
                  [6.0, 0.0, 1.0],       # inside the blob: e0/e2 far above the threshold
                  [20.05, 20.0, 5.0],    # two neighbours only
                  [0.3, -0.4, 0.5]],     # 0.5 m above the plane: neighbours beyond the search radius of 0.4
                 np.float32)
    r = oracle_c.match_pt2pl_knn(m, q, I12, 0.1, 1e-2, 0.4, 10, 6)
    assert r["local_idx"].tolist() == [0]
    assert abs(r["normal"][0][2]) > 0.999 and abs(r["centroid"][0][2]) < 0.01
