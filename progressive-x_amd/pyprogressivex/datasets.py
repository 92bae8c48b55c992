"""Synthetic workloads of BASELINE.json (sizes/distributions: SURVEY.md §8d) and the reference's data-file format.

Everything is float64 and generated with numpy's PCG64 from an explicit seed, so the GPU box and this container build
identical inputs from the same numpy version.  Nothing here touches /root/reference at run time.
"""
import numpy as np

from . import _lib

TLESS_K = np.array([[1075.65087891, 0.0, 370.068878174],
                    [0.0, 1073.90344238, 278.721588135],
                    [0.0, 0.0, 1.0]])


# ---------------------------------------------------------------------------------------------------------------------
# geometry helpers
# ---------------------------------------------------------------------------------------------------------------------
def rodrigues(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def random_rotation(rng, max_angle=np.pi):
    axis = rng.normal(size=3)
    return rodrigues(axis, rng.uniform(0, max_angle))


def normalize_pnp(x1y1, x2y2z2, K):
    """find6DPoses_ marshalling (progressivex_python.cpp:64-98): rows (u_n, v_n, X, Y, Z), threshold scale f."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    Kinv = np.linalg.inv(K)
    x = np.asarray(x1y1, dtype=np.float64)
    un = Kinv[0, 0] * x[:, 0] + Kinv[0, 1] * x[:, 1] + Kinv[0, 2]
    vn = Kinv[1, 0] * x[:, 0] + Kinv[1, 1] * x[:, 1] + Kinv[1, 2]
    pts = np.column_stack([un, vn, np.asarray(x2y2z2, dtype=np.float64)])
    f = 0.5 * (K[0, 0] + K[1, 1])
    return np.ascontiguousarray(pts), f


# ---------------------------------------------------------------------------------------------------------------------
# C1  2D multi-line
# ---------------------------------------------------------------------------------------------------------------------
def make_lines(n_per_line=500, n_lines=3, n_outliers=500, sigma=0.5, size=1000.0, seed=0):
    rng = np.random.default_rng(seed)
    pts, labels, models = [], [], []
    for k in range(n_lines):
        a, b = rng.uniform(0, size, 2), rng.uniform(0, size, 2)
        dirv = (b - a) / np.linalg.norm(b - a)
        nrm = np.array([-dirv[1], dirv[0]])
        t = rng.uniform(0, 1, n_per_line)[:, None]
        p = a + t * (b - a) + rng.normal(0, sigma, n_per_line)[:, None] * nrm
        pts.append(p)
        labels.append(np.full(n_per_line, k + 1))
        models.append(np.array([nrm[0], nrm[1], -nrm @ a]))
    pts.append(rng.uniform(0, size, (n_outliers, 2)))
    labels.append(np.zeros(n_outliers, dtype=int))
    return np.ascontiguousarray(np.vstack(pts)), np.concatenate(labels).astype(np.int32), np.array(models)


# ---------------------------------------------------------------------------------------------------------------------
# C2  multi-homography
# ---------------------------------------------------------------------------------------------------------------------
def make_homographies(n_per_plane=600, n_planes=5, n_outliers=2000, sigma=0.5, size=1000.0, seed=0):
    rng = np.random.default_rng(seed)
    pts, labels, models = [], [], []
    for k in range(n_planes):
        H = np.eye(3)
        H[:2, :2] += rng.uniform(-0.2, 0.2, (2, 2))
        H[:2, 2] = rng.uniform(-50, 50, 2)
        H[2, :2] = rng.uniform(-1e-4, 1e-4, 2)
        c = rng.uniform(0.2 * size, 0.8 * size, 2)
        x = c + rng.uniform(-0.15 * size, 0.15 * size, (n_per_plane, 2))
        xh = np.column_stack([x, np.ones(n_per_plane)]) @ H.T
        y = xh[:, :2] / xh[:, 2:3] + rng.normal(0, sigma, (n_per_plane, 2))
        pts.append(np.column_stack([x, y]))
        labels.append(np.full(n_per_plane, k + 1))
        models.append(H.reshape(-1))
    pts.append(rng.uniform(0, size, (n_outliers, 4)))
    labels.append(np.zeros(n_outliers, dtype=int))
    return np.ascontiguousarray(np.vstack(pts)), np.concatenate(labels).astype(np.int32), np.array(models)


# ---------------------------------------------------------------------------------------------------------------------
# C3  multi two-view motion
# ---------------------------------------------------------------------------------------------------------------------
def make_two_view_motions(n_per_motion=10000, n_motions=8, n_outliers=20000, sigma=0.5, f=800.0, size=1000.0, seed=0):
    rng = np.random.default_rng(seed)
    K = np.array([[f, 0, size / 2], [0, f, size / 2], [0, 0, 1.0]])
    Kinv = np.linalg.inv(K)
    pts, labels, models = [], [], []
    for k in range(n_motions):
        R = random_rotation(rng, np.deg2rad(15))
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        center = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(4, 8)])
        X = center + rng.uniform(-0.7, 0.7, (n_per_motion, 3))
        x1 = X @ K.T
        x1 = x1[:, :2] / x1[:, 2:3]
        X2 = X @ R.T + t
        x2 = X2 @ K.T
        x2 = x2[:, :2] / x2[:, 2:3] + rng.normal(0, sigma, (n_per_motion, 2))
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        F = Kinv.T @ tx @ R @ Kinv
        F /= np.linalg.norm(F)
        pts.append(np.column_stack([x1, x2]))
        labels.append(np.full(n_per_motion, k + 1))
        models.append(F.reshape(-1))
    pts.append(rng.uniform(0, size, (n_outliers, 4)))
    labels.append(np.zeros(n_outliers, dtype=int))
    return np.ascontiguousarray(np.vstack(pts)), np.concatenate(labels).astype(np.int32), np.array(models)


# ---------------------------------------------------------------------------------------------------------------------
# C4  multi 6D pose + the metric batch of 2048 hypotheses
# ---------------------------------------------------------------------------------------------------------------------
def make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, sigma=1.0, seed=0, K=TLESS_K):
    """Returns raw inputs of find6DPoses (pixels, mm) + ground truth poses [n_objects, 12] and labels (0 = outlier)."""
    rng = np.random.default_rng(seed)
    x1, x2, labels, poses = [], [], [], []
    for k in range(n_objects):
        R = random_rotation(rng)
        tz = rng.uniform(600, 900)
        # object centre projects inside the 720 x 540 image
        cu, cv = rng.uniform(100, 620), rng.uniform(80, 460)
        t = np.array([(cu - K[0, 2]) / K[0, 0] * tz, (cv - K[1, 2]) / K[1, 1] * tz, tz])
        X = rng.normal(size=(n_per_object, 3))
        X = X / np.linalg.norm(X, axis=1, keepdims=True) * (50.0 * rng.uniform(0, 1, (n_per_object, 1)) ** (1 / 3))
        Xc = X @ R.T + t
        uv = Xc @ K.T
        uv = uv[:, :2] / uv[:, 2:3] + rng.normal(0, sigma, (n_per_object, 2))
        x1.append(uv)
        x2.append(X)
        labels.append(np.full(n_per_object, k + 1))
        poses.append(np.column_stack([R, t]).reshape(-1))
    x1.append(np.column_stack([rng.uniform(0, 720, n_outliers), rng.uniform(0, 540, n_outliers)]))
    Xo = rng.normal(size=(n_outliers, 3))
    Xo = Xo / np.linalg.norm(Xo, axis=1, keepdims=True) * (50.0 * rng.uniform(0, 1, (n_outliers, 1)) ** (1 / 3))
    x2.append(Xo)
    labels.append(np.zeros(n_outliers, dtype=int))
    return (np.ascontiguousarray(np.vstack(x1)), np.ascontiguousarray(np.vstack(x2)), np.array(K, dtype=np.float64),
            np.concatenate(labels).astype(np.int32), np.array(poses))


def make_pose_hypotheses(gt_poses, M=2048, max_angle_deg=30.0, max_shift_mm=100.0, seed=1):
    """Metric batch: the GT poses followed by poses perturbed by U(0,max_angle) / U(0,max_shift)."""
    rng = np.random.default_rng(seed)
    gt = np.asarray(gt_poses, dtype=np.float64).reshape(-1, 3, 4)
    out = [p.reshape(-1) for p in gt[:M]]
    while len(out) < M:
        P = gt[rng.integers(0, gt.shape[0])]
        dR = rodrigues(rng.normal(size=3), np.deg2rad(rng.uniform(0, max_angle_deg)))
        dt = rng.normal(size=3)
        dt = dt / np.linalg.norm(dt) * rng.uniform(0, max_shift_mm)
        out.append(np.column_stack([dR @ P[:, :3], P[:, 3] + dt]).reshape(-1))
    return np.ascontiguousarray(np.array(out[:M]))


# ---------------------------------------------------------------------------------------------------------------------
# C5  multi vanishing point
# ---------------------------------------------------------------------------------------------------------------------
def make_vanishing_points(n_inliers=100000, n_vps=6, n_outliers=100000, sigma_deg=0.5, size=1000.0, seed=0):
    rng = np.random.default_rng(seed)
    vps = []
    for k in range(n_vps):
        if k < n_vps // 2:  # finite, within +-3000 px
            v = np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), 1.0])
        else:  # near-infinite
            ang = rng.uniform(0, np.pi)
            v = np.array([np.cos(ang), np.sin(ang), rng.uniform(-1e-6, 1e-6)])
        vps.append(v / np.linalg.norm(v))
    segs, labels = [], []
    per = n_inliers // n_vps
    for k, v in enumerate(vps):
        m = rng.uniform(0, size, (per, 2))
        # direction from the midpoint towards the vanishing point (homogeneous: v_xy - v_z * m)
        dirv = v[:2] - v[2] * m
        ang = np.arctan2(dirv[:, 1], dirv[:, 0]) + np.deg2rad(rng.normal(0, sigma_deg, per))
        half = rng.uniform(20, 120, per)[:, None] / 2
        dvec = np.column_stack([np.cos(ang), np.sin(ang)])
        segs.append(np.column_stack([m - half * dvec, m + half * dvec]))
        labels.append(np.full(per, k + 1))
    m = rng.uniform(0, size, (n_outliers, 2))
    ang = rng.uniform(0, np.pi, n_outliers)
    half = rng.uniform(20, 120, n_outliers)[:, None] / 2
    dvec = np.column_stack([np.cos(ang), np.sin(ang)])
    segs.append(np.column_stack([m - half * dvec, m + half * dvec]))
    labels.append(np.zeros(n_outliers, dtype=int))
    return np.ascontiguousarray(np.vstack(segs)), np.concatenate(labels).astype(np.int32), np.array(vps)


# ---------------------------------------------------------------------------------------------------------------------
# reference data-file format (progx_utils.h:32-96 / dataset_comparison/utils.py:15-27)
# ---------------------------------------------------------------------------------------------------------------------
def load_points_with_labels(path):
    """7-column AdelaideRMF text file `x1 y1 1 x2 y2 1 label` -> (corrs[n,4], labels[n]); label 0 = outlier."""
    M = np.loadtxt(path)
    corrs = np.ascontiguousarray(np.concatenate((M[:, :2], M[:, 3:5]), axis=1))
    return corrs, M[:, -1].astype(np.int32)


def misclassification(segmentation, ref_segmentation):
    """dataset_comparison/utils.py:51-66 — min over label permutations of the fraction of disagreeing points.

    Implemented with the Hungarian algorithm on the confusion matrix, which gives the same minimum as enumerating the
    permutations of the reference ids but scales past 8 labels."""
    from scipy.optimize import linear_sum_assignment
    seg = np.asarray(segmentation).astype(np.int64)
    ref = np.asarray(ref_segmentation).astype(np.int64)
    n = int(ref.max()) + 1
    conf = np.zeros((n, n), dtype=np.int64)  # the reference only permutes ids 0..n-1; other predicted ids never match
    ok = seg < n
    np.add.at(conf, (ref[ok], seg[ok]), 1)
    r, c = linear_sum_assignment(-conf)
    return 1.0 - conf[r, c].sum() / len(seg)


def misclassification_labeling(labeling, annotation, K, K_annot):
    """progx_utils.h:201-274, the (labeling, annotation) overload of getMisclassificationError, restated literally:
    predicted label l means cluster l + 1 (labels with l + 1 > K are left unassigned, -1 = outlier stays 0), every
    ground-truth cluster 1..K_annot greedily takes the unused predicted cluster with the largest overlap (first maximum
    wins; the first candidate is always taken because the reference compares an int -1 with a size_t), matched points
    take the annotation's id, the error is the PERCENTAGE of points whose ids still differ."""
    lab = np.asarray(labeling).astype(np.int64)
    ann = np.asarray(annotation).astype(np.int64)
    all1 = np.full(lab.shape[0], -1, dtype=np.int64)
    ok = lab + 1 <= K
    all1[ok] = lab[ok] + 1
    used = np.zeros(K + 1, dtype=bool)
    pair = {0: 0}
    for i in range(1, K_annot + 1):
        best, best_size = -1, -1
        for j in range(1, K + 1):
            if used[j]:
                continue
            size = int(np.count_nonzero((all1 == j) & (ann == i)))
            if best == -1 or best_size < size:
                best, best_size = j, size
        if best == -1:
            continue
        used[best] = True
        pair[i] = best
    gt_pair = np.array([pair.get(int(a), 0) for a in ann], dtype=np.int64)
    hit = (lab != -1) & (all1 == gt_pair)
    all1 = np.where(hit, ann, all1)
    return 100.0 * float(np.count_nonzero(all1 != ann)) / lab.shape[0]


def misclassification_models(preferences, annotation, K_annot):
    """progx_utils.h:98-199, the (models, annotation) overload: a point is claimed by the FIRST model whose preference
    exceeds FLT_EPSILON; over all permutations of the model ids the smallest number of points whose claimed id differs
    from the annotation (unclaimed points are right iff the annotation is 0), as a PERCENTAGE.  The permutation loop is an
    assignment problem and is solved as one; -1 for more than 9 models, as in the reference."""
    from scipy.optimize import linear_sum_assignment
    pref = np.asarray(preferences, dtype=np.float64)
    ann = np.asarray(annotation).astype(np.int64)
    K = pref.shape[0]
    if K > 9:
        return -1.0
    claimed = pref > np.finfo(np.float32).eps
    owner = np.where(claimed.any(axis=0), np.argmax(claimed, axis=0), -1)      # first model with preference 1
    mk = max(K, K_annot)
    gain = np.zeros((mk, mk), dtype=np.int64)                                   # gain[m, id-1] = points of m annotated id
    for m in range(K):
        ids, cnt = np.unique(ann[owner == m], return_counts=True)
        for a, c in zip(ids, cnt):
            if 1 <= a <= mk:
                gain[m, a - 1] += c
    r, c = linear_sum_assignment(-gain)
    right = int(gain[r, c].sum()) + int(np.count_nonzero((owner == -1) & (ann == 0)))
    return 100.0 * float(ann.shape[0] - right) / ann.shape[0]


MODEL_TYPES = dict(line=_lib.LINE2D, homography=_lib.HOMOGRAPHY, fundamental=_lib.FUNDAMENTAL, pnp=_lib.PNP,
                   vanishing_point=_lib.VANISHING_POINT)
