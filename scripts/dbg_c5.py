"""Failure analysis of the synthetic C5 run: findVanishingPoints on 2e5 segments / 6 vanishing points / 50 %
outliers returns 9 instances (ME 0.33).  Prints the confusion matrix (rows = returned instance, columns = ground-truth vanishing
point, 0 = outlier), the angular distance of every returned instance to its nearest ground-truth point, and how the result
moves with the arguments that plausibly matter (label cost = minimum_point_number, threshold, spatial weight, model cap)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'progressive-x_amd')]
import numpy as np
import pyprogressivex as px
from pyprogressivex import datasets
pts, gt, vps = datasets.make_vanishing_points(seed=0)
G = vps / np.linalg.norm(vps, axis=1, keepdims=True)
base = dict(threshold=1.5, conf=0.99, sampler_id=0, seed=1, minimum_point_number=2000, spatial_coherence_weight=0.05,
            neighborhood_ball_radius=10.0)
runs = [dict(), dict(minimum_point_number=8000), dict(spatial_coherence_weight=0.0), dict(threshold=1.0),
        dict(maximum_model_number=6), dict(minimum_point_number=8000, maximum_model_number=6)]
if len(sys.argv) > 1:
    runs = runs[:int(sys.argv[1])]
for kw in runs:
    args = dict(base)
    args.update(kw)
    V, lab = px.findVanishingPoints(pts, np.array(0), 1000, 1000, **args)
    K = V.shape[0]
    conf = np.zeros((K + 1, len(vps) + 1), int)
    np.add.at(conf, (lab, gt), 1)
    Vn = V / np.linalg.norm(V, axis=1, keepdims=True)
    ang = np.degrees(np.arccos(np.clip(np.abs(Vn @ G.T), 0, 1)))           # [K, 6] angle between homogeneous directions
    print(kw, "models", K, "ME", round(float(datasets.misclassification(np.where(lab == K, 0, lab + 1), gt)), 4))
    print(" nearest gt (index, degrees):", [(int(a.argmin()) + 1, round(float(a.min()), 3)) for a in ang])
    print(conf)
