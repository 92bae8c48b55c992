import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'progressive-x_amd')]
import numpy as np
import pyprogressivex as px
from pyprogressivex import datasets
pts, gt, Fs = datasets.make_two_view_motions(seed=0)
for kw in (dict(), dict(spatial_coherence_weight=0.1, neighborhood_ball_radius=20.0), dict(threshold=0.5)):
    args = dict(threshold=0.75, conf=0.99, sampler_id=0, seed=1, minimum_point_number=1000, max_iters=2000)
    args.update(kw)
    F, lab = px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, **args)
    K = F.shape[0] // 3
    conf = np.zeros((K + 1, 9), int)
    np.add.at(conf, (lab, gt), 1)
    print(kw, "models", K, "ME", datasets.misclassification(np.where(lab == K, 0, lab + 1), gt))
    print(conf)
