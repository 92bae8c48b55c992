"""Sensitivity of the bundled two-view scenes to the inlier threshold (docs/experiments-cubetoy.md section 6): the notebook's call with
threshold scaled, CPU-oracle harness, 10 seeds.  usage: python scripts/exp_cubetoy_threshold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'scripts'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import eval_scenes as E
import numpy as np
from oracle_ctx import OracleContext
E._api._ctx = OracleContext()
px, datasets = E.px, E.datasets
def run(scene, seed, thr):
    corrs, gt = datasets.load_points_with_labels(os.path.join(E.SCENES, f"{scene}.txt"))
    order = E.density_order(corrs, 50.0)
    corrs, gt = np.ascontiguousarray(corrs[order]), gt[order]
    F, lab = px.findTwoViewMotions(corrs, 1024, 768, 1024, 768, threshold=thr, conf=0.5, spatial_coherence_weight=0.5,
                                   neighborhood_ball_radius=50.0, maximum_tanimoto_similarity=0.4, max_iters=10000,
                                   minimum_point_number=7, maximum_model_number=4, sampler_id=2, scoring_exponent=1.0, seed=seed)
    return round(float(datasets.misclassification(lab, gt)), 3), F.shape[0] // 3
for thr in (float(x) for x in (sys.argv[1:] or ["0.75", "0.5", "0.45", "0.375", "0.3", "0.25"])):
    for scene in ("cubetoy", "breadcube", "book"):
        r = [run(scene, s, thr) for s in range(10)]
        print(f"thr {thr:5.3f} {scene:10s} median {np.median([x[0] for x in r]):.3f}  {[x[0] for x in r]}  models {[x[1] for x in r]}", flush=True)
