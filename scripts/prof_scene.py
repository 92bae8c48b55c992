"""cProfile of the bundled scenes under the notebooks' arguments.  usage: python scripts/prof_scene.py book unionhouse ..."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scripts")]
import eval_scenes as E
import numpy as np
if os.environ.get("SCENE_RNG"):
    E.EXTRA["sampler_rng"] = os.environ["SCENE_RNG"]
E.px.findLines(np.random.default_rng(0).random((50, 2)) * 100, np.array(0), 100, 100, sampler_id=0, seed=0)
for scene in sys.argv[1:]:
    fn = E.homography_scene if scene in E.RECORDED_H else (E.two_view_scene if scene in E.RECORDED_F else (lambda sc, seed: E.tless(seed)))
    fn(scene, 0)   # warm (graph buffers, kernels)
    pr = cProfile.Profile()
    pr.enable()
    fn(scene, 1)
    pr.disable()
    print("=====", scene, E.WALL[scene])
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
