"""OracleContext — a stand-in for pyprogressivex._lib.Context whose every method is answered by the CPU oracle.

TEST INFRASTRUCTURE: lets the host control flow (ProgressiveX.run / Pearl.run / ProposalEngine) be exercised on a box
without a GPU, and lets the GPU tests assert that the whole pipeline returns the same model set and labelling as the
CPU restatement for the same hypothesis list.  Never imported by the product package.
"""
import numpy as np

import pgx_oracle as O


class OracleContext:
    def __init__(self, device_id=0):
        self.model_type = None
        self.n = 0
        self.M = 0
        self.nranks, self.rank = 1, 0
        self.slots = {}
        self.comp = None
        self.graph = None
        self.labels = None
        self.Dq = None
        self._stats = dict(mincuts=0, sweeps=0, global_relabels=0, bfs_levels=0, relabelled_sites=0)

    def close(self):
        pass

    def sync(self):
        pass

    def set_points(self, model_type, points):
        self.pts = np.ascontiguousarray(points, dtype=np.float64)
        self.model_type = model_type
        self.n = self.pts.shape[0]
        self.comp = np.zeros(self.n)
        self.slots = {}

    def set_compound(self, compound=None):
        self.comp = np.zeros(self.n) if compound is None else np.asarray(compound, dtype=np.float64).copy()

    def get_compound(self):
        return self.comp.copy()

    def score(self, models, T2, has_compound=False, exponent=2, want_masks=False):
        r = O.score(self.model_type, self.pts, models, T2, compound=self.comp, has_compound=has_compound,
                    exponent=exponent, want_masks=want_masks)
        self.M = len(r["counts"])
        return r

    def preference(self, model, T2, slot, want_pref=False):
        p = O.preference(self.model_type, self.pts, model, T2)
        self.slots[slot] = p
        d, a, b = O.tanimoto_terms(p, self.comp)
        return dict(pref=p.copy() if want_pref else None, dot=d, pref_sqnorm=a, comp_sqnorm=b)

    def get_preference(self, slot):
        return self.slots[slot].copy()

    def compound_update(self, slots, want_compound=False):
        slots = list(slots)
        if len(slots) > 0:
            self.comp = O.compound_max(np.stack([self.slots[s] for s in slots]))
        return self.comp.copy() if want_compound else None

    def pearl_unary(self, models, threshold, lam, want_table=False):
        models = np.zeros((0, O.PARAM_DIM[self.model_type])) if models is None else models
        self.Dq = O.unary_q(self.model_type, self.pts, models, threshold, lam)
        self.L = self.Dq.shape[1]
        return self.Dq.copy() if want_table else None

    def set_unary_q(self, Dq):
        self.Dq = np.ascontiguousarray(Dq, dtype=np.int64)
        self.L = self.Dq.shape[1]

    def set_graph(self, off, idx, mult):
        self.graph = (np.asarray(off, np.int32), np.asarray(idx, np.int32), np.asarray(mult, np.int32))

    def solve_minimal(self, samples, fetch=True):
        self.models = O.solve_minimal(self.model_type, self.pts, samples)
        self.M = len(self.models)
        return self.models.copy() if fetch else None

    def score_launch(self, T2, has_compound=False, want_masks=False):
        self._launched = O.score(self.model_type, self.pts, self.models, T2, compound=self.comp, has_compound=has_compound,
                                 exponent=2, want_masks=want_masks)
        self._launch_has_compound = has_compound

    def score_fetch(self, exponent=2, want_masks=False):
        r = dict(self._launched)
        r["scores"] = r["values"] - np.power(r["shared"], float(exponent)) if self._launch_has_compound else r["values"].copy()
        return r

    def gram(self, kind, sel, params=None, weights=None, wpow=2):
        index = np.asarray(sel[1], dtype=np.int64) if sel[0] == "index" else np.nonzero(self.labels == int(sel[1]))[0]
        return O.gram(kind, self.pts, index, params=params, weights=weights, wpow=wpow)

    def gram_labels(self, kind, K, params=None, weights=None, wpow=2):
        res = [self.gram(kind, ("label", k), params=None if params is None else np.asarray(params)[k], weights=weights, wpow=wpow)
               for k in range(K)]
        return (np.array([r[0] for r in res]), np.array([r[1] for r in res], dtype=np.int64),
                np.array([r[2] for r in res], dtype=np.int64))

    def epipolar_support(self, F, T2, S2):
        return O.epipolar_support(self.pts, F, T2, S2)

    def residual_sums(self, models):
        return np.array([self.residual_sum(m, k) for k, m in enumerate(np.asarray(models))])

    def gram_batch(self, kind, index, params=None, weights=None, wpow=2):
        index = np.asarray(index, dtype=np.int64)
        res = [O.gram(kind, self.pts, index[b], params=None if params is None else np.asarray(params)[b], weights=weights,
                      wpow=wpow) for b in range(index.shape[0])]
        return np.array([r[0] for r in res]), np.array([r[2] for r in res], dtype=np.int32)

    def eigh_smallest_batch(self, A):
        vec, val, _ = O.eigh_smallest(A)
        return vec, val

    def graph_build(self, points, kind, radius=0.0, k=5, fetch=True):
        self.graph = O.graph_build(points, kind, radius=radius, k=k)
        return self.graph if fetch else len(self.graph[1])

    def set_labels(self, labels):
        self.labels = np.asarray(labels, dtype=np.int32).copy()

    def get_labels(self):
        return self.labels.copy()

    def _g(self, lam):
        if lam > 0 and self.graph is not None and len(self.graph[1]) > 0:
            return self.graph
        return (np.zeros(self.Dq.shape[0] + 1, np.int32), np.zeros(1, np.int32), np.ones(1, np.int32))

    def energy(self, lam, label_cost):
        e = O.energy(self.Dq, self._g(lam), O.quantize_lambda(lam), O.quantize(label_cost), self.labels)
        return e, e / 2.0 ** 32

    def expand_alpha(self, lam, label_cost, alpha):
        self.labels, ch, _ = O.expand_alpha(self.Dq, self._g(lam), O.quantize_lambda(lam), O.quantize(label_cost),
                                            alpha, self.labels)
        return ch

    def expansion(self, lam, label_cost, max_cycles=1000):
        self.labels, e, cyc = O.expansion(self.Dq, self._g(lam), O.quantize_lambda(lam), O.quantize(label_cost),
                                          self.labels, max_cycles)
        return e, e / 2.0 ** 32, cyc

    def greedy_labeling(self, label_cost):
        self.labels, e, opened = O.greedy_labeling(self.Dq, O.quantize(label_cost))
        return e, e / 2.0 ** 32, opened

    def expansion_stats(self):
        return dict(self._stats)

    def bucket(self, L, want_order=True):
        counts, order = O.bucket(self.labels, L)
        return counts, (order if want_order else None)

    def gc_labeling(self, model, T2, lam):
        return O.gc_labeling(self.model_type, self.pts, model, T2, lam, self.graph)

    def gc_inliers(self, model, T2, lam):
        """the GPU context's pgx_gc_inliers: the same cut as ascending indices"""
        return np.flatnonzero(self.gc_labeling(model, T2, lam) != 0).astype(np.int64)

    def pnp_refine_batch(self, inits, index, weights=None, wpow=2, iterations=10):
        """the GPU context's pgx_pnp_refine_batch: the Gauss-Newton iteration of PnPEstimator._fit_many (the host statement of the
        same mathematics: numpy pinv instead of the kernel's Jacobi pseudo-inverse) on the oracle's Gram sums"""
        from pyprogressivex import _estimators
        index = np.asarray(index, dtype=np.int64)
        inits = np.asarray(inits, dtype=np.float64).reshape(index.shape[0], 12)

        def gram(kind, prm, use_w, wp, rows):
            G, bad = self.gram_batch(kind, index[rows], params=prm, weights=weights if use_w else None, wpow=wp)
            return G, np.full(len(rows), index.shape[1], dtype=np.int64), bad
        fits = _estimators.PnPEstimator()._fit_many(gram, index.shape[0], [inits[b] for b in range(index.shape[0])], iterations=int(iterations))
        ok = np.array([len(f) == 1 for f in fits], dtype=bool)
        P = np.array([f[0] if len(f) == 1 else inits[b] for b, f in enumerate(fits)])
        return P, ok

    def residual_sum(self, model, label):
        return O.residual_sum(self.model_type, self.pts, model, self.labels, label)
