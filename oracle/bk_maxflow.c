/*
 * bk_maxflow.c — Boykov–Kolmogorov augmenting-path max-flow with two search trees (PAMI 26(9), 2004), int64 capacities.
 *
 * TEST INFRASTRUCTURE ONLY (part of libpgx_oracle.so).  This is the ALGORITHM the reference reaches through
 * GCoptimizationGeneralGraph::expansion (/root/reference/src/pyprogressivex/include/PEARL.h:550-551 -> GCO-v3 -> its
 * bundled maxflow-v3 `Graph` class); the GCO / maxflow sources live in the absent graph-cut-ransac submodule, so this file is
 * written from the published algorithm: two trees S and T grown from the terminals (growth), a path found where they touch
 * (augmentation), orphans re-attached or freed (adoption), with the paper's timestamp / distance heuristics for the adoption
 * stage.  It exists for ONE reason: the CPU labelling baseline of bench.py must be timed on the solver
 * family the reference uses, not on the oracle's Dinic — BK is several times faster on these shallow vision graphs and a
 * ratio against Dinic would flatter the GPU.  Correctness: the minimal sink side of a maximum flow is unique, so BK and Dinic
 * must return the same labels; tests/test_oracle_bk.py checks flow values and cuts against Dinic and scipy.
 *
 * Terminal arcs are kept as one signed residual per node (tr_cap > 0: residual from the source, < 0: residual to the sink),
 * flow through s -> i -> t is cancelled when the capacities are added — as in the paper's implementation notes.
 */
#include "bk_maxflow.h"

#include <stdlib.h>
#include <string.h>

#define BK_NONE (-1)
#define BK_TERMINAL (-2)   /* parent "arc": the node hangs directly off its terminal */
#define BK_ORPHAN (-3)
#define BK_INF_D 1000000000

struct bk_graph {
    int nn;
    int64_t na, arc_cap;
    /* nodes */
    int32_t *first;        /* first outgoing arc */
    int64_t *parent;       /* arc towards the parent (an index into the arc arrays), BK_TERMINAL, BK_ORPHAN or BK_NONE (free) */
    int32_t *next_active;  /* intrusive FIFO of active nodes; == own index marks the tail; BK_NONE = not queued */
    int32_t *ts, *dist;
    uint8_t *is_sink;
    int64_t *tr_cap;
    /* arcs (pairs a, a ^ 1) */
    int32_t *head, *next;
    int64_t *r_cap;
    /* queues */
    int32_t q_first[2], q_last[2];
    int32_t *orphans; int64_t n_orph, orph_cap;  /* FIFO processed front to back; adoption may append */
    int64_t orph_head;
    int32_t time;
    int64_t flow;
};

bk_graph *bk_create(int nn, int64_t max_arcs)
{
    bk_graph *g = (bk_graph *)calloc(1, sizeof(bk_graph));
    const size_t N = (size_t)(nn > 0 ? nn : 1), A = (size_t)(max_arcs > 0 ? max_arcs : 1) * 2;
    g->nn = nn; g->na = 0; g->arc_cap = (int64_t)A;
    g->first = (int32_t *)malloc(N * sizeof(int32_t));
    g->parent = (int64_t *)malloc(N * sizeof(int64_t));
    g->next_active = (int32_t *)malloc(N * sizeof(int32_t));
    g->ts = (int32_t *)calloc(N, sizeof(int32_t));
    g->dist = (int32_t *)calloc(N, sizeof(int32_t));
    g->is_sink = (uint8_t *)calloc(N, 1);
    g->tr_cap = (int64_t *)calloc(N, sizeof(int64_t));
    g->head = (int32_t *)malloc(A * sizeof(int32_t));
    g->next = (int32_t *)malloc(A * sizeof(int32_t));
    g->r_cap = (int64_t *)malloc(A * sizeof(int64_t));
    for (int i = 0; i < nn; ++i) { g->first[i] = BK_NONE; g->parent[i] = BK_NONE; g->next_active[i] = BK_NONE; }
    g->orph_cap = 1024;
    g->orphans = (int32_t *)malloc((size_t)g->orph_cap * sizeof(int32_t));
    return g;
}

void bk_destroy(bk_graph *g)
{
    if (!g) return;
    free(g->first); free(g->parent); free(g->next_active); free(g->ts); free(g->dist); free(g->is_sink); free(g->tr_cap);
    free(g->head); free(g->next); free(g->r_cap); free(g->orphans);
    free(g);
}

void bk_add_tweights(bk_graph *g, int i, int64_t cap_source, int64_t cap_sink)
{
    const int64_t delta = g->tr_cap[i];
    if (delta > 0) cap_source += delta; else cap_sink -= delta;
    g->flow += cap_source < cap_sink ? cap_source : cap_sink;
    g->tr_cap[i] = cap_source - cap_sink;
}

int bk_add_edge(bk_graph *g, int i, int j, int64_t cap, int64_t rev_cap)
{
    if (g->na + 2 > g->arc_cap) return -1;
    const int64_t a = g->na;
    g->head[a] = j; g->r_cap[a] = cap; g->next[a] = g->first[i]; g->first[i] = (int32_t)a;
    g->head[a + 1] = i; g->r_cap[a + 1] = rev_cap; g->next[a + 1] = g->first[j]; g->first[j] = (int32_t)(a + 1);
    g->na += 2;
    return 0;
}

/* ---- active nodes: two FIFO lists; nodes found active during a pass go to the second and are taken when the first is empty */
static void set_active(bk_graph *g, int i)
{
    if (g->next_active[i] != BK_NONE) return;            /* already queued */
    if (g->q_last[1] != BK_NONE) g->next_active[g->q_last[1]] = i; else g->q_first[1] = i;
    g->q_last[1] = i;
    g->next_active[i] = i;                               /* tail marker */
}

static int next_active(bk_graph *g)
{
    for (;;) {
        int i = g->q_first[0];
        if (i == BK_NONE) {
            g->q_first[0] = i = g->q_first[1];
            g->q_last[0] = g->q_last[1];
            g->q_first[1] = g->q_last[1] = BK_NONE;
            if (i == BK_NONE) return BK_NONE;
        }
        if (g->next_active[i] == i) g->q_first[0] = g->q_last[0] = BK_NONE;   /* it was the tail */
        else g->q_first[0] = g->next_active[i];
        g->next_active[i] = BK_NONE;
        if (g->parent[i] != BK_NONE) return i;           /* still in a tree: really active */
    }
}

static void push_orphan(bk_graph *g, int i)
{
    if (g->n_orph == g->orph_cap) {
        g->orph_cap *= 2;
        g->orphans = (int32_t *)realloc(g->orphans, (size_t)g->orph_cap * sizeof(int32_t));
    }
    g->parent[i] = BK_ORPHAN;
    g->orphans[g->n_orph++] = i;
}

/* ---- augmentation along (source tree) -> middle arc -> (sink tree) */
static void augment(bk_graph *g, int64_t middle)
{
    int64_t bottleneck = g->r_cap[middle];
    int i;
    int64_t a;
    /* source tree: from the tail of `middle` up to the source */
    for (i = g->head[middle ^ 1];;) {
        a = g->parent[i];
        if (a == BK_TERMINAL) break;
        if (bottleneck > g->r_cap[a ^ 1]) bottleneck = g->r_cap[a ^ 1];
        i = g->head[a];
    }
    if (bottleneck > g->tr_cap[i]) bottleneck = g->tr_cap[i];
    /* sink tree: from the head of `middle` down to the sink */
    for (i = g->head[middle];;) {
        a = g->parent[i];
        if (a == BK_TERMINAL) break;
        if (bottleneck > g->r_cap[a]) bottleneck = g->r_cap[a];
        i = g->head[a];
    }
    if (bottleneck > -g->tr_cap[i]) bottleneck = -g->tr_cap[i];

    /* push */
    g->r_cap[middle ^ 1] += bottleneck;
    g->r_cap[middle] -= bottleneck;
    for (i = g->head[middle ^ 1];;) {
        a = g->parent[i];
        if (a == BK_TERMINAL) break;
        g->r_cap[a] += bottleneck;
        g->r_cap[a ^ 1] -= bottleneck;
        if (g->r_cap[a ^ 1] == 0) push_orphan(g, i);
        i = g->head[a];
    }
    g->tr_cap[i] -= bottleneck;
    if (g->tr_cap[i] == 0) push_orphan(g, i);
    for (i = g->head[middle];;) {
        a = g->parent[i];
        if (a == BK_TERMINAL) break;
        g->r_cap[a ^ 1] += bottleneck;
        g->r_cap[a] -= bottleneck;
        if (g->r_cap[a] == 0) push_orphan(g, i);
        i = g->head[a];
    }
    g->tr_cap[i] += bottleneck;
    if (g->tr_cap[i] == 0) push_orphan(g, i);
    g->flow += bottleneck;
}

/* ---- adoption: find a new parent of the same tree whose path to the terminal is intact, else free the node */
static void process_orphan(bk_graph *g, int i, int sink)
{
    int64_t a0_min = BK_NONE;
    int d_min = BK_INF_D;
    for (int64_t a0 = g->first[i]; a0 != BK_NONE; a0 = g->next[a0]) {
        const int64_t toward = sink ? a0 : (a0 ^ 1);      /* the arc that would carry flow: j -> i (source tree) / i -> j (sink tree) */
        if (g->r_cap[toward] == 0) continue;
        int j = g->head[a0];
        if (g->is_sink[j] != sink || g->parent[j] == BK_NONE) continue;
        /* walk j's path to the terminal; valid if it does not run into an orphan */
        int d = 0;
        for (;;) {
            if (g->ts[j] == g->time) { d += g->dist[j]; break; }
            const int64_t a = g->parent[j];
            d++;
            if (a == BK_TERMINAL) { g->ts[j] = g->time; g->dist[j] = 1; break; }
            if (a == BK_ORPHAN) { d = BK_INF_D; break; }
            j = g->head[a];
        }
        if (d < BK_INF_D) {
            if (d < d_min) { a0_min = a0; d_min = d; }
            /* stamp the walked path */
            for (j = g->head[a0]; g->ts[j] != g->time; j = g->head[g->parent[j]]) {
                g->ts[j] = g->time;
                g->dist[j] = d--;
            }
        }
    }
    g->parent[i] = a0_min;
    if (a0_min != BK_NONE) {
        g->ts[i] = g->time;
        g->dist[i] = d_min + 1;
        return;
    }
    /* no parent: the node becomes free; its children become orphans, neighbours that could re-grow into it become active */
    for (int64_t a0 = g->first[i]; a0 != BK_NONE; a0 = g->next[a0]) {
        const int j = g->head[a0];
        const int64_t a = g->parent[j];
        if (g->is_sink[j] == sink && a != BK_NONE) {
            const int64_t toward = sink ? a0 : (a0 ^ 1);
            if (g->r_cap[toward] != 0) set_active(g, j);
            if (a != BK_TERMINAL && a != BK_ORPHAN && g->head[a] == i) push_orphan(g, j);
        }
    }
}

int64_t bk_maxflow(bk_graph *g)
{
    g->q_first[0] = g->q_last[0] = g->q_first[1] = g->q_last[1] = BK_NONE;
    g->n_orph = 0; g->orph_head = 0;
    g->time = 0;
    for (int i = 0; i < g->nn; ++i) {
        g->next_active[i] = BK_NONE;
        g->ts[i] = 0;
        if (g->tr_cap[i] > 0) { g->is_sink[i] = 0; g->parent[i] = BK_TERMINAL; g->dist[i] = 1; set_active(g, i); }
        else if (g->tr_cap[i] < 0) { g->is_sink[i] = 1; g->parent[i] = BK_TERMINAL; g->dist[i] = 1; set_active(g, i); }
        else g->parent[i] = BK_NONE;
    }
    int current = BK_NONE;
    for (;;) {
        int i = current;
        if (i != BK_NONE) {
            g->next_active[i] = BK_NONE;                  /* taken out of the queue to be looked at again */
            if (g->parent[i] == BK_NONE) i = BK_NONE;
        }
        if (i == BK_NONE) {
            i = next_active(g);
            if (i == BK_NONE) break;
        }
        /* growth */
        int64_t found = BK_NONE;
        if (!g->is_sink[i]) {
            for (int64_t a = g->first[i]; a != BK_NONE; a = g->next[a]) {
                if (g->r_cap[a] == 0) continue;
                const int j = g->head[a];
                if (g->parent[j] == BK_NONE) {
                    g->is_sink[j] = 0; g->parent[j] = a ^ 1; g->ts[j] = g->ts[i]; g->dist[j] = g->dist[i] + 1;
                    set_active(g, j);
                } else if (g->is_sink[j]) { found = a; break; }
                else if (g->ts[j] <= g->ts[i] && g->dist[j] > g->dist[i]) {   /* heuristic: a shorter way to the source */
                    g->parent[j] = a ^ 1; g->ts[j] = g->ts[i]; g->dist[j] = g->dist[i] + 1;
                }
            }
        } else {
            for (int64_t a = g->first[i]; a != BK_NONE; a = g->next[a]) {
                if (g->r_cap[a ^ 1] == 0) continue;
                const int j = g->head[a];
                if (g->parent[j] == BK_NONE) {
                    g->is_sink[j] = 1; g->parent[j] = a ^ 1; g->ts[j] = g->ts[i]; g->dist[j] = g->dist[i] + 1;
                    set_active(g, j);
                } else if (!g->is_sink[j]) { found = a ^ 1; break; }
                else if (g->ts[j] <= g->ts[i] && g->dist[j] > g->dist[i]) {
                    g->parent[j] = a ^ 1; g->ts[j] = g->ts[i]; g->dist[j] = g->dist[i] + 1;
                }
            }
        }
        g->time++;
        if (found != BK_NONE) {
            g->next_active[i] = i;                        /* keep it out of the queues while it stays `current` */
            current = i;
            augment(g, found);
            while (g->orph_head < g->n_orph) {
                const int o = g->orphans[g->orph_head++];
                process_orphan(g, o, g->is_sink[o]);
            }
            g->n_orph = 0; g->orph_head = 0;
        } else
            current = BK_NONE;
    }
    return g->flow;
}

int bk_in_sink_tree(const bk_graph *g, int i)
{
    return g->parent[i] != BK_NONE && g->is_sink[i];
}
