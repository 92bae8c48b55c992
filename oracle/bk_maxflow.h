/* bk_maxflow.h — Boykov-Kolmogorov max-flow (see bk_maxflow.c).  TEST INFRASTRUCTURE ONLY. */
#ifndef BK_MAXFLOW_H
#define BK_MAXFLOW_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct bk_graph bk_graph;
bk_graph *bk_create(int nn, int64_t max_arc_pairs);
void bk_destroy(bk_graph *g);
void bk_add_tweights(bk_graph *g, int i, int64_t cap_source, int64_t cap_sink);
int bk_add_edge(bk_graph *g, int i, int j, int64_t cap, int64_t rev_cap);      /* -1: arc capacity exceeded */
int64_t bk_maxflow(bk_graph *g);
int bk_in_sink_tree(const bk_graph *g, int i);   /* what_segment(i, default = SOURCE) == SINK */
#ifdef __cplusplus
}
#endif
#endif
