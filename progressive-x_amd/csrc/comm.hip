// comm.hip — multi-GPU exchange for the sharded proposal loop (no reference counterpart: the reference
// is single-threaded CPU code, /root/reference/src/pyprogressivex/include/progressive_x.h:251-489).
//
// One process per GPU.  Hypotheses are sharded over ranks, every rank holds all points and the compound preference
// vector, so the only data-path exchange per batch is an RCCL all-gather of the per-hypothesis (count, value, shared)
// triples (24 B x M per rank) over xGMI; the compound vector is kept identical on all ranks by recomputing it
// redundantly (max is exact in any order) or, if ranks accepted different models, by an all-reduce(max).
//
// RCCL is bound lazily with dlopen so that single-GPU use never depends on it being loadable.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>

#include "pgx_internal.h"

namespace pgx {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int load_rccl(pgx_ctx* ctx)
{
    if (g_rccl.handle) return PGX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(ctx, PGX_ERR_COMM, "cannot load librccl: %s", dlerror());
    RcclApi a;
    a.handle = h;
#define BIND(field, sym)                                                                         \
    *(void**)(&a.field) = dlsym(h, sym);                                                         \
    if (!a.field) { dlclose(h); return fail(ctx, PGX_ERR_COMM, "librccl lacks symbol %s", sym); }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllGather, "ncclAllGather")
    BIND(AllReduce, "ncclAllReduce")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    g_rccl = a;
    return PGX_OK;
}

// One in-flight exchange (pgx_score_allgather_begin / _end): the (count | value | shared) block of a launch is copied aside on
// the context's stream, all-gathered and copied into pinned memory on the EXCHANGE stream, so the next batch is scored meanwhile.
struct ExchangeSlot {
    DevBuf stage, gathered;
    void* host = nullptr;
    size_t host_cap = 0;
    hipEvent_t scored = nullptr, done = nullptr;
    int M = 0, Mpad = 0, has_compound = 0, busy = 0;
    int reduced = 0;   // begun by pgx_score_allreduce_begin: `host` holds ONE block counts | values | shared, not one per rank
    int local_fail = 0; // this rank contributed the poison word instead of accumulators (export_or_poison)
};

struct CommState {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    DevBuf tmp;
    hipStream_t xstream = nullptr;
    ExchangeSlot slot[2];
};

// Collectives on ONE communicator must be issued in the same order on every rank.  The pipelined exchanges run on the exchange
// stream; a collective on the context's stream while one of them is in flight could overtake it on some ranks only and hang the
// job, so those calls are refused until the slots have been collected.
static bool exchange_in_flight(const CommState* cs) { return cs->slot[0].busy || cs->slot[1].busy; }
#define PGX_NO_EXCHANGE(ctx, name)                                                                                        \
    if (exchange_in_flight((ctx)->comm))                                                                                  \
        return fail(ctx, PGX_ERR_INVALID, name ": a pipelined exchange is still in flight (collect it with pgx_score_all*_end first)")

#define PGX_NCCL(ctx, call)                                                                              \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fail(ctx, PGX_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r_));           \
    } while (0)

void comm_free(pgx_ctx* ctx)
{
    if (!ctx->comm) return;
    if (ctx->comm->xstream) (void)hipStreamSynchronize(ctx->comm->xstream);
    if (ctx->comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm->comm);
    release(ctx->comm->tmp);
    for (ExchangeSlot& e : ctx->comm->slot) {
        release(e.stage);
        release(e.gathered);
        if (e.host) (void)hipHostFree(e.host);
        if (e.scored) (void)hipEventDestroy(e.scored);
        if (e.done) (void)hipEventDestroy(e.done);
    }
    if (ctx->comm->xstream) (void)hipStreamDestroy(ctx->comm->xstream);
    delete ctx->comm;
    ctx->comm = nullptr;
}

}  // namespace pgx

using namespace pgx;

static_assert(sizeof(ncclUniqueId) == PGX_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" {

int pgx_comm_unique_id(uint8_t id[PGX_UNIQUE_ID_BYTES])
{
    if (!id) return fail(nullptr, PGX_ERR_INVALID, "pgx_comm_unique_id: NULL");
    PGX_TRY(load_rccl(nullptr));
    ncclUniqueId u;
    PGX_NCCL(nullptr, g_rccl.GetUniqueId(&u));
    memcpy(id, &u, PGX_UNIQUE_ID_BYTES);
    return PGX_OK;
}

int pgx_comm_init(pgx_ctx* ctx, int nranks, int rank, const uint8_t id[PGX_UNIQUE_ID_BYTES])
{
    if (!ctx || !id) return fail(ctx, PGX_ERR_INVALID, "pgx_comm_init: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, PGX_ERR_INVALID, "pgx_comm_init: bad rank %d/%d", rank, nranks);
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    PGX_TRY(load_rccl(ctx));
    comm_free(ctx);
    ctx->comm = new CommState();
    ctx->comm->nranks = nranks;
    ctx->comm->rank = rank;
    ncclUniqueId u;
    memcpy(&u, id, PGX_UNIQUE_ID_BYTES);
    PGX_NCCL(ctx, g_rccl.CommInitRank(&ctx->comm->comm, nranks, u, rank));
    return PGX_OK;
}

int pgx_comm_destroy(pgx_ctx* ctx)
{
    if (!ctx) return fail(nullptr, PGX_ERR_INVALID, "ctx is NULL");
    (void)hipSetDevice(ctx->device);
    comm_free(ctx);
    return PGX_OK;
}

int pgx_comm_barrier(pgx_ctx* ctx)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_comm_barrier: communicator not initialised");
    PGX_NO_EXCHANGE(ctx, "pgx_comm_barrier");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    PGX_TRY(ensure(ctx, ctx->comm->tmp, 64));
    PGX_HIP(ctx, hipMemsetAsync(ctx->comm->tmp.p, 0, 8, ctx->stream));
    PGX_NCCL(ctx, g_rccl.AllReduce(ctx->comm->tmp.p, ctx->comm->tmp.p, 1, ncclInt32, ncclSum, ctx->comm->comm, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

int pgx_comm_allreduce_max_f64(pgx_ctx* ctx, double* value)
{
    if (!ctx || !ctx->comm || !value) return fail(ctx, PGX_ERR_INVALID, "pgx_comm_allreduce_max_f64: communicator not initialised");
    PGX_NO_EXCHANGE(ctx, "pgx_comm_allreduce_max_f64");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    PGX_TRY(ensure(ctx, ctx->comm->tmp, 64));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->comm->tmp.p, value, 8, hipMemcpyHostToDevice, ctx->stream));
    PGX_NCCL(ctx, g_rccl.AllReduce(ctx->comm->tmp.p, ctx->comm->tmp.p, 1, ncclFloat64, ncclMax, ctx->comm->comm, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(value, ctx->comm->tmp.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

} // extern "C"

// common head of the two pipelined exchanges: the slot's stream, events and buffers (stage: what leaves this rank, gathered: what
// comes back, host: its pinned copy)
static int slot_prepare(pgx_ctx* ctx, const char* who, int slot, size_t stage_bytes, size_t result_bytes, ExchangeSlot** out)
{
    if (slot < 0 || slot > 1) return fail(ctx, PGX_ERR_INVALID, "%s: slot %d (0 or 1)", who, slot);
    if (ctx->M <= 0 || !ctx->counts.p) return fail(ctx, PGX_ERR_INVALID, "%s: nothing launched", who);
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    CommState* cs = ctx->comm;
    ExchangeSlot& e = cs->slot[slot];
    if (e.busy) return fail(ctx, PGX_ERR_INVALID, "%s: slot %d is still in flight (collect it with the matching _end call)", who, slot);
    if (!cs->xstream) PGX_HIP(ctx, hipStreamCreateWithFlags(&cs->xstream, hipStreamNonBlocking));
    if (!e.scored) PGX_HIP(ctx, hipEventCreateWithFlags(&e.scored, hipEventDisableTiming));
    if (!e.done) PGX_HIP(ctx, hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
    PGX_TRY(ensure(ctx, e.stage, stage_bytes));
    PGX_TRY(ensure(ctx, e.gathered, result_bytes));
    if (e.host_cap < result_bytes) {
        if (e.host) (void)hipHostFree(e.host);
        e.host = nullptr; e.host_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&e.host, result_bytes * 2, hipHostMallocDefault));
        e.host_cap = result_bytes * 2;
    }
    *out = &e;
    return PGX_OK;
}

// rows of M (count, value, shared) triples out of a [3][Mpad] block, with the score of scoring_function_with_compound_model.h:110,120
static void unpack_block(const void* block, size_t M, size_t Mp, int has_compound, int exponent, int64_t* counts, double* values, double* shared, double* scores)
{
    const int64_t* c = (const int64_t*)block;
    const double* v = (const double*)block + Mp;
    const double* sh = (const double*)block + 2 * Mp;
    if (counts) memcpy(counts, c, M * 8);
    if (values) memcpy(values, v, M * 8);
    if (shared) memcpy(shared, sh, M * 8);
    if (scores)
        for (size_t m = 0; m < M; ++m)
            // (shared == +0: pow(+0, e) = +0 and v - (+0) = v bit for bit - capi.hip finish_scores)
            scores[m] = has_compound && !(sh[m] == 0.0 && !std::signbit(sh[m]) && exponent > 0) ? v[m] - std::pow(sh[m], (double)exponent) : v[m];
}

extern "C" {

// counts | values | shared of a launch are ONE allocation of 3 x Mpad words (score_launch): one all-gather of that block per
// step (three grouped ones before) into [rank][3][Mpad], one copy back into pinned memory (three copies into pageable
// vectors before: staged, ~15 us each).  Every rank scores a shard of the same padded length, so Mpad agrees across ranks.
int pgx_score_allgather(pgx_ctx* ctx)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allgather: communicator not initialised");
    PGX_NO_EXCHANGE(ctx, "pgx_score_allgather");
    if (ctx->M <= 0 || !ctx->counts.p) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allgather: nothing launched");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t W = (size_t)3 * (size_t)ctx->Mpad, G = (size_t)ctx->comm->nranks;
    PGX_TRY(ensure(ctx, ctx->g_counts, G * W * 8));
    PGX_NCCL(ctx, g_rccl.AllGather(ctx->counts.p, ctx->g_counts.p, W, ncclInt64, ctx->comm->comm, ctx->stream));
    return PGX_OK;
}

int pgx_score_fetch_all(pgx_ctx* ctx, int exponent, int64_t* counts, double* values, double* shared, double* scores)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_fetch_all: communicator not initialised");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)ctx->M, Mp = (size_t)ctx->Mpad, G = (size_t)ctx->comm->nranks;
    const size_t T = M * G, need = G * 3 * Mp * 8;
    if (T == 0 || !ctx->g_counts.p || ctx->g_counts.cap < need) return fail(ctx, PGX_ERR_INVALID, "pgx_score_fetch_all: nothing gathered");
    if (ctx->h_res_cap < need) {
        if (ctx->h_res) (void)hipHostFree(ctx->h_res);
        ctx->h_res = nullptr; ctx->h_res_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&ctx->h_res, need * 2, hipHostMallocDefault));
        ctx->h_res_cap = need * 2;
    }
    PGX_HIP(ctx, hipMemcpyAsync(ctx->h_res, ctx->g_counts.p, need, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t r = 0; r < G; ++r)   // rank-major [M] rows out of [rank][3][Mpad]
        unpack_block((const int64_t*)ctx->h_res + r * 3 * Mp, M, Mp, ctx->score_has_compound, exponent, counts ? counts + r * M : nullptr,
                     values ? values + r * M : nullptr, shared ? shared + r * M : nullptr, scores ? scores + r * M : nullptr);
    return PGX_OK;
}

// ---- the same exchange, overlapped: two batches in flight ----------------------------------------------------------------------
// begin(slot): right behind pgx_score_launch.  The result block is copied aside on the context's stream (48 KB, in order behind
// the finish kernel: the next launch may overwrite the block at once), an event hands it to the exchange stream, which runs the
// all-gather and the copy into the slot's pinned buffer.  end(slot) waits for that slot only and unpacks it.  Between the two
// the caller uploads / generates and launches the NEXT batch: its scoring hides the exchange of this one (the all-gather of a
// 48 KB block is latency - ~16-20 us with the copy - not bandwidth).  Results are those of pgx_score_allgather + _fetch_all.
int pgx_score_allgather_begin(pgx_ctx* ctx, int slot)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allgather_begin: communicator not initialised");
    CommState* cs = ctx->comm;
    const size_t W = (size_t)3 * (size_t)ctx->Mpad, need = (size_t)cs->nranks * W * 8;
    ExchangeSlot* ep = nullptr;
    PGX_TRY(slot_prepare(ctx, "pgx_score_allgather_begin", slot, W * 8, need, &ep));
    ExchangeSlot& e = *ep;
    PGX_HIP(ctx, hipMemcpyAsync(e.stage.p, ctx->counts.p, W * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PGX_HIP(ctx, hipEventRecord(e.scored, ctx->stream));
    PGX_HIP(ctx, hipStreamWaitEvent(cs->xstream, e.scored, 0));
    PGX_NCCL(ctx, g_rccl.AllGather(e.stage.p, e.gathered.p, W, ncclInt64, cs->comm, cs->xstream));
    PGX_HIP(ctx, hipMemcpyAsync(e.host, e.gathered.p, need, hipMemcpyDeviceToHost, cs->xstream));
    PGX_HIP(ctx, hipEventRecord(e.done, cs->xstream));
    e.M = ctx->M; e.Mpad = ctx->Mpad; e.has_compound = ctx->score_has_compound; e.busy = 1; e.reduced = 0;
    return PGX_OK;
}

int pgx_score_allgather_end(pgx_ctx* ctx, int slot, int exponent, int64_t* counts, double* values, double* shared, double* scores)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allgather_end: communicator not initialised");
    if (slot < 0 || slot > 1 || !ctx->comm->slot[slot].busy || ctx->comm->slot[slot].reduced)
        return fail(ctx, PGX_ERR_INVALID, "pgx_score_allgather_end: slot %d holds no gathered table", slot);
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    ExchangeSlot& e = ctx->comm->slot[slot];
    PGX_HIP(ctx, hipEventSynchronize(e.done));
    e.busy = 0;
    const size_t M = (size_t)e.M, Mp = (size_t)e.Mpad, G = (size_t)ctx->comm->nranks;
    for (size_t r = 0; r < G; ++r)   // rank-major [M] rows out of [rank][3][Mpad]
        unpack_block((const int64_t*)e.host + r * 3 * Mp, M, Mp, e.has_compound, exponent, counts ? counts + r * M : nullptr,
                     values ? values + r * M : nullptr, shared ? shared + r * M : nullptr, scores ? scores + r * M : nullptr);
    return PGX_OK;
}

// ---- point-sharded scoring: every rank scores ALL hypotheses against ITS slice of the points ---------------------------------
// north_star: "RCCL all-reduce of per-model inlier counts / compound-preference vectors".  The accumulators of a launch are
// integers (count, 2^-q fixed-point value and shared support), so ncclAllReduce(sum, uint64) over 3 x Mpad words is exact in
// any order and the reduced table is bitwise the table of ONE GPU scoring all the points (the ranks agree on q through
// pgx_score_set_global_n).  Cull, dispatch and the group kernel's work all divide by the number of ranks, which sharding the
// hypotheses does not do (the fixed ~60 us of a step).  After the call pgx_score_fetch returns the reduced table.
// A rank whose launch could not take the integer-accumulator path (NaN / Inf in ITS slice of the points, the per-slice U / T
// guard, filters switched off) must not skip the collective - the other ranks would wait in it for ever.  Every rank
// always enters the all-reduce: the block carries one extra word, 0 from a rank that exported its accumulators and 1 (behind an
// all-zero block) from a rank that could not; a non-zero sum raises PGX_ERR_INVALID on EVERY rank after the reduction.
// local_rc: what went wrong HERE when the export itself failed (a launch error inside score_acc_export): the rank still enters the
// collective behind a zero block + the poison word and returns that error afterwards (any early return before the
// all-reduce leaves the other ranks waiting in it).  What cannot be routed this way is a rank that cannot size or hold the block at
// all - nothing launched (no Mpad to agree on), the block's allocation failing, a slot misused: caller errors, documented in pgx.h.
static int export_or_poison(pgx_ctx* ctx, unsigned long long* blk, size_t W, hipStream_t stream, int* local_fail, int* local_rc)
{
    *local_fail = 0;
    *local_rc = PGX_OK;
    if (ctx->last_acc != nullptr && ctx->last_score_path == 2 && ctx->last_acc_M == ctx->M && ctx->last_acc_Mpad == ctx->Mpad) {
        int rc = score_acc_export(ctx, blk, stream);
        if (rc == PGX_OK && hipMemsetAsync(blk + W, 0, 64, stream) != hipSuccess) rc = fail(ctx, PGX_ERR_HIP, "score exchange: clearing the poison word failed");
        if (rc == PGX_OK) return PGX_OK;
        *local_rc = rc;
    }
    *local_fail = 1;
    PGX_HIP(ctx, hipMemsetAsync(blk, 0, (W + 8) * 8, stream));
    PGX_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(blk + W), 1u, 1, stream));
    return PGX_OK;
}

static int poisoned(pgx_ctx* ctx, const char* who, unsigned long long bad_ranks, int local_fail)
{
    return fail(ctx, PGX_ERR_INVALID, "%s: %llu rank(s) could not take the group-major path (integer accumulators)%s; the reduced table is invalid. "
                                      "The path needs sorted points, the f32 filter and the cull (the defaults) and finite points in every slice",
                who, bad_ranks, local_fail ? " - this rank is one of them" : "");
}

int pgx_score_allreduce(pgx_ctx* ctx)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allreduce: communicator not initialised");
    PGX_NO_EXCHANGE(ctx, "pgx_score_allreduce");
    if (ctx->M <= 0 || !ctx->counts.p) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allreduce: nothing launched");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t W = (size_t)3 * (size_t)ctx->Mpad;
    PGX_TRY(ensure(ctx, ctx->g_counts, (W + 8) * 8));
    unsigned long long* blk = (unsigned long long*)ctx->g_counts.p;
    int local_fail = 0, local_rc = PGX_OK;
    PGX_TRY(export_or_poison(ctx, blk, W, ctx->stream, &local_fail, &local_rc));
    PGX_NCCL(ctx, g_rccl.AllReduce(blk, blk, W + 8, ncclUint64, ncclSum, ctx->comm->comm, ctx->stream));
    unsigned long long bad = 0;
    PGX_HIP(ctx, hipMemcpyAsync(&bad, blk + W, 8, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (local_rc != PGX_OK) return local_rc;     // (this rank's own failure, reported after the collective every rank was waiting in)
    if (bad) return poisoned(ctx, "pgx_score_allreduce", bad, local_fail);
    PGX_TRY(score_acc_import(ctx, blk, ctx->M, ctx->Mpad, ctx->last_qscale, ctx->counts.as<long long>(), ctx->values.as<double>(),
                             ctx->shared.as<double>(), ctx->stream));
    ctx->mirror_valid = 0;   // the host mirror holds this rank's partial table
    return PGX_OK;
}

// The same, overlapped with the next launch (two in flight, like pgx_score_allgather_begin / _end): the accumulators are exported
// on the context's stream right behind the launch (the next launch's cull kernel zeroes them), reduced and converted on the
// exchange stream.  _end returns M rows - one table, not one per rank.
int pgx_score_allreduce_begin(pgx_ctx* ctx, int slot)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allreduce_begin: communicator not initialised");
    CommState* cs = ctx->comm;
    const size_t W = (size_t)3 * (size_t)ctx->Mpad, need = W * 8;
    ExchangeSlot* ep = nullptr;
    PGX_TRY(slot_prepare(ctx, "pgx_score_allreduce_begin", slot, need + 64, need + 64, &ep));
    ExchangeSlot& e = *ep;
    unsigned long long* blk = (unsigned long long*)e.stage.p;
    int local_fail = 0, local_rc = PGX_OK;
    PGX_TRY(export_or_poison(ctx, blk, W, ctx->stream, &local_fail, &local_rc));     // never skips the collective (see above)
    PGX_HIP(ctx, hipEventRecord(e.scored, ctx->stream));
    PGX_HIP(ctx, hipStreamWaitEvent(cs->xstream, e.scored, 0));
    PGX_NCCL(ctx, g_rccl.AllReduce(blk, blk, W + 8, ncclUint64, ncclSum, cs->comm, cs->xstream));
    long long* res = (long long*)e.gathered.p;
    PGX_TRY(score_acc_import(ctx, blk, ctx->M, ctx->Mpad, local_fail ? 1.0 : ctx->last_qscale, res, (double*)res + ctx->Mpad,
                             (double*)res + 2 * (size_t)ctx->Mpad, cs->xstream));
    PGX_HIP(ctx, hipMemcpyAsync(e.host, e.gathered.p, need, hipMemcpyDeviceToHost, cs->xstream));
    PGX_HIP(ctx, hipMemcpyAsync((char*)e.host + need, blk + W, 8, hipMemcpyDeviceToHost, cs->xstream));
    PGX_HIP(ctx, hipEventRecord(e.done, cs->xstream));
    e.M = ctx->M; e.Mpad = ctx->Mpad; e.has_compound = ctx->score_has_compound; e.busy = 1; e.reduced = 1; e.local_fail = local_fail;
    return local_rc;     // (a failed export is reported here, AFTER the collective was entered; _end then reports the poisoned table on every rank)
}

int pgx_score_allreduce_end(pgx_ctx* ctx, int slot, int exponent, int64_t* counts, double* values, double* shared, double* scores)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_score_allreduce_end: communicator not initialised");
    if (slot < 0 || slot > 1 || !ctx->comm->slot[slot].busy || !ctx->comm->slot[slot].reduced)
        return fail(ctx, PGX_ERR_INVALID, "pgx_score_allreduce_end: slot %d holds no reduced table", slot);
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    ExchangeSlot& e = ctx->comm->slot[slot];
    PGX_HIP(ctx, hipEventSynchronize(e.done));
    e.busy = 0; e.reduced = 0;
    const unsigned long long bad = *(const unsigned long long*)((const char*)e.host + (size_t)3 * (size_t)e.Mpad * 8);
    if (bad) return poisoned(ctx, "pgx_score_allreduce_end", bad, e.local_fail);
    unpack_block(e.host, (size_t)e.M, (size_t)e.Mpad, e.has_compound, exponent, counts, values, shared, scores);
    return PGX_OK;
}

int pgx_compound_allreduce_max(pgx_ctx* ctx)
{
    if (!ctx || !ctx->comm) return fail(ctx, PGX_ERR_INVALID, "pgx_compound_allreduce_max: communicator not initialised");
    PGX_NO_EXCHANGE(ctx, "pgx_compound_allreduce_max");
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_compound_allreduce_max: points not set");
    PGX_HIP(ctx, hipSetDevice(ctx->device));
    PGX_NCCL(ctx, g_rccl.AllReduce(ctx->comp.p, ctx->comp.p, (size_t)ctx->n, ncclFloat64, ncclMax, ctx->comm->comm, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->comp_dirty = 1;
    return PGX_OK;
}

}  // extern "C"
