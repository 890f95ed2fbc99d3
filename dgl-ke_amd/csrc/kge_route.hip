// kge_route.hip - device side of the range-sharded multi-GPU step (dglke_amd/dist.py): everything the
// parameter-server semantics of the reference (KEModel.pull_model / push_gradient, models/general_models.py:650-680; server-side
// Adagrad, kvserver.py:41-51) needs around the collectives, with NO host round trip:
//   route_build   the sorted unique entity ids of a batch (the plan of kge_sample_batches / dglke_amd/plan.py) are cut into owner
//                 buckets of FIXED capacity -> request ids per owner (-1 padded), and the batch re-addressed to "cache rows"
//                 (owner * cap + position in that owner's bucket), so that the fixed-size all-to-all results ARE the row cache;
//   gather_req    owner side of the pull: rows of the requested ids (global id - shard offset), pads skipped;
//   apply_merged  owner side of the push: the messages of all source ranks are applied by the wavefront that owns the FIRST
//                 occurrence of a row, in source-rank order (a k-ary search finds the same row in the other ranks' sorted
//                 buckets) - one launch, no atomics, bit-reproducible, the order of ExternalEmbedding.update per trace.
#include "kge_common.hpp"

using namespace kge;

static inline int check_launch_r() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }
int kge_fail(int code, const char *msg);          // kge_api.hip: records the message kge_last_error() returns

#define RT_THREADS 256
#define RT_MAX_WORLD 64

struct RouteArgs {
    int UEmax, B, CN, world, cap;
    int64_t per;                               // entity rows per shard (owner = id / per)
    const int64_t *ue_id; const int32_t *ue_rec; const int32_t *ue_pos_adj, *ue_neg_slot; const int32_t *counts_dev;
    int64_t *req_ids;                          // [world * cap] global ids requested from each owner, -1 padded
    int64_t *h_loc, *t_loc, *neg_loc;          // [B], [B], [C*N] cache rows of the edge ends / negative slots
    int64_t *ue_loc;                           // [UEmax] cache row of union entry u
    int32_t *ue_rec_loc;                       // [UEmax][8] plan records with the id words replaced by the cache row
    int32_t *overflow;                         // += entries that did not fit their owner's bucket (must stay 0)
    // packed single-trace messages (round 6): message rows per union entry, or null
    int cap2; int32_t *ue_msg;                 // [UEmax][2] {row of the (first) message, bucket position of the second message in the extra region or -1}
};
// a union entry that is in BOTH traces of its batch (positive list and negative list non-empty): packed messages give it two rows
__device__ __forceinline__ bool rec_both(const int32_t *ue_rec, int u) {
    const int4 r0 = reinterpret_cast<const int4 *>(ue_rec)[2 * u];
    const int2 r1 = reinterpret_cast<const int2 *>(ue_rec)[4 * u + 2];
    return r0.w > r0.z && r1.y > r1.x;
}

// grid: enough workgroups of RT_THREADS threads for max(UEmax, world * cap) items.  Every workgroup finds the bucket boundaries
// itself (a scan of the <= 4096 sorted ids: independent loads, ~1 us) instead of waiting for one workgroup to publish them.
__global__ __launch_bounds__(RT_THREADS) void route_build_kernel(RouteArgs a_in, int64_t in_stride, int64_t out_stride) {
    __shared__ int start[RT_MAX_WORLD + 1];
    __shared__ int bcnt[RT_MAX_WORLD];         // packed messages: both-trace entries of owner o in FRONT of this workgroup's slice
    __shared__ unsigned char s_both[RT_THREADS]; __shared__ unsigned char s_own[RT_THREADS];
    const int t = threadIdx.x;
    // group form (round 4): blockIdx.y = batch k of a sampled group - its plan arrays lie k * in_stride bytes behind batch 0's
    // (consecutive sampler slots), its outputs k * out_stride bytes behind batch 0's (one pool per group)
    RouteArgs a = a_in;
    if (blockIdx.y) {
        const int64_t ki = (int64_t)blockIdx.y * in_stride, ko = (int64_t)blockIdx.y * out_stride;
#define RT_IN(p) p = reinterpret_cast<decltype(p)>(reinterpret_cast<const char *>(p) + ki)
#define RT_OUT(p) p = reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + ko)
        RT_IN(a.ue_id); RT_IN(a.ue_rec); RT_IN(a.ue_pos_adj); RT_IN(a.ue_neg_slot); RT_IN(a.counts_dev);
        RT_OUT(a.req_ids); RT_OUT(a.h_loc); RT_OUT(a.t_loc); RT_OUT(a.neg_loc); RT_OUT(a.ue_loc); RT_OUT(a.ue_rec_loc);
        if (a.ue_msg) RT_OUT(a.ue_msg);
#undef RT_IN
#undef RT_OUT
    }
    const int cnt = a.counts_dev ? min(a.counts_dev[0], a.UEmax) : a.UEmax;
    const bool small = a.per < 0x7fffffffLL && (a.per * a.world) < 0x7fffffffLL;     // 32-bit owner arithmetic when the ids fit
    auto owner_of = [&](int64_t id) -> int {
        return small ? (int)((uint32_t)id / (uint32_t)a.per) : (int)(id / a.per);
    };
    for (int o = t; o <= a.world; o += RT_THREADS) start[o] = cnt;
    for (int o = t; o < a.world; o += RT_THREADS) bcnt[o] = 0;
    __syncthreads();
    // bucket boundaries: ue_id is sorted, so owner o's entries are one contiguous run; entry u opens the runs of all owners in
    // (owner(u-1), owner(u)]
    const int slice0 = (int)blockIdx.x * RT_THREADS;
    for (int u = t; u < cnt; u += RT_THREADS) {
        const int o = owner_of(a.ue_id[u]);
        const int op = u ? owner_of(a.ue_id[u - 1]) : -1;
        for (int k = op + 1; k <= o && k < a.world; ++k) start[k] = u;
        // (counts commute: the LDS atomics give the same numbers in any order)
        if (a.ue_msg && u < slice0 && rec_both(a.ue_rec, u)) atomicAdd(&bcnt[min(o, a.world - 1)], 1);
    }
    const int u = slice0 + t;
    if (a.ue_msg) {
        const bool in = u < cnt;
        s_both[t] = (in && rec_both(a.ue_rec, u)) ? 1 : 0;
        s_own[t] = in ? (unsigned char)min(owner_of(a.ue_id[u]), a.world - 1) : 255;
    }
    __syncthreads();
    if (u < cnt) {
        const int64_t id = a.ue_id[u];
        const int o = min(owner_of(id), a.world - 1);
        const int pos = u - start[o];
        const int4 r0 = reinterpret_cast<const int4 *>(a.ue_rec)[2 * u];
        const int4 r1 = reinterpret_cast<const int4 *>(a.ue_rec)[2 * u + 1];
        if (pos >= a.cap) {                    // does not fit: counted; the entry trains against the dump row world * cap this step
            atomicAdd(a.overflow, 1);
        } else {
            a.req_ids[(int64_t)o * a.cap + pos] = id;
        }
        const int64_t cr = pos < a.cap ? (int64_t)o * a.cap + pos : (int64_t)a.world * a.cap;
        a.ue_loc[u] = cr;
        if (a.ue_msg) {
            const int capT = a.cap + a.cap2;
            int extra = -1;
            if (s_both[t]) {                   // rank among the both-trace entries of owner o's bucket: in front of the slice + inside it
                int rank = bcnt[o];
                for (int q = 0; q < t; ++q) rank += (s_own[q] == o && s_both[q]) ? 1 : 0;
                if (rank < a.cap2 && pos < a.cap) extra = rank;      // (position inside the bucket's extra region: row o * capT + cap + rank)
                else if (pos < a.cap) atomicAdd(a.overflow, 1);          // (its second trace would be lost: the caller sizes cap2 first)
            }
            a.ue_msg[2 * u] = pos < a.cap ? o * capT + pos : a.world * capT;
            a.ue_msg[2 * u + 1] = extra;
        }
        int4 q0 = r0;
        q0.x = (int)(cr & 0xffffffff); q0.y = (int)(cr >> 32);
        reinterpret_cast<int4 *>(a.ue_rec_loc)[2 * u] = q0;
        reinterpret_cast<int4 *>(a.ue_rec_loc)[2 * u + 1] = r1;
        // first entries of the two lists sit in the record itself; the (rare) rest is read from the list arrays
        if (r0.w > r0.z) {
            ((r1.z & 1) ? a.t_loc : a.h_loc)[r1.z >> 1] = cr;
            for (int p = r0.z + 1; p < r0.w; ++p) {
                const int adj = a.ue_pos_adj[p];
                ((adj & 1) ? a.t_loc : a.h_loc)[adj >> 1] = cr;
            }
        }
        if (r1.y > r1.x) {
            a.neg_loc[r1.w] = cr;
            for (int p = r1.x + 1; p < r1.y; ++p) a.neg_loc[a.ue_neg_slot[p]] = cr;
        }
    }
    // pads of the request buckets
    const int k = (int)blockIdx.x * RT_THREADS + t;
    if (k < a.world * a.cap) {
        const int o = k / a.cap, pos = k % a.cap;
        const int n = min(start[o + 1], cnt) - min(start[o], cnt);
        if (pos >= n) a.req_ids[k] = -1;
    }
}

// largest owner-bucket fill over a GROUP of batches (one workgroup per batch; batch k's sorted id array / counts sit `stride` bytes
// behind batch k-1's - consecutive slots of the device sampler - or there is one batch).  The caller reads the word once per group,
// BEFORE the group's steps run, and grows the bucket capacity when needed: no entry ever trains against the dump row.
__global__ __launch_bounds__(RT_THREADS) void route_fill_kernel(const char *ue0, const char *cnt0, const char *rec0, int64_t stride, int UEmax,
                                                                int world, int64_t per, int32_t *max_fill) {
    __shared__ int start[RT_MAX_WORLD + 1];
    __shared__ int bcnt[RT_MAX_WORLD];
    const int t = threadIdx.x;
    const int32_t *ue_rec = rec0 ? reinterpret_cast<const int32_t *>(rec0 + (int64_t)blockIdx.x * stride) : nullptr;
    const int64_t *ue_id = reinterpret_cast<const int64_t *>(ue0 + (int64_t)blockIdx.x * stride);
    const int cnt = cnt0 ? min(reinterpret_cast<const int32_t *>(cnt0 + (int64_t)blockIdx.x * stride)[0], UEmax) : UEmax;
    const bool small = per < 0x7fffffffLL && (per * world) < 0x7fffffffLL;
    auto owner_of = [&](int64_t id) -> int {
        return small ? (int)((uint32_t)id / (uint32_t)per) : (int)(id / per);
    };
    for (int o = t; o <= world; o += RT_THREADS) start[o] = cnt;
    for (int o = t; o < world; o += RT_THREADS) bcnt[o] = 0;
    __syncthreads();
    for (int u = t; u < cnt; u += RT_THREADS) {
        const int o = owner_of(ue_id[u]);
        const int op = u ? owner_of(ue_id[u - 1]) : -1;
        for (int k = op + 1; k <= o && k < world; ++k) start[k] = u;
        if (ue_rec && rec_both(ue_rec, u)) atomicAdd(&bcnt[min(o, world - 1)], 1);
    }
    __syncthreads();
    // (ids beyond the last shard's range belong to the last owner, like in route_build_kernel)
    for (int o = t; o < world; o += RT_THREADS) {
        const int hi = o + 1 < world ? start[o + 1] : cnt;
        atomicMax(max_fill, hi - start[o]);
        atomicMax(max_fill + 1, bcnt[o]);      // ABI 8: entries of one bucket that are in both traces (the packed messages' extra region)
    }
}

template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void gather_req_kernel(const float *__restrict__ table, int dim, const int64_t *__restrict__ ids,
                                                               int64_t id_offset, int64_t n_rows, int64_t n, float *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t id = ids[k] - id_offset;
    if (ids[k] < 0 || id < 0 || id >= n_rows) return;          // pad (or an id this shard does not own): row left as it is
    const float *src = table + id * (int64_t)dim;
    float *dst = out + k * (int64_t)dim;
    // every pack of the row is requested before the first store (lane offsets clamped): taken one pack at a time the four
    // dependent load -> store rounds of an 800-float row were the kernel's duration (8.8 us for 3072 rows of a 34-GB shard)
    const int nit = dim / V;
    if (nit <= 256) {
        Pack<V> v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld<V>(src + min(lane + 64 * q, nit - 1) * V);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (lane + 64 * q < nit) st<V>(dst + (lane + 64 * q) * V, v[q]);
        return;
    }
    for (int it = lane; it < nit; it += 64) st<V>(dst + it * V, ld<V>(src + it * V));
}

// ---- merged owner-side apply -----------------------------------------------------------------------------------------
struct MergeArgs {
    float *table, *state;
    int64_t n_rows, id_offset;
    int dim, nsrc, cap, ld, ntraces;
    int cap_extra;                             // > 0: packed single-trace messages - bucket stride cap + cap_extra message rows, header [gs | link]
    const int32_t *idw; int64_t id_stride;     // id of message k: words idw[k * id_stride], idw[k * id_stride + 1] (lo, hi)
    const float *msg;                          // [nsrc * cap][ld]: [g_0 | .. | g_{T-1} | gs_0 .. gs_{T-1} | ...]
    float lr, eps;
};
__device__ __forceinline__ int64_t merge_id(const MergeArgs &a, int64_t k) {
    const int32_t *w = a.idw + k * a.id_stride;
    return (int64_t)(uint32_t)w[0] | ((int64_t)w[1] << 32);
}
__device__ __forceinline__ uint64_t merge_key(int64_t id) { return id < 0 ? ~0ull : (uint64_t)id; }    // pads sort last

template <int NIT>      // row width <= 256 * NIT floats, dim % 4 == 0
__device__ __forceinline__ void apply_merged_body(const MergeArgs &a, int bid) {
    const int64_t k = (int64_t)bid * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (k >= (int64_t)a.nsrc * a.cap) return;
    const int lane = threadIdx.x & 63;
    const int src = (int)(k / a.cap);
    const int64_t gid = merge_id(a, k);
    if (gid < 0) return;
    const int64_t id = gid - a.id_offset;
    if (id < 0 || id >= a.n_rows) return;
    // ---- the same id in the other sources' sorted buckets: a k-ary search per source, Lg lanes probing together ----
    int lgs = 1;
    while (lgs * 2 * a.nsrc <= 64) lgs *= 2;            // lanes per source (power of two), nsrc <= 64
    const int g = lane / lgs, j = lane % lgs;
    const bool active = g < a.nsrc && g != src;
    int lo = 0, hi = active ? a.cap : 0, found = -1;
    const uint64_t X = (uint64_t)gid;
    const uint64_t gmask = lgs == 64 ? ~0ull : ((1ull << lgs) - 1ull);
    while (__any(hi > lo)) {
        const int n = hi - lo;
        const bool fin = n <= lgs;
        const int p = fin ? lo + j : lo + (int)(((int64_t)(j + 1) * n) / (lgs + 1));
        const bool valid = hi > lo && (!fin || j < n);
        const uint64_t v = valid ? merge_key(merge_id(a, (int64_t)g * a.cap + p)) : ~0ull;
        const uint64_t beq = __ballot(valid && v == X), blt = __ballot(valid && v < X);
        const uint64_t geq = (beq >> (g * lgs)) & gmask, glt = (blt >> (g * lgs)) & gmask;
        if (hi > lo) {
            if (geq) {
                const int jj = __builtin_ctzll(geq);
                found = fin ? lo + jj : lo + (int)(((int64_t)(jj + 1) * n) / (lgs + 1));
                lo = hi;
            } else if (fin) {
                lo = hi;
            } else {
                const int c = __builtin_popcountll(glt);      // probes are ascending: the first c are < X
                const int nlo = c > 0 ? lo + (int)(((int64_t)c * n) / (lgs + 1)) + 1 : lo;
                const int nhi = c < lgs ? lo + (int)(((int64_t)(c + 1) * n) / (lgs + 1)) : hi;
                lo = nlo; hi = nhi;
            }
        }
    }
    // a lower source holds the row too: its wavefront applies every occurrence
    const uint64_t bf = __ballot(found >= 0 && j == 0);
    for (int s = 0; s < src; ++s) if ((bf >> (s * lgs)) & 1ull) return;
    // ---- apply: own message, then the higher sources' in rank order; the row stays in registers ----
    const int d = a.dim, nit = d >> 2;
    float *row = a.table + id * (int64_t)d;
    Pack<4> x[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) x[q] = ld<4>(row + min(lane + 64 * q, nit - 1) * 4);
    float st = a.state[id];
    bool any = false;
    for (int s = src; s < a.nsrc; ++s) {
        int pos;
        if (s == src) pos = (int)(k % a.cap);
        else {
            if (!((bf >> (s * lgs)) & 1ull)) continue;
            pos = __builtin_amdgcn_readlane(found, s * lgs);
        }
        const int capT = a.cap + a.cap_extra;
        const float *m = a.msg + ((int64_t)s * capT + pos) * a.ld;
        // packed (cap_extra > 0): one trace per message; `link` >= 0 names the row's second message (its negative trace) in the
        // bucket's extra region - applied behind the first, the reference's trace order
        const int link = a.cap_extra ? __float_as_int(m[d + 1]) : -1;
        const int ntr = a.cap_extra ? (link >= 0 ? 2 : 1) : a.ntraces;
        for (int t = 0; t < ntr; ++t) {
            const float *mt = (a.cap_extra && t) ? a.msg + ((int64_t)s * capT + a.cap + link) * a.ld : m;
            const float inc = a.cap_extra ? mt[d] : m[(int64_t)a.ntraces * d + t];
            if (inc == 0.f) continue;
            any = true;
            st += inc;
            const float kf = -a.lr / (sqrtf(st) + a.eps);
            const float *gp = a.cap_extra ? mt : m + (int64_t)t * d;
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const Pack<4> gv = ld<4>(gp + min(lane + 64 * q, nit - 1) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[q].v[e] = fmaf(gv.v[e], kf, x[q].v[e]);
            }
        }
    }
    if (!any) return;
#pragma unroll
    for (int q = 0; q < NIT; ++q) if (lane + 64 * q < nit) st_wt<4>(row + (lane + 64 * q) * 4, x[q]);
    if (lane == 0) a.state[id] = st;
}
template <int NIT>
__global__ __launch_bounds__(KGE_BLOCK) void apply_merged_kernel(MergeArgs a) { apply_merged_body<NIT>(a, (int)blockIdx.x); }
// the entity-shard apply and the relation-replica apply of one step in ONE launch (independent tables, independent messages): one
// launch boundary less, and the two fill the chip together (round 4)
template <int NIT>
__global__ __launch_bounds__(KGE_BLOCK) void apply_merged_pair_kernel(MergeArgs a, MergeArgs b, int nbA) {
    if ((int)blockIdx.x < nbA) apply_merged_body<NIT>(a, (int)blockIdx.x);
    else apply_merged_body<NIT>(b, (int)blockIdx.x - nbA);
}

extern "C" {

int kge_route_build(const kge_batch *b, int world, int64_t rows_per_shard, int cap, int64_t *req_ids, int64_t *h_loc,
                    int64_t *t_loc, int64_t *neg_loc, int64_t *ue_loc, int32_t *ue_rec_loc, int32_t *overflow, int cap2,
                    int32_t *ue_msg, void *stream) {
    if (!b || !req_ids || !h_loc || !t_loc || !neg_loc || !ue_loc || !ue_rec_loc || !overflow || world < 1 ||
        world > RT_MAX_WORLD || rows_per_shard <= 0 || cap <= 0 || !b->ue_id || !b->ue_rec || !b->ue_pos_adj || !b->ue_neg_slot ||
        (ue_msg && (cap2 < 1 || (int64_t)(world + 1) * ((int64_t)cap + cap2) > 0x7fffffffLL)))
        return kge_fail(KGE_ERR_ARG, "kge_route_build: null pointer, world outside [1, 64] or non-positive shard size / capacity");
    RouteArgs a{};
    a.UEmax = b->UE; a.B = b->B; a.CN = b->C * b->N; a.world = world; a.cap = cap; a.per = rows_per_shard;
    a.ue_id = b->ue_id; a.ue_rec = b->ue_rec; a.ue_pos_adj = b->ue_pos_adj; a.ue_neg_slot = b->ue_neg_slot;
    a.counts_dev = b->counts_dev;
    a.req_ids = req_ids; a.h_loc = h_loc; a.t_loc = t_loc; a.neg_loc = neg_loc; a.ue_loc = ue_loc; a.ue_rec_loc = ue_rec_loc;
    a.overflow = overflow; a.cap2 = ue_msg ? cap2 : 0; a.ue_msg = ue_msg;
    const int items = a.UEmax > world * cap ? a.UEmax : world * cap;
    hipLaunchKernelGGL(route_build_kernel, dim3((items + RT_THREADS - 1) / RT_THREADS), dim3(RT_THREADS), 0, (hipStream_t)stream, a,
                       (int64_t)0, (int64_t)0);
    return check_launch_r();
}

int kge_route_build_group(const kge_batch *b0, int n_batches, size_t in_stride_bytes, int world, int64_t rows_per_shard, int cap,
                          int64_t *req_ids, int64_t *h_loc, int64_t *t_loc, int64_t *neg_loc, int64_t *ue_loc, int32_t *ue_rec_loc,
                          size_t out_stride_bytes, int32_t *overflow, int cap2, int32_t *ue_msg, void *stream) {
    if ((ue_msg && (cap2 < 1 || (int64_t)(world + 1) * ((int64_t)cap + cap2) > 0x7fffffffLL)) || !b0 || !req_ids || !h_loc || !t_loc || !neg_loc || !ue_loc || !ue_rec_loc || !overflow || world < 1 || world > RT_MAX_WORLD ||
        rows_per_shard <= 0 || cap <= 0 || !b0->ue_id || !b0->ue_rec || !b0->ue_pos_adj || !b0->ue_neg_slot || n_batches < 1 ||
        n_batches > 65535 || (n_batches > 1 && (!b0->counts_dev || in_stride_bytes == 0 || out_stride_bytes == 0)))
        return kge_fail(KGE_ERR_ARG, "kge_route_build_group: null pointer, world outside [1, 64], non-positive shard size / capacity, or a "
                                     "group of several batches without device-built plans / strides");
    RouteArgs a{};
    a.UEmax = b0->UE; a.B = b0->B; a.CN = b0->C * b0->N; a.world = world; a.cap = cap; a.per = rows_per_shard;
    a.ue_id = b0->ue_id; a.ue_rec = b0->ue_rec; a.ue_pos_adj = b0->ue_pos_adj; a.ue_neg_slot = b0->ue_neg_slot;
    a.counts_dev = b0->counts_dev;
    a.req_ids = req_ids; a.h_loc = h_loc; a.t_loc = t_loc; a.neg_loc = neg_loc; a.ue_loc = ue_loc; a.ue_rec_loc = ue_rec_loc;
    a.overflow = overflow; a.cap2 = ue_msg ? cap2 : 0; a.ue_msg = ue_msg;
    const int items = a.UEmax > world * cap ? a.UEmax : world * cap;
    hipLaunchKernelGGL(route_build_kernel, dim3((items + RT_THREADS - 1) / RT_THREADS, n_batches), dim3(RT_THREADS), 0, (hipStream_t)stream,
                       a, (int64_t)in_stride_bytes, (int64_t)out_stride_bytes);
    return check_launch_r();
}

int kge_route_fill(const kge_batch *b0, int n_batches, size_t stride_bytes, int world, int64_t rows_per_shard, int32_t *max_fill,
                   void *stream) {
    if (!b0 || !b0->ue_id || !max_fill || n_batches < 1 || world < 1 || world > RT_MAX_WORLD || rows_per_shard <= 0 ||
        (n_batches > 1 && (stride_bytes == 0 || !b0->counts_dev)))
        return kge_fail(KGE_ERR_ARG, "kge_route_fill: bad argument (1 <= world <= 64; several batches need a slot stride and device-built plans)");
    hipLaunchKernelGGL(route_fill_kernel, dim3(n_batches), dim3(RT_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const char *>(b0->ue_id), reinterpret_cast<const char *>(b0->counts_dev),
                       reinterpret_cast<const char *>(b0->ue_rec), (int64_t)stride_bytes, b0->UE, world, rows_per_shard, max_fill);
    return check_launch_r();
}

int kge_batch_localized(const kge_batch *b, const int64_t *h_loc, const int64_t *t_loc, const int64_t *neg_loc,
                        const int64_t *ue_loc, const int32_t *ue_rec_loc, kge_batch *out) {
    if (!b || !h_loc || !t_loc || !neg_loc || !ue_loc || !ue_rec_loc || !out) return KGE_ERR_ARG;
    *out = *b;
    out->h_gid = h_loc; out->t_gid = t_loc; out->neg_ids = neg_loc; out->ue_id = ue_loc; out->ue_rec = ue_rec_loc;
    return KGE_OK;
}

int kge_gather_rows_req(const float *table, int64_t n_rows, int dim, const int64_t *ids, int64_t id_offset, int64_t n_ids,
                        float *out, void *stream) {
    if (!table || n_rows < 0 || dim <= 0 || n_ids < 0 || (n_ids && (!ids || !out))) return KGE_ERR_ARG;
    if (n_ids == 0) return KGE_OK;
    const int nb = (int)((n_ids + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    if (dim % 4 == 0)
        hipLaunchKernelGGL(gather_req_kernel<4>, dim3(nb), dim3(KGE_BLOCK), 0, (hipStream_t)stream, table, dim, ids, id_offset, n_rows, n_ids, out);
    else
        hipLaunchKernelGGL(gather_req_kernel<1>, dim3(nb), dim3(KGE_BLOCK), 0, (hipStream_t)stream, table, dim, ids, id_offset, n_rows, n_ids, out);
    return check_launch_r();
}

static int merge_args(MergeArgs &a, const kge_merge_job *j, float lr, float eps) {
    if (!j || !j->table || !j->state_sum || j->n_rows < 0 || j->dim <= 0 || j->dim % 4 || j->dim > 1024 || j->nsrc < 1 ||
        j->nsrc > RT_MAX_WORLD || j->cap <= 0 || !j->id_words || j->id_stride_words < 2 || !j->msg ||
        j->ld < j->ntraces * j->dim + j->ntraces || j->ld % 4 || j->ntraces < 1 || j->cap_extra < 0 ||
        (j->cap_extra > 0 && (j->ntraces != 1 || j->ld < j->dim + 4)))
        return kge_fail(KGE_ERR_ARG, "kge_adagrad_apply_merged_pair: bad job (row width must be a multiple of 4 and <= 1024, 1 <= sources <= 64, "
                                     "message stride >= traces * (width + 1) and a multiple of 4)");
    a = MergeArgs{};
    a.table = j->table; a.state = j->state_sum; a.n_rows = j->n_rows; a.id_offset = j->id_offset; a.dim = j->dim; a.nsrc = j->nsrc;
    a.cap = j->cap; a.ld = j->ld; a.ntraces = j->ntraces; a.cap_extra = j->cap_extra; a.idw = j->id_words; a.id_stride = j->id_stride_words; a.msg = j->msg;
    a.lr = lr; a.eps = eps;
    return KGE_OK;
}

int kge_adagrad_apply_merged_pair(const kge_merge_job *ja, const kge_merge_job *jb, float lr, float eps, void *stream) {
    MergeArgs a, b;
    if (int rc = merge_args(a, ja, lr, eps)) return rc;
    if (int rc = merge_args(b, jb, lr, eps)) return rc;
    const int nbA = (int)(((int64_t)a.nsrc * a.cap + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const int nbB = (int)(((int64_t)b.nsrc * b.cap + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const dim3 g((unsigned)(nbA + nbB)), bl(KGE_BLOCK);
    const int dmax = a.dim > b.dim ? a.dim : b.dim;      // one instance for both jobs: the narrower row just leaves packs unused
    if (dmax <= 256) hipLaunchKernelGGL(apply_merged_pair_kernel<1>, g, bl, 0, (hipStream_t)stream, a, b, nbA);
    else if (dmax <= 512) hipLaunchKernelGGL(apply_merged_pair_kernel<2>, g, bl, 0, (hipStream_t)stream, a, b, nbA);
    else hipLaunchKernelGGL(apply_merged_pair_kernel<4>, g, bl, 0, (hipStream_t)stream, a, b, nbA);
    return check_launch_r();
}

int kge_adagrad_apply_merged(float *table, float *state_sum, int64_t n_rows, int dim, int nsrc, int cap, const int32_t *id_words,
                             int64_t id_stride_words, int64_t id_offset, const float *msg, int ld, int ntraces, int cap_extra,
                             float lr, float eps, void *stream) {
    if (cap_extra < 0 || (cap_extra > 0 && (ntraces != 1 || ld < dim + 4)) || !table || !state_sum || n_rows < 0 || dim <= 0 || dim % 4 || dim > 1024 || nsrc < 1 || nsrc > RT_MAX_WORLD || cap <= 0 ||
        !id_words || id_stride_words < 2 || !msg || ld < ntraces * dim + ntraces || ld % 4 || ntraces < 1)
        return kge_fail(KGE_ERR_ARG, "kge_adagrad_apply_merged: bad argument (row width must be a multiple of 4 and <= 1024, 1 <= sources <= 64, "
                                     "message stride >= traces * (width + 1) and a multiple of 4)");
    MergeArgs a{};
    a.table = table; a.state = state_sum; a.n_rows = n_rows; a.id_offset = id_offset; a.dim = dim; a.nsrc = nsrc; a.cap = cap;
    a.ld = ld; a.ntraces = ntraces; a.cap_extra = cap_extra; a.idw = id_words; a.id_stride = id_stride_words; a.msg = msg; a.lr = lr; a.eps = eps;
    const int64_t n = (int64_t)nsrc * cap;
    const dim3 g((unsigned)((n + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK)), bl(KGE_BLOCK);
    if (dim <= 256) hipLaunchKernelGGL(apply_merged_kernel<1>, g, bl, 0, (hipStream_t)stream, a);
    else if (dim <= 512) hipLaunchKernelGGL(apply_merged_kernel<2>, g, bl, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(apply_merged_kernel<4>, g, bl, 0, (hipStream_t)stream, a);
    return check_launch_r();
}

}  // extern "C"
