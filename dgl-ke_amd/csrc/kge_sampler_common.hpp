// kge_sampler_common.hpp - what the on-device sampler kernel (kge_sampler.hip) and the sampler TAIL workgroups that ride on the step's
// launches (kge_sampler_tail.hpp) share: launch arguments, the slot layout, the counter-based RNG and the epoch permutation.
#pragma once
#include <cstdint>
#include "kge_common.hpp"

#define SP_THREADS 1024
#define SP_MAXE 4096                 // elements (2B + C*N) of a batch in the sampler's fast instance and in the tail path
#define SP_CODE_BITS 12
// (round 6) the sampler LAUNCH also has a wide instance: 8 keys per thread, 13 code bits - the reference's batch-2048 recipes
// (14 of its example scripts: 2 * 2048 + 8 * 256 = 6144 elements) were built on the host at ~1 ms per step
#define SP_MAXE_BIG 8192
#define SP_CODE_BITS_BIG 13

struct SamplerArgs {
    const int64_t *H, *R, *T;        // training triples [n_train]
    const int64_t *perm;             // epoch permutation [n_train] or null (identity)
    int64_t n_train, n_ent;
    int B, C, chunk, N;
    uint64_t seed;
    int64_t *state;                  // device {pos, step, ticket, -}: advanced by the launch's last workgroup
    char *slots; int64_t slot_bytes; // output slots
    int preperm;                     // tail jobs: H / R / T are ALREADY in base-permutation order (perm only says that epochs are shuffled)
};

// slot layout (must match kge_sampler_slot_* in kge_api.hip)
struct SlotLayout {
    int64_t h_gid, t_gid, rel_ids, neg_ids, ue_id, ur_id;                       // int64 arrays
    int64_t ue_pos_ptr, ue_pos_adj, ue_neg_ptr, ue_neg_slot, ur_ptr, ur_edge;   // int32 arrays
    int64_t ue_rec, ur_rec, counts;
    int64_t total;
};
__host__ __device__ inline int64_t al32(int64_t x) { return (x + 31) & ~(int64_t)31; }
__host__ __device__ inline SlotLayout slot_layout(int B, int CN) {
    const int64_t NE = 2 * (int64_t)B + CN;
    SlotLayout L; int64_t o = 0;
    L.h_gid = o; o = al32(o + 8 * B);
    L.t_gid = o; o = al32(o + 8 * B);
    L.rel_ids = o; o = al32(o + 8 * B);
    L.neg_ids = o; o = al32(o + 8 * CN);
    L.ue_id = o; o = al32(o + 8 * NE);
    L.ur_id = o; o = al32(o + 8 * B);
    L.ue_pos_ptr = o; o = al32(o + 4 * (NE + 1));
    L.ue_pos_adj = o; o = al32(o + 4 * 2 * B);
    L.ue_neg_ptr = o; o = al32(o + 4 * (NE + 1));
    L.ue_neg_slot = o; o = al32(o + 4 * CN);
    L.ur_ptr = o; o = al32(o + 4 * (B + 1));
    L.ur_edge = o; o = al32(o + 4 * B);
    L.ue_rec = o; o = al32(o + 32 * NE);
    L.ur_rec = o; o = al32(o + 32 * B);
    L.counts = o; o = al32(o + 16);
    L.total = al32(o);
    return L;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

// (pos * mul + add) mod n for pos, add < n: 64-bit arithmetic when n < 2^32 (then mul < 2^32 too), else a shift-and-add
// product modulo n (graphs beyond 4 G triples: 64 iterations per index, once per sampled edge)
__device__ __forceinline__ uint64_t epoch_index(uint64_t pos, uint64_t mul, uint64_t add, uint64_t n) {
    if (n < (1ull << 32)) return (pos * mul + add) % n;
    uint64_t acc = add % n, x = pos % n, m = mul;
    while (m) {
        if (m & 1ull) { acc += x; if (acc >= n) acc -= n; }
        x += x; if (x >= n) x -= n;
        m >>= 1;
    }
    return acc;
}
__device__ __forceinline__ uint64_t gcd_u64(uint64_t x, uint64_t y) {
    while (y) { const uint64_t r = x % y; x = y; y = r; }
    return x;
}


// the affine bijection of [0, n_train) that turns the base permutation into epoch `ep`'s edge order (identity for epoch 0 or
// without a base permutation): multiplier AND offset hashed from (seed, epoch)
__device__ __forceinline__ void epoch_affine(const SamplerArgs &a, int64_t ep, uint64_t &mul, uint64_t &add) {
    mul = 1; add = 0;
    if (a.perm && ep > 0) {
        const uint64_t n = (uint64_t)a.n_train;
        add = mix64(a.seed ^ (0xD6E8FEB86659FD93ULL * (uint64_t)ep)) % n;
        uint64_t m = mix64(a.seed ^ (0xA0761D6478BD642FULL * (uint64_t)(ep + 1)));
        m = (n < (1ull << 32)) ? (m & 0xffffffffull) : (m % n);          // (keeps pos * mul inside what epoch_index multiplies)
        m |= 1ull;
        if (m < 3) m = 3;
        while (gcd_u64(m, n) != 1) m += 2;                               // bijection needs gcd(mul, n_train) = 1
        mul = m;
    }
}
// ---- the tail path's cached epoch constants: state[4..7] = {epoch, mul, add, floor((2^64 - 1) / n_train)} (round 5) ----
// epoch_affine costs a search with 64-bit remainders (microseconds for one thread) and epoch_index a 64-bit remainder per edge -
// nothing inside a 33-us sampler launch, too much for a tail workgroup that must finish under a 9-us launch: phase 3 of every job
// caches the constants of the NEXT job's epoch, phase 1 only looks them up, and the remainder is taken with the cached reciprocal
// (exact: two multiplies, at most two corrective subtractions).
struct EpochConst { uint64_t mul, add, rinv; };
__device__ __forceinline__ EpochConst epoch_consts_slow(const SamplerArgs &a, int64_t ep) {
    EpochConst c;
    epoch_affine(a, ep, c.mul, c.add);
    c.rinv = ~0ull / (uint64_t)a.n_train;
    return c;
}
__device__ __forceinline__ EpochConst epoch_consts_cached(const SamplerArgs &a, int64_t ep) {
    const int64_t *st = a.state;
    if (st[4] == ep) { EpochConst c; c.mul = (uint64_t)st[5]; c.add = (uint64_t)st[6]; c.rinv = (uint64_t)st[7]; return c; }
    return epoch_consts_slow(a, ep);
}
// epoch_index with the cached reciprocal (n < 2^32: pos * mul + add < 2^64)
__device__ __forceinline__ uint64_t epoch_index_fast(uint64_t pos, const EpochConst &c, uint64_t n) {
    if (n >= (1ull << 32)) return epoch_index(pos, c.mul, c.add, n);
    const uint64_t x = pos * c.mul + c.add;
    uint64_t r = x - __umul64hi(x, c.rinv) * n;
    while (r >= n) r -= n;
    return r;
}

// uniform negative id j of step `step`
__device__ __forceinline__ int64_t sample_negative(const SamplerArgs &a, int64_t step, int j) {
    const uint64_t x = mix64(mix64(a.seed ^ (uint64_t)step * 0x9E3779B97F4A7C15ULL) + (uint64_t)j);
    return (int64_t)__umul64hi(x, (uint64_t)a.n_ent);                    // uniform in [0, n_ent)
}

// one batch's construction job as tail workgroups of a training step's launches (kge_sampler_tail.hpp)
struct SmpTail {
    SamplerArgs a;                     // triples, base permutation, sizes, seed, state {pos, step, -, -}; a.slots = THIS batch's slot
    char *scratch;                     // kge_sampler_tail_scratch_bytes(): one buffer for all batches (a batch is finished within its step)
    int k;                             // index of the batch in the group under construction: its step number is state[1] + k
    int phase;                         // 0 = no tail workgroups in this launch
    int advance;                       // phase 3, > 0: the group's LAST batch - advance the state by this many batches
    char *slot3;                       // first launch: the slot of the PREVIOUS job of the group, whose phase 3 rides here too (null: none)
};

#define ST_NBK 4                       // id-range buckets per batch
#define ST_THREADS 256                 // = KGE_BLOCK: the host launches' workgroup size
#define ST_P1_WGS 4
#define ST_P2_WGS (ST_NBK + 1)
#define ST_P3_WGS (ST_NBK + 1)

// ---- scratch layout (bytes), shared by the three phases ----
struct TailScratch { int64_t hdr, ekeys, esort, escan, ust, rkeys, total; };
__host__ __device__ inline TailScratch tail_scratch(int B, int CN, bool key64) {
    const int64_t NE = 2 * (int64_t)B + CN, ks = key64 ? 8 : 4;
    TailScratch L; int64_t o = 0;
    L.hdr = o; o = al32(o + 4 * 64);
    L.ekeys = o; o = al32(o + ks * ST_NBK * NE);          // [bucket][edge part: 2B | negative part: CN]
    L.esort = o; o = al32(o + ks * ST_NBK * NE);          // [bucket][NE] sorted keys
    L.escan = o; o = al32(o + 4 * ST_NBK * NE);           // [bucket][NE] positives in front of every element (big buckets only)
    L.ust = o; o = al32(o + 4 * ST_NBK * (NE + 1));       // [bucket][NE + 1] first element of every unique (big buckets only)
    L.rkeys = o; o = al32(o + 8 * B);
    L.total = al32(o);
    return L;
}
// header words
#define ST_H_CNT(b, src) (3 * (b) + (src))                // phase 1: keys of bucket b from the two edge workgroups (0, 1) / the negatives (2)
#define ST_H_NB(b) (12 + (b))                             // phase 2: elements, uniques, positives of bucket b
#define ST_H_UB(b) (16 + (b))
#define ST_H_PB(b) (20 + (b))

