// stack_sort.h - compile-time sorting networks (float32 and packed uint16) and static-index multiplexers for register-resident columns.
// Part of the stack kernels (see stack_kernels.h for the overall design).
#pragma once
#include "common.h"

#include <utility>

namespace apgpu_stack {

using namespace apgpu;


// zero-cost section markers in the generated assembly (tools/isa_sections.py counts instructions per section)
#ifdef APGPU_PROFILE_SECTIONS
#define APGPU_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; APGPU_SECTION " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define APGPU_MARK(name) asm volatile("; APGPU_SECTION " name)
#endif

// Wave-wide votes as one compare into an SGPR pair + a scalar test.  HIP's __any / __all go through an int (v_cndmask 0/1 +
// v_cmp_ne) before the ballot: two extra VALU instructions per vote, and the clip loop votes ~12 times per pass.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

// -------------------------------------------------------------------------------------------------
// Batcher odd-even merge sorting network, generated at compile time.  For NP that is not a power of two the
// network of the next power of two is pruned to its first NP wires: the missing inputs are +inf, which never
// move from the top wires, so every compare-exchange that touches one of them is a no-op and can be dropped.
// -------------------------------------------------------------------------------------------------
struct CE {
    unsigned char a, b;
};

template <int NP>
struct Net {
    CE ce[NP * 12 + 1]; // NP=128 needs 1471 < 1536
    int n;
};

constexpr int next_pow2(int n)
{
    int p = 1;
    while (p < n) p *= 2;
    return p;
}

// P0: first merge level generated (1 = the whole sort; 4 = the merges after every aligned group of four has been sorted).
template <int NP, int P0 = 1>
constexpr Net<NP> make_net()
{
    constexpr int P2 = next_pow2(NP);
    Net<NP> net{};
    int c = 0;
    for (int p = P0; p < P2; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P2 - 1 - k; j += 2 * k) {
                int lim = (k - 1 < P2 - j - k - 1) ? k - 1 : P2 - j - k - 1;
                for (int i = 0; i <= lim; i++)
                    if ((i + j) / (p * 2) == (i + j + k) / (p * 2) && i + j + k < NP) {
                        net.ce[c].a = (unsigned char)(i + j);
                        net.ce[c].b = (unsigned char)(i + j + k);
                        c++;
                    }
            }
    net.n = c;
    return net;
}

// The network pruned to what the float32 fast path of the lean reduction reads (stack_reduce.h, clip_fast32): the T lowest
// and T highest outputs and the middle window [(NP-T-1)/2, (NP+T)/2] in sorted order; the core in between is only ever
// SUMMED, so its internal order is irrelevant - it only has to hold the right set of values, which it does: a
// compare-exchange permutes values, and dropping one whose two outputs feed no needed output (backward liveness over the
// network) changes nothing on the needed wires.  64 slots: 469 of 543 compare-exchanges.  A wave that leaves the fast path
// sorts the column completely (sort_column) before the exact path reads it.
template <int NP, int P0, int T>
constexpr Net<NP> make_pruned_net()
{
    constexpr Net<NP> full = make_net<NP, P0>();
    bool live[NP] = {};
    for (int i = 0; i < T; i++) live[i] = live[NP - 1 - i] = true;
    for (int i = (NP - T - 1) / 2; i <= (NP + T) / 2; i++) live[i] = true;
    bool keep[NP * 12 + 1] = {};
    for (int c = full.n - 1; c >= 0; c--) {
        const int a = full.ce[c].a, b = full.ce[c].b;
        if (live[a] || live[b]) {
            keep[c] = true;
            live[a] = live[b] = true;
        }
    }
    Net<NP> net{};
    int n = 0;
    for (int c = 0; c < full.n; c++)
        if (keep[c]) net.ce[n++] = full.ce[c];
    net.n = n;
    return net;
}

// The network pruned to the outputs [LO, HI) (same backward liveness): the merge levels of stack_chunks.hip, where only the
// middle of the merged list is ever read.
template <int NP, int P0, int LO, int HI>
constexpr Net<NP> make_window_net()
{
    constexpr Net<NP> full = make_net<NP, P0>();
    bool live[NP] = {};
    for (int i = LO; i < HI; i++) live[i] = true;
    bool keep[NP * 12 + 1] = {};
    for (int c = full.n - 1; c >= 0; c--) {
        const int a = full.ce[c].a, b = full.ce[c].b;
        if (live[a] || live[b]) {
            keep[c] = true;
            live[a] = live[b] = true;
        }
    }
    Net<NP> net{};
    int n = 0;
    for (int c = 0; c < full.n; c++)
        if (keep[c]) net.ce[n++] = full.ce[c];
    net.n = n;
    return net;
}

// Compare-exchange.  Written as the two machine instructions: through fminf/fmaxf the compiler has to
// quiet possible signalling NaNs first (IEEE mode) and adds a v_max_f32 x, x, x canonicalisation per
// network input (~120 instructions per column); the columns are NaN-free by construction here.
__device__ __forceinline__ void cmpx(float &x, float &y)
{
    float lo, hi;
    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    x = lo;
    y = hi;
}

// Four values sorted with the three-input instructions: min3 / med3 / max3 sort three (one instruction per output where a
// compare-exchange network needs two per pair), and the fourth is inserted with min, med3, med3, max - 7 instructions
// instead of the 10 of the 5-comparator network (all of the same 4-cycle class, tools/issue_cost.hip).
__device__ __forceinline__ void sort4(float &a, float &b, float &c, float &d)
{
    float x0, x1, x2, o0, o1, o2, o3;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(a), "v"(b), "v"(c));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(a), "v"(b), "v"(c));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(x2) : "v"(a), "v"(b), "v"(c));
    asm("v_min_f32 %0, %1, %2" : "=v"(o0) : "v"(x0), "v"(d));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(o1) : "v"(x0), "v"(x1), "v"(d));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(o2) : "v"(x1), "v"(x2), "v"(d));
    asm("v_max_f32 %0, %1, %2" : "=v"(o3) : "v"(x2), "v"(d));
    a = o0;
    b = o1;
    c = o2;
    d = o3;
}

template <int NP, int P0, int T, int BASE, int... I>
__device__ __forceinline__ void net_chunk(float (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = T > 0 ? make_pruned_net<NP, P0, (T > 0 ? T : 1)>() : make_net<NP, P0>();
    (cmpx(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int NP, int P0, int T, int BASE>
__device__ __forceinline__ void net_from(float (&v)[NP])
{
    constexpr int total = T > 0 ? make_pruned_net<NP, P0, (T > 0 ? T : 1)>().n : make_net<NP, P0>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        net_chunk<NP, P0, T, BASE>(v, std::make_integer_sequence<int, len>{});
        net_from<NP, P0, T, BASE + len>(v);
    }
}

template <int NP, int P0, int LO, int HI, int BASE, int... I>
__device__ __forceinline__ void window_net_chunk(float (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = make_window_net<NP, P0, LO, HI>();
    (cmpx(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

// Runs the merge levels p = P0, 2 P0, .. of the Batcher network on v, pruned to the outputs [LO, HI).
template <int NP, int P0, int LO, int HI, int BASE = 0>
__device__ __forceinline__ void window_net_from(float (&v)[NP])
{
    constexpr int total = make_window_net<NP, P0, LO, HI>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        window_net_chunk<NP, P0, LO, HI, BASE>(v, std::make_integer_sequence<int, len>{});
        window_net_from<NP, P0, LO, HI, BASE + len>(v);
    }
}

// -------------------------------------------------------------------------------------------------
// Round 4: networks of 2-, 3- and 4-sorters for the float32 columns.  A compare-exchange costs two instructions for two
// outputs; v_min3 / v_med3 / v_max3 sort THREE values in three instructions, and a layer of 3-sorters moves every value past
// two others where a layer of compare-exchanges moves it past one.  tools/netsearch/netsearch.cpp (beam search, cost =
// instructions, verified on every 0-1 input the pre-sorted structure allows) found:
//   16 values: sort4 on the rows and on the columns of the 4 x 4 matrix (8 x 7) + ten 3-sorters = 86 instructions
//              (four sort4 + Batcher's 4+4 and 8+8 merges: 114);
//   8 values: eight 3-sorters = 24 (two sort4 + merge: 32);
//   merge of two sorted 16-blocks: 116 (Batcher: 65 compare-exchanges = 130).
// Above that the merge levels p = 32, 64 stay Batcher's.  A network for NP wires is the network for the next multiple of 16
// RESTRICTED to the wires below NP: the missing inputs are +inf, an inf never leaves the highest wire of a sorter, so a
// sorter simply loses those wires (a 3-sorter becomes a compare-exchange, a one-wire sorter disappears).
// -------------------------------------------------------------------------------------------------
struct NetOp {
    unsigned char w[4];     // wires in rank order (ascending); the lowest gets the minimum
    unsigned char k;        // 2: compare-exchange, 3: min3 / med3 / max3, 4: sort4
    unsigned char stage;    // 0: sorts a 16-block, 1: merges two 16-blocks, 2: Batcher merge levels p >= 32 (tools/netsearch/verify_opnet.cpp)
};

template <int NP>
struct OpNet {
    NetOp op[NP * 9 + 24];  // NP = 128: 8 x 18 + 4 x 46 + 2 x 161 + 321 = 971 < 1176
    int n;
};

constexpr NetOp kSort16Ops[18] = {
    {{0, 1, 2, 3}, 4},    {{4, 5, 6, 7}, 4},    {{8, 9, 10, 11}, 4},  {{12, 13, 14, 15}, 4},       // rows
    {{0, 4, 8, 12}, 4},   {{1, 5, 9, 13}, 4},   {{2, 6, 10, 14}, 4},  {{3, 7, 11, 15}, 4},         // columns
    {{3, 5, 12, 0}, 3},   {{7, 10, 13, 0}, 3},  {{5, 6, 9, 0}, 3},    {{2, 5, 8, 0}, 3},   {{1, 2, 4, 0}, 3},
    {{7, 9, 12, 0}, 3},   {{11, 13, 14, 0}, 3}, {{6, 7, 8, 0}, 3},    {{10, 11, 12, 0}, 3}, {{3, 4, 5, 0}, 3}};
constexpr NetOp kSort8Ops[8] = {{{0, 1, 2, 0}, 3}, {{3, 4, 5, 0}, 3}, {{0, 6, 7, 0}, 3}, {{1, 4, 6, 0}, 3},
                                {{2, 5, 7, 0}, 3}, {{0, 1, 3, 0}, 3}, {{2, 3, 4, 0}, 3}, {{4, 5, 6, 0}, 3}};
// two sorted 16-blocks (wires 0..15, 16..31) -> 32 sorted
constexpr NetOp kMerge16Ops[] = {
    {{9, 23, 0, 0}, 2}, {{8, 22, 0, 0}, 2}, {{7, 21, 0, 0}, 2}, {{10, 24, 0, 0}, 2}, {{6, 20, 0, 0}, 2}, {{11, 25, 0, 0}, 2},
    {{5, 19, 0, 0}, 2}, {{12, 26, 0, 0}, 2}, {{4, 12, 18, 0}, 3}, {{13, 19, 27, 0}, 3}, {{3, 17, 0, 0}, 2}, {{14, 20, 28, 0}, 3},
    {{15, 21, 29, 0}, 3}, {{2, 10, 16, 0}, 3}, {{7, 11, 17, 0}, 3}, {{1, 9, 13, 0}, 3}, {{0, 8, 12, 0}, 3}, {{18, 22, 30, 0}, 3},
    {{19, 23, 31, 0}, 3}, {{20, 24, 0, 0}, 2}, {{14, 16, 18, 0}, 3}, {{13, 15, 17, 0}, 3}, {{21, 23, 25, 0}, 3}, {{24, 26, 30, 0}, 3},
    {{6, 8, 10, 0}, 3}, {{1, 3, 5, 0}, 3}, {{0, 4, 6, 0}, 3}, {{25, 27, 31, 0}, 3}, {{5, 6, 7, 0}, 3}, {{15, 17, 19, 0}, 3},
    {{19, 20, 22, 0}, 3}, {{12, 14, 16, 0}, 3}, {{9, 11, 12, 0}, 3}, {{27, 28, 30, 0}, 3}, {{0, 1, 2, 0}, 3}, {{29, 30, 31, 0}, 3},
    {{13, 14, 0, 0}, 2}, {{7, 8, 0, 0}, 2}, {{21, 22, 0, 0}, 2}, {{23, 24, 0, 0}, 2}, {{25, 26, 0, 0}, 2}, {{17, 18, 0, 0}, 2},
    {{9, 10, 0, 0}, 2}, {{15, 16, 0, 0}, 2}, {{3, 4, 0, 0}, 2}};

// appends `o` shifted by `off`, restricted to the wires below NP
template <int NP>
constexpr void add_op(OpNet<NP> &net, const NetOp &o, int off, int stage)
{
    NetOp r{};
    r.stage = (unsigned char)stage;
    int k = 0;
    for (int i = 0; i < o.k; i++)
        if (o.w[i] + off < NP) r.w[k++] = (unsigned char)(o.w[i] + off);
    if (k >= 2) {
        r.k = (unsigned char)k;
        net.op[net.n++] = r;
    }
}

// T > 0: pruned to what clip_fast32 reads (see make_pruned_net): backward liveness over the sorters.
template <int NP, int T>
constexpr OpNet<NP> make_opnet()
{
    OpNet<NP> net{};
    net.n = 0;
    for (int base = 0; base < NP; base += 16) {
        if (NP - base <= 8) {
            for (const NetOp &o : kSort8Ops) add_op<NP>(net, o, base, 0);
        } else {
            for (const NetOp &o : kSort16Ops) add_op<NP>(net, o, base, 0);
        }
    }
    for (int base = 0; base + 16 < NP; base += 32)
        for (const NetOp &o : kMerge16Ops) add_op<NP>(net, o, base, 1);
    constexpr int P2 = next_pow2(NP);
    for (int p = 32; p < P2; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P2 - 1 - k; j += 2 * k) {
                const int lim = (k - 1 < P2 - j - k - 1) ? k - 1 : P2 - j - k - 1;
                for (int i = 0; i <= lim; i++)
                    if ((i + j) / (p * 2) == (i + j + k) / (p * 2)) {
                        const NetOp ce = {{(unsigned char)(i + j), (unsigned char)(i + j + k), 0, 0}, 2, 2};
                        add_op<NP>(net, ce, 0, 2);
                    }
            }
    if (T > 0) {
        bool live[NP] = {};
        for (int i = 0; i < T; i++) live[i] = live[NP - 1 - i] = true;
        for (int i = (NP - T - 1) / 2; i <= (NP + T) / 2; i++) live[i] = true;
        bool keep[NP * 9 + 24] = {};
        for (int c = net.n - 1; c >= 0; c--) {
            bool any = false;
            for (int i = 0; i < net.op[c].k; i++) any = any || live[net.op[c].w[i]];
            if (any) {
                keep[c] = true;
                for (int i = 0; i < net.op[c].k; i++) live[net.op[c].w[i]] = true;
            }
        }
        int n = 0;
        for (int c = 0; c < net.n; c++)
            if (keep[c]) net.op[n++] = net.op[c];
        net.n = n;
    }
    return net;
}

// (evaluated once per (NP, T), not once per sorter: the build would take three times as long)
template <int NP, int T>
inline constexpr OpNet<NP> kOpNet = make_opnet<NP, T>();

__device__ __forceinline__ void sort3(float &a, float &b, float &c)
{
    float x0, x1, x2;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(a), "v"(b), "v"(c));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(a), "v"(b), "v"(c));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(x2) : "v"(a), "v"(b), "v"(c));
    a = x0;
    b = x1;
    c = x2;
}

template <int NP, int T, int I>
__device__ __forceinline__ void opnet_apply(float (&v)[NP])
{
    constexpr NetOp o = kOpNet<NP, T>.op[I];
    if constexpr (o.k == 2) cmpx(v[o.w[0]], v[o.w[1]]);
    else if constexpr (o.k == 3) sort3(v[o.w[0]], v[o.w[1]], v[o.w[2]]);
    else sort4(v[o.w[0]], v[o.w[1]], v[o.w[2]], v[o.w[3]]);
}

template <int NP, int T, int BASE, int... I>
__device__ __forceinline__ void opnet_chunk(float (&v)[NP], std::integer_sequence<int, I...>)
{
    (opnet_apply<NP, T, BASE + I>(v), ...);
}

template <int NP, int T, int BASE = 0>
__device__ __forceinline__ void opnet_from(float (&v)[NP])
{
    constexpr int total = kOpNet<NP, T>.n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        opnet_chunk<NP, T, BASE>(v, std::make_integer_sequence<int, len>{});
        opnet_from<NP, T, BASE + len>(v);
    }
}

// PRUNE_T > 0: only the outputs clip_fast32 reads come out sorted (make_pruned_net / make_opnet).
template <int NP, int PRUNE_T = 0>
__device__ __forceinline__ void sort_column(float (&v)[NP])
{
#ifndef APGPU_VARIANT_BATCHER_ONLY
    if constexpr (NP >= 8 && NP % 4 == 0) {
        opnet_from<NP, PRUNE_T>(v);
    } else
#endif
    if constexpr (NP >= 4 && NP % 4 == 0) {
#pragma unroll
        for (int g = 0; g < NP; g += 4) sort4(v[g], v[g + 1], v[g + 2], v[g + 3]);
        net_from<NP, 4, PRUNE_T, 0>(v);
    } else if constexpr (NP > 1) {
        net_from<NP, 1, PRUNE_T, 0>(v);
    }
}

// v[LO + rel] for a per-lane rel in [0, LEN): binary multiplexer tree (LEN-1 v_cndmask), static
// register indices only (a runtime-indexed register array would be demoted to scratch memory).
template <int LO, int LEN, int NP>
__device__ __forceinline__ float pick_rel(const float (&v)[NP], int rel)
{
    if constexpr (LEN == 1) {
        return v[LO];
    } else if constexpr ((LEN & (LEN - 1)) == 0) {
        constexpr int H = LEN / 2;
        float lo = pick_rel<LO, H, NP>(v, rel);
        float hi = pick_rel<LO + H, H, NP>(v, rel);
        return (rel & H) ? hi : lo;
    } else {
        constexpr int H = next_pow2(LEN) / 2;               // 48 = 32 + 16, 96 = 64 + 32, 24 = 16 + 8, 12 = 8 + 4
        float lo = pick_rel<LO, H, NP>(v, rel);
        float hi = pick_rel<LO + H, LEN - H, NP>(v, rel - H);
        return (rel >= H) ? hi : lo;
    }
}

template <int NP>
__device__ __forceinline__ float pick_at(const float (&v)[NP], int idx)
{
    idx = idx < 0 ? 0 : (idx > NP - 1 ? NP - 1 : idx);
    return pick_rel<0, NP, NP>(v, idx);
}

// The two middle elements v[i1], v[i2] (i2 = i1 or i1 + 1) of the survivor range.  The clip trims a
// few values off either end, so the middle stays within a few slots of NP/2: if every lane of the
// wave is inside the 8-slot window around NP/2 the multiplexer needs 2 x 7 selects instead of
// 2 x (NP-1); otherwise the whole wave takes the full tree.
// 8-slot window [4K, 4K + 8) chosen at run time by a wave-uniform K (static register indices per case).
template <int K, int NP>
__device__ __forceinline__ void pick_window(const float (&v)[NP], int k, int i1, int i2, float &m1, float &m2)
{
    if (k == K) {
        m1 = pick_rel<4 * K, 8, NP>(v, i1 - 4 * K);
        m2 = pick_rel<4 * K, 8, NP>(v, i2 - 4 * K);
    } else if constexpr (4 * (K + 1) + 8 <= NP) {
        pick_window<K + 1, NP>(v, k, i1, i2, m1, m2);
    }
}

template <int NP>
__device__ __forceinline__ void pick_middle(const float (&v)[NP], int i1, int i2, float &m1, float &m2)
{
    if constexpr (NP <= 8) {
        m1 = pick_at<NP>(v, i1);
        m2 = pick_at<NP>(v, i2);
    } else {
        constexpr int WLO = NP / 2 - 4;
        const bool inside = (i1 >= WLO) && (i2 < WLO + 8);
        if (wave_all(inside)) {
            m1 = pick_rel<WLO, 8, NP>(v, i1 - WLO);
            m2 = pick_rel<WLO, 8, NP>(v, i2 - WLO);
        } else {
            // stacks with fewer frames than slots (or many rejected values) have their middle elsewhere:
            // take the 8-slot window around the first lane's middle if it holds every lane's
            int k = (__builtin_amdgcn_readfirstlane(i1) - 2) >> 2;
            k = k < 0 ? 0 : (k > NP / 4 - 2 ? NP / 4 - 2 : k);
            const bool inside_k = (i1 >= 4 * k) && (i2 < 4 * k + 8);
            if (wave_all(inside_k)) {
                pick_window<0, NP>(v, k, i1, i2, m1, m2);
            } else {
                m1 = pick_at<NP>(v, i1);
                m2 = pick_at<NP>(v, i2);
            }
        }
    }
}

// Packed uint16 compare-exchange network: two columns (the two halves of every register) sorted at once.
__device__ __forceinline__ void cmpx_pk16(uint32_t &x, uint32_t &y)
{
    uint32_t lo, hi;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    x = lo;
    y = hi;
}

template <int NP, int BASE, int... I>
__device__ __forceinline__ void net_chunk_pk16(uint32_t (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = make_net<NP>();
    (cmpx_pk16(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int NP, int BASE>
__device__ __forceinline__ void net_from_pk16(uint32_t (&v)[NP])
{
    constexpr int total = make_net<NP>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        net_chunk_pk16<NP, BASE>(v, std::make_integer_sequence<int, len>{});
        net_from_pk16<NP, BASE + len>(v);
    }
}

// The packed network pruned to what clip_fast32 reads (make_pruned_net), both columns at once.
template <int NP, int T, int BASE, int... I>
__device__ __forceinline__ void pruned_chunk_pk16(uint32_t (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = make_pruned_net<NP, 1, T>();
    (cmpx_pk16(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int NP, int T, int BASE = 0>
__device__ __forceinline__ void pruned_net_pk16(uint32_t (&v)[NP])
{
    constexpr int total = make_pruned_net<NP, 1, T>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        pruned_chunk_pk16<NP, T, BASE>(v, std::make_integer_sequence<int, len>{});
        pruned_net_pk16<NP, T, BASE + len>(v);
    }
}

// v[LO + rel] for rel in [0, LEN): select tree with static register indices (a branchy binary search over the
// registers gets turned into a run-time indexed array by the compiler, i.e. the column is demoted to scratch).
template <int LO, int LEN, int NP>
__device__ __forceinline__ uint32_t pick_rel_u32(const uint32_t (&v)[NP], int rel)
{
    if constexpr (LEN == 1) {
        return v[LO];
    } else if constexpr ((LEN & (LEN - 1)) == 0) {
        constexpr int H = LEN / 2;
        const uint32_t lo = pick_rel_u32<LO, H, NP>(v, rel);
        const uint32_t hi = pick_rel_u32<LO + H, H, NP>(v, rel);
        return (rel & H) ? hi : lo;
    } else {
        constexpr int H = next_pow2(LEN) / 2;
        const uint32_t lo = pick_rel_u32<LO, H, NP>(v, rel);
        const uint32_t hi = pick_rel_u32<LO + H, LEN - H, NP>(v, rel - H);
        return (rel >= H) ? hi : lo;
    }
}

}  // namespace apgpu_stack
