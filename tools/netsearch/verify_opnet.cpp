// Development aid / CPU test helper: prints the instruction count of every sort_column network (make_opnet) and verifies it
// by the 0-1 principle, stage by stage: each 16-block exhaustively (2^16 inputs), each merge of two sorted 16-blocks on its
// 17 x 17 inputs, the Batcher levels on every combination of sorted 32-blocks; pruned networks (T > 0) must deliver the ranks
// clip_fast32 reads on their wires.   hipcc -std=c++17 -I include -I astrophotography_amd/csrc tools/netsearch/verify_opnet.cpp -o /tmp/verify_opnet
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "stack_sort.h"

using namespace apgpu_stack;
typedef unsigned __int128 u128;

static u128 apply(u128 v, const NetOp &o)
{
    int ones = 0;
    u128 mask = 0;
    for (int i = 0; i < o.k; i++) {
        ones += (int)((v >> o.w[i]) & 1);
        mask |= (u128)1 << o.w[i];
    }
    v &= ~mask;
    for (int i = o.k - ones; i < o.k; i++) v |= (u128)1 << o.w[i];
    return v;
}

static int popc(u128 v) { return __builtin_popcountll((uint64_t)v) + __builtin_popcountll((uint64_t)(v >> 64)); }

template <int NP, int T>
static bool verify()
{
    constexpr OpNet<NP> net = make_opnet<NP, T>();
    int instr = 0, cnt[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < net.n; c++) {
        instr += net.op[c].k == 4 ? 7 : net.op[c].k;
        cnt[net.op[c].k]++;
    }
    bool ok = true;
    // stage 0: every block, exhaustively, with the full (unpruned) network's block ops - pruning only removes ops, so test the
    // pruned network end to end below on the structured inputs, and the full one stage by stage
    constexpr OpNet<NP> full = make_opnet<NP, 0>();
    for (int base = 0; base < NP; base += 16) {
        const int len = NP - base < 16 ? NP - base : 16;
        for (uint32_t x = 0; x < (1u << len); x++) {
            u128 v = (u128)x << base;
            for (int c = 0; c < full.n; c++)
                if (full.op[c].stage == 0) v = apply(v, full.op[c]);
            const uint32_t y = (uint32_t)(v >> base) & ((1u << len) - 1);
            const int ones = __builtin_popcount(x);
            if (y != (((1u << len) - 1) >> (len - ones) << (len - ones))) ok = false;
        }
    }
    // stage 1: every pair of sorted 16-blocks
    for (int base = 0; base + 16 < NP; base += 32) {
        const int l0 = 16, l1 = NP - base - 16 < 16 ? NP - base - 16 : 16;
        for (int o0 = 0; o0 <= l0; o0++)
            for (int o1 = 0; o1 <= l1; o1++) {
                u128 v = 0;
                for (int i = l0 - o0; i < l0; i++) v |= (u128)1 << (base + i);
                for (int i = l1 - o1; i < l1; i++) v |= (u128)1 << (base + 16 + i);
                for (int c = 0; c < full.n; c++)
                    if (full.op[c].stage == 1) v = apply(v, full.op[c]);
                const int len = l0 + l1, ones = o0 + o1;
                for (int i = 0; i < len; i++)
                    if ((int)((v >> (base + i)) & 1) != (i >= len - ones)) ok = false;
            }
    }
    // stages 1 + 2 of the network under test (pruned or not) on every combination of sorted 16-blocks (NP <= 64) or, for larger
    // NP, stage 2 alone on every combination of sorted 32-blocks
    const int B = NP <= 64 ? 16 : 32;
    const int nb = (NP + B - 1) / B;
    std::vector<int> ones(nb, 0);
    long cases = 0;
    while (true) {
        u128 v = 0;
        int tot = 0;
        for (int b = 0; b < nb; b++) {
            const int len = NP - b * B < B ? NP - b * B : B;
            for (int i = len - ones[b]; i < len; i++) v |= (u128)1 << (b * B + i);
            tot += ones[b];
        }
        for (int c = 0; c < net.n; c++)
            if (net.op[c].stage == 2 || (B == 16 && net.op[c].stage == 1)) v = apply(v, net.op[c]);
        cases++;
        if (T == 0) {
            for (int i = 0; i < NP; i++)
                if ((int)((v >> i) & 1) != (i >= NP - tot)) ok = false;
        } else {
            if (popc(v) != tot) ok = false;
            for (int i = 0; i < NP; i++) {
                const bool needed = i < T || i >= NP - T || (i >= (NP - T - 1) / 2 && i <= (NP + T) / 2);
                if (needed && (int)((v >> i) & 1) != (i >= NP - tot)) ok = false;
            }
        }
        int k = 0;
        while (k < nb) {
            const int len = NP - k * B < B ? NP - k * B : B;
            if (ones[k] < len) { ones[k]++; break; }
            ones[k] = 0;
            k++;
        }
        if (k == nb) break;
    }
    // what the old scheme cost: sort4 groups + Batcher from p = 4 (pruned alike)
    int old_instr = 0;
    if constexpr (T > 0) old_instr = (NP / 4) * 7 + 2 * make_pruned_net<NP, 4, (T > 0 ? T : 1)>().n;
    else old_instr = (NP / 4) * 7 + 2 * make_net<NP, 4>().n;
    printf("NP %3d T %d: %4d instructions (%3d sort4, %3d sort3, %3d compare-exchanges; sort4 + Batcher: %4d), %ld structured cases: %s\n", NP, T, instr,
           cnt[4], cnt[3], cnt[2], old_instr, cases, ok ? "ok" : "FAILED");
    return ok;
}

int main()
{
    bool ok = true;
    ok &= verify<8, 0>();
    ok &= verify<12, 0>();
    ok &= verify<16, 0>();
    ok &= verify<20, 0>();
    ok &= verify<24, 0>();
    ok &= verify<28, 0>();
    ok &= verify<36, 0>();
    ok &= verify<44, 0>();
    ok &= verify<52, 0>();
    ok &= verify<60, 0>();
    ok &= verify<88, 0>();
    ok &= verify<120, 0>();
    ok &= verify<32, 0>();
    ok &= verify<40, 0>();
    ok &= verify<48, 0>();
    ok &= verify<56, 0>();
    ok &= verify<64, 0>();
    ok &= verify<72, 0>();
    ok &= verify<80, 0>();
    ok &= verify<96, 0>();
    ok &= verify<104, 0>();
    ok &= verify<112, 0>();
    ok &= verify<128, 0>();
    ok &= verify<16, 4>();
    ok &= verify<20, 4>();
    ok &= verify<24, 4>();
    ok &= verify<28, 4>();
    ok &= verify<36, 4>();
    ok &= verify<44, 4>();
    ok &= verify<52, 4>();
    ok &= verify<60, 4>();
    ok &= verify<88, 4>();
    ok &= verify<32, 4>();
    ok &= verify<40, 4>();
    ok &= verify<48, 4>();
    ok &= verify<56, 4>();
    ok &= verify<64, 4>();
    ok &= verify<72, 4>();
    ok &= verify<80, 4>();
    ok &= verify<96, 4>();
    ok &= verify<24, 8>();
    ok &= verify<28, 8>();
    ok &= verify<36, 8>();
    ok &= verify<44, 8>();
    ok &= verify<52, 8>();
    ok &= verify<60, 8>();
    ok &= verify<88, 8>();
    ok &= verify<32, 8>();
    ok &= verify<40, 8>();
    ok &= verify<48, 8>();
    ok &= verify<56, 8>();
    ok &= verify<64, 8>();
    ok &= verify<72, 8>();
    ok &= verify<80, 8>();
    ok &= verify<96, 8>();
    printf(ok ? "ALL OK\n" : "FAILURES\n");
    return ok ? 0 : 1;
}
