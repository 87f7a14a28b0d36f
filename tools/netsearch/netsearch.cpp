// Development aid (round 4): beam search for cheap sorting / merging / selection networks built from 2-sorters (v_min + v_max,
// 2 instructions) and 3-sorters (v_min3 + v_med3 + v_max3, 3 instructions) - cost = VALU instructions, checked on every 0-1 input
// the pre-sorted structure allows (0-1 principle; a 3-sorter is a sub-network of compare-exchanges).
//   g++ -O3 -march=native -fopenmp -std=c++17 tools/netsearch/netsearch.cpp -o /tmp/netsearch
//   netsearch <n> blocks <len> <len> ... | grid <rows> <cols>   [--need r,r,r..] [--beam B] [--maxcost C] [--prefix a-b,a-b-c,..] [--no3]
// Wires are in rank order (wire i should end with rank i); blocks / rows are sorted ascending on consecutive wires.
// --need: only these ranks have to arrive on their wires (selection); default: all (sorting / merging).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_set>
#include <vector>
#include <omp.h>

typedef uint64_t u64;
struct Op { int a, b, c; };                                 // c < 0: 2-sorter
struct Cand { std::vector<u64> st; std::vector<Op> path; double score; };

static int n;
static std::vector<int> need;

static inline u64 apply(u64 v, const Op &o)
{
    if (o.c < 0) {
        const u64 x = (v >> o.a) & 1, y = (v >> o.b) & 1;
        if (x > y) v ^= (1ull << o.a) | (1ull << o.b);
        return v;
    }
    const int ones = (int)((v >> o.a) & 1) + (int)((v >> o.b) & 1) + (int)((v >> o.c) & 1);
    v &= ~((1ull << o.a) | (1ull << o.b) | (1ull << o.c));
    if (ones >= 1) v |= 1ull << o.c;
    if (ones >= 2) v |= 1ull << o.b;
    if (ones >= 3) v |= 1ull << o.a;
    return v;
}

static double score(const std::vector<u64> &st, long *viol_out = nullptr)
{
    long viol = 0;
    for (u64 v : st) {
        const int z = n - __builtin_popcountll(v);
        for (int r : need) viol += (((v >> r) & 1) != (u64)(r >= z));
    }
    if (viol_out) *viol_out = viol;
    return (double)viol * 4.0 + (double)st.size();
}

static u64 hash_states(const std::vector<u64> &st)
{
    u64 h = 1469598103934665603ull;
    for (u64 v : st) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; }
    return h;
}

int main(int argc, char **argv)
{
    n = atoi(argv[1]);
    std::vector<u64> init;
    int ai = 2;
    if (!strcmp(argv[ai], "blocks")) {
        ai++;
        std::vector<int> len;
        while (ai < argc && argv[ai][0] != '-') len.push_back(atoi(argv[ai++]));
        init.push_back(0);
        int base = 0;
        for (int L : len) {
            std::vector<u64> nxt;
            for (u64 v : init)
                for (int ones = 0; ones <= L; ones++) {
                    u64 w = v;
                    for (int i = L - ones; i < L; i++) w |= 1ull << (base + i);
                    nxt.push_back(w);
                }
            init.swap(nxt);
            base += L;
        }
    } else {                                                // grid R C: rows and columns sorted, wire = r * C + c
        const int R = atoi(argv[ai + 1]), C = atoi(argv[ai + 2]);
        ai += 3;
        std::vector<int> z(R, 0);
        // zeros per row non-increasing down the rows
        std::vector<std::vector<int>> all;
        std::vector<int> cur(R);
        std::function<void(int, int)> *rec = nullptr;
        (void)rec;
        // iterative enumeration
        std::vector<int> zz(R, C);
        while (true) {
            bool ok = true;
            for (int i = 0; i + 1 < R; i++) ok = ok && zz[i] >= zz[i + 1];
            if (ok) {
                u64 v = 0;
                for (int i = 0; i < R; i++)
                    for (int j = zz[i]; j < C; j++) v |= 1ull << (i * C + j);
                init.push_back(v);
            }
            int k = R - 1;
            while (k >= 0 && zz[k] == 0) { zz[k] = C; k--; }
            if (k < 0) break;
            zz[k]--;
        }
    }
    int beam = 64, maxcost = 200;
    bool use3 = true;
    std::vector<Op> prefix;
    for (int i = 0; i < n; i++) need.push_back(i);
    for (; ai < argc; ai++) {
        if (!strcmp(argv[ai], "--beam")) beam = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--maxcost")) maxcost = atoi(argv[++ai]);
        else if (!strcmp(argv[ai], "--no3")) use3 = false;
        else if (!strcmp(argv[ai], "--need")) {
            need.clear();
            char *s = argv[++ai];
            for (char *t = strtok(s, ","); t; t = strtok(nullptr, ",")) need.push_back(atoi(t));
        } else if (!strcmp(argv[ai], "--prefix")) {
            char *s = argv[++ai];
            for (char *t = strtok(s, ","); t; t = strtok(nullptr, ",")) {
                Op o{-1, -1, -1};
                sscanf(t, "%d-%d-%d", &o.a, &o.b, &o.c);
                prefix.push_back(o);
            }
        }
    }
    std::sort(init.begin(), init.end());
    init.erase(std::unique(init.begin(), init.end()), init.end());
    int pcost = 0;
    for (const Op &o : prefix) {
        for (u64 &v : init) v = apply(v, o);
        pcost += o.c < 0 ? 2 : 3;
    }
    std::sort(init.begin(), init.end());
    init.erase(std::unique(init.begin(), init.end()), init.end());
    printf("n %d, %zu 0-1 inputs, %zu needed ranks, beam %d, prefix cost %d\n", n, init.size(), need.size(), beam, pcost);
    std::vector<Op> moves;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++) {
            moves.push_back({a, b, -1});
            if (use3) for (int c = b + 1; c < n; c++) moves.push_back({a, b, c});
        }
    std::map<int, std::vector<Cand>> levels;
    levels[0].push_back({init, {}, score(init)});
    for (int cost = 0; cost <= maxcost; cost++) {
        auto it = levels.find(cost);
        if (it == levels.end()) continue;
        std::vector<Cand> cands;
        cands.swap(it->second);
        levels.erase(it);
        std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) { return x.score < y.score; });
        if ((int)cands.size() > beam) cands.resize(beam);
        long viol;
        score(cands[0].st, &viol);
        printf("cost %3d: %zu candidates, best: %zu states, %ld violations\n", cost + pcost, cands.size(), cands[0].st.size(), viol);
        fflush(stdout);
        if (viol == 0) {
            printf("FOUND cost %d (+ prefix %d = %d):", cost, pcost, cost + pcost);
            for (const Op &o : cands[0].path) o.c < 0 ? printf(" %d-%d", o.a, o.b) : printf(" %d-%d-%d", o.a, o.b, o.c);
            printf("\n");
            return 0;
        }
        std::unordered_set<u64> seen2, seen3;
        for (auto &c : levels[cost + 2]) seen2.insert(hash_states(c.st));
        for (auto &c : levels[cost + 3]) seen3.insert(hash_states(c.st));
        for (const Cand &c : cands) {
            std::vector<std::vector<Cand>> found(omp_get_max_threads());
#pragma omp parallel for schedule(dynamic, 64)
            for (size_t m = 0; m < moves.size(); m++) {
                const Op &o = moves[m];
                std::vector<u64> ns(c.st.size());
                bool changed = false;
                for (size_t i = 0; i < c.st.size(); i++) {
                    ns[i] = apply(c.st[i], o);
                    changed = changed || ns[i] != c.st[i];
                }
                if (!changed) continue;
                std::sort(ns.begin(), ns.end());
                ns.erase(std::unique(ns.begin(), ns.end()), ns.end());
                Cand nc{std::move(ns), c.path, 0.0};
                nc.path.push_back(o);
                nc.score = score(nc.st);
                found[omp_get_thread_num()].push_back(std::move(nc));
            }
            for (auto &fv : found)
                for (auto &nc : fv) {
                    const int cc = cost + (nc.path.back().c < 0 ? 2 : 3);
                    if (cc > maxcost) continue;
                    auto &seen = nc.path.back().c < 0 ? seen2 : seen3;
                    const u64 h = hash_states(nc.st);
                    if (!seen.insert(h).second) continue;
                    levels[cc].push_back(std::move(nc));
                }
        }
        // keep memory bounded: trim the next levels to a multiple of the beam
        for (int d = 2; d <= 3; d++) {
            auto &lv = levels[cost + d];
            if ((int)lv.size() > beam * 8) {
                std::nth_element(lv.begin(), lv.begin() + beam * 8, lv.end(), [](const Cand &x, const Cand &y) { return x.score < y.score; });
                lv.resize(beam * 8);
            }
        }
    }
    printf("nothing found up to cost %d\n", maxcost);
    return 1;
}
