// Host-side helper (no device code): the first k entries of torch.randperm(n) on the CPU generator, bit-exact,
// without materialising the permutation.
//
// Detectron2's subsample_labels draws torch.randperm(#negatives)[:256] with #negatives ~ 268k per image
// (reached from the reference at aldi/distill.py:157,162,200-202); the sampling indices must reproduce that
// stream, but only the first <= 512 entries are ever used.  torch's CPU randperm is a forward Fisher-Yates
// driven by the generator's mt19937 (`z = random() % (n - i); swap(r[i], r[z + i])`), so position i is final
// after iteration i: run k iterations on a sparse map, then discard the remaining n-1-k draws by advancing the
// Mersenne state.  Operates on the 5056-byte blob of torch.get_rng_state() / set_rng_state().
#include "common.h"
#include <string.h>
#include <unordered_map>

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t s[MT_N];
    int left;
    uint64_t next;
    void regen() {
        auto twist = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
        uint32_t* p = s;
        for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ twist(p[0], p[1]);
        for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ twist(p[0], p[1]);
        *p = p[MT_M - MT_N] ^ twist(p[0], s[0]);
        left = MT_N;
        next = 0;
    }
    uint32_t draw() {
        if (--left == 0) regen();
        uint32_t y = s[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    void discard(long d) {
        while (d > 0) {
            if (left > 1) {
                long t = d < left - 1 ? d : left - 1;
                left -= (int)t;
                next += (uint64_t)t;
                d -= t;
            } else {
                (void)draw();
                --d;
            }
        }
    }
};

}  // namespace

// state: CPUGeneratorImplState blob {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; ...}.  out: k int64.
extern "C" int aldi_torch_randperm_prefix(unsigned char* state, long n, long k, long* out) {
    if (k > n) k = n;
    if (!state || n < 0 || k < 0 || (k > 0 && !out)) return aldi_set_error_msg(ALDI_ERR_ARG, "torch_randperm_prefix: bad args");
    Mt mt;
    int32_t left;
    uint64_t next, w;
    memcpy(&left, state + 8, 4);
    memcpy(&next, state + 16, 8);
    for (int i = 0; i < MT_N; ++i) { memcpy(&w, state + 24 + 8 * i, 8); mt.s[i] = (uint32_t)w; }
    mt.left = left;
    mt.next = next;
    std::unordered_map<long, long> moved;
    moved.reserve((size_t)(2 * k + 8));
    auto get = [&](long p) { auto it = moved.find(p); return it == moved.end() ? p : it->second; };
    const long iters = n > 0 ? n - 1 : 0;           // the reference loop: for (i = 0; i < n - 1; i++)
    const long run = k < iters ? k : iters;
    for (long i = 0; i < run; ++i) {
        long z = (long)(mt.draw() % (uint64_t)(n - i));
        long j = z + i;
        long vi = get(i), vj = get(j);
        moved[i] = vj;
        moved[j] = vi;
    }
    mt.discard(iters - run);
    for (long i = 0; i < k; ++i) out[i] = get(i);
    left = mt.left;
    next = mt.next;
    memcpy(state + 8, &left, 4);
    memcpy(state + 16, &next, 8);
    for (int i = 0; i < MT_N; ++i) { w = mt.s[i]; memcpy(state + 24 + 8 * i, &w, 8); }
    return ALDI_OK;
}
