// Host-side helper (no device code): the first k entries of torch.randperm(n) on the CPU generator, bit-exact,
// without materialising the permutation.
//
// Detectron2's subsample_labels draws torch.randperm(#negatives)[:256] with #negatives ~ 268k per image
// (reached from the reference at aldi/distill.py:157,162,200-202); the sampling indices must reproduce that
// stream, but only the first <= 512 entries are ever used.  torch's CPU randperm is a forward Fisher-Yates
// driven by the generator's mt19937 (`z = random() % (n - i); swap(r[i], r[z + i])`), so position i is final
// after iteration i: run k iterations on a sparse map, then discard the remaining n-1-k draws by advancing the
// Mersenne state.  Operates on the 5056-byte blob of torch.get_rng_state() / set_rng_state().
#include <stdint.h>
#include "../../include/aldi_hip.h"
int aldi_set_error_msg(int code, const char* msg);   // core.hip
#include <string.h>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t s[MT_N];
    int left;
    uint64_t next;
    // The state refill is what `discard` spends its time in (a 268k-entry permutation skips ~430 refills per image).  Within
    // a refill, element i reads s[i+1] (not yet rewritten) and s[i+397] / s[i-227] (old / rewritten >= 227 elements ago), so
    // blocks of W consecutive elements are independent: W = 8 with AVX2 (chosen at run time), 4 with baseline SSE2.
#define MT_LD(p) ({ vu v_; memcpy(&v_, (p), sizeof(v_)); v_; })
#define MT_TWIST(u, v) ((((u) & 0x80000000u) | ((v) & 0x7fffffffu)) >> 1) ^ ((0u - ((v) & 1u)) & 0x9908b0dfu)
    template <int W>
    __attribute__((always_inline)) inline void regen_w() {
        typedef uint32_t vu __attribute__((vector_size(W * 4)));
        constexpr int A = MT_N - MT_M;                         // 227: elements whose partner is s[i + 397]
        constexpr int A_VEC = A / W * W, B_VEC = A + (MT_N - 1 - A) / W * W;
        for (int i = 0; i < A_VEC; i += W) { vu a = MT_LD(s + i), b = MT_LD(s + i + 1), r = MT_LD(s + i + MT_M) ^ (MT_TWIST(a, b)); memcpy(s + i, &r, sizeof(r)); }
        for (int i = A_VEC; i < A; ++i) s[i] = s[i + MT_M] ^ twist1(s[i], s[i + 1]);
        for (int i = A; i < B_VEC; i += W) { vu a = MT_LD(s + i), b = MT_LD(s + i + 1), r = MT_LD(s + i - A) ^ (MT_TWIST(a, b)); memcpy(s + i, &r, sizeof(r)); }
        for (int i = B_VEC; i < MT_N - 1; ++i) s[i] = s[i - A] ^ twist1(s[i], s[i + 1]);
        s[MT_N - 1] = s[MT_M - 1] ^ twist1(s[MT_N - 1], s[0]);
        left = MT_N;
        next = 0;
    }
#undef MT_LD
#undef MT_TWIST
    static inline uint32_t twist1(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
    __attribute__((target("avx2"))) void regen_avx2() { regen_w<8>(); }
    void regen_sse2() { regen_w<4>(); }
    void regen() {
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) regen_avx2(); else regen_sse2();
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

namespace {

constexpr int kBlobBytes = 5056;          // CPUGeneratorImplState: legacy state (5048 B) + float normal sample + its valid flag

void load_blob(const unsigned char* state, Mt& mt) {
    int32_t left;
    uint64_t next, w;
    memcpy(&left, state + 8, 4);
    memcpy(&next, state + 16, 8);
    for (int i = 0; i < MT_N; ++i) { memcpy(&w, state + 24 + 8 * i, 8); mt.s[i] = (uint32_t)w; }
    mt.left = left;
    mt.next = next;
}
void store_blob(unsigned char* state, const Mt& mt) {
    const int32_t left = mt.left;
    const uint64_t next = mt.next;
    uint64_t w;
    memcpy(state + 8, &left, 4);
    memcpy(state + 16, &next, 8);
    for (int i = 0; i < MT_N; ++i) { w = mt.s[i]; memcpy(state + 24 + 8 * i, &w, 8); }
}
// at::mt19937(seed): what torch.manual_seed leaves in the engine (left = 1: the first draw regenerates the state)
void seed_mt(Mt& mt, uint64_t seed) {
    mt.s[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < MT_N; ++j) mt.s[j] = 1812433253u * (mt.s[j - 1] ^ (mt.s[j - 1] >> 30)) + (uint32_t)j;
    mt.left = 1;
    mt.next = 0;
}

// A pre-generated Mersenne stream (aldi_torch_rng_prefetch): the raw state words of consecutive refills, block 0 = the start
// state as it was.  Draw t of the consumer is word `origin + t`; skipping n draws is an addition, which is what takes the
// 268k-entry negative lists off the critical path between the step's two device phases.
struct Stream {
    bool seeded = false;
    uint64_t seed = 0;
    Mt start;
    long origin = 0, capacity = 0;          // position of the first draw in block 0; draws available
    std::vector<uint32_t> raw;
    void fill(long max_draws) {
        // a seeded engine (left == 1) refills before its first draw; otherwise next + left == 625 and the next draw is s[next]
        origin = start.left == 1 ? MT_N : (long)start.next;
        const long nb = (origin + max_draws) / MT_N + 2;
        if (raw.size() < (size_t)nb * MT_N) raw.resize((size_t)nb * MT_N);      // (kept across calls: fresh pages cost more than the refills)
        memcpy(raw.data(), start.s, sizeof(start.s));
        Mt t = start;
        for (long b = 1; b < nb; ++b) {
            t.regen();
            memcpy(raw.data() + (size_t)b * MT_N, t.s, sizeof(t.s));
        }
        capacity = nb * MT_N - origin;
    }
};
struct Cursor {
    const Stream* st;
    long pos;
    uint32_t draw() {
        uint32_t y = st->raw[(size_t)(st->origin + pos++)];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    void discard(long d) { pos += d; }
    // the engine a sequential consumer would be left with
    void leave(Mt& mt) const {
        if (pos == 0) { mt = st->start; return; }
        const long p = st->origin + pos - 1, b = p / MT_N, i = p % MT_N;
        memcpy(mt.s, st->raw.data() + (size_t)b * MT_N, sizeof(mt.s));
        mt.next = (uint64_t)(i + 1);
        mt.left = (int)(MT_N - i);
    }
};

constexpr int kMaxStreams = 8;
// The pre-generated streams and their filler threads are one object: `mu` serialises the entry points that touch it (two
// Python threads may call into the library), and the destructor joins -- a process that exits between a prefetch and the
// script that would have joined it (an exception in the caller, the last step of a run) would otherwise leave joinable
// std::threads in static storage and die in std::terminate, masking the real error.
struct StreamSet {
    std::mutex mu;
    Stream streams[kMaxStreams];        // buffers persist; the first `n` are valid for the next script
    int n = 0;
    long hits = 0;
    std::vector<std::thread> fillers;
    void join() {
        for (auto& t : fillers)
            if (t.joinable()) t.join();
        fillers.clear();
    }
    ~StreamSet() { join(); }
};
StreamSet g_set;
Stream* const g_streams = g_set.streams;
int& g_nstreams = g_set.n;
long& g_stream_hits = g_set.hits;
std::vector<std::thread>& g_fillers = g_set.fillers;
void join_fillers() { g_set.join(); }

// first k entries of torch.randperm(n) from `mt` (an engine or a stream cursor), which ends where torch's generator would
template <typename OutT, typename Gen>
void randperm_prefix(Gen& mt, long n, long k, OutT* out) {
    if (k > n) k = n;
    const long iters = n > 0 ? n - 1 : 0;           // the reference loop: for (i = 0; i < n - 1; i++)
    const long run = k < iters ? k : iters;
    if (n <= 16384) {
        // short lists (the ROI samples: a few thousand candidates): the permutation array itself, swapped in place
        static thread_local std::vector<int> perm;
        if ((long)perm.size() < n) perm.resize((size_t)n);
        for (long i = 0; i < n; ++i) perm[(size_t)i] = (int)i;
        for (long i = 0; i < run; ++i) {
            const long j = (long)(mt.draw() % (uint64_t)(n - i)) + i;
            const int t = perm[(size_t)i]; perm[(size_t)i] = perm[(size_t)j]; perm[(size_t)j] = t;
        }
        mt.discard(iters - run);
        for (long i = 0; i < k; ++i) out[i] = (OutT)perm[(size_t)i];
        return;
    }
    // sparse image of the permutation array: position -> value for the <= 2k positions touched so far (open addressing in a
    // table kept per thread and wiped slot by slot afterwards; a node-based map costs more than the shuffle itself)
    size_t cap = 64;
    while (cap < (size_t)(4 * k + 16)) cap <<= 1;
    static thread_local std::vector<long> keys, vals;
    static thread_local std::vector<uint32_t> used;
    if (keys.size() < cap) { keys.assign(cap, -1); vals.resize(cap); }
    used.clear();
    const size_t hmask = cap - 1;
    auto slot = [&](long pos) {
        size_t h = ((uint64_t)pos * 0x9E3779B97F4A7C15ull >> 20) & hmask;
        while (keys[h] != -1 && keys[h] != pos) h = (h + 1) & hmask;
        return h;
    };
    auto get = [&](long pos) { size_t h = slot(pos); return keys[h] == pos ? vals[h] : pos; };
    auto put = [&](long pos, long v) { size_t h = slot(pos); if (keys[h] == -1) used.push_back((uint32_t)h); keys[h] = pos; vals[h] = v; };
    for (long i = 0; i < run; ++i) {
        long z = (long)(mt.draw() % (uint64_t)(n - i));
        long j = z + i;
        long vi = get(i), vj = get(j);
        put(i, vj);
        put(j, vi);
    }
    mt.discard(iters - run);
    for (long i = 0; i < k; ++i) out[i] = (OutT)get(i);
    for (uint32_t h : used) keys[h] = -1;
}

}  // namespace

// state: CPUGeneratorImplState blob {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; ...}.  out: k int64.
extern "C" int aldi_torch_randperm_prefix(unsigned char* state, long n, long k, long* out) {
    if (k > n) k = n;
    if (!state || n < 0 || k < 0 || (k > 0 && !out)) return aldi_set_error_msg(ALDI_ERR_ARG, "torch_randperm_prefix: bad args");
    Mt mt;
    load_blob(state, mt);
    randperm_prefix<long>(mt, n, k, out);
    store_blob(state, mt);
    return ALDI_OK;
}

// A whole iteration's sampling draws in one call.  script: nops rows of {kind, a, b, out_off}:
//   kind 0  draw: the first min(b, a) entries of torch.randperm(a) -> out[out_off ...] (int32); out_off < 0 discards them
//   kind 1  torch.manual_seed(a)
// The generator state after a manual_seed does not depend on anything before it, so the script splits into segments at the
// seeds and the segments run on `threads` host threads (the cost is the Mersenne-Twister skip-ahead of the 268k-entry negative
// lists: ~80 us each, six per iteration); the blob ends as the sequential execution leaves it.
static int run_script(unsigned char* state, const long* script, int nops, int* out, int threads) {
    struct Seg { int begin, end; bool seeded; uint64_t seed; Mt mt; };
    std::vector<Seg> segs;
    segs.push_back(Seg{0, 0, false, 0, Mt()});
    for (int i = 0; i < nops; ++i) {
        const long kind = script[4 * i];
        if (kind == 1) {
            segs.back().end = i;
            segs.push_back(Seg{i + 1, i + 1, true, (uint64_t)script[4 * i + 1], Mt()});
        } else if (kind != 0 || script[4 * i + 1] < 0 || script[4 * i + 2] < 0) {
            return aldi_set_error_msg(ALDI_ERR_ARG, "torch_rng_script: bad op");
        }
    }
    segs.back().end = nops;
    std::lock_guard<std::mutex> lock(g_set.mu);
    join_fillers();
    auto ops = [&](Seg& sg, auto& gen) {
        for (int i = sg.begin; i < sg.end; ++i) {
            const long n = script[4 * i + 1], k = script[4 * i + 2], off = script[4 * i + 3];
            if (off < 0) {                               // discarded draws: only the generator moves (randperm(n) consumes n - 1 values)
                gen.discard(n > 0 ? n - 1 : 0);
                continue;
            }
            randperm_prefix<int>(gen, n, k, out + off);
        }
    };
    auto run = [&](Seg& sg) { ops(sg, sg.mt); };
    // segments whose stream was pre-generated (same seed / same engine state, long enough) cost a few hundred draws each and
    // run here; the others skip through their Mersenne refills, on their own threads when there are several
    std::vector<Seg*> slow;
    for (auto& sg : segs) {
        if (sg.seeded) seed_mt(sg.mt, sg.seed);
        else load_blob(state, sg.mt);
        long total = 0;
        for (int i = sg.begin; i < sg.end; ++i) total += script[4 * i + 1] > 0 ? script[4 * i + 1] - 1 : 0;
        const Stream* hit = nullptr;
        for (int si = 0; si < g_nstreams; ++si) {
            const Stream& st = g_streams[si];
            const bool same = sg.seeded ? (st.seeded && st.seed == sg.seed)
                                        : (!st.seeded && st.start.left == sg.mt.left && st.start.next == sg.mt.next &&
                                           memcmp(st.start.s, sg.mt.s, sizeof(sg.mt.s)) == 0);
            if (same && total <= st.capacity) { hit = &st; break; }
        }
        if (hit) {
            Cursor cur{hit, 0};
            ops(sg, cur);
            cur.leave(sg.mt);
            g_stream_hits++;
        } else {
            slow.push_back(&sg);
        }
    }
    if (threads > 1 && slow.size() > 1) {
        std::vector<std::thread> pool;
        for (size_t i = 1; i < slow.size(); ++i) pool.emplace_back(run, std::ref(*slow[i]));
        run(*slow[0]);
        for (auto& t : pool) t.join();
    } else {
        for (Seg* sg : slow) run(*sg);
    }
    const Seg& last = segs.back();
    if (last.seeded) {                     // what torch.manual_seed(seed) + the draws leave in the blob
        const uint64_t seed = last.seed;
        const int32_t one = 1, zero = 0;
        memcpy(state, &seed, 8);
        memcpy(state + 12, &one, 4);       // seeded
        memset(state + 24 + 8 * MT_N, 0, 24);                 // normal_x, normal_y, normal_rho
        memcpy(state + 24 + 8 * MT_N + 24, &zero, 4);         // normal_is_valid
        memset(state + kBlobBytes - 8, 0, 8);                 // next_float_normal_sample + its valid flag
    }
    store_blob(state, last.mt);
    return ALDI_OK;
}

extern "C" int aldi_torch_rng_script(unsigned char* state, const long* script, int nops, int* out, int threads) {
    if (!state || !script || nops < 0 || (!out && nops > 0)) return aldi_set_error_msg(ALDI_ERR_ARG, "torch_rng_script: bad args");
    return run_script(state, script, nops, out, threads);
}

// The whole host phase of one fused ALDI iteration in one call: from the device's list lengths to the sampling positions,
// their counts, the ROI row offsets and the distillation normalisers, written into the pinned upload buffer.  Order of the
// draws on the global CPU generator (aldi/distill.py:148-162,200-202, aldi/helpers.py:17-26), chunk by chunk
// (= micro-step by micro-step of the reference schedule, aldi/trainer.py:51-52,86-89):
//   [distillation chunk: manual_seed(current seed) by the teacher's eval pass, then the seeder is reset: next seed] RPN sample
//   (positives, negatives per image), manual_seed(current seed) by the student's roi_heads hook, ROI sample; a distillation
//   chunk continues with manual_seed(current seed), the teacher's identical ROI draws (discarded) and the fresh RPN sample of
//   get_rpn_losses.
// counts: [N][2] RPN (positives, negatives) then [N][2] ROI.  chunks: nch rows {kind (1 = distillation), n0, n1}.
// seeds: the ManualSeed hook's seed before the iteration, then after each distillation chunk's reset_seed (1 + #distillation chunks).
// word0: int32 word offsets into `words` of {rsel [N][2][rpn_batch], rnsel [N][2], osel [N][2][roi_batch], onsel [N][2],
// row_off [N], dsel [Nd][2][rpn_batch], dnsel [Nd][2], nvf [#distillation chunks][2]} (Nd = images of the distillation chunks,
// in chunk order).  rows_out [N]: sampled ROI rows per image.
extern "C" int aldi_step_draws(unsigned char* state, const int* counts, int N, const int* chunks, int nch, const long* seeds, int nseeds,
                               int rpn_batch, int rpn_pos_cap, int roi_batch, int roi_pos_cap, int* words, const int* word0, int* rows_out,
                               int threads) {
    if (!state || !counts || !chunks || !seeds || !words || !word0 || !rows_out || N < 1 || nch < 1 || nseeds < 1)
        return aldi_set_error_msg(ALDI_ERR_ARG, "step_draws: bad args");
    std::vector<long> sc;
    sc.reserve(64 * (size_t)N);
    auto sample = [&](int o_sel, int o_nsel, int row0, const int* cnt, int nimg, int batch, int pos_cap, int* sums) {
        for (int i = 0; i < nimg; ++i) {
            const int npos = cnt[2 * i], nneg = cnt[2 * i + 1];
            const int num_pos = npos < pos_cap ? npos : pos_cap;
            const int num_neg = nneg < batch - num_pos ? nneg : batch - num_pos;
            const long base = o_sel < 0 ? -1 : (long)o_sel + (long)(row0 + i) * 2 * batch;
            sc.insert(sc.end(), {0, (long)npos, (long)num_pos, base, 0, (long)nneg, (long)num_neg, base < 0 ? -1 : base + batch});
            if (o_nsel >= 0) { words[o_nsel + 2 * (row0 + i)] = num_pos; words[o_nsel + 2 * (row0 + i) + 1] = num_neg; }
            if (sums) { sums[2 * i] = num_pos; sums[2 * i + 1] = num_neg; }
        }
    };
    const int* rpn = counts;
    const int* roi = counts + 2 * N;
    int k = 0, d0 = 0;                              // resets so far / distillation images so far
    std::vector<int> pn(2 * (size_t)N), dn;
    for (int c = 0; c < nch; ++c) {
        const int kind = chunks[3 * c], n0 = chunks[3 * c + 1], n1 = chunks[3 * c + 2];
        if (n0 < 0 || n1 > N || n1 <= n0) return aldi_set_error_msg(ALDI_ERR_ARG, "step_draws: bad chunk");
        if (kind == 1) {
            if (k + 1 >= nseeds) return aldi_set_error_msg(ALDI_ERR_ARG, "step_draws: one seed per distillation chunk (+ the initial one)");
            sc.insert(sc.end(), {1, seeds[k], 0, 0});
            ++k;
        }
        sample(word0[0], word0[1], n0, rpn + 2 * n0, n1 - n0, rpn_batch, rpn_pos_cap, nullptr);
        sc.insert(sc.end(), {1, seeds[k], 0, 0});
        sample(word0[2], word0[3], n0, roi + 2 * n0, n1 - n0, roi_batch, roi_pos_cap, pn.data() + 2 * n0);
        if (kind == 1) {
            sc.insert(sc.end(), {1, seeds[k], 0, 0});
            sample(-1, -1, 0, roi + 2 * n0, n1 - n0, roi_batch, roi_pos_cap, nullptr);
            dn.assign(2 * (size_t)(n1 - n0), 0);
            sample(word0[5], word0[6], d0, rpn + 2 * n0, n1 - n0, rpn_batch, rpn_pos_cap, dn.data());
            int n_valid = 0, n_fg = 0;
            for (int i = 0; i < n1 - n0; ++i) { n_fg += dn[2 * i]; n_valid += dn[2 * i] + dn[2 * i + 1]; }
            words[word0[7] + 2 * (k - 1)] = n_valid;
            words[word0[7] + 2 * (k - 1) + 1] = n_fg;
            d0 += n1 - n0;
        }
    }
    int off = 0;
    for (int i = 0; i < N; ++i) {
        rows_out[i] = pn[2 * i] + pn[2 * i + 1];
        words[word0[4] + i] = off;
        off += rows_out[i];
    }
    return run_script(state, sc.data(), (int)(sc.size() / 4), words, threads);
}

// Pre-generates, on background threads, the Mersenne streams the next aldi_torch_rng_script call will consume: the one that
// continues `state` (the generator blob as it is NOW; NULL = none) and one per torch.manual_seed value in `seeds`, each
// `max_draws` long.  The caller issues this while the device is busy with the phase whose results size the draws; the
// script then only indexes into the streams.  A stream that does not match the script's actual engine state / seeds, or is
// too short, is ignored (the script falls back to skipping through the refills itself).
extern "C" int aldi_torch_rng_prefetch(const unsigned char* state, const long* seeds, int nseeds, long max_draws) {
    if (nseeds < 0 || (nseeds > 0 && !seeds) || max_draws < 0 || max_draws > (1L << 28))
        return aldi_set_error_msg(ALDI_ERR_ARG, "torch_rng_prefetch: bad args");
    std::lock_guard<std::mutex> lock(g_set.mu);
    join_fillers();
    if (nseeds + (state ? 1 : 0) > kMaxStreams) return aldi_set_error_msg(ALDI_ERR_ARG, "torch_rng_prefetch: too many streams");
    int j = 0;
    if (state) {
        Stream& st = g_streams[j];
        st.seeded = false;
        load_blob(state, st.start);
        // (an engine state this code knows how to continue: seeded-and-untouched, or in the middle of a block)
        if (st.start.left >= 1 && st.start.left <= MT_N && (st.start.left == 1 || (long)st.start.next + st.start.left == MT_N + 1)) ++j;
    }
    for (int i = 0; i < nseeds; ++i) {
        Stream& st = g_streams[j++];
        st.seeded = true;
        st.seed = (uint64_t)seeds[i];
        seed_mt(st.start, st.seed);
    }
    g_nstreams = j;
    for (int i = 0; i < j; ++i) {
        Stream* st = &g_streams[i];
        g_fillers.emplace_back([st, max_draws] { st->fill(max_draws); });
    }
    return ALDI_OK;
}

// Reaps the background fillers of aldi_torch_rng_prefetch (blocks until their streams are complete).  The script joins them itself; a
// caller that has nothing else to do before the list lengths arrive calls this first, so that the joins are not paid between the phases.
extern "C" int aldi_torch_rng_prefetch_wait(void) {
    std::lock_guard<std::mutex> lock(g_set.mu);
    join_fillers();
    return ALDI_OK;
}

// how many script segments were served from a pre-generated stream since the library was loaded (tests, bench stats)
extern "C" int aldi_torch_rng_prefetch_hits(void) {
    std::lock_guard<std::mutex> lock(g_set.mu);
    return (int)(g_stream_hits & 0x7fffffff);
}
