// Finder-level entry point: whole-batch drop-in for GeneFinder.find_genes() in meta or single
// mode (ref: lib.pyx:5281-5469).  Host orchestration only; every per-base / per-node stage runs in
// the HIP kernels of pipeline.hip and dp.hip.  The tail of the reference's _dynamic_programming
// (traceback untangling), eliminate_bad_genes, Genes._extract and Genes._tweak_final_starts runs on
// the device as well (tail.inl and the k_tail_* kernels below); the same functions, compiled for the
// host too, remain as a cross-check (PGA_TAIL=host: one contig per worker thread over the winning
// models' nodes brought home).
#include "pga_internal.h"
#include "pipeline.h"
#include "dpw_core.h"
#include "dev_common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace {

struct Buf { void* p = nullptr; size_t cap = 0; };

}  // namespace

// Device arrays of the last stage-level run (single contig), kept for the training driver, which continues from them.
struct LastRun {
    const uint8_t* d_dig = nullptr;
    GroupArrays ga{};
    ChainArrays ca{};
    int n_nodes = 0;
    int len = 0;
};
// Host threads that stay around between calls (the per-contig tail work is short: starting threads for it every call
// would cost more than the work).  run(fn, k): fn runs on the caller and on k - 1 pool threads, returns when all are done.
class WorkerPool {
public:
    ~WorkerPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(const std::function<void()>& fn, int k) {
        k = std::max(1, k);
        {
            std::lock_guard<std::mutex> g(mu_);
            while ((int)th_.size() < k - 1) th_.emplace_back([this, id = (int)th_.size()] { loop(id); });
            job_ = &fn; want_ = k - 1; left_ = k - 1; gen_++;
        }
        if (k > 1) cv_.notify_all();
        fn();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return left_ == 0; });
        job_ = nullptr;
    }
private:
    void loop(const int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void()>* job = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (id < want_) job = job_;
            }
            if (!job) continue;
            (*job)();
            { std::lock_guard<std::mutex> g(mu_); if (--left_ == 0) done_.notify_all(); }
        }
    }
    std::mutex mu_; std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void()>* job_ = nullptr;
    int want_ = 0, left_ = 0; unsigned long long gen_ = 0; bool stop_ = false;
};

struct FinderState {
    bool stage_full = false;        // a batch of this context did not fit the half-density staging of the extraction: full staging from then on
    LastRun last;
    WorkerPool pool;
    std::map<std::string, Buf> dev, pin;
    std::vector<int> model_group;   // model -> translation-table group
    std::vector<int> group_tt;
    ModelScoreConst* d_msc = nullptr;
    double* d_model_gc = nullptr;   // GC of every loaded model, and its translation-table group: which groups a contig needs
    int32_t* d_model_grp = nullptr;
    // the hexamer tables (gene_dc) of each group's models interleaved: d_gil + gil_off[g], [4096][gil_stride[g]]; d_model_rank[m] = column
    double* d_gil = nullptr; int32_t* d_model_rank = nullptr;
    std::vector<size_t> gil_off; std::vector<int> gil_stride; std::vector<int32_t> model_rank;
    unsigned* d_sd_lut = nullptr;   // RBS search table (pga_launch_sd_lut), filled when the context's finder state is created
    std::mutex spare_mu; void* spare_p = nullptr; size_t spare_cap = 0;   // the letters' allocation of the last batch that was freed
    // Round 6: an upload (pga_batch_create) may run BESIDE a call of the same context -- a caller that keeps the next batch on its way
    // while the current one is being worked on: double buffering of H2D against the kernels.  So the upload has a stream, a pinned
    // staging area and a worker pool of its own and touches nothing else of the context; one upload at a time per context (up_mu).
    std::mutex up_mu; WorkerPool up_pool; hipStream_t up_stream = nullptr; void* up_pin = nullptr; size_t up_pin_cap = 0;
    hipEvent_t e_start = nullptr, e_stop = nullptr, e_dp0[4] = {}, e_dp1[4] = {};
    hipEvent_t e_aux[20] = {};      // around the topology / schedule launches of each group (pga_dp_timings), and the first of two connection-scoring launches
};

namespace {

struct StageTimer {
    bool on; std::chrono::steady_clock::time_point t0, tbegin; const char* names[32]; double ms[32]; int n = 0;
    StageTimer() : on(getenv("PGA_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), tbegin(t0) {}
    void mark(const char* name) {
        if (!on || n >= 32) return;
        auto t1 = std::chrono::steady_clock::now();
        names[n] = name; ms[n++] = std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1;
    }
    ~StageTimer() {
        if (!on) return;
        fprintf(stderr, "[pga timing]");
        for (int i = 0; i < n; i++) fprintf(stderr, " %s=%.2fms", names[i], ms[i]);
        fprintf(stderr, " | total=%.2fms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tbegin).count());
    }
};

#define HT(ctx, expr) do { int rc__ = pga_hip_try_(ctx, (expr), #expr); if (rc__ != PGA_OK) return rc__; } while (0)

// Buffers of destroyed contexts wait in a process-wide cache for the next context: a caller that makes a context per file pays the
// 120 allocations of a context (50 ms, and as much again to free them) once.  Blocks enter the cache only when their context is
// released, after its stream has drained; the cache is bounded (PGA_CACHE_GB, default 16 GB of device and 2 GB of pinned memory)
// and pga_release_cached() empties it.
struct CachedBlock { void* p; size_t cap; int dev; };
static std::mutex g_cache_mu;
static std::vector<CachedBlock> g_cache[2];          // 0: device, 1: pinned
static size_t g_cached[2] = {0, 0};
static size_t cache_limit(int kind) {
    static const size_t gb = [] { const char* e = getenv("PGA_CACHE_GB"); return e ? (size_t)std::max(0, atoi(e)) : (size_t)16; }();
    return kind == 0 ? gb << 30 : std::min<size_t>(gb << 30, (size_t)2 << 30);
}
static void* cache_take(int kind, int dev, size_t want, size_t* cap, size_t roomy = 2) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    std::vector<CachedBlock>& v = g_cache[kind];
    int best = -1;
    for (int k = 0; k < (int)v.size(); k++)
        if (v[k].dev == dev && v[k].cap >= want && v[k].cap <= roomy * want + ((size_t)1 << 20) && (best < 0 || v[k].cap < v[best].cap)) best = k;
    if (best < 0) return nullptr;
    void* p = v[best].p; *cap = v[best].cap;
    g_cached[kind] -= v[best].cap;
    v.erase(v.begin() + best);
    return p;
}
static bool cache_put(int kind, int dev, void* p, size_t cap) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    if (g_cached[kind] + cap > cache_limit(kind) || g_cache[kind].size() >= 4096) return false;
    g_cache[kind].push_back(CachedBlock{p, cap, dev});
    g_cached[kind] += cap;
    return true;
}
// a pinned block of a freed result: always taken -- when the cache is full the oldest pinned result blocks go back to the runtime first
// (a workload with results of another size would otherwise pay hipHostMalloc / hipHostFree on every call: round 6, seen as 7 ms steps of a
// 4 ms workload behind a job that had filled the cache)
static void cache_put_result(void* p, size_t cap) {
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        std::vector<CachedBlock>& v = g_cache[1];
        for (size_t k = 0; k < v.size() && (g_cached[1] + cap > cache_limit(1) || v.size() >= 4096); ) {
            if (v[k].dev != -1) { k++; continue; }
            drop.push_back(v[k].p); g_cached[1] -= v[k].cap; v.erase(v.begin() + (long)k);
        }
        if (g_cached[1] + cap <= cache_limit(1) && v.size() < 4096) { v.push_back(CachedBlock{p, cap, -1}); g_cached[1] += cap; p = nullptr; }
    }
    for (void* q : drop) (void)hipHostFree(q);
    if (p != nullptr) (void)hipHostFree(p);
}
extern "C" void pga_release_cached(void) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    for (auto& b : g_cache[0]) hipFree(b.p);
    for (auto& b : g_cache[1]) hipHostFree(b.p);
    g_cache[0].clear(); g_cache[1].clear(); g_cached[0] = g_cached[1] = 0;
}
// An allocation that fails while blocks are parked here (a cached block only serves requests of about its own size) gets a second
// try after the parked blocks of that kind have gone back to the runtime.
template <class Alloc>
static hipError_t alloc_or_evict(int kind, Alloc alloc) {
    hipError_t e = alloc();
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();                          // the failed attempt is not this call's error
    bool had;
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        had = !g_cache[kind].empty();
        for (auto& b : g_cache[kind]) { if (kind == 0) hipFree(b.p); else hipHostFree(b.p); }
        g_cache[kind].clear(); g_cached[kind] = 0;
    }
    return had ? alloc() : e;
}

int ensure_dev(pga_ctx* c, const char* name, size_t bytes, void** out) {
    Buf& b = c->finder->dev[name];
    if (b.cap < bytes || !b.p) {
        if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }     // (not to the cache: kernels of this call may still read it; hipFree waits)
        // room to grow: the calls of a job are about the same size, so the big buffers get a sixteenth on top, the small ones a quarter
        size_t want = bytes + (bytes >= ((size_t)64 << 20) ? bytes / 16 : bytes / 4) + 256;
        b.p = cache_take(0, c->device, want, &want);
        if (!b.p) HT(c, alloc_or_evict(0, [&] { return hipMalloc(&b.p, want); }));
        b.cap = want;
    }
    *out = b.p;
    return PGA_OK;
}
int ensure_pin(pga_ctx* c, const char* name, size_t bytes, void** out) {
    Buf& b = c->finder->pin[name];
    if (b.cap < bytes || !b.p) {
        if (b.p) { hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        b.p = cache_take(1, c->device, want, &want);
        if (!b.p) HT(c, alloc_or_evict(1, [&] { return hipHostMalloc(&b.p, want, hipHostMallocDefault); }));
        b.cap = want;
    }
    *out = b.p;
    return PGA_OK;
}
#define DEVBUF(var, type, name, count) type* var; { void* p__; int rc__ = ensure_dev(c, name, sizeof(type) * (size_t)(count) + 64, &p__); if (rc__) return rc__; var = (type*)p__; }
#define PINBUF(var, type, name, count) type* var; { void* p__; int rc__ = ensure_pin(c, name, sizeof(type) * (size_t)(count) + 64, &p__); if (rc__) return rc__; var = (type*)p__; }

// ---- tail: nodes of one contig for its winning model (device functions; one thread walks a contig's path) ----
struct NodeView {
    int n;
    const int32_t* ndx; const int32_t* stop_val; const uint8_t* type; const int8_t* strand; const uint8_t* edge;
    double* cscore; double* sscore; const double* rscore; const double* uscore; const double* tscore;
    const int32_t* star_ptr; int32_t* traceb; int32_t* tracef; int8_t* ov_mark; const double* score; uint8_t* elim;
};
struct GeneRec { int begin, end, start_ndx, stop_ndx; };

__host__ __device__ inline bool is_stop_n(const NodeView& v, int i) { return v.type[i] == PGA_T_STOP; }

// ref: _connection.h:52-78
__host__ __device__ double igm_same_h(const NodeView& v, int a, int b, double st_wt) {
    const int dist = abs(v.ndx[a] - v.ndx[b]);
    const bool ovl = v.ndx[a] + 2 * v.strand[a] >= v.ndx[b];
    double r = 0.0;
    if (v.ndx[a] + 2 == v.ndx[b] || v.ndx[a] == v.ndx[b] + 1) {
        if (v.strand[a] == 1) { if (v.rscore[b] < 0) r -= v.rscore[b]; if (v.uscore[b] < 0) r -= v.uscore[b]; }
        else                  { if (v.rscore[a] < 0) r -= v.rscore[a]; if (v.uscore[a] < 0) r -= v.uscore[a]; }
    }
    if (dist > 3 * PGA_OPER_DIST) r -= 0.15 * st_wt;
    else if ((dist <= PGA_OPER_DIST && !ovl) || dist * 4 < PGA_OPER_DIST) r += (2.0 - (double)dist / PGA_OPER_DIST) * 0.15 * st_wt;
    return r;
}
__host__ __device__ inline double igm_h(const NodeView& v, int a, int b, double st_wt) {   // ref: _connection.h:81-91
    return v.strand[a] == v.strand[b] ? igm_same_h(v, a, b, st_wt) : -0.15 * st_wt;
}

// The reference walks down from node `from` until it meets a node at position `pos` (the other end of the ORF, which
// always exists): positions are sorted, so the first hit of that walk is the LAST node at or below `from` with
// ndx == pos, which a binary search finds in O(log n) loads instead of one dependent load per node of the gene.
__host__ __device__ inline int walk_down_to(const NodeView& v, int from, int pos) {
    int a = 0, b = from + 1;                      // first index in [0, from] with ndx > pos
    while (a < b) { const int m = (a + b) >> 1; if (v.ndx[m] <= pos) a = m + 1; else b = m; }
    return a - 1;
}

// ref: lib.pyx:1253-1295 (_disentangle_overlaps, _max_forward_pointers)
__host__ __device__ int untangle(NodeView& v, int mx, int32_t* __restrict__ path) {
    // Every test reads its operands first, unconditionally: the loads of one step then leave together instead of one
    // round trip per `&&`.
    for (int p = mx, nx = v.traceb[p]; nx != -1; p = nx, nx = v.traceb[p]) {
        const int sp = v.strand[p], sn = v.strand[nx], ov = v.ov_mark[p], np = v.ndx[p], nn = v.ndx[nx];
        const bool stp = is_stop_n(v, p), stn = is_stop_n(v, nx);
        if ((sp == -1) & stp & (sn == 1) & stn & (ov != -1) & (np > nn)) {
            const int tmp = v.star_ptr[3 * p + ov];
            const int k = walk_down_to(v, tmp, v.stop_val[tmp]);
            v.traceb[p] = tmp; v.traceb[tmp] = k; v.ov_mark[k] = -1; v.traceb[k] = nx;
            nx = tmp;                                  // the walk continues through the nodes just spliced in
        }
    }
    for (int p = mx, nx = v.traceb[p]; nx != -1; p = nx, nx = v.traceb[p]) {
        const int sp = v.strand[p], sn = v.strand[nx], np = v.ndx[p], nn = v.ndx[nx];
        const bool stp = is_stop_n(v, p), stn = is_stop_n(v, nx);
        const bool p_rb = sp == -1 && !stp, p_fs = sp == 1 && stp, p_rs = sp == -1 && stp;
        const bool n_fs = sn == 1 && stn, n_rs = sn == -1 && stn;
        int ins = -1;
        if (p_rb && n_fs) ins = walk_down_to(v, p, v.stop_val[p]);
        if (p_fs && n_fs) ins = v.star_ptr[3 * nx + np % 3];
        if (p_rs && n_rs) ins = v.star_ptr[3 * p + nn % 3];
        if (ins != -1) { v.traceb[p] = ins; v.traceb[ins] = nx; nx = ins; }
    }
    // forward pointers; the nodes of the path are also listed (from mx back to its head) so that the later passes
    // read them with independent loads instead of chasing pointers again
    int cnt = 0;
    int p = mx;
    for (; v.traceb[p] != -1; p = v.traceb[p]) { path[cnt++] = p; v.tracef[v.traceb[p]] = p; }
    path[cnt++] = p;
    return cnt;
}

// Prodigal dprog.c eliminate_bad_genes (call sites ref: lib.pyx:5308, 5369).  path[cnt-1] is the head of the path,
// path[0] its last node (ipath).
__host__ __device__ void eliminate_bad_genes(NodeView& v, const int32_t* __restrict__ path, int cnt, double st_wt) {
    for (int q = cnt - 1; q >= 1; q--) {
        const int p = path[q], f = path[q - 1];
        const int sp = v.strand[p]; const bool stp = is_stop_n(v, p);
        if (sp == 1 && stp) v.sscore[f] += igm_h(v, p, f, st_wt);
        if (sp == -1 && !stp) v.sscore[p] += igm_h(v, p, f, st_wt);
    }
    for (int q = cnt - 1; q >= 1; q--) {
        const int p = path[q], f = path[q - 1];
        const int sp = v.strand[p]; const bool stp = is_stop_n(v, p);
        const double gp = v.cscore[p] + v.sscore[p], gf = v.cscore[f] + v.sscore[f];
        if (sp == 1 && !stp && gp < 0) { v.elim[p] = 1; v.elim[f] = 1; }
        if (sp == -1 && stp && gf < 0) { v.elim[p] = 1; v.elim[f] = 1; }
    }
}

// The same on several host threads (long paths): every node's start score receives at most two terms, in path order
// (as the node after a forward stop, then as a reverse start), so nodes are independent; the elimination pass reads the
// finished scores.
void eliminate_bad_genes_mt(NodeView& v, const int32_t* path, int cnt, double st_wt, WorkerPool& pool, int threads) {
    if (cnt < 4096 || threads <= 1) { eliminate_bad_genes(v, path, cnt, st_wt); return; }
    std::atomic<int> next(cnt - 1);
    const std::function<void()> add = [&]() {
        for (;;) {
            const int hi = next.fetch_sub(512);
            if (hi < 0) break;
            for (int q = hi; q > hi - 512 && q >= 0; q--) {          // node x = path[q]
                const int x = path[q];
                if (q + 1 <= cnt - 1) {                               // x follows p = path[q + 1]
                    const int p = path[q + 1];
                    if (v.strand[p] == 1 && is_stop_n(v, p)) v.sscore[x] += igm_h(v, p, x, st_wt);
                }
                if (q >= 1 && v.strand[x] == -1 && !is_stop_n(v, x)) v.sscore[x] += igm_h(v, x, path[q - 1], st_wt);
            }
        }
    };
    pool.run(add, threads);
    next.store(cnt - 1);
    const std::function<void()> mark = [&]() {
        for (;;) {
            const int hi = next.fetch_sub(512);
            if (hi < 1) break;
            for (int q = hi; q > hi - 512 && q >= 1; q--) {
                const int p = path[q], f = path[q - 1];
                const int sp = v.strand[p]; const bool stp = is_stop_n(v, p);
                const double gp = v.cscore[p] + v.sscore[p], gf = v.cscore[f] + v.sscore[f];
                if (sp == 1 && !stp && gp < 0) { v.elim[p] = 1; v.elim[f] = 1; }
                if (sp == -1 && stp && gf < 0) { v.elim[p] = 1; v.elim[f] = 1; }
            }
        }
    };
    pool.run(mark, threads);
}

// ref: lib.pyx:3231-3270 (Genes._extract); returns the number of genes written to `out`
__host__ __device__ int extract_genes(const NodeView& v, const int32_t* __restrict__ path, int cnt, GeneRec* out) {
    int b = 0, e = 0, s = 0, t = 0, ng = 0;
    for (int q = cnt - 1; q >= 0; q--) {
        const int p = path[q];
        const int el = v.elim[p], sp = v.strand[p], np = v.ndx[p]; const bool stp = is_stop_n(v, p);
        if (el == 1) continue;
        if (sp == 1) {
            if (!stp) { b = np + 1; s = p; }
            else { e = np + 3; t = p; out[ng++] = GeneRec{b, e, s, t}; }
        } else {
            if (!stp) { e = np + 1; s = p; out[ng++] = GeneRec{b, e, s, t}; }
            else { b = np - 1; t = p; }
        }
    }
    return ng;
}

// ref: lib.pyx:3272-3401 (Genes._tweak_final_starts)
// one gene; `prev` / `next` are its neighbours as the reference's in-order loop would see them
// (prev already tweaked, next not yet)
__host__ __device__ void tweak_one(const NodeView& v, const GeneRec* prev, GeneRec& cur, const GeneRec* next, double w, int maxov) {
    const int nn = v.n;
    {
        const int ndx = cur.start_ndx;
        const double sc = v.sscore[ndx] + v.cscore[ndx];
        double ig = 0.0;
        const bool prev_fwd = prev && v.strand[prev->start_ndx] == 1, prev_rev = prev && v.strand[prev->start_ndx] == -1;
        const bool next_fwd = next && v.strand[next->start_ndx] == 1, next_rev = next && v.strand[next->start_ndx] == -1;
        if (v.strand[ndx] == 1 && prev_fwd) ig = igm_same_h(v, prev->stop_ndx, ndx, w);
        if (v.strand[ndx] == 1 && prev_rev) ig = -0.15 * w;
        if (v.strand[ndx] == -1 && next_fwd) ig = -0.15 * w;
        if (v.strand[ndx] == -1 && next_rev) ig = igm_same_h(v, ndx, next->stop_ndx, w);
        int mi[2] = {-1, -1}; double ms[2] = {0, 0}, mg[2] = {0, 0};
        const int sv_ndx = v.stop_val[ndx];
        // the 200 neighbours forty at a time: what rules nearly all of them out (a stop node, or a node of another ORF) is asked for
        // in one go -- one memory round trip per forty candidates instead of one per candidate, on the device one thread walks them
        constexpr int TWB = 40;             // 200 = 5 x 40
        for (int j0 = ndx - 100; j0 < ndx + 100; j0 += TWB) {
          unsigned long long pass = 0;          // start nodes of the gene's own ORF among the forty, then taken in index order
#pragma unroll
          for (int q = 0; q < TWB; q++) {
            const int j = j0 + q;
            const int jj = j < 0 ? 0 : (j >= nn ? nn - 1 : j);
            const bool ok = !(is_stop_n(v, jj) | (v.stop_val[jj] != sv_ndx)) && j >= 0 && j < nn && j != ndx;
            pass |= (unsigned long long)ok << q;
          }
          while (pass) {
            const int q = __builtin_ctzll(pass);
            pass &= pass - 1ull;
            const int j = j0 + q;
            double tg = 0.0;
            if (v.strand[j] == 1 && prev_fwd) {
                if (v.ndx[prev->stop_ndx] - v.ndx[j] > maxov) continue;
                tg = igm_same_h(v, prev->stop_ndx, j, w);
            }
            if (v.strand[j] == 1 && prev_rev) { if (v.ndx[prev->start_ndx] - v.ndx[j] >= 0) continue; tg = -0.15 * w; }
            if (v.strand[j] == -1 && next_fwd) { if (v.ndx[j] - v.ndx[next->start_ndx] >= 0) continue; tg = -0.15 * w; }
            if (v.strand[j] == -1 && next_rev) {
                if (v.ndx[j] - v.ndx[next->stop_ndx] > maxov) continue;
                tg = igm_same_h(v, j, next->stop_ndx, w);
            }
            const double cs = v.cscore[j] + v.sscore[j];
            if (mi[0] == -1) { mi[0] = j; ms[0] = cs; mg[0] = tg; }
            else if (cs + tg > ms[0]) { mi[1] = mi[0]; ms[1] = ms[0]; mg[1] = mg[0]; mi[0] = j; ms[0] = cs; mg[0] = tg; }
            else if (mi[1] == -1 || cs + tg > ms[1]) { mi[1] = j; ms[1] = cs; mg[1] = tg; }
          }
        }
        for (int k = 0; k < 2; k++) {
            const int m = mi[k];
            if (m == -1) continue;
            if (v.tscore[m] < v.tscore[ndx] && ms[k] - v.tscore[m] >= sc - v.tscore[ndx] + w && v.rscore[m] > v.rscore[ndx] &&
                v.uscore[m] > v.uscore[ndx] && v.cscore[m] > v.cscore[ndx] && abs(v.ndx[m] - v.ndx[ndx]) > 15) {
                ms[k] += v.tscore[ndx] - v.tscore[m];
            } else if (abs(v.ndx[m] - v.ndx[ndx]) <= 15 && v.rscore[m] + v.tscore[m] > v.rscore[ndx] + v.tscore[ndx] &&
                       v.edge[ndx] == 0 && v.edge[m] == 0) {
                if (v.cscore[ndx] > v.cscore[m]) ms[k] += v.cscore[ndx] - v.cscore[m];
                if (v.uscore[ndx] > v.uscore[m]) ms[k] += v.uscore[ndx] - v.uscore[m];
                if (ig > mg[k]) ms[k] += ig - mg[k];
            } else ms[k] = -1000.0;
        }
        int pick = -1;
        for (int k = 0; k < 2; k++) {
            if (mi[k] == -1) continue;
            if (pick == -1 && ms[k] + mg[k] > sc + ig) pick = k;
            else if (pick >= 0 && ms[k] + mg[k] > ms[pick] + mg[pick]) pick = k;
        }
        if (pick != -1 && v.strand[mi[pick]] == 1) { cur.start_ndx = mi[pick]; cur.begin = v.ndx[mi[pick]] + 1; }
        else if (pick != -1 && v.strand[mi[pick]] == -1) { cur.start_ndx = mi[pick]; cur.end = v.ndx[mi[pick]] + 1; }
    }
}

// In-order semantics of the reference loop, evaluated in parallel when a contig has many genes: every
// gene is first tweaked against its ORIGINAL neighbours; a gene whose predecessor did change is then
// redone in order against the predecessor's final record (tweaks are rare, so few are redone).
void tweak_final_starts(const NodeView& v, std::vector<GeneRec>& g, double w, int maxov, int inner_threads = 1, WorkerPool* pool = nullptr) {
    const int ng = (int)g.size();
    if (inner_threads <= 1 || ng < 2048 || pool == nullptr) {
        for (int i = 0; i < ng; i++) tweak_one(v, i > 0 ? &g[i - 1] : nullptr, g[i], i < ng - 1 ? &g[i + 1] : nullptr, w, maxov);
        return;
    }
    const std::vector<GeneRec> orig(g);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i0 = next.fetch_add(256);
            if (i0 >= ng) break;
            for (int i = i0; i < std::min(ng, i0 + 256); i++)
                tweak_one(v, i > 0 ? &orig[i - 1] : nullptr, g[i], i < ng - 1 ? &orig[i + 1] : nullptr, w, maxov);
        }
    };
    pool->run(work, inner_threads);
    for (int i = 1; i < ng; i++) {
        if (g[i - 1].start_ndx == orig[i - 1].start_ndx) continue;       // predecessor unchanged: the parallel result stands
        g[i] = orig[i];
        tweak_one(v, &g[i - 1], g[i], i < ng - 1 ? &orig[i + 1] : nullptr, w, maxov);
    }
}

// ---- tail kernels --------------------------------------------------------------------------------------
struct OutArrays;
struct TailDesc {          // one per contig
    int64_t out_off;       // first node of the contig's winning chain in the gathered arrays
    int64_t gene_off;      // first slot of the contig's gene records (capacity n / 2 + 2)
    int32_t n;             // nodes (0: no winning model)
    int32_t mx;            // _find_max_index of the winning pass
    double  st_wt;
    // where the winner's final-pass node fields and its topology sit in the arrays the scorers wrote (k_emit_genes reads the
    // two nodes of a gene from there when the winners were gathered without them)
    int64_t fin_off, topo_off;
    int32_t group, _pad;
    int64_t dp_off;        // the winning pass's chain offset (what the connection scoring saw)
};
// ---- gather kernel: pack the winning chains' node fields contiguously for one D2H per field ----
struct WinDesc {
    int64_t out_off;     // first output node
    int64_t dp_off;      // chain offset of the pass that won (scores as seen by the DP)
    int64_t fin_off;     // chain offset of the fresh re-score (== dp_off when the winner was `first`)
    int64_t topo_off;
    int32_t n;
    int32_t _pad;
};
struct GcPtrs { const float* p[4]; };      // GroupArrays::gc_cont of every translation-table group
struct OutArrays {
    int32_t* ndx; int32_t* stop_val; uint8_t* type; int8_t* strand; float* gc_cont;
    uint8_t* edge_dp; double* cscore_dp; double* sscore_dp; double* rscore_dp; double* uscore_dp; double* tscore_dp;
    int32_t* star_ptr; int32_t* traceb; int8_t* ov_mark; double* score;
    uint8_t* edge; double* cscore; double* sscore; double* rscore; double* uscore; double* tscore; double* mot_score;
    int32_t* mot_ndx; uint8_t* rbs; uint8_t* mot_len; uint8_t* mot_spacer; uint8_t* mot_spacendx;
    // direct != 0 (nobody reads the node arrays after the call): only the fields the tail WRITES -- sscore_dp, traceb, ov_mark -- are
    // gathered; what it only reads stays where the scorers left it: the topology of the contig's group at TailDesc::topo_off, the
    // DP pass's node fields and scores at TailDesc::dp_off
    int32_t direct;
    const int32_t* g_ndx[4]; const int32_t* g_stop_val[4]; const uint8_t* g_type[4]; const int8_t* g_strand[4];
    const uint8_t* c_edge; const double* c_cscore; const double* c_rscore; const double* c_uscore; const double* c_tscore;
    const int32_t* c_star_ptr; const double* d_score;
};

__global__ void __launch_bounds__(256)
k_gather_winners(const WinDesc* __restrict__ wd, int n_win, int64_t out_begin, int64_t total, GroupArrays ga, ChainArrays ca,
                 DpBuffers dp, OutArrays o, const int lean /* 1: only what the device tail walks */) {
    __shared__ int s_w0;
    const int64_t blk0 = out_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t g = blk0 + threadIdx.x;
    int lo = block_search_le([&](const int k) { return wd[k].out_off; }, n_win, blk0, &s_w0);      // one search per workgroup, then a short walk per thread
    if (g >= out_begin + total) return;
    while (lo + 1 < n_win && wd[lo + 1].out_off <= g) lo++;
    const WinDesc w = wd[lo];
    const int i = (int)(g - w.out_off);
    const int64_t t = w.topo_off + i, a = w.dp_off + i, f = w.fin_off + i;
    if (o.direct) { o.sscore_dp[g] = ca.sscore[a]; o.traceb[g] = dp.traceb[a]; o.ov_mark[g] = dp.ov_mark[a]; return; }
    o.ndx[g] = ga.ndx[t]; o.stop_val[g] = ga.stop_val[t]; o.type[g] = ga.type[t]; o.strand[g] = ga.strand[t];
    o.edge_dp[g] = ca.edge[a]; o.cscore_dp[g] = ca.cscore[a]; o.sscore_dp[g] = ca.sscore[a]; o.rscore_dp[g] = ca.rscore[a];
    o.uscore_dp[g] = ca.uscore[a]; o.tscore_dp[g] = ca.tscore[a];
    o.star_ptr[3 * g] = ca.star_ptr[3 * a]; o.star_ptr[3 * g + 1] = ca.star_ptr[3 * a + 1]; o.star_ptr[3 * g + 2] = ca.star_ptr[3 * a + 2];
    o.traceb[g] = dp.traceb[a]; o.ov_mark[g] = dp.ov_mark[a]; o.score[g] = dp.score[a];
    // the final-pass fields: fourteen of the thirty per node.  Only the two nodes of a gene are ever looked at unless the caller
    // wants the node arrays, and k_emit_genes can fetch those from where the scorer left them
    if (lean) return;
    o.gc_cont[g] = ga.gc_cont[t];
    o.edge[g] = ca.edge[f]; o.cscore[g] = ca.cscore[f]; o.sscore[g] = ca.sscore[f]; o.rscore[g] = ca.rscore[f];
    o.uscore[g] = ca.uscore[f]; o.tscore[g] = ca.tscore[f]; o.mot_score[g] = ca.mot_score[f]; o.mot_ndx[g] = ca.mot_ndx[f];
    o.rbs[2 * g] = ca.rbs[2 * f]; o.rbs[2 * g + 1] = ca.rbs[2 * f + 1];
    o.mot_len[g] = ca.mot_len[f]; o.mot_spacer[g] = ca.mot_spacer[f]; o.mot_spacendx[g] = ca.mot_spacendx[f];
}

// Final-pass fields of the two nodes of every gene (start, stop), fetched after the host tail so that
// the full per-node arrays of the re-score need not cross PCIe unless the caller asks for nodes.
struct GeneNodeAttr {
    double cscore, sscore, rscore, uscore, tscore, mot_score;
    float gc_cont; int32_t mot_ndx;
    uint8_t edge, rbs0, rbs1, mot_len, mot_spacer, _pad[3];
};
__global__ void __launch_bounds__(256)
k_gather_gene_nodes(const int64_t* __restrict__ idx, int n, OutArrays o, GeneNodeAttr* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t g = idx[t];
    GeneNodeAttr a;
    a.cscore = o.cscore[g]; a.sscore = o.sscore[g]; a.rscore = o.rscore[g]; a.uscore = o.uscore[g]; a.tscore = o.tscore[g];
    a.mot_score = o.mot_score[g]; a.gc_cont = o.gc_cont[g]; a.mot_ndx = o.mot_ndx[g];
    a.edge = o.edge[g]; a.rbs0 = o.rbs[2 * g]; a.rbs1 = o.rbs[2 * g + 1]; a.mot_len = o.mot_len[g]; a.mot_spacer = o.mot_spacer[g];
    a._pad[0] = a._pad[1] = a._pad[2] = 0;
    out[t] = a;
}

__device__ inline NodeView node_view(const TailDesc& d, const OutArrays& o, int32_t* tracef, uint8_t* elim) {
    const int64_t oo = d.out_off;
    if (o.direct) {
        const int64_t t = d.topo_off, a = d.dp_off; const int g = d.group;
        return NodeView{d.n, o.g_ndx[g] + t, o.g_stop_val[g] + t, o.g_type[g] + t, o.g_strand[g] + t, o.c_edge + a,
                        const_cast<double*>(o.c_cscore) + a, o.sscore_dp + oo, o.c_rscore + a, o.c_uscore + a, o.c_tscore + a,
                        o.c_star_ptr + 3 * a, o.traceb + oo, tracef + oo, o.ov_mark + oo, o.d_score + a, elim + oo};
    }
    return NodeView{d.n, o.ndx + oo, o.stop_val + oo, o.type + oo, o.strand + oo, o.edge_dp + oo,
                    o.cscore_dp + oo, o.sscore_dp + oo, o.rscore_dp + oo, o.uscore_dp + oo, o.tscore_dp + oo,
                    o.star_ptr + 3 * oo, o.traceb + oo, tracef + oo, o.ov_mark + oo, o.score + oo, elim + oo};
}

// One thread per contig: untangle the traceback, eliminate bad genes, list the genes
// (ref: lib.pyx:1253-1311, 5308 / 5369, 3231-3270).  A path is a pointer chase, so contigs are the parallel axis.
__global__ void __launch_bounds__(64)
k_tail_path(const TailDesc* __restrict__ td, int n_contigs, OutArrays o, int32_t* tracef, uint8_t* elim, int32_t* __restrict__ path,
            GeneRec* __restrict__ genes, int32_t* __restrict__ n_genes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_contigs) return;
    const TailDesc d = td[i];
    int ng = 0;
    if (d.n > 0 && d.mx >= 0) {
        NodeView v = node_view(d, o, tracef, elim);
        int32_t* pl = path + d.out_off;                  // at most n entries
        const int cnt = untangle(v, d.mx, pl);
        if (v.traceb[d.mx] != -1) {                      // ipath != -1
            eliminate_bad_genes(v, pl, cnt, d.st_wt);
            ng = extract_genes(v, pl, cnt, genes + d.gene_off);
        }
    }
    n_genes[i] = ng;
}

__device__ inline int contig_of_slot(const TailDesc* __restrict__ td, int n_contigs, int64_t slot) {
    int lo = 0, hi = n_contigs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (td[mid].gene_off <= slot) lo = mid; else hi = mid - 1; }
    return lo;
}

// ref: lib.pyx:3272-3401 (Genes._tweak_final_starts).  The reference loop runs in gene order: gene i sees its
// predecessor already tweaked and its successor untouched.  Here every gene is first tweaked against its ORIGINAL
// neighbours (one thread per gene), then one thread per contig redoes, in order, the genes whose predecessor did
// change (tweaks are rare).
__global__ void __launch_bounds__(256)
k_tail_tweak(const TailDesc* __restrict__ td, int n_contigs, int64_t n_slots, OutArrays o, int32_t* tracef, uint8_t* elim,
             const GeneRec* __restrict__ orig, GeneRec* __restrict__ out, const int32_t* __restrict__ n_genes, int maxov,
             uint8_t* __restrict__ changed, int32_t* __restrict__ n_changed) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const int c = contig_of_slot(td, n_contigs, slot);
    const TailDesc d = td[c];
    const int g = (int)(slot - d.gene_off), ng = n_genes[c];
    if (g >= ng) return;
    const NodeView v = node_view(d, o, tracef, elim);
    const GeneRec* base = orig + d.gene_off;
    GeneRec cur = base[g];
    tweak_one(v, g > 0 ? &base[g - 1] : nullptr, cur, g < ng - 1 ? &base[g + 1] : nullptr, d.st_wt, maxov);
    out[slot] = cur;
    // only a reverse-strand predecessor hands its START to the next gene's tweak (a forward one hands its stop, which
    // never moves): that is the one case the in-order pass has to revisit
    const bool moved = cur.start_ndx != base[g].start_ndx && v.strand[cur.start_ndx] == -1;
    changed[slot] = moved ? 1 : 0;
    if (moved) atomicAdd(&n_changed[c], 1);
}
// The same over the genes themselves instead of over their (mostly empty) slots: gene number q of the batch is gene q - gpre[c]
// of the contig c with gpre[c] <= q < gpre[c + 1] (gpre = running sum of n_genes, k_gene_prefix).  With many short contigs the
// genes are a few per contig: one thread per slot leaves a handful of busy lanes in each of twenty thousand wavefronts, each of
// which takes as long as its slowest lane; packed, the same lanes fill a few hundred.
// the tail's starting state in one launch (it was four or five memsets): tracef = -1, elim = 0 for every gathered node; the per-contig
// counters of changed genes and of genes; the per-chain counters of the many-launch path
__global__ void __launch_bounds__(256)
k_tail_init(int32_t* __restrict__ tracef, uint8_t* __restrict__ elim, const int64_t n_nodes, int32_t* __restrict__ nchanged, int32_t* __restrict__ ngenes /* or nullptr */,
            const int n_contigs, int32_t* __restrict__ seg_cnt /* or nullptr */, const int n_segs) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = t; k <= n_nodes; k += stride) { tracef[k] = -1; elim[k] = 0; }
    for (int64_t k = t; k <= n_contigs; k += stride) { nchanged[k] = 0; if (ngenes != nullptr) ngenes[k] = 0; }
    if (seg_cnt != nullptr) for (int64_t k = t; k < n_segs; k += stride) seg_cnt[k] = 0;
}

__global__ void __launch_bounds__(1024)
k_gene_prefix(const int32_t* __restrict__ n_genes, int n_contigs, int32_t* __restrict__ gpre) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_contigs; base += 1024) {
        const int v = base + t < n_contigs ? n_genes[base + t] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o, 64); if (lane >= o) inc += x; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int run = s_carry + inc - v;
        for (int k = 0; k < w; k++) run += s_w[k];
        if (base + t < n_contigs) gpre[base + t] = run;
        __syncthreads();
        if (t == 1023) s_carry = run + v;
        __syncthreads();
    }
    if (t == 0) gpre[n_contigs] = s_carry;
}
__global__ void __launch_bounds__(256)
k_tail_tweak_packed(const TailDesc* __restrict__ td, int n_contigs, OutArrays o, int32_t* tracef, uint8_t* elim,
                    const GeneRec* __restrict__ orig, GeneRec* __restrict__ out, const int32_t* __restrict__ n_genes, int maxov,
                    uint8_t* __restrict__ changed, int32_t* __restrict__ n_changed, const int32_t* __restrict__ gpre) {
    const int total = gpre[n_contigs];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        int lo = 0, hi = n_contigs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (gpre[mid] <= q) lo = mid; else hi = mid - 1; }
        const int c = lo, g = q - gpre[c], ng = n_genes[c];
        const TailDesc d = td[c];
        const NodeView v = node_view(d, o, tracef, elim);
        const GeneRec* base = orig + d.gene_off;
        GeneRec cur = base[g];
        tweak_one(v, g > 0 ? &base[g - 1] : nullptr, cur, g < ng - 1 ? &base[g + 1] : nullptr, d.st_wt, maxov);
        out[d.gene_off + g] = cur;
        const bool moved = cur.start_ndx != base[g].start_ndx && v.strand[cur.start_ndx] == -1;
        changed[d.gene_off + g] = moved ? 1 : 0;
        if (moved) atomicAdd(&n_changed[c], 1);
    }
}
// tweak_one with the wavefront on ONE gene: the 200 candidate nodes are filtered and priced by the lanes (that is where the
// memory traffic is), the few that pass are then merged one after the other in index order, as the reference's loop
// meets them -- its two-best bookkeeping depends on that order.  Every lane returns the same record.
__device__ void tweak_one_wave(const NodeView& v, const GeneRec* prev, GeneRec& cur, const GeneRec* next, double w, int maxov, const int lane) {
    const int nn = v.n;
    const int ndx = cur.start_ndx;
    const double sc = v.sscore[ndx] + v.cscore[ndx];
    double ig = 0.0;
    const bool prev_fwd = prev && v.strand[prev->start_ndx] == 1, prev_rev = prev && v.strand[prev->start_ndx] == -1;
    const bool next_fwd = next && v.strand[next->start_ndx] == 1, next_rev = next && v.strand[next->start_ndx] == -1;
    if (v.strand[ndx] == 1 && prev_fwd) ig = igm_same_h(v, prev->stop_ndx, ndx, w);
    if (v.strand[ndx] == 1 && prev_rev) ig = -0.15 * w;
    if (v.strand[ndx] == -1 && next_fwd) ig = -0.15 * w;
    if (v.strand[ndx] == -1 && next_rev) ig = igm_same_h(v, ndx, next->stop_ndx, w);
    int mi[2] = {-1, -1}; double ms[2] = {0, 0}, mg[2] = {0, 0};
    const int sv_ndx = v.stop_val[ndx];
    for (int base = ndx - 100; base < ndx + 100; base += 64) {
        const int j = base + lane;
        bool cand = j >= 0 && j < nn && j != ndx && j < ndx + 100;
        double tg = 0.0, cs = 0.0;
        if (cand) cand = !(is_stop_n(v, j) | (v.stop_val[j] != sv_ndx));
        if (cand) {
            const int sj = v.strand[j];
            if (sj == 1 && prev_fwd) {
                if (v.ndx[prev->stop_ndx] - v.ndx[j] > maxov) cand = false;
                else tg = igm_same_h(v, prev->stop_ndx, j, w);
            }
            if (cand && sj == 1 && prev_rev) { if (v.ndx[prev->start_ndx] - v.ndx[j] >= 0) cand = false; else tg = -0.15 * w; }
            if (cand && sj == -1 && next_fwd) { if (v.ndx[j] - v.ndx[next->start_ndx] >= 0) cand = false; else tg = -0.15 * w; }
            if (cand && sj == -1 && next_rev) {
                if (v.ndx[j] - v.ndx[next->stop_ndx] > maxov) cand = false;
                else tg = igm_same_h(v, j, next->stop_ndx, w);
            }
            if (cand) cs = v.cscore[j] + v.sscore[j];
        }
        unsigned long long m = __ballot(cand);
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1ull;
            const int jk = base + k;
            const double csk = __shfl(cs, k, 64), tgk = __shfl(tg, k, 64);
            if (mi[0] == -1) { mi[0] = jk; ms[0] = csk; mg[0] = tgk; }
            else if (csk + tgk > ms[0]) { mi[1] = mi[0]; ms[1] = ms[0]; mg[1] = mg[0]; mi[0] = jk; ms[0] = csk; mg[0] = tgk; }
            else if (mi[1] == -1 || csk + tgk > ms[1]) { mi[1] = jk; ms[1] = csk; mg[1] = tgk; }
        }
    }
    for (int k = 0; k < 2; k++) {
        const int m = mi[k];
        if (m == -1) continue;
        if (v.tscore[m] < v.tscore[ndx] && ms[k] - v.tscore[m] >= sc - v.tscore[ndx] + w && v.rscore[m] > v.rscore[ndx] &&
            v.uscore[m] > v.uscore[ndx] && v.cscore[m] > v.cscore[ndx] && abs(v.ndx[m] - v.ndx[ndx]) > 15) {
            ms[k] += v.tscore[ndx] - v.tscore[m];
        } else if (abs(v.ndx[m] - v.ndx[ndx]) <= 15 && v.rscore[m] + v.tscore[m] > v.rscore[ndx] + v.tscore[ndx] &&
                   v.edge[ndx] == 0 && v.edge[m] == 0) {
            if (v.cscore[ndx] > v.cscore[m]) ms[k] += v.cscore[ndx] - v.cscore[m];
            if (v.uscore[ndx] > v.uscore[m]) ms[k] += v.uscore[ndx] - v.uscore[m];
            if (ig > mg[k]) ms[k] += ig - mg[k];
        } else ms[k] = -1000.0;
    }
    int pick = -1;
    for (int k = 0; k < 2; k++) {
        if (mi[k] == -1) continue;
        if (pick == -1 && ms[k] + mg[k] > sc + ig) pick = k;
        else if (pick >= 0 && ms[k] + mg[k] > ms[pick] + mg[pick]) pick = k;
    }
    if (pick != -1 && v.strand[mi[pick]] == 1) { cur.start_ndx = mi[pick]; cur.begin = v.ndx[mi[pick]] + 1; }
    else if (pick != -1 && v.strand[mi[pick]] == -1) { cur.start_ndx = mi[pick]; cur.end = v.ndx[mi[pick]] + 1; }
}

// k_tail_tweak for long contigs: a genome has few thousand genes, too few for one thread each to hide the latency of a
// 200-candidate scan, so every gene gets a wavefront (tweak_one_wave); blockIdx.y = contig, four genes per workgroup.
__global__ void __launch_bounds__(256)
k_tail_tweak_wave(const TailDesc* __restrict__ td, int n_contigs, OutArrays o, int32_t* tracef, uint8_t* elim,
                  const GeneRec* __restrict__ orig, GeneRec* __restrict__ out, const int32_t* __restrict__ n_genes, int maxov,
                  uint8_t* __restrict__ changed, int32_t* __restrict__ n_changed) {
    const int c = blockIdx.y, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), ng = n_genes[c];
    if (g >= ng) return;
    const TailDesc d = td[c];
    const NodeView v = node_view(d, o, tracef, elim);
    const GeneRec* base = orig + d.gene_off;
    GeneRec cur = base[g];
    GeneRec prev{}, nxt{};
    if (g > 0) prev = base[g - 1];
    if (g < ng - 1) nxt = base[g + 1];
    tweak_one_wave(v, g > 0 ? &prev : nullptr, cur, g < ng - 1 ? &nxt : nullptr, d.st_wt, maxov, lane);
    if (lane != 0) return;
    out[d.gene_off + g] = cur;
    const bool moved = cur.start_ndx != base[g].start_ndx && v.strand[cur.start_ndx] == -1;
    changed[d.gene_off + g] = moved ? 1 : 0;
    if (moved) atomicAdd(&n_changed[c], 1);
}

// The in-order pass over the genes whose predecessor moved (the reference's loop semantics, ref: lib.pyx:3272-3401): one
// workgroup per contig, windows of 64 genes.  A window only depends on the one before it through its first gene's
// predecessor, and only if that predecessor -- the last gene of the window before -- was itself redone: so every wavefront
// first runs its windows assuming it was not (phase 1, all windows side by side), then one wavefront goes over the
// windows in order and redoes, from their parallel-pass state, those whose assumption failed (phase 2, rare).  Within a
// window genes are taken in order; a redone gene's 200 candidates are priced by the lanes (tweak_one_wave).
//   par / ch0   the parallel pass (k_tail_tweak): records against the ORIGINAL neighbours, "start moved on the reverse strand"
//   fin / chf   the final records and flags
#define PGA_FIX_MAXWIN 4096
__global__ void __launch_bounds__(1024)
k_tail_tweak_fixup(const TailDesc* __restrict__ td, int n_contigs, OutArrays o, int32_t* tracef, uint8_t* elim,
                   const GeneRec* __restrict__ orig, const GeneRec* __restrict__ par_all, const int32_t* __restrict__ n_genes, int maxov,
                   const uint8_t* __restrict__ ch0_all, const int32_t* __restrict__ n_changed, GeneRec* __restrict__ fin_all,
                   uint8_t* __restrict__ chf_all) {
    __shared__ uint8_t s_redone[PGA_FIX_MAXWIN];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (c >= n_contigs) return;
    const TailDesc d = td[c];
    const int ng = n_genes[c];
    const GeneRec* ob = orig + d.gene_off; const GeneRec* par = par_all + d.gene_off; const uint8_t* ch0 = ch0_all + d.gene_off;
    GeneRec* fin = fin_all + d.gene_off; uint8_t* chf = chf_all + d.gene_off;
    for (int g = threadIdx.x; g < ng; g += blockDim.x) { fin[g] = par[g]; chf[g] = ch0[g]; }
    if (n_changed[c] == 0 || ng < 2) return;
    __threadfence_block();
    __syncthreads();
    const NodeView v = node_view(d, o, tracef, elim);
    const int nwin = (ng - 1 + 63) >> 6;                  // genes 1 .. ng-1
    // one window, in gene order; `assume`: its first gene's predecessor is as the parallel pass left it.  Returns whether the
    // window's last gene was redone (then the next window may not assume that).
    auto run_window = [&](const int w, const bool assume) -> bool {
        const int g0 = 1 + (w << 6), g1 = min(ng, g0 + 64);
        const int g = g0 + lane;
        if (!assume) {
            if (g < g1) { fin[g] = par[g]; chf[g] = ch0[g]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const int pf = g < g1 ? ((lane == 0 && !assume) ? chf[g - 1] : ch0[g - 1]) : 0;
        unsigned long long mask = __ballot(pf != 0);
        bool last = false;
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int gg = g0 + k;
            GeneRec cur = ob[gg];
            const GeneRec prev = (k == 0 && assume) ? par[gg - 1] : fin[gg - 1];
            GeneRec nxt{};
            if (gg < ng - 1) nxt = ob[gg + 1];
            tweak_one_wave(v, &prev, cur, gg < ng - 1 ? &nxt : nullptr, d.st_wt, maxov, lane);
            const int nf = cur.start_ndx != ob[gg].start_ndx && v.strand[cur.start_ndx] == -1;
            if (lane == 0) { fin[gg] = cur; chf[gg] = (uint8_t)nf; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // fin[gg] is read back by all lanes if gg + 1 is redone
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // the next gene's predecessor flag is this gene's new flag
            if (gg + 1 < g1) { if (nf) mask |= 1ull << (k + 1); else mask &= ~(1ull << (k + 1)); }
            if (gg == g1 - 1) last = true;
        }
        return last;
    };
    if (nwin <= PGA_FIX_MAXWIN) {
        for (int w = wave; w < nwin; w += nwaves) { const bool r = run_window(w, true); if (lane == 0) s_redone[w] = r; }
        __threadfence_block();
        __syncthreads();
        if (wave == 0)
            for (int w = 1; w < nwin; w++)
                if (s_redone[w - 1]) { const bool r = run_window(w, false); if (lane == 0) s_redone[w] = r; __builtin_amdgcn_wave_barrier(); }
    } else if (wave == 0) {
        for (int w = 0; w < nwin; w++) run_window(w, false);       // a contig of more than 262 k genes: plainly in order
    }
}

// The public gene records, packed in (contig, gene) order (ref: lib.pyx:2644-2830 for what Gene reads).
__global__ void __launch_bounds__(256)
k_emit_genes(const TailDesc* __restrict__ td, int n_contigs, int64_t n_slots, OutArrays o, const GeneRec* __restrict__ fin,
             const int32_t* __restrict__ n_genes, const int64_t* __restrict__ gene_begin, int single, pga_gene* __restrict__ out,
             const int lean, ChainArrays ca, GcPtrs gcs) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const int c = contig_of_slot(td, n_contigs, slot);
    const TailDesc d = td[c];
    const int g = (int)(slot - d.gene_off);
    if (g >= n_genes[c]) return;
    const GeneRec gr = fin[slot];
    const int64_t sn = d.out_off + gr.start_ndx, en = d.out_off + gr.stop_ndx;
    pga_gene G;
    memset(&G, 0, sizeof G);
    G.contig = c; G.begin = gr.begin; G.end = gr.end; G.start_ndx = gr.start_ndx; G.stop_ndx = gr.stop_ndx;
    if (o.direct) {
        // (lean as well) the DP pass's read-only fields were not gathered either
        const int64_t ts = d.topo_off + gr.start_ndx, as = d.dp_off + gr.start_ndx, ae = d.dp_off + gr.stop_ndx;
        const int64_t fs = d.fin_off + gr.start_ndx, fe = d.fin_off + gr.stop_ndx;
        G.strand = o.g_strand[d.group][ts];
        const uint8_t se = single ? o.c_edge[as] : ca.edge[fs], ee = single ? o.c_edge[ae] : ca.edge[fe];
        G.partial_begin = G.strand == 1 ? se : ee; G.partial_end = G.strand == 1 ? ee : se;
        G.start_type = se ? 3 : o.g_type[d.group][ts];
        G.rbs[0] = ca.rbs[2 * fs]; G.rbs[1] = ca.rbs[2 * fs + 1];
        G.mot_len = ca.mot_len[fs]; G.mot_spacer = ca.mot_spacer[fs]; G.mot_ndx = ca.mot_ndx[fs]; G.mot_score = ca.mot_score[fs];
        G.gc_cont = gcs.p[d.group][ts];
        G.cscore = single ? o.c_cscore[as] : ca.cscore[fs]; G.sscore = single ? o.sscore_dp[sn] : ca.sscore[fs];
        G.rscore = single ? o.c_rscore[as] : ca.rscore[fs]; G.uscore = single ? o.c_uscore[as] : ca.uscore[fs];
        G.tscore = single ? o.c_tscore[as] : ca.tscore[fs];
        out[gene_begin[c] + g] = G;
        return;
    }
    G.strand = o.strand[sn];
    // single mode keeps the nodes of the DP pass (ref: lib.pyx:5296-5311); meta mode re-scores (5380-5394)
    if (lean) {
        // the winners were gathered without their final-pass fields: the start node's sit where the scorer wrote them
        const int64_t fs = d.fin_off + gr.start_ndx, fe = d.fin_off + gr.stop_ndx;
        const uint8_t se = single ? o.edge_dp[sn] : ca.edge[fs], ee = single ? o.edge_dp[en] : ca.edge[fe];
        G.partial_begin = G.strand == 1 ? se : ee; G.partial_end = G.strand == 1 ? ee : se;
        G.start_type = se ? 3 : o.type[sn];
        G.rbs[0] = ca.rbs[2 * fs]; G.rbs[1] = ca.rbs[2 * fs + 1];
        G.mot_len = ca.mot_len[fs]; G.mot_spacer = ca.mot_spacer[fs]; G.mot_ndx = ca.mot_ndx[fs]; G.mot_score = ca.mot_score[fs];
        G.gc_cont = gcs.p[d.group][d.topo_off + gr.start_ndx];
        G.cscore = single ? o.cscore_dp[sn] : ca.cscore[fs]; G.sscore = single ? o.sscore_dp[sn] : ca.sscore[fs];
        G.rscore = single ? o.rscore_dp[sn] : ca.rscore[fs]; G.uscore = single ? o.uscore_dp[sn] : ca.uscore[fs];
        G.tscore = single ? o.tscore_dp[sn] : ca.tscore[fs];
        out[gene_begin[c] + g] = G;
        return;
    }
    const uint8_t se = single ? o.edge_dp[sn] : o.edge[sn], ee = single ? o.edge_dp[en] : o.edge[en];
    G.partial_begin = G.strand == 1 ? se : ee; G.partial_end = G.strand == 1 ? ee : se;
    G.start_type = se ? 3 : o.type[sn];
    G.rbs[0] = o.rbs[2 * sn]; G.rbs[1] = o.rbs[2 * sn + 1];
    G.mot_len = o.mot_len[sn]; G.mot_spacer = o.mot_spacer[sn]; G.mot_ndx = o.mot_ndx[sn]; G.mot_score = o.mot_score[sn];
    G.gc_cont = o.gc_cont[sn];
    G.cscore = single ? o.cscore_dp[sn] : o.cscore[sn]; G.sscore = single ? o.sscore_dp[sn] : o.sscore[sn];
    G.rscore = single ? o.rscore_dp[sn] : o.rscore[sn]; G.uscore = single ? o.uscore_dp[sn] : o.uscore[sn];
    G.tscore = single ? o.tscore_dp[sn] : o.tscore[sn];
    out[gene_begin[c] + g] = G;
}

#include "tail.inl"

struct ResultOwner {
    pga_result pub;
    std::vector<pga_contig_result> contigs;
    std::vector<pga_gene> genes;
    // Round 6: the gene records of a large result land in PINNED memory that the result owns (from the process-wide cache of pinned
    // blocks, back to it when the result is freed): the read-back of 16 MB per 125 Mbp call is one DMA to where the records stay, where
    // a std::vector meant a value-initialised allocation (page faults) and a copy staged through the runtime's own pinned buffers,
    // 0.8 .. 4 ms of a 9 ms call.
    pga_gene* pin_genes = nullptr; size_t pin_cap = 0, pin_n = 0;
    std::vector<pga_nodes> nodes;
    std::vector<int32_t> mask_off, masks;
    std::vector<void*> blocks;
    ~ResultOwner() {
        for (void* b : blocks) free(b);
        if (pin_genes != nullptr) cache_put_result(pin_genes, pin_cap);
    }
    // room for n gene records; pinned when the result is large enough for it to matter
    pga_gene* gene_records(const size_t n) {
        const size_t bytes = n * sizeof(pga_gene);
        if (bytes < ((size_t)256 << 10) || getenv("PGA_PAGEABLE_RESULTS")) { genes.resize(n); return genes.data(); }
        size_t cap = 0;
        void* p = cache_take(1, -1, bytes, &cap, 8);         // (a block up to eight times the size will do: results vary, pinned allocations cost milliseconds)
        if (p == nullptr) {
            cap = (bytes + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); genes.resize(n); return genes.data(); }
        }
        pin_genes = (pga_gene*)p; pin_cap = cap; pin_n = n;
        return pin_genes;
    }
};

// one zeroed block per contig for the SoA node arrays of a result
int alloc_nodes(ResultOwner* R, pga_nodes& N, const int n) {
    memset(&N, 0, sizeof N);
    N.n = n;
    const size_t bytes = (size_t)n * (4 * 8 + 10 + 4 + 8 * 7) + 8 * 32;   // 8 int32, 10 bytes, 1 float, 7 doubles per node + padding
    char* blk = (char*)calloc(1, bytes);
    if (!blk) return PGA_ENOMEM;
    R->blocks.push_back(blk);
    char* q = blk;
    auto take = [&](size_t b) { char* r = q; q += (b + 7) & ~(size_t)7; return (void*)r; };
    N.ndx = (int32_t*)take(4 * n); N.stop_val = (int32_t*)take(4 * n); N.traceb = (int32_t*)take(4 * n); N.tracef = (int32_t*)take(4 * n);
    N.star_ptr = (int32_t*)take(12 * n); N.mot_ndx = (int32_t*)take(4 * n);
    N.type = (uint8_t*)take(n); N.edge = (uint8_t*)take(n); N.elim = (uint8_t*)take(n); N.rbs = (uint8_t*)take(2 * n);
    N.strand = (int8_t*)take(n); N.ov_mark = (int8_t*)take(n); N.mot_len = (uint8_t*)take(n); N.mot_spacer = (uint8_t*)take(n); N.mot_spacendx = (uint8_t*)take(n);
    N.gc_cont = (float*)take(4 * n);
    N.cscore = (double*)take(8 * n); N.sscore = (double*)take(8 * n); N.rscore = (double*)take(8 * n); N.uscore = (double*)take(8 * n);
    N.tscore = (double*)take(8 * n); N.score = (double*)take(8 * n); N.mot_score = (double*)take(8 * n);
    return PGA_OK;
}

void fill_model_score_const(ModelScoreConst* m, const pga_training* t) {   // ref: lib.pyx:2136-2147, 2209-2210
    double no_stop;
    if (t->trans_table != 11) {
        no_stop = ((1 - t->gc) * (1 - t->gc) * t->gc) / 8.0;
        no_stop += ((1 - t->gc) * (1 - t->gc) * (1 - t->gc)) / 8.0;
    } else {
        no_stop = ((1 - t->gc) * (1 - t->gc) * t->gc) / 4.0;
        no_stop += ((1 - t->gc) * (1 - t->gc) * (1 - t->gc)) / 8.0;
    }
    no_stop = 1 - no_stop;
    m->lfac_max = log((1 - pow(no_stop, 1000.0)) / pow(no_stop, 1000.0));
    m->lfac_min = log((1 - pow(no_stop, 80)) / pow(no_stop, 80));
    for (int n = 0; n <= 1000; n++) {
        const double tmp = pow(no_stop, (double)n);
        m->lfac_tab[n] = log((1 - tmp) / tmp) - m->lfac_min;
    }
}

}  // namespace

void pga_finder_release(pga_ctx* c) {
    if (!c->finder) return;
    if (getenv("PGA_PRINT_BUFFERS")) {
        // diagnostics: what this context held on the device, largest first
        std::vector<std::pair<size_t, std::string>> v; size_t tot = 0;
        for (auto& kv : c->finder->dev) if (kv.second.p) { v.push_back({kv.second.cap, kv.first}); tot += kv.second.cap; }
        std::sort(v.begin(), v.end(), [](const std::pair<size_t, std::string>& a, const std::pair<size_t, std::string>& b) { return a.first > b.first; });
        fprintf(stderr, "[pga] context buffers: %.2f GB in %zu allocations\n", tot / 1e9, v.size());
        for (size_t k = 0; k < v.size() && k < 48; k++) fprintf(stderr, "[pga]   %-22s %9.1f MB\n", v[k].second.c_str(), v[k].first / 1e6);
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);        // nothing of this context still reads what goes to the cache
    for (auto& kv : c->finder->dev) if (kv.second.p && !cache_put(0, c->device, kv.second.p, kv.second.cap)) hipFree(kv.second.p);
    for (auto& kv : c->finder->pin) if (kv.second.p && !cache_put(1, c->device, kv.second.p, kv.second.cap)) hipHostFree(kv.second.p);
    if (c->finder->d_msc) hipFree(c->finder->d_msc);
    if (c->finder->d_model_gc) hipFree(c->finder->d_model_gc);
    if (c->finder->d_model_grp) hipFree(c->finder->d_model_grp);
    if (c->finder->d_gil) hipFree(c->finder->d_gil);
    if (c->finder->d_model_rank) hipFree(c->finder->d_model_rank);
    if (c->finder->d_sd_lut) hipFree(c->finder->d_sd_lut);
    if (c->finder->spare_p) hipFree(c->finder->spare_p);
    if (c->finder->up_stream) { (void)hipStreamSynchronize(c->finder->up_stream); (void)hipStreamDestroy(c->finder->up_stream); }
    if (c->finder->up_pin && !cache_put(1, c->device, c->finder->up_pin, c->finder->up_pin_cap)) hipHostFree(c->finder->up_pin);
    if (c->finder->e_start) hipEventDestroy(c->finder->e_start);
    if (c->finder->e_stop) hipEventDestroy(c->finder->e_stop);
    for (int i = 0; i < 4; i++) { if (c->finder->e_dp0[i]) hipEventDestroy(c->finder->e_dp0[i]); if (c->finder->e_dp1[i]) hipEventDestroy(c->finder->e_dp1[i]); }
    for (int i = 0; i < 20; i++) if (c->finder->e_aux[i]) hipEventDestroy(c->finder->e_aux[i]);
    delete c->finder;
    c->finder = nullptr;
}

int pga_finder_models_changed(pga_ctx* c) {
    if (!c->finder) {
        c->finder = new (std::nothrow) FinderState();
        if (!c->finder) return PGA_ENOMEM;
        HT(c, hipEventCreate(&c->finder->e_start)); HT(c, hipEventCreate(&c->finder->e_stop));
        for (int i = 0; i < 4; i++) { HT(c, hipEventCreate(&c->finder->e_dp0[i])); HT(c, hipEventCreate(&c->finder->e_dp1[i])); }
        for (int i = 0; i < 20; i++) HT(c, hipEventCreate(&c->finder->e_aux[i]));
        HT(c, hipMalloc((void**)&c->finder->d_sd_lut, sizeof(unsigned) * 1920));
        pga_launch_sd_lut(c->finder->d_sd_lut, c->stream);
        HT(c, hipStreamSynchronize(c->stream));
    }
    FinderState* f = c->finder;
    f->model_group.clear(); f->group_tt.clear();
    if (f->d_msc) { hipFree(f->d_msc); f->d_msc = nullptr; }
    if (f->d_model_gc) { hipFree(f->d_model_gc); f->d_model_gc = nullptr; }
    if (f->d_model_grp) { hipFree(f->d_model_grp); f->d_model_grp = nullptr; }
    if (f->d_gil) { hipFree(f->d_gil); f->d_gil = nullptr; }
    if (f->d_model_rank) { hipFree(f->d_model_rank); f->d_model_rank = nullptr; }
    f->gil_off.clear(); f->gil_stride.clear();
    const int nm = c->n_models;
    if (nm == 0) return PGA_OK;
    std::vector<ModelScoreConst> msc(nm);
    for (int m = 0; m < nm; m++) {
        const int tt = c->models[m].trans_table;
        int g = -1;
        for (size_t k = 0; k < f->group_tt.size(); k++) if (f->group_tt[k] == tt) g = (int)k;
        if (g < 0) { g = (int)f->group_tt.size(); f->group_tt.push_back(tt); }
        f->model_group.push_back(g);
        fill_model_score_const(&msc[m], &c->models[m]);
    }
    if (f->group_tt.size() > 4) { c->err = "pga_set_models: more than 4 distinct translation tables"; return PGA_EINVAL; }
    HT(c, hipMalloc((void**)&f->d_msc, sizeof(ModelScoreConst) * nm));
    HT(c, hipMemcpy(f->d_msc, msc.data(), sizeof(ModelScoreConst) * nm, hipMemcpyHostToDevice));
    std::vector<double> mgc(nm);
    for (int m = 0; m < nm; m++) mgc[m] = c->models[m].gc;
    HT(c, hipMalloc((void**)&f->d_model_gc, sizeof(double) * nm));
    HT(c, hipMalloc((void**)&f->d_model_grp, sizeof(int32_t) * nm));
    HT(c, hipMemcpy(f->d_model_gc, mgc.data(), sizeof(double) * nm, hipMemcpyHostToDevice));
    HT(c, hipMemcpy(f->d_model_grp, f->model_group.data(), sizeof(int32_t) * nm, hipMemcpyHostToDevice));
    {
        const int ng = (int)f->group_tt.size();
        std::vector<int32_t> rank(nm, 0), count(ng, 0);
        for (int m = 0; m < nm; m++) rank[m] = count[f->model_group[m]]++;
        size_t total_il = 0;
        for (int g = 0; g < ng; g++) { f->gil_stride.push_back((count[g] + 1) & ~1); f->gil_off.push_back(total_il); total_il += (size_t)4096 * f->gil_stride[g]; }
        std::vector<double> il(total_il + 8, 0.0);               // a pair load at the last column of the last row stays inside
        for (int m = 0; m < nm; m++) {
            const int g = f->model_group[m];
            for (int h = 0; h < 4096; h++) il[f->gil_off[g] + (size_t)h * f->gil_stride[g] + rank[m]] = c->models[m].gene_dc[h];
        }
        HT(c, hipMalloc((void**)&f->d_gil, sizeof(double) * il.size()));
        HT(c, hipMalloc((void**)&f->d_model_rank, sizeof(int32_t) * nm));
        HT(c, hipMemcpy(f->d_gil, il.data(), sizeof(double) * il.size(), hipMemcpyHostToDevice));
        HT(c, hipMemcpy(f->d_model_rank, rank.data(), sizeof(int32_t) * nm, hipMemcpyHostToDevice));
        f->model_rank = rank;
    }
    return PGA_OK;
}

struct pga_batch {
    pga_ctx* ctx;
    int32_t n;
    int64_t total;
    std::vector<ContigDesc> ct;   // n + 1 entries
    char* d_seq;                  // packed ASCII, resident in HBM
    size_t d_seq_cap = 0;         // bytes of that allocation (it goes back to the context's spare slot)
    TileDesc* d_tiles;            // extraction tiles of every contig (behind the letters, in their allocation)
    int32_t* d_tile0;             // first tile of every contig, n + 1 entries (behind d_tiles)
    int32_t n_tiles;
};

// Tiles of a batch: every position of a contig of at least three bases lies in one tile.  Host vectors for one upload.
static void batch_tiles(const pga_batch* b, std::vector<TileDesc>& tiles, std::vector<int32_t>& tile0) {
    const int TS = pga_extract_tile_size();
    tile0.resize((size_t)b->n + 1);
    for (int i = 0; i < b->n; i++) {
        tile0[i] = (int32_t)tiles.size();
        const int64_t L = b->ct[i].len;
        if (L >= 3) for (int64_t s0 = 0; s0 < L; s0 += TS) tiles.push_back(TileDesc{i, (int32_t)s0});
    }
    tile0[b->n] = (int32_t)tiles.size();
}
// the tile list and the per-contig first-tile table live behind the letters, in the same device allocation (`at`, 256-byte aligned)
static size_t batch_tiles_bytes(const std::vector<TileDesc>& tiles, const std::vector<int32_t>& tile0) {
    return sizeof(TileDesc) * tiles.size() + sizeof(int32_t) * tile0.size() + 64 + 256;
}
static hipError_t batch_upload_tiles(pga_batch* b, char* at, const std::vector<TileDesc>& tiles, const std::vector<int32_t>& tile0, hipStream_t st) {
    b->n_tiles = (int32_t)tiles.size();
    const size_t tb = sizeof(TileDesc) * tiles.size(), zb = sizeof(int32_t) * tile0.size();
    b->d_tiles = (TileDesc*)(((uintptr_t)at + 255) & ~(uintptr_t)255);
    b->d_tile0 = (int32_t*)((char*)b->d_tiles + tb);
    hipError_t e = hipSuccess;
    if (tb) e = hipMemcpyAsync(b->d_tiles, tiles.data(), tb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b->d_tile0, tile0.data(), zb, hipMemcpyHostToDevice, st);
    return e;
}

extern "C" int pga_dp_start_order(int32_t n_chains, const int32_t* nodes_per_chain, int32_t* order) {
    if (n_chains < 0 || (n_chains > 0 && (!nodes_per_chain || !order))) return PGA_EINVAL;
    // counting sort on walk batches (nodes / 64), most first; stable, so equals keep the launch order
    int maxb = 0;
    for (int k = 0; k < n_chains; k++) maxb = std::max(maxb, std::max(nodes_per_chain[k], 0) >> 6);
    std::vector<int32_t> first((size_t)maxb + 2, 0);
    for (int k = 0; k < n_chains; k++) first[(size_t)(maxb - (std::max(nodes_per_chain[k], 0) >> 6)) + 1]++;
    for (int b = 0; b <= maxb; b++) first[(size_t)b + 1] += first[(size_t)b];
    for (int k = 0; k < n_chains; k++) order[(size_t)first[(size_t)(maxb - (std::max(nodes_per_chain[k], 0) >> 6))]++] = k;
    return PGA_OK;
}

// The start order made XCD-aware: workgroup b runs on XCD b % 8, and chains that share a key (a contig under one translation table:
// the same topology arrays) should meet in one XCD's L2.  `order` is a start order (pga_dp_start_order); every key goes to the XCD
// with the fewest nodes so far, in that order; out[8 k + x] = k-th chain of XCD x, -1 where a queue has ended.  Returns the number
// of entries written (a multiple of 8, at most 8 * n_chains), or a negative error code.
extern "C" int64_t pga_dp_xcd_order(int32_t n_chains, const int32_t* order, const int32_t* nodes_per_chain, const int32_t* key_of_chain, int32_t n_keys,
                                    int32_t* out, int64_t out_cap) {
    if (n_chains < 0 || n_keys < 0 || (n_chains > 0 && (!order || !nodes_per_chain || !key_of_chain || !out))) return -(int64_t)PGA_EINVAL;
    std::vector<int32_t> queue[8];
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int8_t> xcd_of((size_t)n_keys, (int8_t)-1);
    for (int p = 0; p < n_chains; p++) {
        const int c = order[p];
        if (c < 0 || c >= n_chains || key_of_chain[c] < 0 || key_of_chain[c] >= n_keys) return -(int64_t)PGA_EINVAL;
        int8_t& x = xcd_of[(size_t)key_of_chain[c]];
        if (x < 0) { int best = 0; for (int q = 1; q < 8; q++) if (load[q] < load[best]) best = q; x = (int8_t)best; }
        queue[x].push_back(c);
        load[x] += std::max(nodes_per_chain[c], 0);
    }
    size_t longest = 0;
    for (int q = 0; q < 8; q++) longest = std::max(longest, queue[q].size());
    if ((int64_t)(longest * 8) > out_cap) return -(int64_t)PGA_EINVAL;
    for (size_t k = 0; k < longest * 8; k++) out[k] = -1;
    for (int q = 0; q < 8; q++) for (size_t k = 0; k < queue[q].size(); k++) out[k * 8 + (size_t)q] = queue[q][k];
    return (int64_t)(longest * 8);
}

extern "C" int pga_cs_task_summary(int32_t n_contigs, const int32_t* nodes_per_contig, const int32_t* first_column,
                                   const int32_t* models_per_contig, int32_t task_nodes, int64_t out[5]) {
    if (n_contigs < 0 || !out || (n_contigs > 0 && (!nodes_per_contig || !first_column || !models_per_contig)) || task_nodes < 1) return PGA_EINVAL;
    // the inputs in the shape pga_cs_tasks plans from: one chain per (contig, model), columns = model ranks
    std::vector<int2> cc((size_t)n_contigs); std::vector<ChainDesc> chains; std::vector<int32_t> cbase((size_t)n_contigs + 1, 0), rank;
    int max_col = 0;
    for (int i = 0; i < n_contigs; i++) max_col = std::max(max_col, first_column[i] + std::max(models_per_contig[i], 0));
    rank.resize((size_t)max_col + 1);
    for (int m = 0; m <= max_col; m++) rank[(size_t)m] = m;
    for (int i = 0; i < n_contigs; i++) {
        cbase[(size_t)i + 1] = cbase[(size_t)i] + std::max(nodes_per_contig[i], 0);
        cc[(size_t)i] = make_int2((int)chains.size(), std::max(models_per_contig[i], 0));
        for (int m = 0; m < models_per_contig[i]; m++) { ChainDesc ch{}; ch.model = first_column[i] + m; ch.contig = i; ch.n = nodes_per_contig[i]; chains.push_back(ch); }
    }
    std::vector<int32_t> tasks, entries;
    if (!pga_cs_tasks(cc.data(), n_contigs, chains.data(), cbase.data(), rank.data(), task_nodes, tasks, entries)) return PGA_EINVAL;
    const size_t nt = tasks.size() / 4, ne = entries.size() / 4;
    int64_t biggest = 0, total = 0; bool falling = true;
    for (size_t t = 0; t < nt; t++) {
        int64_t nodes = 0;
        for (int e = 0; e < tasks[4 * t + 2]; e++) nodes += entries[4 * (size_t)(tasks[4 * t + 1] + e) + 3];
        biggest = std::max(biggest, nodes); total += nodes;
        if (t > 0 && tasks[4 * t] > tasks[4 * (t - 1)]) falling = false;
    }
    out[0] = (int64_t)nt; out[1] = (int64_t)ne; out[2] = biggest; out[3] = total; out[4] = falling ? 1 : 0;
    return PGA_OK;
}

// The letters of a batch live in one device allocation.  hipMalloc / hipFree per call cost a device-wide synchronisation each
// (hipFree waits for every stream of the device, i.e. for the other contexts' kernels), so a context keeps the allocation of the
// last batch it freed and hands it to the next one that fits.
static hipError_t batch_take_dev(pga_ctx* c, size_t bytes, char** out, size_t* cap) {
    FinderState* f = c->finder;
    {
        std::lock_guard<std::mutex> g(f->spare_mu);
        if (f->spare_p && f->spare_cap >= bytes && f->spare_cap <= 2 * bytes + (64u << 20)) {
            *out = (char*)f->spare_p; *cap = f->spare_cap; f->spare_p = nullptr; f->spare_cap = 0;
            return hipSuccess;
        }
    }
    const size_t want = bytes + bytes / 8 + 256;
    *cap = want;
    return hipMalloc((void**)out, want);
}
static void batch_give_dev(pga_ctx* c, char* p, size_t cap) {
    if (!p) return;
    FinderState* f = c->finder;
    void* drop = p;
    if (f) {
        std::lock_guard<std::mutex> g(f->spare_mu);
        if (!f->spare_p || f->spare_cap < cap) { drop = f->spare_p; f->spare_p = p; f->spare_cap = cap; }
    }
    if (drop) hipFree(drop);
}

// the upload's own stream and pinned staging area (FinderState::up_*; the caller holds up_mu)
static int upload_resources(pga_ctx* c, size_t pin_bytes, hipStream_t* st, char** pin) {
    FinderState* f = c->finder;
    if (!f->up_stream) HT(c, hipStreamCreateWithFlags(&f->up_stream, hipStreamNonBlocking));
    *st = f->up_stream;
    if (pin_bytes > 0 && (f->up_pin_cap < pin_bytes || !f->up_pin)) {
        if (f->up_pin) { hipHostFree(f->up_pin); f->up_pin = nullptr; f->up_pin_cap = 0; }
        size_t want = pin_bytes + pin_bytes / 4 + 256;
        f->up_pin = cache_take(1, c->device, want, &want);
        if (!f->up_pin) HT(c, alloc_or_evict(1, [&] { return hipHostMalloc(&f->up_pin, want, hipHostMallocDefault); }));
        f->up_pin_cap = want;
    }
    if (pin) *pin = (char*)f->up_pin;
    return PGA_OK;
}

extern "C" int pga_batch_create(pga_ctx* c, int32_t n_contigs, const char* const* seqs, const int64_t* lens, pga_batch** out) {
    if (out) *out = nullptr;
    if (!c || !out || n_contigs < 0 || (n_contigs > 0 && (!seqs || !lens))) { if (c) c->err = "pga_batch_create: bad arguments"; return PGA_EINVAL; }
    if (!c->finder) { int rc = pga_finder_models_changed(c); if (rc) return rc; }
    HT(c, hipSetDevice(c->device));
    pga_batch* b = new (std::nothrow) pga_batch();
    if (!b) return PGA_ENOMEM;
    b->ctx = c; b->n = n_contigs; b->d_seq = nullptr; b->d_tiles = nullptr; b->d_tile0 = nullptr; b->n_tiles = 0; b->ct.resize((size_t)n_contigs + 1);
    int64_t total = 0;
    for (int i = 0; i < n_contigs; i++) {
        if (lens[i] < 0 || lens[i] > 0x7fff0000LL || (lens[i] > 0 && !seqs[i])) { delete b; c->err = "pga_batch_create: bad contig length"; return PGA_EINVAL; }
        b->ct[i].base = total; b->ct[i].len = (int32_t)lens[i]; b->ct[i]._pad = 0;
        total += lens[i];
    }
    b->ct[n_contigs].base = total; b->ct[n_contigs].len = 0; b->ct[n_contigs]._pad = 0;
    b->total = total;
    if (total >= 0x7fffffffLL) { delete b; c->err = "pga_batch_create: batch larger than 2^31 bases; split it"; return PGA_EINVAL; }
    if (total > 0) {
        std::lock_guard<std::mutex> up(c->finder->up_mu);          // one upload at a time per context; a call of the context may run beside it
        hipStream_t st = nullptr; char* h_seq = nullptr;
        { const int rc = upload_resources(c, (size_t)total + 16, &st, &h_seq); if (rc) { delete b; return rc; } }
        std::vector<TileDesc> tiles; std::vector<int32_t> tile0;
        batch_tiles(b, tiles, tile0);
        if (batch_take_dev(c, (size_t)total + 16 + batch_tiles_bytes(tiles, tile0), &b->d_seq, &b->d_seq_cap) != hipSuccess) { delete b; c->err = "pga_batch_create: hipMalloc failed"; return PGA_ENOMEM; }
        // Packing is a host memcpy per contig and the upload a DMA from the pinned copy: the contigs are cut into slices of about
        // 8 MB, a few host threads pack slices, and each slice goes on its way to the device as soon as it is packed -- the DMA of
        // slice k runs under the packing of the slices after it (it was: one thread packing everything, then one DMA, 12 + 3 ms
        // per 125 Mbp).
        std::vector<int> cut{0};
        { int64_t acc = 0; for (int i = 0; i < n_contigs; i++) { acc += lens[i]; if (acc >= (8 << 20)) { cut.push_back(i + 1); acc = 0; } } }
        if (cut.back() != n_contigs) cut.push_back(n_contigs);
        const int n_slices = (int)cut.size() - 1;
        std::atomic<int> next{0};
        std::atomic<int> first_err{(int)hipSuccess};
        const int dev = c->device;
        char* d_seq = b->d_seq;
        const ContigDesc* ct = b->ct.data();
        std::mutex issue_mu;
        auto work = [&]() {
            (void)hipSetDevice(dev);
            for (;;) {
                const int k = next.fetch_add(1);
                if (k >= n_slices) return;
                for (int i = cut[k]; i < cut[k + 1]; i++) if (lens[i] > 0) memcpy(h_seq + ct[i].base, seqs[i], (size_t)lens[i]);
                const int64_t lo = ct[cut[k]].base, hi = ct[cut[k + 1]].base;
                if (hi > lo) {
                    std::lock_guard<std::mutex> g(issue_mu);
                    const hipError_t e = hipMemcpyAsync(d_seq + lo, h_seq + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, st);
                    if (e != hipSuccess) { int z = (int)hipSuccess; first_err.compare_exchange_strong(z, (int)e); }
                }
            }
        };
        const int threads = n_slices >= 2 ? std::min(n_slices, 6) : 1;
        hipError_t e;
        {
            // One upload at a time per process: several contexts that start their calls together (a job dealt to eight contexts) would
            // otherwise share the host's memory bandwidth and the PCIe link eight ways, every batch would arrive late, and the device
            // would idle until the first one is complete.  In turn, the first batch is on the device after one eighth of that time and
            // the uploads of the others run under its kernels.  (PGA_UPLOAD_TURNS=0: no turns.)
            static std::mutex upload_turns[64];        // one per device: only contexts that share a link take turns
            static const bool turns = !(getenv("PGA_UPLOAD_TURNS") && atoi(getenv("PGA_UPLOAD_TURNS")) == 0);
            std::unique_lock<std::mutex> turn(upload_turns[c->device & 63], std::defer_lock);
            if (turns && total >= (8 << 20)) turn.lock();
            if (threads == 1) work(); else c->finder->up_pool.run(work, threads);
            e = (hipError_t)first_err.load();
            if (e == hipSuccess) e = batch_upload_tiles(b, b->d_seq + total + 16, tiles, tile0, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) { batch_give_dev(c, b->d_seq, b->d_seq_cap); delete b; return pga_hip_try_(c, e, "upload of the batch"); }
    }
    *out = b;
    return PGA_OK;
}

extern "C" int pga_batch_create_packed(pga_ctx* c, int32_t n_contigs, const char* packed, const int64_t* offs, const int64_t* lens, pga_batch** out) {
    if (out) *out = nullptr;
    if (!c || !out || n_contigs < 0 || (n_contigs > 0 && (!packed || !offs || !lens))) { if (c) c->err = "pga_batch_create_packed: bad arguments"; return PGA_EINVAL; }
    for (int i = 0; i < n_contigs; i++) {
        if (lens[i] < 0 || lens[i] > 0x7fff0000LL || offs[i] != offs[0] + (i ? offs[i - 1] - offs[0] + lens[i - 1] : 0)) {
            c->err = "pga_batch_create_packed: contigs must lie back to back"; return PGA_EINVAL;
        }
    }
    if (!c->finder) { int rc = pga_finder_models_changed(c); if (rc) return rc; }
    HT(c, hipSetDevice(c->device));
    pga_batch* b = new (std::nothrow) pga_batch();
    if (!b) return PGA_ENOMEM;
    b->ctx = c; b->n = n_contigs; b->d_seq = nullptr; b->d_tiles = nullptr; b->d_tile0 = nullptr; b->n_tiles = 0; b->ct.resize((size_t)n_contigs + 1);
    int64_t total = 0;
    for (int i = 0; i < n_contigs; i++) { b->ct[i].base = total; b->ct[i].len = (int32_t)lens[i]; b->ct[i]._pad = 0; total += lens[i]; }
    b->ct[n_contigs].base = total; b->ct[n_contigs].len = 0; b->ct[n_contigs]._pad = 0;
    b->total = total;
    if (total >= 0x7fffffffLL) { delete b; c->err = "pga_batch_create_packed: batch larger than 2^31 bases; split it"; return PGA_EINVAL; }
    if (total > 0) {
        std::vector<TileDesc> tiles; std::vector<int32_t> tile0;
        batch_tiles(b, tiles, tile0);
        if (batch_take_dev(c, (size_t)total + 16 + batch_tiles_bytes(tiles, tile0), &b->d_seq, &b->d_seq_cap) != hipSuccess) { delete b; c->err = "pga_batch_create_packed: hipMalloc failed"; return PGA_ENOMEM; }
        // the letters go straight from the caller's (pinned) buffer: one DMA, overlapped with the tile list's host work
        std::lock_guard<std::mutex> up(c->finder->up_mu);
        hipStream_t st = nullptr;
        { const int rc = upload_resources(c, 0, &st, nullptr); if (rc) { batch_give_dev(c, b->d_seq, b->d_seq_cap); delete b; return rc; } }
        hipError_t e = hipMemcpyAsync(b->d_seq, packed + offs[0], (size_t)total, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = batch_upload_tiles(b, b->d_seq + total + 16, tiles, tile0, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { batch_give_dev(c, b->d_seq, b->d_seq_cap); delete b; return pga_hip_try_(c, e, "upload of the packed batch"); }
    }
    *out = b;
    return PGA_OK;
}

// what translate.hip needs of a batch
struct pga_batch_view { pga_ctx* ctx; int32_t n; int64_t total; const ContigDesc* ct; const char* d_seq; };
pga_batch_view pga_batch_peek(const pga_batch* b) { return pga_batch_view{b->ctx, b->n, b->total, b->ct.data(), b->d_seq}; }

extern "C" void pga_batch_free(pga_batch* b) {
    if (!b) return;
    if (b->d_seq) { hipSetDevice(b->ctx->device); batch_give_dev(b->ctx, b->d_seq, b->d_seq_cap); }
    delete b;
}

extern "C" int pga_find_genes_batch(pga_ctx* c, int32_t n_contigs, const char* const* seqs, const int64_t* lens,
                                    const pga_params* pp, pga_result** out) {
    if (out) *out = nullptr;
    pga_batch* b = nullptr;
    int rc = pga_batch_create(c, n_contigs, seqs, lens, &b);
    if (rc != PGA_OK) return rc;
    rc = pga_find_genes(c, b, pp, out);
    pga_batch_free(b);
    return rc;
}

extern "C" void pga_result_free(pga_result* r) {
    if (r) delete reinterpret_cast<ResultOwner*>(r);
}

// hand the result over to the caller
static int publish(ResultOwner* R, ResultOwner*& guarded, const pga_params& P, pga_result** out) {
    R->pub.contigs = R->contigs.data();
    R->pub.genes = R->pin_genes != nullptr ? R->pin_genes : R->genes.data();
    R->pub.n_genes = (int64_t)(R->pin_genes != nullptr ? R->pin_n : R->genes.size());
    R->pub.nodes = !R->nodes.empty() ? R->nodes.data() : nullptr;
    R->pub.mask_off = P.mask ? R->mask_off.data() : nullptr;
    R->pub.masks = P.mask ? R->masks.data() : nullptr;
    guarded = nullptr;
    *out = &R->pub;
    return PGA_OK;
}

// stage 0 = the whole path; PGA_STAGE_* = stop after that stage and return the node arrays (single chain per
// contig scored with model 0; P.meta then only selects the meta-mode start penalties of Nodes.score)
static int find_impl(pga_ctx* c, const pga_batch* batch, const pga_params* pp, const int stage, const int tt_override, pga_result** out) {
    if (out) *out = nullptr;
    if (!c || !out || !pp || !batch || batch->ctx != c) {
        if (c) c->err = "pga_find_genes: bad arguments";
        return PGA_EINVAL;
    }
    const int32_t n_contigs = batch->n;
    if (!c->finder) { int rc0 = pga_finder_models_changed(c); if (rc0) return rc0; }
    const bool meta_run = pp->meta && stage == 0;
    // meta mode over an empty bin collection is legal and finds nothing (ref: tests/test_gene_finder.py:316-324)
    if (c->n_models <= 0 && !meta_run && stage != PGA_STAGE_EXTRACT && stage != PGA_STAGE_SEQUENCE) { c->err = "pga_find_genes: no model loaded (call pga_set_models first)"; return PGA_EINVAL; }
    const pga_params P = *pp;
    if (P.mask && P.min_mask < 0) { c->err = "pga_find_genes: min_mask must be positive"; return PGA_EINVAL; }   // ref: lib.pyx:5175-5176
    if (!P.closed && P.min_edge_gene > 0 && P.min_edge_gene < 4) {
        // with open ends and min_edge_gene <= 3 the reference emits a start node AND the virtual edge stop node at the same
        // position and strand; the per-position node layout of this path holds one of them.  Nobody calls genes of one codon.
        c->err = "pga_find_genes: min_edge_gene below 4 is not supported with open ends";
        return PGA_EINVAL;
    }
    if (P.min_gene <= 0 || P.min_edge_gene <= 0 || P.max_overlap < 0 || P.max_overlap > P.min_gene) {
        c->err = "pga_find_genes: invalid min_gene / min_edge_gene / max_overlap";   // ref: lib.pyx:5169-5181
        return PGA_EINVAL;
    }
    HT(c, hipSetDevice(c->device));
    FinderState* f = c->finder;
    hipStream_t st = c->stream;
    const int NC = n_contigs, NM = c->n_models, NG = meta_run ? (int)f->group_tt.size() : 1;
    if (NG > 4) { c->err = "pga_find_genes: more than 4 distinct translation tables loaded"; return PGA_EINVAL; }
    if (const char* fault = getenv("PGA_FAULT_CONTIG_LEN")) {
        // diagnostics: a call that carries a contig of exactly this many bases fails (the host layer's error paths under test)
        const long fl = atol(fault);
        for (int i = 0; i < NC; i++) if ((long)batch->ct[i].len == fl) { c->err = "pga_find_genes: fault injected by PGA_FAULT_CONTIG_LEN"; return PGA_EDEVICE; }
    }

    ResultOwner* R = new (std::nothrow) ResultOwner();
    if (!R) return PGA_ENOMEM;
    struct Guard { ResultOwner* r; ~Guard() { delete r; } } guard{R};
    R->contigs.assign(NC, pga_contig_result{-1, 0, 0, 0, 0, 0.0, 0.0});
    R->mask_off.assign(NC + 1, 0);
    memset(&R->pub, 0, sizeof R->pub);
    R->pub.n_contigs = NC;

    const std::vector<ContigDesc>& ct = batch->ct;
    const int64_t total = batch->total;
    StageTimer tm;

    if (NC > 0 && total > 0) {
        const char* d_seq = batch->d_seq;
        DEVBUF(d_dig, uint8_t, "d_dig", total + 16);
        DEVBUF(d_ct, ContigDesc, "d_ct", NC + 1);
        // What the host reads back after the extraction -- the staging-overflow flag, the contigs' GC / unknown counts, the node and stop-node
        // offsets of every group, the groups a contig is extracted under -- sits in ONE device allocation and ONE pinned one with the same
        // layout: one read-back instead of five, one memset instead of two (round 6: a lone 20 kbp call was 56 launches, 31 of them the
        // runtime's own copy and fill kernels).
        const size_t xa_cnt = 4, xa_cbase = xa_cnt + 2 * (size_t)NC, xa_sbase = xa_cbase + (size_t)NG * (NC + 1), xa_en = xa_sbase + (size_t)NG * (NC + 1);
        const size_t xa_words = xa_en + ((size_t)NG * NC + 3) / 4 + 1;
        DEVBUF(d_xa, int32_t, "d_extract_info", xa_words);
        PINBUF(h_xa, int32_t, "h_extract_info", xa_words);
        int32_t* const d_st_overflow = d_xa; int32_t* const h_st_overflow = h_xa;
        int32_t* const d_cnt = d_xa + xa_cnt; int32_t* const h_cnt = h_xa + xa_cnt;
        const int64_t gc_blocks = pga_gc_blocks(total);
        DEVBUF(d_p16, int32_t, "d_gc_p16", total / 16 + 2);
        DEVBUF(d_gc_bsum, int32_t, "d_gc_bsum", gc_blocks + 1);
        DEVBUF(d_gc_boff, int32_t, "d_gc_boff", gc_blocks + 1);
        int32_t* const d_cbase = d_xa + xa_cbase; int32_t* const h_cbase = h_xa + xa_cbase;
        DEVBUF(d_tile_first, int32_t, "d_tile_first", (size_t)6 * batch->n_tiles);
        DEVBUF(d_tile_last, int32_t, "d_tile_last", (size_t)6 * batch->n_tiles);
        DEVBUF(d_tile_count, int32_t, "d_tile_count", (size_t)NG * (batch->n_tiles + 1));
        DEVBUF(d_tile_off, int32_t, "d_tile_off", (size_t)NG * (batch->n_tiles + 1));
        DEVBUF(d_tile_scount, int32_t, "d_tile_scount", (size_t)NG * (batch->n_tiles + 1));
        DEVBUF(d_tile_soff, int32_t, "d_tile_soff", (size_t)NG * (batch->n_tiles + 1));
        int32_t* const d_sbase = d_xa + xa_sbase; int32_t* const h_sbase = h_xa + xa_sbase;
        uint8_t* const d_enabled = (uint8_t*)(d_xa + xa_en); uint8_t* const h_enabled = (uint8_t*)(h_xa + xa_en);

        GroupArrays ga[4];
        for (int g = 0; g < NG; g++) {
            char nm[32];
#define GBUF(field, type, count) { snprintf(nm, sizeof nm, #field "%d", g); void* p__; int rc__ = ensure_dev(c, nm, sizeof(type) * (size_t)(count) + 64, &p__); if (rc__) return rc__; ga[g].field = (type*)p__; }
            GBUF(df, uint8_t, total + 16)
            GBUF(c16, int32_t, (size_t)batch->n_tiles * 192 + 2)
            ga[g].tile0 = batch->d_tile0;
            ga[g].ndx = nullptr; ga[g].stop_val = nullptr; ga[g].type = nullptr; ga[g].strand = nullptr; ga[g].edge0 = nullptr; ga[g].gc_cont = nullptr;
        }

        HT(c, hipEventRecord(f->e_start, st));
        HT(c, hipMemcpyAsync(d_ct, ct.data(), sizeof(ContigDesc) * (NC + 1), hipMemcpyHostToDevice, st));
        HT(c, hipMemsetAsync(d_xa, 0, sizeof(int32_t) * xa_cbase, st));          // the overflow flag and the counts
        pga_launch_digitize(d_seq, d_dig, total, d_ct, NC, d_cnt, d_cnt + NC, st);
        pga_launch_gc_prefix(d_dig, total, d_gc_bsum, d_gc_boff, d_p16, st);
        MaskList masks{nullptr, nullptr};
        if (P.mask) {
            // masked regions: found on the device, ordered on the host (a handful of intervals)
            const int cap = (int)std::min<int64_t>(total / std::max(1, P.min_mask) + NC + 1, (int64_t)1 << 28);
            DEVBUF(d_runs, MaskRun, "d_mask_runs", cap + 1);
            DEVBUF(d_nruns, int32_t, "d_mask_count", 4);
            DEVBUF(d_moff, int32_t, "d_mask_off", NC + 2);
            DEVBUF(d_miv, int2, "d_mask_iv", cap + 1);
            pga_launch_find_masks(d_dig, d_ct, NC, batch->d_tiles, batch->n_tiles, P.min_mask, d_runs, d_nruns, cap, st);
            int32_t nruns = 0;
            HT(c, hipMemcpyAsync(&nruns, d_nruns, 4, hipMemcpyDeviceToHost, st));
            HT(c, hipStreamSynchronize(st));
            nruns = std::min(nruns, cap);
            std::vector<MaskRun> runs((size_t)nruns);
            if (nruns > 0) HT(c, hipMemcpy(runs.data(), d_runs, sizeof(MaskRun) * nruns, hipMemcpyDeviceToHost));
            std::sort(runs.begin(), runs.end(), [](const MaskRun& a, const MaskRun& b) { return a.contig != b.contig ? a.contig < b.contig : a.begin < b.begin; });
            std::vector<int32_t> moff(NC + 1, 0);
            std::vector<int2> miv((size_t)nruns + 1);
            for (int k = 0; k < nruns; k++) { moff[runs[k].contig + 1]++; miv[k] = make_int2(runs[k].begin, runs[k].end); }
            for (int i = 0; i < NC; i++) moff[i + 1] += moff[i];
            R->mask_off = moff;
            R->masks.resize(2 * (size_t)nruns);
            for (int k = 0; k < nruns; k++) { R->masks[2 * k] = runs[k].begin; R->masks[2 * k + 1] = runs[k].end; }
            HT(c, hipMemcpy(d_moff, moff.data(), sizeof(int32_t) * (NC + 1), hipMemcpyHostToDevice));
            if (nruns > 0) HT(c, hipMemcpy(d_miv, miv.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice));
            masks = MaskList{d_moff, d_miv};
        }
        if (stage == PGA_STAGE_SEQUENCE) {
            HT(c, hipMemcpyAsync(h_cnt, d_cnt, sizeof(int32_t) * 2 * (size_t)NC, hipMemcpyDeviceToHost, st));
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            for (int i = 0; i < NC; i++) {
                R->contigs[i].gc = ct[i].len > 0 ? (double)h_cnt[i] / (double)ct[i].len : 0.0;
                R->contigs[i].n_unknown = h_cnt[NC + i];
            }
            return publish(R, guard.r, P, out);
        }
        // meta mode: a contig is extracted under a translation table only if a model with that table lies in its GC window
        if (meta_run && NM > 0) pga_launch_group_enable(d_ct, NC, d_cnt, f->d_model_gc, f->d_model_grp, NM, NG, d_enabled, st);
        // Staging of the extraction: one slot per two positions of a tile (sequence has a node every 25 positions or so; the full
        // two-slots-per-position staging was half of a context's memory).  A tile that does not fit raises a flag, the batch is then
        // extracted again with full staging, and the context keeps that (PGA_STAGE_FULL=1: from the start).
        if (getenv("PGA_STAGE_FULL")) f->stage_full = true;
        bool stage_full = f->stage_full;
        c->extract_passes = 0;
        for (;;) {
            const bool half = !stage_full;
            // (PGA_STAGE_SHIFT=5: one slot per 32 positions, so that ordinary sequence overflows and the tests see the second pass)
            const int shift = half ? std::max(1, std::min(8, getenv("PGA_STAGE_SHIFT") ? atoi(getenv("PGA_STAGE_SHIFT")) : 1)) : 0;
            const int64_t st_slots = half ? (total >> shift) + (int64_t)PGA_STAGE_SLACK * batch->n_tiles + 8 : 2 * total + 2;
            for (int g = 0; g < NG; g++) {
                char nm[32];
                GBUF(st_ndx, int32_t, st_slots) GBUF(st_sv, int32_t, st_slots) GBUF(st_info, uint8_t, st_slots)
                ga[g].st_half = shift; ga[g].st_overflow = d_st_overflow;
            }
            if (c->extract_passes > 0) HT(c, hipMemsetAsync(d_st_overflow, 0, sizeof(int32_t), st));      // (the first pass: cleared with the counts)
            for (int g = 0; g < NG; g++) {
                const int tt = meta_run ? f->group_tt[g] : (stage == PGA_STAGE_EXTRACT ? tt_override : c->models[0].trans_table);
                pga_launch_extract(d_dig, total, d_ct, NC, tt, P, ga[g], batch->d_tiles, batch->n_tiles, batch->d_tile0, d_tile_first, d_tile_last,
                                   d_tile_count + (size_t)g * (batch->n_tiles + 1), d_tile_off + (size_t)g * (batch->n_tiles + 1), d_cbase + (size_t)g * (NC + 1),
                                   d_tile_scount + (size_t)g * (batch->n_tiles + 1), d_tile_soff + (size_t)g * (batch->n_tiles + 1), d_sbase + (size_t)g * (NC + 1),
                                   masks, st, (meta_run && NM > 0) ? d_enabled + (size_t)g * NC : nullptr);
            }
            HT(c, hipMemcpyAsync(h_xa, d_xa, sizeof(int32_t) * xa_words, hipMemcpyDeviceToHost, st));      // flag, counts, offsets, enabled groups
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            c->extract_passes++;
            if (!half || h_st_overflow[0] == 0) break;
            stage_full = true;
            if (!getenv("PGA_STAGE_SHIFT")) f->stage_full = true;       // (the test knob leaves the context as it was)
        }

        tm.mark("extract+sync");
        // ---- topology buffers; the nodes go to their places and get their GC content while the host plans the chains: neither kernel
        //      reads anything of the plan (round 6: the device sat idle through the 0.7 ms of planning of a 6 250-contig call)
        int64_t group_nodes[4] = {0, 0, 0, 0};
        for (int g = 0; g < NG; g++) {
            char nm[32];
            group_nodes[g] = h_cbase[(size_t)g * (NC + 1) + NC];
            const int64_t n = group_nodes[g] + 1;
            GBUF(ndx, int32_t, n) GBUF(stop_val, int32_t, n) GBUF(type, uint8_t, n) GBUF(strand, int8_t, n) GBUF(edge0, uint8_t, n) GBUF(gc_cont, float, n) GBUF(contig_of, int32_t, n)
            GBUF(stop_list, int32_t, (int64_t)h_sbase[(size_t)g * (NC + 1) + NC] + 1)
            GBUF(start_list, int32_t, group_nodes[g] - (int64_t)h_sbase[(size_t)g * (NC + 1) + NC] + 1)
            GBUF(ovl_topo, uint32_t, (int64_t)h_sbase[(size_t)g * (NC + 1) + NC] + 1)
            GBUF(srank, int32_t, n + 4)
        }
        for (int g = 0; g < NG; g++) {
            pga_launch_place(d_ct, batch->d_tiles, batch->n_tiles, d_tile_off + (size_t)g * (batch->n_tiles + 1),
                             d_tile_soff + (size_t)g * (batch->n_tiles + 1), ga[g], st);
            pga_launch_orf_gc(d_ct, NC, d_dig, d_p16, ga[g], (int)group_nodes[g], d_cbase + (size_t)g * (NC + 1), st);
        }
        // ---- plan the (contig, model) chains (ref: lib.pyx:5335-5362) ------------------------
        std::vector<std::vector<ChainDesc>> gch(NG);     // per group, in (contig, model) order
        std::vector<double> mgc((size_t)NM); std::vector<int> mtt((size_t)NM);     // the models are 558 KB apart: keep what the loop reads together
        for (int m = 0; m < NM; m++) { mgc[m] = c->models[m].gc; mtt[m] = c->models[m].trans_table; }
        for (int g = 0; g < NG; g++) gch[g].reserve(meta_run ? (size_t)NC * 6 : (size_t)NC);
        for (int i = 0; i < NC; i++) {
            const int L = ct[i].len;
            const double gc = L > 0 ? (double)h_cnt[i] / (double)L : 0.0;
            R->contigs[i].gc = gc;
            R->contigs[i].n_unknown = h_cnt[NC + i];
            if (!meta_run) {
                const int32_t* cb = h_cbase;
                ChainDesc ch{0, cb[i], cb[i + 1] - cb[i], 0, i, 1};
                gch[0].push_back(ch);
                continue;
            }
            const double low = fmin(0.65, 0.88495 * gc - 0.0102337), high = fmax(0.35, 0.86596 * gc + 0.1131991);
            int tt_prev = -1;
            for (int m = 0; m < NM; m++) {
                if (mgc[m] < low || mgc[m] > high) continue;
                const int g = f->model_group[m];
                if (!h_enabled[(size_t)g * NC + i]) { c->err = "pga_find_genes: host and device disagree on a GC window"; return PGA_EDEVICE; }
                const int32_t* cb = h_cbase + (size_t)g * (NC + 1);
                ChainDesc ch{0, cb[i], cb[i + 1] - cb[i], m, i, mtt[m] != tt_prev ? 1 : 0};
                ch.group = g;
                tt_prev = mtt[m];
                gch[g].push_back(ch);
            }
        }
        std::vector<ChainDesc> chains;
        { size_t tot = 0; for (int g = 0; g < NG; g++) tot += gch[g].size(); chains.reserve(tot); }
        std::vector<int> g_c0(NG + 1, 0);
        std::vector<int64_t> g_n0(NG + 1, 0);
        int64_t tot_chain_nodes = 0, tot_chain_stops = 0;
        std::vector<int64_t> g_s0(NG + 1, 0);            // (chain, stop node) pairs before the chains of group g
        for (int g = 0; g < NG; g++) {
            g_c0[g] = (int)chains.size(); g_n0[g] = tot_chain_nodes;
            for (ChainDesc& ch : gch[g]) {
                ch.off = tot_chain_nodes; tot_chain_nodes += ch.n;
                ch.soff = tot_chain_stops;
                const int32_t* sb = h_sbase + (size_t)(meta_run ? g : 0) * (NC + 1);
                tot_chain_stops += sb[ch.contig + 1] - sb[ch.contig];
                chains.push_back(ch);
            }
            g_s0[g + 1] = tot_chain_stops;
        }
        g_c0[NG] = (int)chains.size(); g_n0[NG] = tot_chain_nodes;
        const int NCH = (int)chains.size();
        R->pub.node_passes = tot_chain_nodes;
        R->pub.n_chains = NCH;
        // space for the fresh re-scores of winners that were not `first` (at most one per contig)
        int64_t max_rescore = 0;
        if (meta_run) for (int g = 0; g < NG; g++) max_rescore += h_cbase[(size_t)g * (NC + 1) + NC];   // loose bound: every node once per group
        const int64_t chain_cap = tot_chain_nodes + max_rescore + 64;

        // ---- chain buffers ---------------------------------------------------------------------
        ChainArrays ca;
        {
            DEVBUF(a0, double, "ca_cscore", chain_cap) DEVBUF(a1, double, "ca_sscore", chain_cap) DEVBUF(a2, double, "ca_rscore", chain_cap)
            DEVBUF(a3, double, "ca_uscore", chain_cap) DEVBUF(a4, double, "ca_tscore", chain_cap) DEVBUF(a5, double, "ca_mot_score", chain_cap)
            DEVBUF(a6, int32_t, "ca_star_ptr", 3 * chain_cap) DEVBUF(a7, int32_t, "ca_mot_ndx", chain_cap) DEVBUF(a8, uint8_t, "ca_rbs", 2 * chain_cap)
            DEVBUF(a9, uint8_t, "ca_edge", chain_cap) DEVBUF(a10, uint8_t, "ca_mot_len", chain_cap) DEVBUF(a11, uint8_t, "ca_mot_spacer", chain_cap)
            DEVBUF(a12, uint8_t, "ca_mot_spacendx", chain_cap)
            int64_t max_contig_nodes = 0;
            for (int g = 0; g < NG; g++) for (int i = 0; i < NC; i++)
                max_contig_nodes = std::max<int64_t>(max_contig_nodes, h_cbase[(size_t)g * (NC + 1) + i + 1] - h_cbase[(size_t)g * (NC + 1) + i]);
            DEVBUF(a13, double, "ca_cscore_raw", chain_cap + max_contig_nodes + 64)
            ca = ChainArrays{a0, a1, a2, a3, a4, a5, a13, a6, a7, a8, a9, a10, a11, a12, a13 + chain_cap};
        }
        // few long chains: their walks are cut into segments that run side by side (dp.hip "segmented chains")
        DpSegPlan seg_plan;
        // many chains: one wavefront each, records and far-field structures of dp_wave.hip
        const bool use_wave = stage == 0 && pga_dp_use_wave(NCH);
        // Round 6 -- few LONG chains whose segments the wave-batch kernel walks (dp.hip, launch_dp_segmented): a wavefront per segment, 2048
        // segments at once where the chain kernel's workgroup per compute unit allows 252.  Every segment pays the same 4096-node warm-up and
        // a wavefront walks a node more slowly than the sixteen of a chain-kernel workgroup, so this pays where the chains are long enough --
        // from PGA_DP_SEG_WAVE_MIN (2.5 M) nodes in chains of 16 k nodes or more; PGA_DP_SEG_WAVE=1 / 0: always / never (tests).
        bool seg_wave = false;
        if (stage == 0 && !use_wave && pga_dpw_use_sched() && !getenv("PGA_DP_KERNEL")) {
            int64_t cand = 0;
            const int min_chain = std::max(256, getenv("PGA_DP_SEG_MIN") ? atoi(getenv("PGA_DP_SEG_MIN")) : 16384);      // (as pga_dp_plan)
            for (int k = 0; k < NCH; k++) if (chains[(size_t)k].n >= min_chain) cand += chains[(size_t)k].n;
            const char* e = getenv("PGA_DP_SEG_WAVE");
            const char* m = getenv("PGA_DP_SEG_WAVE_MIN");
            seg_wave = e ? atoi(e) != 0 : cand >= (m ? atoll(m) : 2500000ll);
        }
        // what the wave-batch kernel reads -- topology, step schedule, cs, extras -- is made for its own launches and for such segments
        bool wave_prep = use_wave || seg_wave;
        // the step schedule of the wave-batch scorer: one header per 64-node batch of every contig of a group (dpw_core.h)
        bool use_sched = wave_prep && pga_dpw_use_sched();
        std::vector<int32_t> h_bbase;
        if (use_sched) {
            h_bbase.resize((size_t)NG * (NC + 1));
            for (int g = 0; g < NG; g++) {
                const int32_t* cb = h_cbase + (size_t)g * (NC + 1);
                int32_t* bb = h_bbase.data() + (size_t)g * (NC + 1);
                bb[0] = 0;
                for (int i = 0; i < NC; i++) bb[i + 1] = bb[i] + ((cb[i + 1] - cb[i] + 63) >> 6);
            }
            for (ChainDesc& ch : chains) ch.sched_b0 = h_bbase[(size_t)ch.group * (NC + 1) + ch.contig];
        }
        int max_batches[4] = {0, 0, 0, 0};
        if (use_sched) for (int g = 0; g < NG; g++) for (int i = 0; i < NC; i++)
            max_batches[g] = std::max(max_batches[g], h_bbase[(size_t)g * (NC + 1) + i + 1] - h_bbase[(size_t)g * (NC + 1) + i]);
        bool segmented = stage == 0 && !use_wave && pga_dp_plan(chains.data(), NCH, tot_chain_nodes, seg_plan, seg_wave);
        if (seg_wave && !segmented) { seg_wave = false; wave_prep = use_wave; use_sched = false; h_bbase.clear(); }       // (nothing to cut after all)
        // lean: the tail runs on the device and nobody asked for node arrays, so only what the tail writes is gathered; direct: it
        // also reads the winners' node fields where the scorers left them.  (One definition: the scorers skip the star_ptr fill under
        // exactly the condition under which the tail never looks at it.)
        const bool lean_gather = !P.want_nodes && !(getenv("PGA_TAIL") && strcmp(getenv("PGA_TAIL"), "host") == 0) && !getenv("PGA_FULL_GATHER");
        const bool direct_gather = lean_gather && !getenv("PGA_GATHER_ALL_DP");
        const int64_t dp_cap = tot_chain_nodes + seg_plan.extra + 1;
        const int64_t dp_slots = NCH + (int64_t)seg_plan.segs.size() + 1;
        // records and far-field arrays of the tree / chain kernels (sub-chains walked by the wave-batch kernel need none of their own)
        const int64_t tree_cap = use_wave ? 1 : (seg_wave ? tot_chain_nodes + 1 : dp_cap);
        DpBuffers dp;
        {
            DEVBUF(b0, DpSrc, "dp_src", tree_cap) DEVBUF(b1, DpTgt, "dp_tgt", tree_cap)
            DEVBUF(b2, double, "dp_score", dp_cap) DEVBUF(b3, int32_t, "dp_traceb", dp_cap)
            DEVBUF(b4, int32_t, "dp_tbn", dp_cap) DEVBUF(b5, int8_t, "dp_ov", dp_cap)
            // (what the host reads back of a chain -- best gene end, its score, the path's start -- in one allocation: one read-back)
            DEVBUF(b7, double, "dp_chain_results", 2 * dp_slots + 2)
            int32_t* const b6 = (int32_t*)(b7 + dp_slots); int32_t* const b8 = b6 + dp_slots;
            DEVBUF(b9, double, "dp_A", tree_cap) DEVBUF(b10, double, "dp_V0", tree_cap) DEVBUF(b11, double, "dp_V1", tree_cap)
            DEVBUF(b12, double, "dp_V2", tree_cap) DEVBUF(b13, double, "dp_hv", tree_cap) DEVBUF(b14, int32_t, "dp_hi", tree_cap)
            dp = DpBuffers{b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, {b10, b11, b12}, b13, b14, nullptr};
        }
        DpwGroupPtrs wgroups{};
        DpwBuffers wbuf{};
        if (wave_prep) {
            for (int g = 0; g < NG; g++) {
                char nm[32];
                const int64_t n = group_nodes[g] + 1;
#define WBUF(field, type) { snprintf(nm, sizeof nm, "dpw_" #field "%d", g); void* p__; int rc__ = ensure_dev(c, nm, sizeof(type) * (size_t)n + 64, &p__); if (rc__) return rc__; wgroups.g[g].field = (type*)p__; }
                WBUF(kf, uint8_t) WBUF(lo, int32_t) WBUF(q1, int32_t) WBUF(q2, int32_t)
#undef WBUF
                wgroups.g[g].ndx = ga[g].ndx; wgroups.g[g].stop_val = ga[g].stop_val; wgroups.g[g].srank = ga[g].srank;
                if (use_sched) {
                    // headers: one per batch; lists: DPW_SCHED_STRIDE 32-byte slots per batch (what does not fit is counted and the launch
                    // falls back to k_dpw_dyn)
                    const int64_t nbat = h_bbase[(size_t)g * (NC + 1) + NC];
                    void* p__; int rc__;
                    snprintf(nm, sizeof nm, "dpw_shdr%d", g); rc__ = ensure_dev(c, nm, sizeof(DpwSchedHdr) * (size_t)(nbat + 1), &p__); if (rc__) return rc__; wgroups.g[g].shdr = (DpwSchedHdr*)p__;
                    snprintf(nm, sizeof nm, "dpw_sent%d", g); rc__ = ensure_dev(c, nm, sizeof(DpwSlot) * (size_t)DPW_SCHED_STRIDE * (size_t)(nbat + 1) + 256, &p__); if (rc__) return rc__; wgroups.g[g].sent = (uint4*)p__;
                    snprintf(nm, sizeof nm, "dpw_scur%d", g); rc__ = ensure_dev(c, nm, 64, &p__); if (rc__) return rc__; wgroups.g[g].scur = (uint32_t*)p__;
                }
            }
            DEVBUF(w0, double, "dpw_cs", dp_cap + 2) DEVBUF(w1, DpwExt, "dpw_ext", tot_chain_stops + 4)      /* one extras record per (chain, stop node) pair */ DEVBUF(w2, double, "dpw_sfxv", dp_cap) DEVBUF(w3, int32_t, "dpw_sfxi", dp_cap)
            wbuf = DpwBuffers{w0, w1, w2, w3};
            // the topology of every group now: it reads the placed nodes and nothing of the plan, and runs under the rest of the planning
            // (start order, the copy of the plan, the coding-score tasks)
            for (int g = 0; g < NG; g++) {
                if (g_c0[g + 1] - g_c0[g] == 0 || g_n0[g + 1] - g_n0[g] == 0 || stage == PGA_STAGE_EXTRACT) continue;
                int max_nodes_g = 0;
                for (int i = 0; i < NC; i++) max_nodes_g = std::max(max_nodes_g, h_cbase[(size_t)g * (NC + 1) + i + 1] - h_cbase[(size_t)g * (NC + 1) + i]);
                if (g < 4) HT(c, hipEventRecord(f->e_aux[4 * g], st));
                pga_launch_dpw_topo(wgroups.g[g], ga[g].type, ga[g].strand, d_cbase + (size_t)g * (NC + 1), NC, (int)group_nodes[g], st, max_nodes_g);
                if (g < 4) HT(c, hipEventRecord(f->e_aux[4 * g + 1], st));
            }
        }
        DpSegDev seg_dev{};
        if (segmented) {
            DEVBUF(seg_arena, char, "dp_seg_arena", pga_dp_seg_bytes(seg_plan, NCH, tot_chain_nodes));
            HT(c, pga_dp_seg_bind(seg_plan, NCH, tot_chain_nodes, seg_arena, st, &seg_dev));
            if (seg_wave) { seg_dev.wave_groups = &wgroups; seg_dev.wave_buf = &wbuf; }
        }
        // start order of the wave-batch scorer: longest chains first (counting sort on nodes / 64 = walk batches)
        int32_t* d_dp_order = nullptr;
        std::vector<int32_t> dp_order;
        if (use_wave && NCH > 1 && !getenv("PGA_DP_NO_ORDER")) {
            std::vector<int32_t> lens((size_t)NCH);
            for (int k = 0; k < NCH; k++) lens[(size_t)k] = chains[k].n;
            dp_order.resize((size_t)NCH);
            pga_dp_start_order(NCH, lens.data(), dp_order.data());
            // Workgroups go to the eight XCDs in turn (b % 8), each with an L2 of its own, and the chains of a contig (one per model of
            // the same translation table) read the same topology arrays.  So every (contig, table) goes to ONE XCD -- the one with
            // the fewest nodes so far, in start order -- and workgroup 8 k + x takes the k-th chain of XCD x's queue; queues that end
            // early are filled up with -1 (the workgroup returns at once).  PGA_DP_XCD=0: the order as it is.
            // Only where the premise holds -- a device of eight XCDs (an MI355X in its default SPX mode; a partitioned one reports fewer) --
            // and where the queues come out even: a handful of keys, or one contig far longer than the rest, would leave XCDs idle
            // behind one long queue, and the plain longest-first order is the better one then.
            static std::atomic<int> xcc_of[64];      // per device: 0 = not asked yet
            int n_xcc = xcc_of[c->device & 63].load();
            if (n_xcc == 0) {
                int v = 0;
                n_xcc = (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, c->device) == hipSuccess && v > 0) ? v : 8;
                xcc_of[c->device & 63].store(n_xcc);
            }
            if (!(getenv("PGA_DP_XCD") && atoi(getenv("PGA_DP_XCD")) == 0) && NCH >= 64 && n_xcc == 8) {
                std::vector<int32_t> key((size_t)NCH), by_xcd((size_t)NCH * 8);
                for (int k = 0; k < NCH; k++) key[(size_t)k] = chains[(size_t)k].group * NC + chains[(size_t)k].contig;
                const int64_t nb = pga_dp_xcd_order(NCH, dp_order.data(), lens.data(), key.data(), NC * NG, by_xcd.data(), (int64_t)by_xcd.size());
                int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0, heaviest = 0;
                for (int64_t k = 0; k < nb; k++) if (by_xcd[(size_t)k] >= 0) { load[k & 7] += lens[(size_t)by_xcd[(size_t)k]]; total += lens[(size_t)by_xcd[(size_t)k]]; }
                for (int q = 0; q < 8; q++) heaviest = std::max(heaviest, load[q]);
                if (nb > 0 && heaviest * 8 <= total + total * 3 / 10) { by_xcd.resize((size_t)nb); dp_order.swap(by_xcd); }       // within 1.3 x the mean
            }
        }
        PINBUF(h_maxscore, double, "h_chain_results", 2 * dp_slots + 2);
        int32_t* const h_maxidx = (int32_t*)(h_maxscore + dp_slots); int32_t* const h_ipath = h_maxidx + dp_slots;
        PINBUF(h_scur, uint32_t, "h_dpw_scur", 16);
        // the workgroups of k_ovl_stops take 256 (chain, stop node) pairs each: the chain of a workgroup's first pair, relative to its group's
        // first chain, comes with the plan (a search on the device was three dependent loads in front of everything the workgroup does)
        std::vector<int64_t> ovl_blk0((size_t)NG + 1, 0);
        for (int g = 0; g < NG; g++) ovl_blk0[(size_t)g + 1] = ovl_blk0[(size_t)g] + (g_s0[g + 1] - g_s0[g] + 255) / 256;
        // The plan of the call goes to the device in ONE copy (round 6; it was four copies and a memset): the cleared flags "scoring turned a
        // start node into an edge node", per group and contig the run of chains scored on it, the batches before each contig (step
        // schedule), the start order of the connection scoring, the chains -- one pinned staging area, one device area, the same layout.
        // (Behind the chains the device area has room for the re-score chains of the winners, which are uploaded later.)
        const size_t pa_conv = 0, pa_cc = (((size_t)NG * NC + 1) + 7) & ~(size_t)7, pa_bb = pa_cc + sizeof(int2) * (size_t)2 * NG * NC,
                     pa_ord = pa_bb + ((sizeof(int32_t) * (h_bbase.size() + 1) + 7) & ~(size_t)7),
                     pa_blk = pa_ord + ((sizeof(int32_t) * (dp_order.size() + 1) + 7) & ~(size_t)7),
                     pa_ch = pa_blk + ((sizeof(int32_t) * ((size_t)ovl_blk0[NG] + 1) + 7) & ~(size_t)7), pa_copy = pa_ch + sizeof(ChainDesc) * (size_t)NCH;
        DEVBUF(d_plan, char, "d_call_plan", pa_copy + sizeof(ChainDesc) * ((size_t)NC + 2));
        PINBUF(h_plan, char, "h_call_plan", pa_copy + 64);
        uint8_t* const d_conv = (uint8_t*)(d_plan + pa_conv);
        int2* const d_cc = (int2*)(d_plan + pa_cc); int2* const h_cc = (int2*)(h_plan + pa_cc);
        int32_t* const d_bbase = use_sched ? (int32_t*)(d_plan + pa_bb) : nullptr;
        if (!dp_order.empty()) d_dp_order = (int32_t*)(d_plan + pa_ord);
        ChainDesc* const d_chains = (ChainDesc*)(d_plan + pa_ch);
        memset(h_plan + pa_conv, 0, pa_cc);
        // per group and contig: the contiguous run of chains (models) scored on that contig
        for (size_t k = 0; k < (size_t)2 * NG * NC; k++) h_cc[k] = make_int2(0, 0);
        for (int k = 0; k < NCH; k++) {
            const int g = meta_run ? f->model_group[chains[k].model] : 0;
            int2& e = h_cc[(size_t)g * NC + chains[k].contig];
            if (e.y == 0) e.x = k;
            e.y++;
        }
        if (use_sched && !h_bbase.empty()) memcpy(h_plan + pa_bb, h_bbase.data(), sizeof(int32_t) * h_bbase.size());
        if (!dp_order.empty()) memcpy(h_plan + pa_ord, dp_order.data(), sizeof(int32_t) * dp_order.size());
        if (NCH > 0) memcpy(h_plan + pa_ch, chains.data(), sizeof(ChainDesc) * (size_t)NCH);
        const int32_t* const d_ovl_blk = (const int32_t*)(d_plan + pa_blk);
        {
            int32_t* const h_blk = (int32_t*)(h_plan + pa_blk);
            for (int g = 0; g < NG; g++) {
                int k = g_c0[g];
                const int64_t nb = ovl_blk0[(size_t)g + 1] - ovl_blk0[(size_t)g];
                for (int64_t b = 0; b < nb; b++) {
                    const int64_t p0 = g_s0[g] + 256 * b;
                    while (k + 1 < g_c0[g + 1] && chains[(size_t)k + 1].soff <= p0) k++;
                    h_blk[ovl_blk0[(size_t)g] + b] = k - g_c0[g];
                }
            }
        }
        HT(c, hipMemcpyAsync(d_plan, h_plan, pa_copy, hipMemcpyHostToDevice, st));

        tm.mark("plan+alloc");
        std::vector<int32_t> cs_tk[4], cs_en[4];
        ScoreParams sp{P.closed, P.meta, P.max_overlap, NM, nullptr, nullptr};
        PINBUF(h_conv, uint8_t, "h_conv_flag", (size_t)NG * NC + 1);
        const pga_training* d_models = (const pga_training*)c->d_models_raw;
        for (int g = 0; g < NG; g++) {
            const int nch = g_c0[g + 1] - g_c0[g];
            const int64_t nn = g_n0[g + 1] - g_n0[g];
            if (nch == 0 || nn == 0 || stage == PGA_STAGE_EXTRACT) continue;
            // the ORF walks of the coding score run from hexamer tables in LDS, contigs bucketed by the four table columns they need
            // (PGA_CS_LDS=0: the global-memory form)
            const void* d_cs_tasks = nullptr; const void* d_cs_entries = nullptr; int n_cs_tasks = 0;
            const char* cs_env = getenv("PGA_CS_LDS");
            const char* cs_tn = getenv("PGA_CS_TASK_NODES");
            const int cs_task_nodes = cs_tn && atoi(cs_tn) >= 256 && atoi(cs_tn) <= 8192 ? atoi(cs_tn) : 5120;     // several tasks per CU and launch (swept again at the end of round 6, one box: 4096 629 us per launch, 4864 581, 5120 586, 5376 584, 5632 600, 6144 592)
            // PGA_CS_LDS=2 (tests): the LDS form whatever the size of the launch
            // (single mode as well: one table column, a genome is cut into tasks of `cs_task_nodes` nodes)
            // (the global-memory form walks every ORF on one lane through the texture path: a launch takes as long as its longest ORF,
            //  a few hundred microseconds on high-GC sequence whatever its size -- so small launches take the LDS form as well)
            if (!(cs_env && atoi(cs_env) == 0) && !f->gil_stride.empty()) {
                std::vector<int32_t>& tk = cs_tk[g]; std::vector<int32_t>& en = cs_en[g];     // alive until the stream is synchronized
                if (pga_cs_tasks(h_cc + (size_t)g * NC, NC, chains.data(), h_cbase + (size_t)g * (NC + 1), f->model_rank.data(), cs_task_nodes, tk, en) && !tk.empty()) {
                    char nm1[32], nm2[32];
                    snprintf(nm1, sizeof nm1, "cs_tasks%d", g); snprintf(nm2, sizeof nm2, "cs_entries%d", g);
                    void* p1; void* p2;
                    { int rc__ = ensure_dev(c, nm1, tk.size() * 4 + 64, &p1); if (rc__) return rc__; }
                    { int rc__ = ensure_dev(c, nm2, en.size() * 4 + 64, &p2); if (rc__) return rc__; }
                    HT(c, hipMemcpyAsync(p1, tk.data(), tk.size() * 4, hipMemcpyHostToDevice, st));
                    HT(c, hipMemcpyAsync(p2, en.data(), en.size() * 4, hipMemcpyHostToDevice, st));
                    d_cs_tasks = p1; d_cs_entries = p2; n_cs_tasks = (int)(tk.size() / 4);
                }
            }
            sp.conv_flag = meta_run ? d_conv + (size_t)g * NC : nullptr;
            // overlapping starts over (chain, stop node) pairs; for the wave-batch scorer the same pass builds the stops' extras and
            // the start scorer leaves cscore + sscore, so its per-chain preparation is done when scoring is
            StopLaunch sl;
            sl.sbase = d_sbase + (size_t)g * (NC + 1); sl.soff_begin = g_s0[g]; sl.n_pairs = g_s0[g + 1] - g_s0[g];
            sl.n_stops = h_sbase[(size_t)g * (NC + 1) + NC];
            // (the path proper; a stage-level call returns the node arrays after any stage, so there every node keeps its thread)
            sl.starts_only = stage == 0 && !(getenv("PGA_SS_STARTS_ONLY") && atoi(getenv("PGA_SS_STARTS_ONLY")) == 0);
            sl.n_starts = (int32_t)(group_nodes[g] - sl.n_stops);
            sl.blk_chain = d_ovl_blk + ovl_blk0[(size_t)g];
            sp.cs_out = nullptr;
            if (wave_prep) {
                sl.topo_q2 = wgroups.g[g].q2; sl.ext = wbuf.ext; sp.cs_out = wbuf.cs;
                if (use_wave) {
                    // (direct mode: the tail reads star_ptr only at the stop nodes, which k_ovl_stops always writes)
                    sl.fill_star_ptr = !(stage == 0 && direct_gather);
                    // (the path proper, node arrays not asked for: a stop node's zero scores are read by nobody -- ScoreParams::lean_stops)
                    sp.lean_stops = (stage == 0 && direct_gather && sl.starts_only && !getenv("PGA_SS_FULL_STOPS")) ? 1 : 0;
                }       // (segments walked by the wave-batch kernel: the re-scoring and verifying kernels read the node arrays in full)
            }
            pga_launch_score(d_chains + g_c0[g], nch, g_n0[g], nn, d_dig, d_ct, ga[g], d_models, f->d_msc, c->d_model_const, ca, sp,
                             d_chains, d_cc + (size_t)g * NC, d_cbase + (size_t)g * (NC + 1), NC, (int)group_nodes[g], f->d_sd_lut, st, 0,
                             (meta_run || n_cs_tasks > 0) ? f->d_gil + f->gil_off[g] : nullptr, (meta_run || n_cs_tasks > 0) ? f->gil_stride[g] : 0, f->d_model_rank,
                             d_cs_tasks, n_cs_tasks, d_cs_entries, &sl);
            if (wave_prep && use_sched) {
                // (behind the scoring launches: the schedule's headers carry the stop nodes' ranks, which k_ovl_topo writes there)
                if (g < 4) HT(c, hipEventRecord(f->e_aux[4 * g + 2], st));
                pga_launch_dpw_sched(wgroups.g[g], d_cbase + (size_t)g * (NC + 1), d_bbase + (size_t)g * (NC + 1), NC, max_batches[g], st);
                if (g < 4) HT(c, hipEventRecord(f->e_aux[4 * g + 3], st));
                HT(c, hipMemcpyAsync(h_scur + 2 * g, wgroups.g[g].scur, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            }
            NodeArrays na{ga[g].ndx, ga[g].stop_val, ga[g].type, ga[g].strand, ca.cscore, ca.sscore, ca.rscore, ca.uscore, ca.star_ptr};
            if (!use_wave && stage == 0) pga_launch_dp_prepare(d_chains + g_c0[g], nch, g_n0[g], nn, na, c->d_model_const, dp, st);
        }
        if (stage != 0) {
            // ---- stage-level call: bring the node arrays home as they are now and stop ---------------
            const int64_t nn = group_nodes[0];
            PINBUF(hs_i32, int32_t, "hs_i32", 6 * nn + 8);         // ndx, stop_val, mot_ndx, star_ptr[3]
            PINBUF(hs_f64, double, "hs_f64", 6 * nn + 8);          // cscore, sscore, rscore, uscore, tscore, mot_score
            PINBUF(hs_u8, uint8_t, "hs_u8", 10 * nn + 8);          // type, strand, edge, rbs[2], mot_len, mot_spacer, mot_spacendx
            PINBUF(hs_f32, float, "hs_f32", nn + 8);
            const bool scored = stage >= PGA_STAGE_SCORE;
            if (nn > 0) {
                HT(c, hipMemcpyAsync(hs_i32, ga[0].ndx, 4 * nn, hipMemcpyDeviceToHost, st));
                HT(c, hipMemcpyAsync(hs_i32 + nn, ga[0].stop_val, 4 * nn, hipMemcpyDeviceToHost, st));
                HT(c, hipMemcpyAsync(hs_u8, ga[0].type, nn, hipMemcpyDeviceToHost, st));
                HT(c, hipMemcpyAsync(hs_u8 + nn, ga[0].strand, nn, hipMemcpyDeviceToHost, st));
                HT(c, hipMemcpyAsync(hs_u8 + 2 * nn, scored ? ca.edge : ga[0].edge0, nn, hipMemcpyDeviceToHost, st));
                if (scored) {
                    HT(c, hipMemcpyAsync(hs_i32 + 2 * nn, ca.mot_ndx, 4 * nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_i32 + 3 * nn, ca.star_ptr, 12 * nn, hipMemcpyDeviceToHost, st));
                    double* const src64[6] = {ca.cscore, ca.sscore, ca.rscore, ca.uscore, ca.tscore, ca.mot_score};
                    for (int q = 0; q < 6; q++) HT(c, hipMemcpyAsync(hs_f64 + q * nn, src64[q], 8 * nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_u8 + 3 * nn, ca.rbs, 2 * nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_u8 + 5 * nn, ca.mot_len, nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_u8 + 6 * nn, ca.mot_spacer, nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_u8 + 7 * nn, ca.mot_spacendx, nn, hipMemcpyDeviceToHost, st));
                    HT(c, hipMemcpyAsync(hs_f32, ga[0].gc_cont, 4 * nn, hipMemcpyDeviceToHost, st));
                }
            }
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            f->last.d_dig = d_dig; f->last.ga = ga[0]; f->last.ca = ca; f->last.n_nodes = (int)nn; f->last.len = ct[0].len;
            R->nodes.resize(NC);
            for (int i = 0; i < NC; i++) {
                pga_nodes& N = R->nodes[i];
                const int64_t oo = h_cbase[i]; const int n = h_cbase[i + 1] - h_cbase[i];
                R->contigs[i].model = stage == PGA_STAGE_EXTRACT ? -1 : 0; R->contigs[i].n_nodes = n;
                if (int rc = alloc_nodes(R, N, n)) return rc;
                memcpy(N.ndx, hs_i32 + oo, 4 * (size_t)n); memcpy(N.stop_val, hs_i32 + nn + oo, 4 * (size_t)n);
                memcpy(N.type, hs_u8 + oo, n); memcpy(N.strand, hs_u8 + nn + oo, n); memcpy(N.edge, hs_u8 + 2 * nn + oo, n);
                for (int j = 0; j < n; j++) { N.traceb[j] = -1; N.tracef[j] = -1; N.ov_mark[j] = -1; }    // ref: reset_node_scores
                if (!scored) continue;
                memcpy(N.mot_ndx, hs_i32 + 2 * nn + oo, 4 * (size_t)n);
                if (stage >= PGA_STAGE_OVERLAP) memcpy(N.star_ptr, hs_i32 + 3 * nn + 3 * oo, 12 * (size_t)n);
                double* const dst64[6] = {N.cscore, N.sscore, N.rscore, N.uscore, N.tscore, N.mot_score};
                for (int q = 0; q < 6; q++) memcpy(dst64[q], hs_f64 + q * nn + oo, 8 * (size_t)n);
                memcpy(N.rbs, hs_u8 + 3 * nn + 2 * oo, 2 * (size_t)n); memcpy(N.mot_len, hs_u8 + 5 * nn + oo, n);
                memcpy(N.mot_spacer, hs_u8 + 6 * nn + oo, n); memcpy(N.mot_spacendx, hs_u8 + 7 * nn + oo, n);
                memcpy(N.gc_cont, hs_f32 + oo, 4 * (size_t)n);
            }
            return publish(R, guard.r, P, out);
        }
        PINBUF(h_segflags, int32_t, "h_segflags", (size_t)(PGA_SEG_ROUNDS + 2) * NCH + 1);      // [rounds][chains] flags, [chains] spine sizes, [chains] a round's verdict
        if (segmented && !(getenv("PGA_DP_SEG_READBACK") && atoi(getenv("PGA_DP_SEG_READBACK")) == 0)) seg_dev.h_round = h_segflags + (size_t)(PGA_SEG_ROUNDS + 1) * NCH;
        // one DP launch over the chains of every group: chains are independent, the more in flight the better
        HT(c, hipEventRecord(f->e_dp0[0], st));
        if (use_wave) {
            // PGA_DP_PROFILE=1: cycles per batch phase of every 64th chain (a synchronising debug aid)
            if (getenv("PGA_DP_PROFILE")) {
                DEVBUF(d_prof, unsigned long long, "dp_prof", 16);
                HT(c, hipMemsetAsync(d_prof, 0, 128, st));
                dp.prof = d_prof;
            }
            pga_launch_dp_wave(d_chains, NCH, wgroups, c->d_model_const, dp, wbuf, st, d_dp_order, (int)dp_order.size(), use_sched);
            if (dp.prof != nullptr) {
                unsigned long long pr[16];
                HT(c, hipStreamSynchronize(st));
                HT(c, hipMemcpy(pr, dp.prof, 128, hipMemcpyDeviceToHost));
                const double nbp = pr[7] ? (double)pr[7] : 1.0;
                fprintf(stderr, "[pga dp profile] %s, %d chains, batches sampled=%llu, cycles/batch: load=%.0f near steps=%.0f far gene ends=%.0f "
                                "carries=%.0f chains=%.0f walk=%.0f finalize=%.0f | total=%.0f\n", use_sched ? "k_dp_wave" : "k_dpw_dyn", NCH, pr[7], pr[0] / nbp, pr[1] / nbp, pr[2] / nbp, pr[3] / nbp,
                        pr[4] / nbp, pr[5] / nbp, pr[6] / nbp, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[5] + pr[6]) / nbp);
                dp.prof = nullptr;
            }
        }
        else pga_launch_dp(d_chains, NCH, c->d_model_const, dp, 1, st, segmented ? &seg_dev : nullptr);
        HT(c, hipEventRecord(f->e_dp1[0], st));
        HT(c, hipMemcpyAsync(h_maxscore, dp.max_score, sizeof(double) * 2 * (size_t)dp_slots, hipMemcpyDeviceToHost, st));       // scores, indices, path starts
        if (meta_run) HT(c, hipMemcpyAsync(h_conv, d_conv, (size_t)NG * NC, hipMemcpyDeviceToHost, st));
        if (segmented) {
            HT(c, hipMemcpyAsync(h_segflags, seg_dev.flags, sizeof(int32_t) * PGA_SEG_ROUNDS * NCH, hipMemcpyDeviceToHost, st));
            HT(c, hipMemcpyAsync(h_segflags + (size_t)PGA_SEG_ROUNDS * NCH, seg_dev.nsp, sizeof(int32_t) * NCH, hipMemcpyDeviceToHost, st));
        }
        HT(c, hipGetLastError());
        HT(c, hipStreamSynchronize(st));
        uint32_t sched_missed = 0;
        double dp_first_ms = 0.0;
        if (use_sched) {
            // a schedule that did not fit its buffer (node-dense input: more than two slots per node on average): the same launch again
            // with the kernel that works the lane masks out itself
            uint32_t missed = 0;
            for (int g = 0; g < NG; g++) if (g_c0[g + 1] > g_c0[g] && g_n0[g + 1] > g_n0[g]) missed += h_scur[2 * g + 1];
            sched_missed = missed;
            if (getenv("PGA_DPW_SCHED_DEBUG")) for (int g = 0; g < NG; g++) fprintf(stderr, "[pga dpw sched] group %d: %u batches missed\n", g, h_scur[2 * g + 1]);
            if (missed && use_wave) {      // (segments walked by the wave-batch kernel: a missed batch ends its segment's claims, the verification does the rest)
                // (the first launch's time counts: t_dp_ms covers both)
                { float ms = 0; HT(c, hipEventElapsedTime(&ms, f->e_dp0[0], f->e_dp1[0])); dp_first_ms = ms; }
                HT(c, hipEventRecord(f->e_dp0[0], st));
                pga_launch_dp_wave(d_chains, NCH, wgroups, c->d_model_const, dp, wbuf, st, d_dp_order, (int)dp_order.size(), false);
                HT(c, hipEventRecord(f->e_dp1[0], st));
                HT(c, hipMemcpyAsync(h_maxscore, dp.max_score, sizeof(double) * 2 * (size_t)dp_slots, hipMemcpyDeviceToHost, st));
                HT(c, hipGetLastError());
                HT(c, hipStreamSynchronize(st));
            }
        }
        pga_dp_note_stats(c, segmented ? &seg_plan : nullptr, h_segflags, NCH);
        c->dp_stats[6] = use_sched ? 1 : 0; c->dp_stats[7] = (int32_t)sched_missed;
        if (segmented && getenv("PGA_DP_SEG_DEBUG")) {
            fprintf(stderr, "[pga dp-seg] %d chains cut into %d segments (<= %d nodes each); nodes rejected per round: %d %d %d; walked serially: %d; spine:",
                    seg_dev.n_big, seg_dev.n_segs, seg_dev.max_seg_nodes, c->dp_stats[2], c->dp_stats[3], c->dp_stats[4], c->dp_stats[5]);
            for (int k : seg_plan.big) fprintf(stderr, " %d/%d", h_segflags[(size_t)PGA_SEG_ROUNDS * NCH + k], chains[(size_t)k].n);
            fprintf(stderr, "\n");
        }
        { float ms = 0; HT(c, hipEventElapsedTime(&ms, f->e_dp0[0], f->e_dp1[0])); R->pub.t_dp_ms = NCH > 0 ? ms + dp_first_ms : 0.0; }
        c->dp_timings[0] = R->pub.t_dp_ms; c->dp_timings[1] = c->dp_timings[2] = c->dp_timings[3] = 0.0;
        if (wave_prep) for (int g = 0; g < NG && g < 4; g++) {
            if (!(g_c0[g + 1] > g_c0[g] && g_n0[g + 1] > g_n0[g])) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, f->e_aux[4 * g], f->e_aux[4 * g + 1]) == hipSuccess) c->dp_timings[1] += ms;
            if (use_sched && hipEventElapsedTime(&ms, f->e_aux[4 * g + 2], f->e_aux[4 * g + 3]) == hipSuccess) c->dp_timings[2] += ms;
        }

        tm.mark("score+dp+sync");
        // ---- pick the winning model per contig (ref: lib.pyx:5364-5367, strict '>' from -100) ---
        std::vector<int> win_chain(NC, -1);
        {
            // the reference visits the models of a contig in model order and keeps a strictly better one: the winner is the
            // highest score, the lowest model among equals -- one pass over the chains, whatever group they sit in
            if (!P.meta) { for (int k = NCH - 1; k >= 0; k--) win_chain[chains[k].contig] = k; }
            else {
                std::vector<double> best(NC, -100.0);
                for (int k = 0; k < NCH; k++) {
                    const ChainDesc& ch = chains[k];
                    if (ch.n <= 0 || h_ipath[k] < 0) continue;
                    const int w = win_chain[ch.contig];
                    if (h_maxscore[k] > best[ch.contig] || (w >= 0 && h_maxscore[k] == best[ch.contig] && ch.model < chains[w].model)) {
                        best[ch.contig] = h_maxscore[k]; win_chain[ch.contig] = k;
                    }
                }
            }
        }
        // ---- fresh re-score of winners that were not the first model of their group (ref: lib.pyx:5380-5394)
        std::vector<ChainDesc> rescore;
        std::vector<std::vector<ChainDesc>> rs_g(NG);
        std::vector<int64_t> fin_off(NC, -1);
        int64_t rs_nodes = 0;
        for (int i = 0; i < NC; i++) {
            const int k = win_chain[i];
            if (k < 0) continue;
            // a later model of a run sees the edge flags its predecessors left (lib.pyx:2424-2434); where the contig has no start
            // node that scoring turns into an edge node, that state is the fresh one and the winning pass already is the re-score
            if (!P.meta || chains[k].first || !h_conv[(size_t)f->model_group[chains[k].model] * NC + i]) { fin_off[i] = chains[k].off; continue; }
            ChainDesc ch = chains[k]; ch.first = 1;
            ch.raw_off = chains[k].off;                 // same contig, same model: the ORF walk of the winning pass stands
            rs_g[f->model_group[ch.model]].push_back(ch);
        }
        std::vector<int> r_c0(NG + 1, 0); std::vector<int64_t> r_n0(NG + 1, 0);
        for (int g = 0; g < NG; g++) {
            r_c0[g] = (int)rescore.size(); r_n0[g] = tot_chain_nodes + rs_nodes;
            for (ChainDesc ch : rs_g[g]) { ch.off = tot_chain_nodes + rs_nodes; fin_off[ch.contig] = ch.off; rs_nodes += ch.n; rescore.push_back(ch); }
        }
        r_c0[NG] = (int)rescore.size(); r_n0[NG] = tot_chain_nodes + rs_nodes;
        if (!rescore.empty()) {
            HT(c, hipMemcpyAsync(d_chains + NCH, rescore.data(), sizeof(ChainDesc) * rescore.size(), hipMemcpyHostToDevice, st));
            int2* h_cc2 = h_cc + (size_t)NG * NC;
            for (size_t k = 0; k < rescore.size(); k++)
                h_cc2[(size_t)f->model_group[rescore[k].model] * NC + rescore[k].contig] = make_int2(NCH + (int)k, 1);
            HT(c, hipMemcpyAsync(d_cc + (size_t)NG * NC, h_cc2, sizeof(int2) * (size_t)NG * NC, hipMemcpyHostToDevice, st));
            sp.conv_flag = nullptr; sp.cs_out = nullptr;
            for (int g = 0; g < NG; g++) {
                const int nch = r_c0[g + 1] - r_c0[g]; const int64_t nn = r_n0[g + 1] - r_n0[g];
                if (nch == 0 || nn == 0) continue;
                pga_launch_score(d_chains + NCH + r_c0[g], nch, r_n0[g], nn, d_dig, d_ct, ga[g], d_models, f->d_msc, c->d_model_const, ca, sp,
                                 d_chains, d_cc + (size_t)NG * NC + (size_t)g * NC, d_cbase + (size_t)g * (NC + 1), NC, (int)group_nodes[g], f->d_sd_lut, st, 1);
            }
        }
        tm.mark("winners+rescore_launch");
        // ---- gather the winners and bring them home ------------------------------------------------
        // the winners' final-pass fields only travel to the gathered arrays when somebody reads more of them than the two nodes
        // of every gene: the caller (node arrays) or the host tail's attribute fetch (PGA_FULL_GATHER=1 keeps the full gather: tests)
        std::vector<std::vector<WinDesc>> wg(NG);
        std::vector<int64_t> out_off(NC, 0);
        int64_t out_nodes = 0;
        std::vector<int64_t> w_o0(NG + 1, 0);
        // output order: group-major then contig, so each group's launch covers a contiguous output range
        for (int g = 0; g < NG; g++) {
            w_o0[g] = out_nodes;
            for (int i = 0; i < NC; i++) {
                const int k = win_chain[i];
                if (k < 0 || (P.meta ? f->model_group[chains[k].model] : 0) != g) continue;
                out_off[i] = out_nodes;
                wg[g].push_back(WinDesc{out_nodes, chains[k].off, fin_off[i], chains[k].topo_off, chains[k].n, 0});
                out_nodes += chains[k].n;
            }
        }
        w_o0[NG] = out_nodes;
        // one device arena (and a pinned mirror, used only when the caller asks for the node arrays)
        OutArrays o, h;
        size_t arena_dp = 0, arena_all = 0;
        {
            const size_t n = (size_t)out_nodes + 1;
            size_t off = 0;
            auto reserve = [&](size_t elem, size_t mult) { const size_t at = off; off += ((elem * mult * n + 255) / 256) * 256; return at; };
#define OFFS(field, type, mult) const size_t at_##field = reserve(sizeof(type), mult);
            OFFS(ndx, int32_t, 1) OFFS(stop_val, int32_t, 1) OFFS(type, uint8_t, 1) OFFS(strand, int8_t, 1)
            OFFS(edge_dp, uint8_t, 1) OFFS(cscore_dp, double, 1) OFFS(sscore_dp, double, 1) OFFS(rscore_dp, double, 1) OFFS(uscore_dp, double, 1) OFFS(tscore_dp, double, 1)
            OFFS(star_ptr, int32_t, 3) OFFS(traceb, int32_t, 1) OFFS(ov_mark, int8_t, 1) OFFS(score, double, 1)
            arena_dp = off;
            OFFS(gc_cont, float, 1)
            OFFS(edge, uint8_t, 1) OFFS(cscore, double, 1) OFFS(sscore, double, 1) OFFS(rscore, double, 1) OFFS(uscore, double, 1) OFFS(tscore, double, 1) OFFS(mot_score, double, 1)
            OFFS(mot_ndx, int32_t, 1) OFFS(rbs, uint8_t, 2) OFFS(mot_len, uint8_t, 1) OFFS(mot_spacer, uint8_t, 1) OFFS(mot_spacendx, uint8_t, 1)
            arena_all = off;
            void* dp__; void* hp__;
            { int rc__ = ensure_dev(c, "o_arena", arena_all + 256, &dp__); if (rc__) return rc__; }
            { int rc__ = ensure_pin(c, "h_arena", arena_all + 256, &hp__); if (rc__) return rc__; }
            char* db = (char*)dp__; char* hb = (char*)hp__;
#define OB(field, type) o.field = (type*)(db + at_##field); h.field = (type*)(hb + at_##field);
            OB(ndx, int32_t) OB(stop_val, int32_t) OB(type, uint8_t) OB(strand, int8_t) OB(gc_cont, float)
            OB(edge_dp, uint8_t) OB(cscore_dp, double) OB(sscore_dp, double) OB(rscore_dp, double) OB(uscore_dp, double) OB(tscore_dp, double)
            OB(star_ptr, int32_t) OB(traceb, int32_t) OB(ov_mark, int8_t) OB(score, double)
            OB(edge, uint8_t) OB(cscore, double) OB(sscore, double) OB(rscore, double) OB(uscore, double) OB(tscore, double) OB(mot_score, double)
            OB(mot_ndx, int32_t) OB(rbs, uint8_t) OB(mot_len, uint8_t) OB(mot_spacer, uint8_t) OB(mot_spacendx, uint8_t)
        }
        // lean and the device tail: gather only what the tail writes
        o.direct = direct_gather ? 1 : 0;
        for (int g = 0; g < 4; g++) {
            const bool in = g < NG;
            o.g_ndx[g] = in ? ga[g].ndx : nullptr; o.g_stop_val[g] = in ? ga[g].stop_val : nullptr;
            o.g_type[g] = in ? ga[g].type : nullptr; o.g_strand[g] = in ? ga[g].strand : nullptr;
        }
        o.c_edge = ca.edge; o.c_cscore = ca.cscore; o.c_rscore = ca.rscore; o.c_uscore = ca.uscore; o.c_tscore = ca.tscore;
        o.c_star_ptr = ca.star_ptr; o.d_score = dp.score;
        h.direct = 0;
        size_t nwin = 0; for (int g = 0; g < NG; g++) nwin += wg[g].size();
        DEVBUF(d_win, WinDesc, "d_win", nwin + 1);
        {
            size_t k0 = 0;
            for (int g = 0; g < NG; g++) {
                if (wg[g].empty()) continue;
                HT(c, hipMemcpyAsync(d_win + k0, wg[g].data(), sizeof(WinDesc) * wg[g].size(), hipMemcpyHostToDevice, st));
                const int64_t nn = w_o0[g + 1] - w_o0[g];
                if (nn > 0)
                    hipLaunchKernelGGL(k_gather_winners, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, d_win + k0, (int)wg[g].size(), w_o0[g], nn, ga[g], ca, dp, o, lean_gather ? 1 : 0);
                k0 += wg[g].size();
            }
        }
        // The tail (traceback untangling, bad-gene elimination, gene list, start tweaks) is a pointer chase per
        // contig.  Many small contigs: one device thread each, only the genes come home.  Few or very long contigs:
        // a single device thread would crawl (one memory round trip per step), so their DP-pass fields come home
        // and host threads walk them.
        int max_n = 0;
        for (int i = 0; i < NC; i++) if (win_chain[i] >= 0) max_n = std::max(max_n, chains[win_chain[i]].n);
        // PGA_TAIL = host | device (one thread per contig) | par (tail.inl, the default)
        const char* tail_env = getenv("PGA_TAIL");
        const bool device_tail = tail_env ? strcmp(tail_env, "host") != 0 : true;
        const bool par_tail = device_tail && !(tail_env && strcmp(tail_env, "device") == 0);
        std::vector<int32_t> tracef;
        std::vector<uint8_t> elim;
        if (!device_tail) {
            tracef.assign((size_t)out_nodes + 1, -1);
            elim.assign((size_t)out_nodes + 1, 0);
            if (out_nodes > 0)
                HT(c, hipMemcpyAsync(h.ndx, o.ndx, P.want_nodes ? arena_all : arena_dp, hipMemcpyDeviceToHost, st));   // arena starts at `ndx`
            HT(c, hipEventRecord(f->e_stop, st));
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            { float ms = 0; HT(c, hipEventElapsedTime(&ms, f->e_start, f->e_stop)); R->pub.t_total_ms = ms; }

            tm.mark("gather+d2h+sync");
            // ---- host tail per contig ------------------------------------------------------------------
            std::vector<std::vector<GeneRec>> cg(NC);
            std::unique_ptr<int32_t[]> pathbuf(new int32_t[(size_t)out_nodes + 1]);      // written before it is read: no zero fill
            std::atomic<int> next(0);
            auto worker = [&]() {
                for (;;) {
                    const int i = next.fetch_add(1);
                    if (i >= NC) break;
                    const int k = win_chain[i];
                    if (k < 0) continue;
                    const int64_t oo = out_off[i];
                    NodeView v{chains[k].n, h.ndx + oo, h.stop_val + oo, h.type + oo, h.strand + oo, h.edge_dp + oo,
                               h.cscore_dp + oo, h.sscore_dp + oo, h.rscore_dp + oo, h.uscore_dp + oo, h.tscore_dp + oo,
                               h.star_ptr + 3 * oo, h.traceb + oo, tracef.data() + oo, h.ov_mark + oo, h.score + oo, elim.data() + oo};
                    const int mx = h_maxidx[k];
                    const double st_wt = c->models[chains[k].model].st_wt;
                    const bool sub = NC == 1 && getenv("PGA_TIMING");
                    auto now = [] { return std::chrono::steady_clock::now(); };
                    auto t0 = now(), t1 = t0, t2 = t0, t3 = t0;
                    if (v.n > 0 && mx >= 0) {
                        int32_t* pl = pathbuf.get() + oo;
                        const int cnt = untangle(v, mx, pl);
                        t1 = now();
                        if (v.traceb[mx] != -1) {
                            if (NC < 4) eliminate_bad_genes_mt(v, pl, cnt, st_wt, f->pool, 16); else eliminate_bad_genes(v, pl, cnt, st_wt);
                            t2 = now();
                            cg[i].resize((size_t)cnt / 2 + 2);           // two path nodes per gene
                            cg[i].resize((size_t)extract_genes(v, pl, cnt, cg[i].data()));
                        }
                    }
                    t3 = now();
                    tweak_final_starts(v, cg[i], st_wt, P.max_overlap, NC < 4 ? 32 : 1, &f->pool);
                    if (sub) {
                        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                        fprintf(stderr, "[pga timing] host tail: untangle=%.2f eliminate=%.2f extract=%.2f tweak=%.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
                    }
                }
            };
            auto run_parallel = [&](const std::function<void()>& fn) {
                int nt = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
                if (NC < 4) nt = 1;
                if (nt == 1) fn(); else f->pool.run(fn, nt);
            };
            run_parallel(worker);
            tm.mark("host_tail");
            // ---- results -----------------------------------------------------------------------------------
            int64_t ngenes = 0;
            for (int i = 0; i < NC; i++) {
                pga_contig_result& cr = R->contigs[i];
                const int k = win_chain[i];
                cr.gene_begin = ngenes; cr.n_genes = (int32_t)cg[i].size();
                ngenes += cr.n_genes;
                if (k < 0) { cr.model = -1; continue; }
                cr.model = chains[k].model; cr.n_nodes = chains[k].n;
                cr.score = P.meta ? 0.0 : (h_ipath[k] >= 0 ? h_maxscore[k] : 0.0);
            }
            R->genes.resize((size_t)ngenes);
            // fields of the gene's start / stop nodes as of the final pass: a second, small gather
            PINBUF(h_gidx, int64_t, "h_gidx", 2 * ngenes + 1);
            PINBUF(h_attr, GeneNodeAttr, "h_attr", 2 * ngenes + 1);
            if (ngenes > 0) {
                DEVBUF(d_gidx, int64_t, "d_gidx", 2 * ngenes + 1);
                DEVBUF(d_attr, GeneNodeAttr, "d_attr", 2 * ngenes + 1);
                for (int i = 0; i < NC; i++) {
                    int64_t gi = R->contigs[i].gene_begin;
                    for (const GeneRec& gr : cg[i]) { h_gidx[2 * gi] = out_off[i] + gr.start_ndx; h_gidx[2 * gi + 1] = out_off[i] + gr.stop_ndx; gi++; }
                }
                HT(c, hipMemcpyAsync(d_gidx, h_gidx, sizeof(int64_t) * 2 * ngenes, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_gather_gene_nodes, dim3((unsigned)((2 * ngenes + 255) / 256)), dim3(256), 0, st, d_gidx, (int)(2 * ngenes), o, d_attr);
                HT(c, hipMemcpyAsync(h_attr, d_attr, sizeof(GeneNodeAttr) * 2 * ngenes, hipMemcpyDeviceToHost, st));
                HT(c, hipGetLastError());
                HT(c, hipStreamSynchronize(st));
            }
            tm.mark("gene_attrs");
            std::atomic<int> next2(0);
            auto filler = [&]() {
                for (;;) {
                    const int i = next2.fetch_add(1);
                    if (i >= NC) break;
                    const int k = win_chain[i];
                    if (k < 0) continue;
                    const int64_t oo = out_off[i];
                    const bool single = !P.meta;
                    int64_t gi = R->contigs[i].gene_begin;
                    for (const GeneRec& gr : cg[i]) {
                        pga_gene& G = R->genes[(size_t)gi];
                        const GeneNodeAttr& as = h_attr[2 * gi]; const GeneNodeAttr& ae = h_attr[2 * gi + 1];
                        gi++;
                        memset(&G, 0, sizeof G);
                        const int64_t sn = oo + gr.start_ndx, en = oo + gr.stop_ndx;
                        G.contig = i; G.begin = gr.begin; G.end = gr.end; G.start_ndx = gr.start_ndx; G.stop_ndx = gr.stop_ndx;
                        G.strand = h.strand[sn];
                        // single mode keeps the nodes of the DP pass (ref: lib.pyx:5296-5311); meta mode re-scores (5380-5394)
                        const uint8_t se = single ? h.edge_dp[sn] : as.edge, ee = single ? h.edge_dp[en] : ae.edge;
                        G.partial_begin = G.strand == 1 ? se : ee; G.partial_end = G.strand == 1 ? ee : se;
                        G.start_type = se ? 3 : h.type[sn];
                        G.rbs[0] = as.rbs0; G.rbs[1] = as.rbs1;
                        G.mot_len = as.mot_len; G.mot_spacer = as.mot_spacer; G.mot_ndx = as.mot_ndx; G.mot_score = as.mot_score;
                        G.gc_cont = as.gc_cont;
                        G.cscore = single ? h.cscore_dp[sn] : as.cscore; G.sscore = single ? h.sscore_dp[sn] : as.sscore;
                        G.rscore = single ? h.rscore_dp[sn] : as.rscore; G.uscore = single ? h.uscore_dp[sn] : as.uscore;
                        G.tscore = single ? h.tscore_dp[sn] : as.tscore;
                    }
                }
            };
            run_parallel(filler);
        } else {
            tm.mark("gather");
            // ---- tail on the device: traceback untangling, bad-gene elimination, gene list, start tweaks ----------
            std::vector<TailDesc> tdv(NC);
            int64_t n_slots = 0;
            for (int i = 0; i < NC; i++) {
                const int k = win_chain[i];
                TailDesc& d = tdv[i];
                d.out_off = k >= 0 ? out_off[i] : 0; d.gene_off = n_slots;
                d.n = k >= 0 ? chains[k].n : 0; d.mx = k >= 0 ? h_maxidx[k] : -1;
                d.st_wt = k >= 0 ? c->models[chains[k].model].st_wt : 0.0;
                d.fin_off = k >= 0 ? fin_off[i] : 0; d.topo_off = k >= 0 ? chains[k].topo_off : 0;
                d.group = k >= 0 && P.meta ? f->model_group[chains[k].model] : 0; d._pad = 0;
                d.dp_off = k >= 0 ? chains[k].off : 0;
                n_slots += d.n / 2 + 2;
            }
            DEVBUF(d_td, TailDesc, "d_taildesc", NC + 1);
            DEVBUF(d_tracef, int32_t, "d_tracef", out_nodes + 1);
            DEVBUF(d_elim, uint8_t, "d_elim", out_nodes + 1);
            DEVBUF(d_gene0, GeneRec, "d_gene0", n_slots + 1);
            DEVBUF(d_gene1, GeneRec, "d_gene1", n_slots + 1);
            DEVBUF(d_ngenes, int32_t, "d_ngenes", NC + 1);
            DEVBUF(d_gbegin, int64_t, "d_gbegin", NC + 1);
            PINBUF(h_ngenes, int32_t, "h_ngenes", NC + 1);
            PINBUF(h_gbegin, int64_t, "h_gbegin", NC + 1);
            HT(c, hipMemcpyAsync(d_td, tdv.data(), sizeof(TailDesc) * NC, hipMemcpyHostToDevice, st));
            DEVBUF(d_path, int32_t, "d_path", out_nodes + 1);
            DEVBUF(d_changed, uint8_t, "d_changed", n_slots + 1);
            DEVBUF(d_nchanged, int32_t, "d_nchanged", NC + 1);
            const unsigned init_blocks = (unsigned)std::min<int64_t>(2048, (std::max<int64_t>(out_nodes, NC) + 256) / 256);
            if (!par_tail) {
                hipLaunchKernelGGL(k_tail_init, dim3(init_blocks), dim3(256), 0, st, d_tracef, d_elim, out_nodes, d_nchanged, (int32_t*)nullptr, NC, (int32_t*)nullptr, 0);
                hipLaunchKernelGGL(k_tail_path, dim3((NC + 63) / 64), dim3(64), 0, st, d_td, NC, o, d_tracef, d_elim, d_path, d_gene0, d_ngenes);
            } else {
                std::vector<TpSeg> segs;
                bool tail_inited = false;
                for (int i = 0; i < NC; i++) if (win_chain[i] >= 0 && chains[win_chain[i]].n > 0) segs.push_back(TpSeg{out_off[i], chains[win_chain[i]].n, i});
                std::sort(segs.begin(), segs.end(), [](const TpSeg& a, const TpSeg& b) { return a.off < b.off; });
                if (!segs.empty() && out_nodes > 0) {
                    if (out_nodes >= (1ll << 30)) { c->err = "pga_find_genes: more than 2^30 nodes in the winning chains of one batch"; return PGA_EINVAL; }
                    int levels = 1;
                    while ((1ll << levels) < max_n) levels++;
                    const unsigned nblk = (unsigned)((out_nodes + 255) / 256);
                    const bool tp_steps = getenv("PGA_TP_STEPS") != nullptr;      // the many-launch form also for short chains (cross-check)
                    const bool tp_one = max_n <= TP_SMALL_MAX && !tp_steps;
                    const int64_t tpn = tp_one ? 0 : out_nodes;                   // the per-node work arrays of the many-launch form
                    DEVBUF(tp_seg, TpSeg, "tp_seg", segs.size()) DEVBUF(tp_up, int32_t, "tp_up", (size_t)2 * tpn)
                    DEVBUF(tp_mark, uint8_t, "tp_mark", tpn) DEVBUF(tp_slots, int32_t, "tp_slots", tpn)
                    DEVBUF(tp_ins, int32_t, "tp_ins", 2 * tpn) DEVBUF(tp_excl, int32_t, "tp_excl", tpn + 1)
                    DEVBUF(tp_bsum, int32_t, "tp_bsum", nblk + 1) DEVBUF(tp_cnt, int32_t, "tp_cnt", segs.size())
                    HT(c, hipMemcpyAsync(tp_seg, segs.data(), sizeof(TpSeg) * segs.size(), hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(k_tail_init, dim3(init_blocks), dim3(256), 0, st, d_tracef, d_elim, out_nodes, d_nchanged, d_ngenes, NC,
                                       tp_one ? (int32_t*)nullptr : tp_cnt, (int)segs.size());
                    tail_inited = true;
                    const TpWork tw{tp_seg, (int)segs.size(), out_nodes, levels, tp_up, tp_mark, tp_slots, tp_ins, tp_excl, tp_bsum, tp_cnt};
                    const dim3 grid(nblk), blk(256);
                    if (tp_one) {
                        const int cap = (int)((max_n + 63) & ~63ll);
                        hipLaunchKernelGGL(k_tp_small, dim3((unsigned)segs.size()), dim3(max_n <= 2048 ? 64 : 256), sizeof(int32_t) * 3 * (size_t)cap, st, tw, d_td, o,
                                           d_tracef, d_elim, d_gene0, d_ngenes, cap);
                    } else {
                    hipLaunchKernelGGL(k_tp_init, grid, blk, 0, st, tw, d_td, o);
                    // pointer jumping: 2^levels >= the longest chain; eight hops per launch while three levels or more are left (PGA_TP_JUMP8=0: doubling only)
                    {
                        const bool j8 = !(getenv("PGA_TP_JUMP8") && atoi(getenv("PGA_TP_JUMP8")) == 0);
                        int k = 0;
                        for (int left = levels; left > 0; k++) {
                            const int32_t* in = tp_up + (size_t)(k & 1) * out_nodes; int32_t* outp = tp_up + (size_t)((k + 1) & 1) * out_nodes;
                            if (j8 && left >= 3) { hipLaunchKernelGGL(k_tp_jump8, grid, blk, 0, st, tw, in, outp); left -= 3; }
                            else { hipLaunchKernelGGL(k_tp_jump, grid, blk, 0, st, tw, in, outp); left -= 1; }
                        }
                    }
                    hipLaunchKernelGGL(k_tp_slots, grid, blk, 0, st, tw, d_td, o);
                    hipLaunchKernelGGL(k_tp_scan1, grid, blk, 0, st, tw);
                    hipLaunchKernelGGL(k_tp_scan2, dim3(1), dim3(1024), 0, st, tw, (int)nblk);
                    hipLaunchKernelGGL(k_tp_scan3, grid, blk, 0, st, tw);
                    hipLaunchKernelGGL(k_tp_fill, grid, blk, 0, st, tw, o, d_path);
                    hipLaunchKernelGGL(k_tp_link, grid, blk, 0, st, tw, o, d_path, d_tracef);
                    hipLaunchKernelGGL(k_tp_elim_a, grid, blk, 0, st, tw, d_td, o, d_path);
                    hipLaunchKernelGGL(k_tp_elim_b, grid, blk, 0, st, tw, d_td, o, d_path, d_elim);
                    // a workgroup per contig: wide for genomes, narrow when there are many short paths
                    if (max_n >= 32768 && segs.size() <= 64 && !getenv("PGA_TP_EXTRACT_ONE")) {
                        // genomes: a workgroup per chunk of 1024 path positions (k_tp_extract: one workgroup per contig, 400 rounds)
                        const int chunks = (max_n + 1023) / 1024;
                        DEVBUF(tp_xsum, TpChunkSum, "tp_xsum", (size_t)chunks * segs.size() + 1);
                        const dim3 xg((unsigned)chunks, (unsigned)segs.size());
                        hipLaunchKernelGGL(k_tp_extract_sum, xg, dim3(1024), 0, st, tw, d_td, o, d_path, d_elim, tp_xsum, chunks);
                        hipLaunchKernelGGL(k_tp_extract_chunk, xg, dim3(1024), 0, st, tw, d_td, o, d_path, d_elim, (const TpChunkSum*)tp_xsum, chunks, d_gene0,
                                           d_ngenes);
                    } else
                    hipLaunchKernelGGL(k_tp_extract, dim3((unsigned)segs.size()), dim3(max_n >= 32768 ? 1024 : 256), 0, st, tw, d_td, o, d_path, d_elim,
                                       d_gene0, d_ngenes);
                    }
                }
                // (no winning chain at all: the counters the tweaks read still start from zero)
                if (!tail_inited) hipLaunchKernelGGL(k_tail_init, dim3(init_blocks), dim3(256), 0, st, d_tracef, d_elim, out_nodes, d_nchanged, d_ngenes, NC, (int32_t*)nullptr, 0);
            }
            if (max_n >= 32768 && NC <= 1024)       // genomes: a wavefront per gene (at most n / 2 + 2 genes per contig)
                hipLaunchKernelGGL(k_tail_tweak_wave, dim3((unsigned)((max_n / 2 + 2 + 3) / 4), (unsigned)NC), dim3(256), 0, st, d_td, NC, o, d_tracef,
                                   d_elim, d_gene0, d_gene1, d_ngenes, P.max_overlap, d_changed, d_nchanged);
            else if (NC >= 256 && !getenv("PGA_TWEAK_SLOTS")) {   // many contigs: one thread per gene of the batch, packed
                DEVBUF(d_gpre, int32_t, "d_gene_prefix", NC + 2);
                hipLaunchKernelGGL(k_gene_prefix, dim3(1), dim3(1024), 0, st, d_ngenes, NC, d_gpre);
                hipLaunchKernelGGL(k_tail_tweak_packed, dim3(1024), dim3(256), 0, st, d_td, NC, o, d_tracef, d_elim, d_gene0, d_gene1, d_ngenes,
                                   P.max_overlap, d_changed, d_nchanged, d_gpre);
            } else
                hipLaunchKernelGGL(k_tail_tweak, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, d_td, NC, n_slots, o, d_tracef, d_elim,
                                   d_gene0, d_gene1, d_ngenes, P.max_overlap, d_changed, d_nchanged);
            DEVBUF(d_gene2, GeneRec, "d_gene2", n_slots + 1);
            DEVBUF(d_chfin, uint8_t, "d_chfin", n_slots + 1);
            hipLaunchKernelGGL(k_tail_tweak_fixup, dim3(NC), dim3(max_n >= 32768 ? 1024 : 128), 0, st, d_td, NC, o, d_tracef, d_elim, d_gene0, d_gene1,
                               d_ngenes, P.max_overlap, d_changed, d_nchanged, d_gene2, d_chfin);
            HT(c, hipMemcpyAsync(h_ngenes, d_ngenes, sizeof(int32_t) * NC, hipMemcpyDeviceToHost, st));
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            tm.mark("tail");
            // ---- results --------------------------------------------------------------------------------------
            int64_t ngenes = 0;
            for (int i = 0; i < NC; i++) {
                pga_contig_result& cr = R->contigs[i];
                const int k = win_chain[i];
                h_gbegin[i] = ngenes;
                cr.gene_begin = ngenes; cr.n_genes = h_ngenes[i];
                ngenes += cr.n_genes;
                if (k < 0) { cr.model = -1; continue; }
                cr.model = chains[k].model; cr.n_nodes = chains[k].n;
                cr.score = P.meta ? 0.0 : (h_ipath[k] >= 0 ? h_maxscore[k] : 0.0);
            }
            pga_gene* const genes_out = R->gene_records((size_t)ngenes);
            if (ngenes > 0) {
                DEVBUF(d_genes, pga_gene, "d_genes_out", ngenes + 1);
                HT(c, hipMemcpyAsync(d_gbegin, h_gbegin, sizeof(int64_t) * NC, hipMemcpyHostToDevice, st));
                GcPtrs gcs{};
                for (int g = 0; g < NG; g++) gcs.p[g] = ga[g].gc_cont;
                hipLaunchKernelGGL(k_emit_genes, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, d_td, NC, n_slots, o, d_gene2, d_ngenes,
                                   d_gbegin, P.meta ? 0 : 1, d_genes, lean_gather ? 1 : 0, ca, gcs);
                HT(c, hipMemcpyAsync(genes_out, d_genes, sizeof(pga_gene) * (size_t)ngenes, hipMemcpyDeviceToHost, st));
            }
            if (P.want_nodes && out_nodes > 0) {
                tracef.resize((size_t)out_nodes + 1); elim.resize((size_t)out_nodes + 1);
                HT(c, hipMemcpyAsync(h.ndx, o.ndx, arena_all, hipMemcpyDeviceToHost, st));   // arena starts at `ndx`
                HT(c, hipMemcpyAsync(tracef.data(), d_tracef, sizeof(int32_t) * (size_t)out_nodes, hipMemcpyDeviceToHost, st));
                HT(c, hipMemcpyAsync(elim.data(), d_elim, (size_t)out_nodes, hipMemcpyDeviceToHost, st));
            }
            HT(c, hipEventRecord(f->e_stop, st));
            HT(c, hipGetLastError());
            HT(c, hipStreamSynchronize(st));
            { float ms = 0; HT(c, hipEventElapsedTime(&ms, f->e_start, f->e_stop)); R->pub.t_total_ms = ms; }
            tm.mark("genes+d2h");

        }
        if (P.want_nodes) {
            R->nodes.resize(NC);
            for (int i = 0; i < NC; i++) {
                pga_nodes& N = R->nodes[i];
                memset(&N, 0, sizeof N);
                const int k = win_chain[i];
                if (k < 0) continue;
                const int n = chains[k].n; const int64_t oo = out_off[i];
                const bool single = !P.meta;
                if (int rc = alloc_nodes(R, N, n)) return rc;
                memcpy(N.ndx, h.ndx + oo, 4 * n); memcpy(N.stop_val, h.stop_val + oo, 4 * n); memcpy(N.type, h.type + oo, n); memcpy(N.strand, h.strand + oo, n);
                memcpy(N.gc_cont, h.gc_cont + oo, 4 * n); memcpy(N.rbs, h.rbs + 2 * oo, 2 * n); memcpy(N.mot_ndx, h.mot_ndx + oo, 4 * n);
                memcpy(N.mot_len, h.mot_len + oo, n); memcpy(N.mot_spacer, h.mot_spacer + oo, n); memcpy(N.mot_spacendx, h.mot_spacendx + oo, n);
                memcpy(N.mot_score, h.mot_score + oo, 8 * n);
                if (single) {   // nodes as left by the DP + eliminate_bad_genes (ref: lib.pyx:5296-5311)
                    memcpy(N.edge, h.edge_dp + oo, n); memcpy(N.cscore, h.cscore_dp + oo, 8 * n); memcpy(N.sscore, h.sscore_dp + oo, 8 * n);
                    memcpy(N.rscore, h.rscore_dp + oo, 8 * n); memcpy(N.uscore, h.uscore_dp + oo, 8 * n); memcpy(N.tscore, h.tscore_dp + oo, 8 * n);
                    memcpy(N.score, h.score + oo, 8 * n); memcpy(N.traceb, h.traceb + oo, 4 * n); memcpy(N.tracef, tracef.data() + oo, 4 * n);
                    memcpy(N.ov_mark, h.ov_mark + oo, n); memcpy(N.elim, elim.data() + oo, n); memcpy(N.star_ptr, h.star_ptr + 3 * oo, 12 * n);
                } else {        // fresh re-score, DP fields reset (ref: lib.pyx:5380-5394)
                    memcpy(N.edge, h.edge + oo, n); memcpy(N.cscore, h.cscore + oo, 8 * n); memcpy(N.sscore, h.sscore + oo, 8 * n);
                    memcpy(N.rscore, h.rscore + oo, 8 * n); memcpy(N.uscore, h.uscore + oo, 8 * n); memcpy(N.tscore, h.tscore + oo, 8 * n);
                    for (int j = 0; j < n; j++) { N.traceb[j] = -1; N.tracef[j] = -1; N.ov_mark[j] = -1; }
                }
            }
        }
    }
    tm.mark("results");
    return publish(R, guard.r, P, out);
}

#include "train.inl"

extern "C" int pga_find_genes(pga_ctx* c, const pga_batch* batch, const pga_params* pp, pga_result** out) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = find_impl(c, batch, pp, 0, 0, out);
    if (getenv("PGA_TIMING")) fprintf(stderr, "[pga timing] pga_find_genes wall=%.2fms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

extern "C" int pga_nodes_stage(pga_ctx* c, const pga_batch* batch, const pga_params* pp, int stage, int translation_table, pga_result** out) {
    if (stage < PGA_STAGE_EXTRACT || stage > PGA_STAGE_SEQUENCE) {
        if (out) *out = nullptr;
        if (c) c->err = "pga_nodes_stage: unknown stage";
        return PGA_EINVAL;
    }
    return find_impl(c, batch, pp, stage, translation_table, out);
}

extern "C" int pga_train(pga_ctx* c, const pga_batch* batch, const pga_params* pp, int translation_table, double start_weight,
                         int force_nonsd, int upto, pga_training* out) {
    return train_impl(c, batch, pp, translation_table, start_weight, force_nonsd, upto <= 0 ? TR_ALL : upto, out);
}
