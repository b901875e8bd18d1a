// C-ABI entry points (include/pyrodigal_amd.h): context, models, scorer-level call.
// Host code only; no CPU compute path exists here -- without a gfx950 device every call fails.
#include "pga_internal.h"
#include <malloc.h>
#include <mutex>
#include "dpw_core.h"

#include <algorithm>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <new>

static int fail(pga_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                           \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return fail(ctx, e__ == hipErrorOutOfMemory ? PGA_ENOMEM : PGA_EDEVICE, "%s failed: %s", \
                        #expr, hipGetErrorString(e__));                                              \
    } while (0)

int pga_hip_try_(pga_ctx* c, hipError_t e, const char* what) {
    if (e == hipSuccess) return PGA_OK;
    return fail(c, e == hipErrorOutOfMemory ? PGA_ENOMEM : PGA_EDEVICE, "%s failed: %s", what, hipGetErrorString(e));
}

extern "C" int pga_create(int device, pga_ctx** out) {
    if (!out) return PGA_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PGA_ENODEVICE;
    if (device < 0 || device >= count) return PGA_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PGA_EDEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PGA_ENODEVICE;   // this library is built for CDNA4 only
    {
        // The host side of a large call plans in megabyte-sized std::vectors (26 000 chain descriptors are 2.3 MB) that live for one call.
        // glibc serves those from mmap and gives them back with munmap -- page faults on the way in, a system call on the way out, about
        // 0.6 ms of a 9 ms call -- unless its thresholds say otherwise; they did by accident while every call also freed a 16 MB vector of
        // gene records (round 6: those records now sit in pinned memory).  Once per process; PGA_MALLOPT=0 leaves the allocator alone.
        static std::once_flag once;
        std::call_once(once, [] {
            const char* e = getenv("PGA_MALLOPT");
            if (e && atoi(e) == 0) return;
            mallopt(M_MMAP_THRESHOLD, 32 << 20);
            mallopt(M_TRIM_THRESHOLD, 512 << 20);
        });
    }
    pga_ctx* c = new (std::nothrow) pga_ctx();
    if (!c) return PGA_ENOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return PGA_EDEVICE;
    }
    *out = c;
    return PGA_OK;
}

extern "C" void pga_destroy(pga_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    pga_finder_release(c);
    if (c->d_models_raw) hipFree(c->d_models_raw);
    if (c->d_model_const) hipFree(c->d_model_const);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* pga_last_error(const pga_ctx* c) { return c ? c->err.c_str() : "no context"; }

extern "C" int pga_dp_stats(const pga_ctx* c, int32_t out[8]) {
    if (!c || !out) return PGA_EINVAL;
    memcpy(out, c->dp_stats, sizeof c->dp_stats);
    return PGA_OK;
}

extern "C" int pga_dp_timings(const pga_ctx* c, double out[4]) {
    if (!c || !out) return PGA_EINVAL;
    memcpy(out, c->dp_timings, sizeof c->dp_timings);
    return PGA_OK;
}

extern "C" int pga_extract_stats(const pga_ctx* c, int32_t out[2]) {
    if (!c || !out) return PGA_EINVAL;
    out[0] = c->extract_passes; out[1] = 0;
    return PGA_OK;
}

void pga_dp_note_stats(pga_ctx* c, const DpSegPlan* plan, const int32_t* h_flags, int stride) {
    memset(c->dp_stats, 0, sizeof c->dp_stats);
    if (!plan || plan->segs.empty()) return;
    c->dp_stats[0] = (int32_t)plan->big.size(); c->dp_stats[1] = (int32_t)plan->segs.size();
    for (int r = 0; r < PGA_SEG_ROUNDS; r++) {
        long long tot = 0; int left = 0;
        for (int k : plan->big) { tot += h_flags[(size_t)r * stride + k]; left += h_flags[(size_t)r * stride + k] != 0; }
        c->dp_stats[2 + r] = (int32_t)std::min<long long>(tot, 0x7fffffff);
        if (r == PGA_SEG_ROUNDS - 1) c->dp_stats[5] = left;
    }
}

extern "C" int pga_device_info(const pga_ctx* c, char* name, int name_len, int* cus, int64_t* hbm_bytes) {
    if (!c) return PGA_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return PGA_EDEVICE;
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PGA_OK;
}

void pga_fill_model_const(ModelConst* mc, double st_wt) {
    memset(mc, 0, sizeof *mc);
    mc->st_wt = st_wt;
    mc->negc = -0.15 * st_wt;                                               // ref: _connection.h:43-49
    for (int d = 0; d <= PGA_OPER_DIST; d++)
        mc->igm[d] = (2.0 - ((double)d / PGA_OPER_DIST)) * 0.15 * st_wt;    // ref: _connection.h:73-75
}

extern "C" int pga_set_models(pga_ctx* c, const pga_training* const* models, int n_models) {
    if (!c || n_models < 0 || (n_models > 0 && !models)) return fail(c, PGA_EINVAL, "pga_set_models: bad arguments");
    // validate before touching the context: a rejected call leaves the loaded model set as it was
    {
        int tts[5]; int ntt = 0;
        for (int i = 0; i < n_models; i++) {
            if (!models[i]) return fail(c, PGA_EINVAL, "pga_set_models: model %d is NULL", i);
            bool seen = false;
            for (int k = 0; k < ntt; k++) seen = seen || tts[k] == models[i]->trans_table;
            if (!seen) { if (ntt == 4) return fail(c, PGA_EINVAL, "pga_set_models: more than 4 distinct translation tables"); tts[ntt++] = models[i]->trans_table; }
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_models_raw) { hipFree(c->d_models_raw); c->d_models_raw = nullptr; }
    if (c->d_model_const) { hipFree(c->d_model_const); c->d_model_const = nullptr; }
    c->models.clear();
    c->n_models = 0;
    if (n_models == 0) return pga_finder_models_changed(c);
    std::vector<ModelConst> mcs(n_models);
    for (int i = 0; i < n_models; i++) {
        c->models.push_back(*models[i]);
        pga_fill_model_const(&mcs[i], models[i]->st_wt);
    }
    HIP_TRY(c, hipMalloc(&c->d_models_raw, sizeof(pga_training) * (size_t)n_models));
    HIP_TRY(c, hipMalloc((void**)&c->d_model_const, sizeof(ModelConst) * (size_t)n_models));
    HIP_TRY(c, hipMemcpy(c->d_models_raw, c->models.data(), sizeof(pga_training) * (size_t)n_models, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_model_const, mcs.data(), sizeof(ModelConst) * (size_t)n_models, hipMemcpyHostToDevice));
    c->n_models = n_models;
    return pga_finder_models_changed(c);
}

namespace {
struct DevBuf {     // frees everything it allocated when the call returns
    std::vector<void*> ptrs;
    ~DevBuf() { for (void* p : ptrs) hipFree(p); }
    template <typename T> hipError_t alloc(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};
}  // namespace

namespace {
// bias . gc_score per node, the factor of the training pass (ref: _connection.h, `final == false` branches)
__global__ void k_gc_factor(int n, const double* __restrict__ gc_score, double b0, double b1, double b2, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = b0 * gc_score[3 * i] + b1 * gc_score[3 * i + 1] + b2 * gc_score[3 * i + 2];
}
}  // namespace

static int score_connections_impl(pga_ctx* c, int32_t n, const int32_t* ndx, const int32_t* stop_val,
                                  const uint8_t* type, const int8_t* strand, const double* cscore,
                                  const double* sscore, const double* rscore, const double* uscore,
                                  const int32_t* star_ptr, double st_wt, int final, const double* gc_score, const double* bias,
                                  double* score, int32_t* traceb, int8_t* ov_mark, int32_t* max_index, double* kernel_ms);

extern "C" int pga_score_connections(pga_ctx* c, int32_t n, const int32_t* ndx, const int32_t* stop_val,
                                     const uint8_t* type, const int8_t* strand, const double* cscore,
                                     const double* sscore, const double* rscore, const double* uscore,
                                     const int32_t* star_ptr, double st_wt, int final, double* score,
                                     int32_t* traceb, int8_t* ov_mark, int32_t* max_index, double* kernel_ms) {
    if (!c) return PGA_EINVAL;
    if (!final) return fail(c, PGA_EINVAL, "pga_score_connections: the training pass (final=0) scores connections from the frame-bias "
                                           "scores of the nodes: use pga_score_connections_training");
    return score_connections_impl(c, n, ndx, stop_val, type, strand, cscore, sscore, rscore, uscore, star_ptr, st_wt, 1, nullptr, nullptr,
                                  score, traceb, ov_mark, max_index, kernel_ms);
}

extern "C" int pga_score_connections_training(pga_ctx* c, int32_t n, const int32_t* ndx, const int32_t* stop_val,
                                              const uint8_t* type, const int8_t* strand, const double* gc_score, const double* bias,
                                              const int32_t* star_ptr, double st_wt, double* score, int32_t* traceb, int8_t* ov_mark,
                                              int32_t* max_index, double* kernel_ms) {
    if (!c) return PGA_EINVAL;
    if (n > 0 && (!gc_score || !bias)) return fail(c, PGA_EINVAL, "pga_score_connections_training: NULL array");
    return score_connections_impl(c, n, ndx, stop_val, type, strand, gc_score, gc_score, gc_score, gc_score, star_ptr, st_wt, 0, gc_score, bias,
                                  score, traceb, ov_mark, max_index, kernel_ms);
}

static int score_connections_impl(pga_ctx* c, int32_t n, const int32_t* ndx, const int32_t* stop_val,
                                  const uint8_t* type, const int8_t* strand, const double* cscore,
                                  const double* sscore, const double* rscore, const double* uscore,
                                  const int32_t* star_ptr, double st_wt, int final, const double* gc_score, const double* bias,
                                  double* score, int32_t* traceb, int8_t* ov_mark, int32_t* max_index, double* kernel_ms) {
    if (n < 0) return fail(c, PGA_EINVAL, "pga_score_connections: negative node count");
    if (max_index) *max_index = -1;
    if (kernel_ms) *kernel_ms = 0.0;
    if (n == 0) return PGA_OK;
    if (!ndx || !stop_val || !type || !strand || !cscore || !sscore || !rscore || !uscore || !star_ptr || !score || !traceb || !ov_mark)
        return fail(c, PGA_EINVAL, "pga_score_connections: NULL array");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf db;
    struct { int32_t* ndx; int32_t* stop_val; uint8_t* type; int8_t* strand; double* cscore; double* sscore; double* rscore; double* uscore; int32_t* star_ptr; } nd{};
    DpBuffers buf{};
    ChainDesc* d_chain; ModelConst* d_mc;
    ChainDesc ch{0, 0, n, 0, 0, 1};
    // a long chain is cut into segments walked side by side (dp.hip "segmented chains")
    DpSegPlan seg_plan;
    const bool use_wave = final && pga_dp_use_wave(1);          // one chain: only when PGA_DP_KERNEL=wave asks for it
    const bool segmented = final && !use_wave && pga_dp_plan(&ch, 1, n, seg_plan);
    const size_t N = (size_t)n + (size_t)seg_plan.extra, NS = 1 + seg_plan.segs.size();
    HIP_TRY(c, db.alloc(&nd.ndx, N)); HIP_TRY(c, db.alloc(&nd.stop_val, N)); HIP_TRY(c, db.alloc(&nd.type, N));
    HIP_TRY(c, db.alloc(&nd.strand, N)); HIP_TRY(c, db.alloc(&nd.cscore, N)); HIP_TRY(c, db.alloc(&nd.sscore, N));
    HIP_TRY(c, db.alloc(&nd.rscore, N)); HIP_TRY(c, db.alloc(&nd.uscore, N)); HIP_TRY(c, db.alloc(&nd.star_ptr, 3 * N));
    HIP_TRY(c, db.alloc(&buf.src, N)); HIP_TRY(c, db.alloc(&buf.tgt, N)); HIP_TRY(c, db.alloc(&buf.score, N));
    HIP_TRY(c, db.alloc(&buf.traceb, N)); HIP_TRY(c, db.alloc(&buf.tbn, N)); HIP_TRY(c, db.alloc(&buf.ov_mark, N));
    HIP_TRY(c, db.alloc(&buf.max_index, NS)); HIP_TRY(c, db.alloc(&buf.max_score, NS)); HIP_TRY(c, db.alloc(&buf.ipath, NS));
    HIP_TRY(c, db.alloc(&buf.A, N)); HIP_TRY(c, db.alloc(&buf.V[0], N)); HIP_TRY(c, db.alloc(&buf.V[1], N)); HIP_TRY(c, db.alloc(&buf.V[2], N));
    HIP_TRY(c, db.alloc(&buf.hv, N)); HIP_TRY(c, db.alloc(&buf.hi, N));
    buf.prof = nullptr;
    if (getenv("PGA_DP_PROFILE")) { HIP_TRY(c, db.alloc(&buf.prof, 16)); HIP_TRY(c, hipMemsetAsync(buf.prof, 0, 128, c->stream)); }
    HIP_TRY(c, db.alloc(&d_chain, 1)); HIP_TRY(c, db.alloc(&d_mc, 1));
    hipStream_t st = c->stream;
#define UP(dst, srcp, bytes) HIP_TRY(c, hipMemcpyAsync(dst, srcp, bytes, hipMemcpyHostToDevice, st))
    const size_t NN = (size_t)n;
    UP(nd.ndx, ndx, 4 * NN); UP(nd.stop_val, stop_val, 4 * NN); UP(nd.type, type, NN); UP(nd.strand, strand, NN);
    double* d_gcb = nullptr;
    if (final) { UP(nd.cscore, cscore, 8 * NN); UP(nd.sscore, sscore, 8 * NN); UP(nd.rscore, rscore, 8 * NN); UP(nd.uscore, uscore, 8 * NN); }
    else {
        double* d_gcs;
        HIP_TRY(c, db.alloc(&d_gcs, 3 * NN)); HIP_TRY(c, db.alloc(&d_gcb, NN));
        UP(d_gcs, gc_score, 24 * NN);
        hipLaunchKernelGGL(k_gc_factor, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, st, n, d_gcs, bias[0], bias[1], bias[2], d_gcb);
    }
    UP(nd.star_ptr, star_ptr, 12 * NN);
    DpSegDev seg_dev{};
    if (segmented) {
        char* arena;
        HIP_TRY(c, db.alloc(&arena, pga_dp_seg_bytes(seg_plan, 1, n)));
        HIP_TRY(c, pga_dp_seg_bind(seg_plan, 1, n, arena, st, &seg_dev));
    }
    ModelConst mc; pga_fill_model_const(&mc, st_wt);
    UP(d_chain, &ch, sizeof ch); UP(d_mc, &mc, sizeof mc);
#undef UP
    NodeArrays na{nd.ndx, nd.stop_val, nd.type, nd.strand, nd.cscore, nd.sscore, nd.rscore, nd.uscore, nd.star_ptr, d_gcb};
    if (use_wave) {
        DpwGroupPtrs wg{};
        DpwBuffers wb{};
        int32_t* d_cbase;
        const int32_t h_cbase[2] = {0, n};
        HIP_TRY(c, db.alloc(&wg.g[0].kf, N)); HIP_TRY(c, db.alloc(&wg.g[0].lo, N)); HIP_TRY(c, db.alloc(&wg.g[0].q1, N)); HIP_TRY(c, db.alloc(&wg.g[0].q2, N));
        HIP_TRY(c, db.alloc(&wb.cs, N + 2)); HIP_TRY(c, db.alloc(&wb.ext, N + 4)); HIP_TRY(c, db.alloc(&wb.sfxv, N)); HIP_TRY(c, db.alloc(&wb.sfxi, N));
        HIP_TRY(c, db.alloc(&d_cbase, 2));
        HIP_TRY(c, hipMemcpyAsync(d_cbase, h_cbase, sizeof h_cbase, hipMemcpyHostToDevice, st));
        wg.g[0].ndx = nd.ndx; wg.g[0].stop_val = nd.stop_val;
        pga_launch_dpw_topo(wg.g[0], nd.type, nd.strand, d_cbase, 1, n, st, n);
        pga_launch_dpw_chain(d_chain, 1, 0, n, na, wg.g[0], d_mc, wb, st);
        bool sched = pga_dpw_use_sched();
        if (sched) {
            // the step schedule of this one contig (the chain's sched_b0 is 0); a buffer it does not fit sends the launch to k_dpw_dyn
            int32_t* d_bbase;
            const int nbat = (n + 63) >> 6;
            const int32_t h_bbase[2] = {0, nbat};
            uint32_t h_cur[2] = {0, 0};
            HIP_TRY(c, db.alloc(&wg.g[0].shdr, (size_t)nbat + 1)); HIP_TRY(c, db.alloc(&wg.g[0].sent, 2 * (size_t)DPW_SCHED_STRIDE * ((size_t)nbat + 1) + 16)); HIP_TRY(c, db.alloc(&wg.g[0].scur, 16));
            HIP_TRY(c, db.alloc(&d_bbase, 2));
            HIP_TRY(c, hipMemcpyAsync(d_bbase, h_bbase, sizeof h_bbase, hipMemcpyHostToDevice, st));
            pga_launch_dpw_sched(wg.g[0], d_cbase, d_bbase, 1, nbat, st);
            HIP_TRY(c, hipMemcpyAsync(h_cur, wg.g[0].scur, sizeof h_cur, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (h_cur[1] != 0) sched = false;
        }
        HIP_TRY(c, hipEventRecord(c->ev0, st));
        pga_launch_dp_wave(d_chain, 1, wg, d_mc, buf, wb, st, nullptr, 0, sched);
        HIP_TRY(c, hipEventRecord(c->ev1, st));
        HIP_TRY(c, hipStreamSynchronize(st));                  // h_cbase lives on this stack frame
    } else {
        pga_launch_dp_prepare(d_chain, 1, 0, n, na, d_mc, buf, st, final);
        HIP_TRY(c, hipEventRecord(c->ev0, st));
        pga_launch_dp(d_chain, 1, d_mc, buf, final, st, segmented ? &seg_dev : nullptr);
        HIP_TRY(c, hipEventRecord(c->ev1, st));
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(score, buf.score, 8 * NN, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(traceb, buf.traceb, 4 * NN, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(ov_mark, buf.ov_mark, NN, hipMemcpyDeviceToHost, st));
    int32_t mi = -1;
    HIP_TRY(c, hipMemcpyAsync(&mi, buf.max_index, 4, hipMemcpyDeviceToHost, st));
    int32_t h_flags[PGA_SEG_ROUNDS] = {};
    if (segmented) HIP_TRY(c, hipMemcpyAsync(h_flags, seg_dev.flags, sizeof h_flags, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    pga_dp_note_stats(c, segmented ? &seg_plan : nullptr, h_flags, 1);
    if (buf.prof && use_wave) {
        unsigned long long pr[16]; HIP_TRY(c, hipMemcpy(pr, buf.prof, 128, hipMemcpyDeviceToHost));
        const double nbp = pr[7] ? (double)pr[7] : 1.0;
        fprintf(stderr, "[pga dp profile] k_dp_wave batches=%llu cycles/batch: load=%.0f near steps=%.0f far gene ends=%.0f carries=%.0f chains=%.0f "
                        "walk=%.0f finalize=%.0f | total=%.0f\n", pr[7], pr[0] / nbp, pr[1] / nbp, pr[2] / nbp, pr[3] / nbp, pr[4] / nbp, pr[5] / nbp,
                pr[6] / nbp, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[5] + pr[6]) / nbp);
    } else if (buf.prof) {
        unsigned long long pr[16]; HIP_TRY(c, hipMemcpy(pr, buf.prof, 128, hipMemcpyDeviceToHost));
        const double nbp = pr[5] ? (double)pr[5] : 1.0;
        fprintf(stderr, "[pga dp profile] batches=%llu cycles/batch: serial wave: previous batch=%.0f walk=%.0f publish=%.0f | weight helper=%.0f | "
                        "far field by target kind=%.0f/%.0f/%.0f/%.0f store=%.0f | iteration=%.0f\n",
                pr[5], pr[0] / nbp, pr[1] / nbp, pr[2] / nbp, pr[3] / nbp, pr[8] / nbp, pr[9] / nbp, pr[10] / nbp, pr[11] / nbp, pr[12] / nbp, pr[6] / nbp);
        fprintf(stderr, "[pga dp profile] kind0 far wave: suffix build=%.0f | previous-batch slice: serial wave=%.0f helper wave=%.0f\n", pr[13] / nbp, pr[14] / nbp, pr[15] / nbp);
    }
    if (max_index) *max_index = mi;
    if (kernel_ms) { float ms = 0; HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1)); *kernel_ms = ms; }
    return PGA_OK;
}
