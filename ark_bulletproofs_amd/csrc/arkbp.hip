// libarkbp_hip.so — context, workspaces, host orchestration and the C ABI (include/arkbp.h).
// There is no CPU fallback: without a HIP device every compute entry point returns BP_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <sys/prctl.h>
#include <time.h>
#include <rccl/rccl.h>   // types only: the entry points are bound with dlsym (rccl_api)
#include <dlfcn.h>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/arkbp.h"
#include <atomic>
#include <chrono>
#include "host_proto.hpp"
#include "r1cs.cuh"
#include "pedersen.cuh"
#include "small.cuh"
#include "vfe.hpp"
#include "vfe_sched.hpp"
static_assert(arkbp::vfe::PB_WORDS == arkbp::VFY_PB_WORDS, "parameter block layout shared by vfe.hip and r1cs.cuh");

using namespace arkbp;
using arkbp::host::A4;
using arkbp::host::F4;
using arkbp::host::J4;

static thread_local std::string g_err;
const char* bp_last_error(void) { return g_err.c_str(); }

// ---- host-only dry run (sanitizer builds on machines without a GPU; include/arkbp.h bp_debug_ctx_create_hostonly) ------------------
// A ctx created without a device runs the HOST side of batch verification — framing, thread pools, shared recordings, transcript
// replay, template cache and eviction, staging — with every HIP call replaced: "device" buffers are zeroed host memory, copies are
// memcpy, kernels / streams / events are skipped.  Nothing is verified in that mode (the entry points return a checksum of what the
// host staged instead of the mega-check); it exists so that ASan / UBSan / TSan see the multi-threaded host layer (tools/sanitize/).
// g_dry is set, per calling thread, by the entry points that accept such a ctx; the library's own pool threads make no HIP calls.
static thread_local bool g_dry = false;
namespace dry {
static char g_token;   // stands for streams and events
template <class T> static inline hipError_t Malloc(T** p, size_t n) { if (g_dry) { *p = (T*)calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; } return ::hipMalloc(p, n); }
static inline hipError_t Free(void* p) { if (g_dry) { free(p); return hipSuccess; } return ::hipFree(p); }
template <class T> static inline hipError_t HostMalloc(T** p, size_t n, unsigned flags = hipHostMallocDefault) { if (g_dry) { *p = (T*)calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; } return ::hipHostMalloc(p, n, flags); }
static inline hipError_t HostFree(void* p) { if (g_dry) { free(p); return hipSuccess; } return ::hipHostFree(p); }
static inline hipError_t MemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind k, hipStream_t st) { if (g_dry) { if (n) memmove(d, s_, n); return hipSuccess; } return ::hipMemcpyAsync(d, s_, n, k, st); }
static inline hipError_t MemsetAsync(void* d, int v, size_t n, hipStream_t st) { if (g_dry) { if (n) memset(d, v, n); return hipSuccess; } return ::hipMemsetAsync(d, v, n, st); }
static inline hipError_t GetLastError() { return g_dry ? hipSuccess : ::hipGetLastError(); }
static inline hipError_t SetDevice(int d) { return g_dry ? hipSuccess : ::hipSetDevice(d); }
static inline hipError_t StreamSynchronize(hipStream_t st) { return g_dry ? hipSuccess : ::hipStreamSynchronize(st); }
static inline hipError_t StreamCreateWithFlags(hipStream_t* st, unsigned f) { if (g_dry) { *st = (hipStream_t)&g_token; return hipSuccess; } return ::hipStreamCreateWithFlags(st, f); }
static inline hipError_t StreamDestroy(hipStream_t st) { return g_dry ? hipSuccess : ::hipStreamDestroy(st); }
static inline hipError_t EventCreate(hipEvent_t* e) { if (g_dry) { *e = (hipEvent_t)&g_token; return hipSuccess; } return ::hipEventCreate(e); }
static inline hipError_t EventCreateWithFlags(hipEvent_t* e, unsigned f) { if (g_dry) { *e = (hipEvent_t)&g_token; return hipSuccess; } return ::hipEventCreateWithFlags(e, f); }
static inline hipError_t EventDestroy(hipEvent_t e) { return g_dry ? hipSuccess : ::hipEventDestroy(e); }
static inline hipError_t EventRecord(hipEvent_t e, hipStream_t st) { return g_dry ? hipSuccess : ::hipEventRecord(e, st); }
static inline hipError_t EventSynchronize(hipEvent_t e) { return g_dry ? hipSuccess : ::hipEventSynchronize(e); }
static inline hipError_t EventQuery(hipEvent_t e) { return g_dry ? hipSuccess : ::hipEventQuery(e); }
static inline hipError_t EventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { if (g_dry) { *ms = 0; return hipSuccess; } return ::hipEventElapsedTime(ms, a, b); }
}  // namespace dry
#define hipMalloc(...) dry::Malloc(__VA_ARGS__)
#define hipFree(...) dry::Free(__VA_ARGS__)
#define hipHostMalloc(...) dry::HostMalloc(__VA_ARGS__)
#define hipHostFree(...) dry::HostFree(__VA_ARGS__)
#define hipMemcpyAsync(...) dry::MemcpyAsync(__VA_ARGS__)
#define hipMemsetAsync(...) dry::MemsetAsync(__VA_ARGS__)
#define hipGetLastError(...) dry::GetLastError(__VA_ARGS__)
#define hipSetDevice(...) dry::SetDevice(__VA_ARGS__)
#define hipStreamSynchronize(...) dry::StreamSynchronize(__VA_ARGS__)
#define hipStreamCreateWithFlags(...) dry::StreamCreateWithFlags(__VA_ARGS__)
#define hipStreamDestroy(...) dry::StreamDestroy(__VA_ARGS__)
#define hipEventCreate(...) dry::EventCreate(__VA_ARGS__)
#define hipEventCreateWithFlags(...) dry::EventCreateWithFlags(__VA_ARGS__)
#define hipEventDestroy(...) dry::EventDestroy(__VA_ARGS__)
#define hipEventRecord(...) dry::EventRecord(__VA_ARGS__)
#define hipEventSynchronize(...) dry::EventSynchronize(__VA_ARGS__)
#define hipEventQuery(...) dry::EventQuery(__VA_ARGS__)
#define hipEventElapsedTime(...) dry::EventElapsedTime(__VA_ARGS__)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) do { if (!g_dry) { hipLaunchKernelGGLInternal(__VA_ARGS__); } } while (0)

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                 \
            return BP_E_HIP;                                                                        \
        }                                                                                           \
    } while (0)
#define BPCHK(x) do { int r_ = (x); if (r_ != BP_OK) return r_; } while (0)

// ---- native collectives: RCCL over xGMI (BASELINE north_star: "final RCCL point-reduce over xGMI") --------------------------------
// The library does not link librccl: it binds the five entry points it needs at run time from the RCCL the process already has
// (a PyTorch-ROCm host brings its own copy) or from the ROCm installation, so a single-GPU host never loads RCCL at all.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};
static RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        // ARKBP_RCCL_LIB names the library to bind instead (tests point it at a path that does not exist to drive the error path)
        const char* forced = getenv("ARKBP_RCCL_LIB");
        std::string last_err;
        auto try_open = [&](const char* n, int flags) {
            h = dlopen(n, flags);
            if (!h) { const char* e = dlerror(); if (e) last_err = e; }   // dlerror() clears the message it returns: read it ONCE
            return h != nullptr;
        };
        if (forced) try_open(forced, RTLD_NOW | RTLD_GLOBAL);
        else {
            for (const char* n : names) if (try_open(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) break;   // the copy the process already mapped
            if (!h) for (const char* n : names) if (try_open(n, RTLD_NOW | RTLD_GLOBAL)) break;
        }
        if (!h) { api.why = std::string("librccl not found: ") + last_err; return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        if (!api.ok) api.why = "librccl lacks an expected entry point";
    });
    return api;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool owned = true;  // false: a view of another ctx's allocation (shared generator tables)
    int ensure(size_t bytes) {
        if (bytes <= cap) return BP_OK;
        if (!owned) { g_err = "shared device buffer too small"; return BP_E_ARG; }
        if (p) HIPCHK(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return BP_OK;
    }
    // exactly `bytes` (the large precomputed tables: the 25 % growth slack of ensure() would be tens of GB)
    int ensure_exact(size_t bytes) {
        if (bytes <= cap) return BP_OK;
        if (!owned) { g_err = "shared device buffer too small"; return BP_E_ARG; }
        if (p) HIPCHK(hipFree(p));
        p = nullptr; cap = 0;
        HIPCHK(hipMalloc(&p, bytes + 256));
        cap = bytes + 256;
        return BP_OK;
    }
    // buffers whose kernels rely on "all zero between uses" (and restore it themselves): zeroed once, when (re)allocated
    int ensure_zeroed(size_t bytes, hipStream_t st) {
        if (bytes <= cap) return BP_OK;
        int rc = ensure(bytes);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(p, 0, cap, st));
        return BP_OK;
    }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
    template <class T> T* as() { return (T*)p; }
};

struct KTimer {
    double ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct IpaState {
    const u32 *d_Q = nullptr, *d_Gf = nullptr, *d_Hf = nullptr;
    u32 *d_G = nullptr, *d_H = nullptr, *d_a = nullptr, *d_b = nullptr;
    size_t n = 0;            // current vector length
    size_t round = 0;
    bool first = true;
    // pending common factors of the resident generator vectors: G_true = gamma_G * Ghat, H_true = gamma_H * Hhat
    F4 gamma_G, gamma_H;
    bool pending = false, h_geo = false;  // h_geo: H_true[i] = gamma_H * rho^i * Hhat[i]
    // optional hints of the R1CS prover (see ipa_create_dev)
    bool have_gf = false, have_rho = false;
    F4 gf_halves[2];
    const F4* rho_pw = nullptr;
    const u32* d_rho_pow = nullptr;
    bool lr_done = false;
    // frozen-generator tail (ipa.cuh): from length n0 on, G and H stay as they are and per-element coefficients fold instead
    bool allow_freeze = false, frozen = false;
    size_t n0 = 0;
    u32 *d_cG = nullptr, *d_cH = nullptr;
    // multi-GPU: how the L / R MSMs of this instance treat the ctx's shard mode (see msm_run): -1 = the ctx default (window
    // partition + point-reduce), 2 = this rank's terms are its own slice of the vectors (index-cyclic IPA): all windows, then the
    // point-reduce, 0 = replicated (every rank computes the same small MSM: the frozen tail after the gather)
    // gens_stride > 0: the working vectors are (a slice of) the ctx's generator tables at round 1: d_G[i] = G[gens_first + i*gens_stride]
    u32 gens_first = 0, gens_stride = 0;
    // round 1 may read the generators straight from the ctx's resident tables (no working copy): non-null until the first fold
    const u32 *d_G_in = nullptr, *d_H_in = nullptr;
    bool have_qw = false;      // Q = qw * B (the R1CS prover's Q, prover.rs:779): lets round 1 run as a fixed-base MSM over the tables
    F4 qw;
    int msm_mode = -1;
    // deferred first fold (ipa.cuh "TWO fold rounds from the tables"): round 1's multipliers are kept, G and H stay the generator
    // tables for one more round, and the second fold produces Ghat'' / Hhat'' straight from the tables
    bool deferred = false;
    F4 def_tG, def_tH;
    size_t min_len = 0;         // index-cyclic slices: the local length at which the ranks gather (a deferred fold needs one more local round)
    bool direct = false;        // frozen from round 1 over the ctx's generator tables, L and R as sums over the direct window tables (small.cuh)
    // direct: the fold a challenge asks for is carried out by the NEXT round's k_dt_round (or by k_ipa_fold_ab after the last round)
    bool fold_pending = false;
    F4 pend_u, pend_ui;
    u32 *d_a_alt = nullptr, *d_b_alt = nullptr;   // second buffer pair: a fused fold reads one pair and writes the other
    F4 geo_k0;                  // index-cyclic slices: the geometric H factor of local element j is K * rho^(rank + j*world) = (K * geo_k0) * (rho^world)^j
    bool have_k0 = false;
};

struct bp_ctx {
    int curve = 0, device = 0;
    bool host_only = false;      // no device behind this ctx: the host-side dry run of batch verification (see g_dry)
    hipStream_t stream = nullptr;
    bool profiling = false;
    size_t tune_fold_batch_min = 65536;   // BP_TUNE_FOLD_BATCH_MIN
    size_t tune_msm_bin_min = 64;         // BP_TUNE_MSM_BIN_MIN
    size_t tune_ipa_freeze_len = 8192;    // BP_TUNE_IPA_FREEZE_LEN (measured at 2^20, one proof at a time: 1024 -> 51.0 ms of inner-product argument, 4096 -> 49.0, 8192 -> 47.5, 16384 -> 46.8 with a slower prove() around it; throughput unchanged)
    size_t tune_msm_wsum_min = (size_t)1 << 18;   // BP_TUNE_MSM_WSUM_MIN: buckets from which running-sum window aggregation replaces the marginals
    KTimer timers[BP_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    // MSM workspaces
    DevBuf canon, hist, lvl_off, totals, cursor, entries, slots, bin_cur, boff, lvA, lvB, Tbuf, io_pts, io_scal, io_out;
    DevBuf fs_bcnt, fs_loff, fs_binch, fs_sums;   // fixed-shape MSM pipeline (msm.cuh 7)
    // IPA workspaces (resident layouts)
    DevBuf ipa_G, ipa_H, ipa_a, ipa_b, ipa_Gf, ipa_Hf, ipa_sL, ipa_sR, ipa_part, ipa_Q, ipa_jac, ipa_pref, ipa_cG, ipa_cH;
    // generator tables (BulletproofGens party 0, PedersenGens), resident layout
    DevBuf d_G, d_H, d_pc;
    DevBuf pc_table;   // fixed-base window tables of B, B_blinding (pedersen.cuh), built on first use
    DevBuf pc_dt;      // direct window tables of B, B_blinding (small.cuh k_dt_commit)
    size_t gens_cap = 0;
    A4 pc_B, pc_Bb;
    // R1CS prover / verifier vectors (resident scalar layout)
    DevBuf r_aL, r_aR, r_aO, r_sL, r_sR, r_wL, r_wR, r_wO, r_msmsc, r_ypow, r_part, r_small, r_g, r_h, r_chal, r_tail;
    // batch verification: per-proof parameter blocks, chunk partials; cached circuit templates (VTemplate<C>)
    DevBuf v_params, v_gpart, v_hpart, v_alpha, v_tables, v_dec;
    DevBuf p_moff, p_ment, p_mc, p_coefs, p_ztab;   // prover-side constraint index of the statement being proved (k_r1cs_flatten)
    // verifier front end on the device (vfe.hip; r1cs_host.inc batch_verify_device): raw inputs, serialized items, challenges, scratch
    DevBuf vfe_in, vfe_msg, vfe_chal, vfe_ws, vfe_small;
    void* h_vfe = nullptr;                    // pinned staging: [proof bytes | commitments | transcript states | small results]
    size_t h_vfe_cap = 0;
    std::map<std::string, std::shared_ptr<void>> vfe_classes;   // (shared recording, m, k, transcript position) -> VfeClassDev<C>
    bool tune_vfy_device = true;              // BP_TUNE_VFY_DEVICE
    uint64_t fb_runs = 0, fb_runs_sharded = 0; // fixed-base MSMs completed on this ctx / of those, on a rank's share of the terms of a sharded proof
    uint64_t vfe_batches = 0, vfe_fallbacks = 0;                 // batches the device front end completed / handed to the host replay
    void* h_vstage[2] = {nullptr, nullptr};   // pinned staging halves of the batch-verify pipeline
    size_t h_vstage_cap[2] = {0, 0};
    hipEvent_t vstage_ev[2] = {nullptr, nullptr};
    hipEvent_t vtail_ev[2] = {nullptr, nullptr};   // the tails of a block go up on aux_stream
    std::unique_ptr<host::HostPool> pool;     // host threads of this ctx's data-parallel loops (created on first use)
    hipEvent_t sync_ev = nullptr;             // blocking-sync event of ctx_stream_wait
    void* h_upload = nullptr;                 // pinned slab for the prover's witness uploads (r1cs_host.inc upload_scalars_pinned)
    size_t h_upload_cap = 0;
    void* h_csc = nullptr;                    // pinned staging of the prover's constraint index
    size_t h_csc_cap = 0;
    void* h_dec = nullptr;                    // pinned: x words, points, flags, ok of a batch's compressed points
    size_t h_dec_cap = 0;
    hipStream_t aux_stream = nullptr;         // point decompression runs beside the main stream
    hipEvent_t dec_ev[2] = {nullptr, nullptr};
    std::map<std::string, std::shared_ptr<void>> templates;          // structure digest -> VTemplate<C>
    std::map<std::string, std::shared_ptr<void>> scenario_sources;   // (scenario, params, publics) -> recorded, frozen verifier
    size_t scenario_source_terms = 0;
    void* h_vaux[2] = {nullptr, nullptr};     // pinned: launch order + per-instance coefficient tables of a block
    size_t h_vaux_cap[2] = {0, 0};
    // window-sharded multi-GPU mode (bp_ctx_set_window_shard): every MSM of this ctx accumulates only this rank's Pippenger
    // windows and the ranks' partial points are summed through the host's collective
    int shard_rank = 0, shard_world = 1;
    bp_point_reduce_cb shard_cb = nullptr;
    void* shard_user = nullptr;
    bp_allgather_cb gather_cb = nullptr;   // optional: lets the prover partition the IPA index-cyclically (bp_ctx_set_shard_allgather)
    void* gather_user = nullptr;
    // native collectives (bp_ctx_rccl_init): both exchanges of the sharded mode as ncclAllGather on the ctx's stream
    ncclComm_t nccl = nullptr;
    DevBuf coll_send, coll_recv;
    void* h_coll = nullptr;                // pinned: [send | recv]
    size_t h_coll_cap = 0;
    uint64_t coll_count = 0;
    double coll_seconds = 0;
    size_t tune_cyclic_min = (size_t)1 << 14;   // BP_TUNE_CYCLIC_MIN: padded size from which a sharded prover partitions the IPA
    size_t tune_msm_fixed_min = (size_t)1 << 20;   // BP_TUNE_MSM_FIXED_MIN: terms from which MSMs over the generator tables use the fixed-base rows
    size_t tune_host_threads = 0;                  // BP_TUNE_HOST_THREADS: size of this ctx's host pool (0 = host_pool_threads())
    hipEvent_t sync_ev_aux = nullptr;              // event of ctx_aux_stream_wait
    unsigned tune_wait_sleep = 0;                  // BP_TUNE_WAIT_SLEEP: host waits poll and sleep this many microseconds in between instead of busy-waiting (see event_wait)
    bool msm_latency_first = false;                // set by the bp_msm* entry points for the duration of the call (see msm_use_quad)
    size_t tune_msm_chunk_cap = 0;                 // BP_TUNE_MSM_CHUNK_CAP: entries per level-1 chunk of the fixed-shape MSM pipeline (0 = fs_chunk_cap's choice)
    size_t tune_fold_quad_max = 0;                 // BP_TUNE_FOLD_QUAD_MAX: fold rounds with at most this many output points run four lanes per point (0 = never)
    size_t tune_msm_glv_min = 256;                 // BP_TUNE_MSM_GLV_MIN: terms from which an MSM on a GLV curve splits its scalars
    DevBuf cyc_a, cyc_b, cyc_Gf, cyc_Hf;
    // fixed-base MSM rows of the generators (bp_gens_msm_tables): row r of a table = 2^(4r) * base, r < FB_ROWS, layout [r][i]
    DevBuf fb_G, fb_H, fb_pc;
    size_t fb_cap = 0;       // bases covered per vector
    // fixed-base tables of the generators for the first fold round (bp_gens_fold_tables; ipa.cuh k_ipa_fold_tab)
    DevBuf ftab_G, ftab_H;
    size_t ftab_n = 0;       // bases covered: G[0..ftab_n), H[0..ftab_n)
    u32 ftab_first = 0, ftab_stride = 1;   // ... or, for a rank's SLICE of the tables (bp_gens_fold_tables_slice): column i = generator ftab_first + i * ftab_stride
    int ftab_w = 0, ftab_nwin = 0;
    // direct window tables of the first generators (small.cuh): bases [B, B_blinding | G[0..dt_cap) | H[0..dt_cap)], built on the first
    // small statement this ctx proves
    DevBuf dt_tab, dt_part, dt_a2, dt_b2, dt_ticket;
    size_t dt_cap = 0;
    size_t tune_direct_max = 8192;   // BP_TUNE_DIRECT_MAX: padded sizes up to this one prove over the direct tables (0 = never)
    u32* h_dt = nullptr;             // pinned: the results of one launch (or its partial points, see msm_direct_launch)
    u32 dt_pending_parts = 1;        // partial points per MSM on their way to h_dt
    uint64_t dt_runs = 0;            // MSMs answered from the direct tables
    uint64_t folds_deferred = 0, folds_tab2 = 0;   // first folds deferred / second folds that came straight from the tables (bp_ctx_fold_stats)
    IpaState ipa_step;         // bp_ipa_begin .. bp_ipa_finish
    bool ipa_step_active = false;
    u32* h_totals = nullptr;  // pinned
    u32* h_T = nullptr;       // pinned
    size_t h_T_cap = 0;
};

static int get_event(bp_ctx* c, hipEvent_t* e) {
    if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); return BP_OK; }
    HIPCHK(hipEventCreate(e));
    return BP_OK;
}
// Host waits.  hipStreamSynchronize AND hipEventSynchronize busy-wait on this stack — also for events created with
// hipEventBlockingSync: measured 1.00 core of process CPU per waiting thread (tools/wait_cpu.py; a 2^22-term MSM: 117.3 ms of wall,
// 117.3 ms of CPU).  With eight proofs in flight that was seven cores of a 16-CPU quota doing nothing, and the quota was what bounded
// the prover pipeline (bench.py config.host_cpu_in_timed_region: 12.5 cores used, the cgroup throttled in a third of its periods).
// BP_TUNE_WAIT_SLEEP = 1 makes a ctx's waits poll the event and sleep ~50 us in between: one 2^20 proof then costs 20 ms of host CPU
// instead of 68 (wall 67.7 -> 68.9 ms) and the pipeline's throughput is unchanged at 7.9 instead of 12.5 cores — the setting for a
// prover service.  Batch verification waits far more often per millisecond of GPU work and lost 10 % with it, so the default stays
// the busy wait (ARKBP_SYNC=sleep / =spin force either everywhere); the bp_msm* entry points always busy-wait.
static void wait_thread_setup() {
    static thread_local bool done = false;
    if (!done) { done = true; (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0); }   // 2 us of timer slack instead of 50: short sleeps stay short
}
static int wait_env() {   // -1: per ctx (BP_TUNE_WAIT_SLEEP), 0: busy wait everywhere, 1: sleeping waits everywhere
    static const int v = !getenv("ARKBP_SYNC") ? -1 : !strcmp(getenv("ARKBP_SYNC"), "sleep") ? 1 : !strcmp(getenv("ARKBP_SYNC"), "spin") ? 0 : -1;
    return v;
}
static unsigned wait_sleep_us(const bp_ctx* c) { const int e = wait_env(); return e == 0 ? 0u : (c && c->tune_wait_sleep) ? c->tune_wait_sleep : e == 1 ? 30u : 0u; }
static bool wait_sleeps(const bp_ctx* c) { return wait_sleep_us(c) != 0; }
static hipError_t event_wait(const bp_ctx* c, hipEvent_t ev) {
    if (!wait_sleeps(c)) return hipEventSynchronize(ev);
    wait_thread_setup();
    for (int it = 0;; it++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return hipSuccess;
        if (e != hipErrorNotReady) return e;
        if (it < 3) continue;                              // (a wait that is over within a few microseconds costs no sleep)
        const long ns = (long)wait_sleep_us(c) * 1000L;
        struct timespec ts = {0, it < 40 ? ns : 2 * ns};
        nanosleep(&ts, nullptr);
    }
}
static hipError_t stream_wait(bp_ctx* c, hipStream_t stream, hipEvent_t& ev) {
    if (!wait_sleeps(c)) return hipStreamSynchronize(stream);
    if (!ev) { hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
    hipError_t e = hipEventRecord(ev, stream);
    if (e != hipSuccess) return e;
    return event_wait(c, ev);
}
static hipError_t ctx_stream_wait(bp_ctx* c) {
    if (c->msm_latency_first) return hipStreamSynchronize(c->stream);
    return stream_wait(c, c->stream, c->sync_ev);
}
static hipError_t ctx_aux_stream_wait(bp_ctx* c) { return stream_wait(c, c->aux_stream, c->sync_ev_aux); }
// all-gather of `bytes` host bytes per rank -> recv (world x bytes, rank order): ncclAllGather on the ctx's stream between a pinned
// H2D and a pinned D2H copy; the stream order puts it behind the kernels already enqueued
static int ctx_native_allgather(bp_ctx* c, const void* send, size_t bytes, void* recv) {
    RcclApi& api = rccl_api();
    const size_t W = (size_t)c->shard_world;
    const auto t0 = std::chrono::steady_clock::now();
    BPCHK(c->coll_send.ensure(bytes)); BPCHK(c->coll_recv.ensure(bytes * W));
    if (c->h_coll_cap < bytes * (W + 1)) {
        if (c->h_coll) HIPCHK(hipHostFree(c->h_coll));
        c->h_coll = nullptr; c->h_coll_cap = 0;
        HIPCHK(hipHostMalloc(&c->h_coll, bytes * (W + 1) + 4096));
        c->h_coll_cap = bytes * (W + 1) + 4096;
    }
    memcpy(c->h_coll, send, bytes);
    HIPCHK(hipMemcpyAsync(c->coll_send.p, c->h_coll, bytes, hipMemcpyHostToDevice, c->stream));
    const ncclResult_t r = api.AllGather(c->coll_send.p, c->coll_recv.p, bytes, ncclUint8, c->nccl, c->stream);
    if (r != ncclSuccess) { g_err = std::string("ncclAllGather: ") + api.GetErrorString(r); return BP_E_HIP; }
    HIPCHK(hipMemcpyAsync((char*)c->h_coll + bytes, c->coll_recv.p, bytes * W, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx_stream_wait(c));
    memcpy(recv, (char*)c->h_coll + bytes, bytes * W);
    c->coll_count++;
    c->coll_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return BP_OK;
}
// the two exchanges of the sharded mode, through RCCL when the ctx has a communicator, else through the host's callbacks
static int shard_allgather(bp_ctx* c, const void* send, size_t bytes, void* recv) {
    if (c->nccl) return ctx_native_allgather(c, send, bytes, recv);
    if (!c->gather_cb) { g_err = "sharded mode: no all-gather installed"; return BP_E_ARG; }
    const int rc = c->gather_cb(c->gather_user, send, bytes, recv);
    if (rc) { g_err = "the all-gather callback failed"; return rc < 0 ? rc : BP_E_ARG; }
    return BP_OK;
}
static inline bool shard_has_allgather(const bp_ctx* c) { return c->nccl || c->gather_cb; }

struct ScopedK {  // records start/stop events around a region when profiling is on
    bp_ctx* c; int which; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
    ScopedK(bp_ctx* c_, int w) : c(c_), which(w), on(c_->profiling) {
        if (on) { if (get_event(c, &e0) || get_event(c, &e1)) { on = false; return; } (void)hipEventRecord(e0, c->stream); }
    }
    void stop() { if (on) { (void)hipEventRecord(e1, c->stream); c->timers[which].pending.push_back({e0, e1}); on = false; } }
    ~ScopedK() { stop(); }
};
static void collect_timers(bp_ctx* c) {
    for (int k = 0; k < BP_K_COUNT; k++) {
        for (auto& pr : c->timers[k].pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->timers[k].ms += ms; c->timers[k].launches++;
            }
            c->event_pool.push_back(pr.first); c->event_pool.push_back(pr.second);
        }
        c->timers[k].pending.clear();
    }
}

extern "C" int bp_points_import(bp_ctx* c, const void* d_in, void* d_out, size_t n);

// ---- MSM orchestration ---------------------------------------------------------------------------
static MsmPlan msm_plan(size_t n, int bits) {
    double best = 1e300; int bc = 4;
    for (int c = 3; c <= 16; c++) {
        double W = bits / c + 1, NBk = std::ldexp(1.0, c - 1), B = W * NBk;
        double cost = n * W + (n * W / MSM_CH + B) * 1.45 + B * (0.5 * c) * 1.45 + 256.0 * c * W * 0.05;
        if (cost < best) { best = cost; bc = c; }
    }
    static const int c_env = getenv("ARKBP_MSM_C") ? atoi(getenv("ARKBP_MSM_C")) : 0;   // experiments
    if (c_env >= 3 && c_env <= 16 && n >= 4096) bc = c_env;
    MsmPlan pl; pl.c = bc; pl.W = bits / bc + 1; pl.NB = 1 << (bc - 1); pl.B = (u32)pl.W * pl.NB; pl.n = (u32)n;
    pl.w_lo = 0; pl.w_hi = pl.W;
    return pl;
}

// quad-cooperative additions in the latency-bound kernels of the fixed-shape pipeline (ecq.cuh); ARKBP_MSM_NOQUAD=1 restores the
// lane-per-addition kernels (A/B: profiles/r03_msm_quad_ab.txt)
// Policy (measured, profiles/r03_msm_quad_glv_ab.txt): the GLV split and the quad-cooperative trees shorten ONE mid-size MSM
// (2^16 terms: 0.607 -> 0.513 ms wall) but do not lower its arithmetic — the split adds an endomorphism product to every second
// mixed addition, a quad addition issues 20 products where a lane issues 16 — so a caller that keeps the GPU full with many MSMs in
// flight (the prover pipeline: eight streams) gains nothing and pays ~5 % on the accumulate.  They are therefore used where the
// caller waits for the one result: the bp_msm* entry points.  ARKBP_MSM_LATENCY=1 / =0 forces them on / off everywhere (A/B).
static int msm_latency_env() { static const int v = getenv("ARKBP_MSM_LATENCY") ? atoi(getenv("ARKBP_MSM_LATENCY")) : -1; return v; }
static bool msm_use_quad(const bp_ctx* ctx) { const int e = msm_latency_env(); return e >= 0 ? e != 0 : ctx->msm_latency_first; }
static bool msm_quad_trees(const bp_ctx* ctx) { static const bool force = getenv("ARKBP_MSM_QUAD_TREES") != nullptr; return force || msm_use_quad(ctx); }   // (experiment: the quad trees without the GLV split)
// (direct launches in both branches: a kernel template named only inside a conditional expression is not emitted for the device)
#define MSM_LAUNCH_REDUCE_FS(red_g, grid, ...) do { if (msm_quad_trees(ctx) && (red_g) == 4u) hipLaunchKernelGGL((k_msm_reduce_fs<C, true>), grid, dim3(256), 0, st, __VA_ARGS__); \
                                                    else hipLaunchKernelGGL((k_msm_reduce_fs<C, false>), grid, dim3(256), 0, st, __VA_ARGS__); } while (0)
#define MSM_LAUNCH_MARGINALS_FS(grid, ...) do { if (msm_quad_trees(ctx)) hipLaunchKernelGGL((k_msm_marginals_fs<C, 256, true>), grid, dim3(256), 0, st, __VA_ARGS__); \
                                                else hipLaunchKernelGGL((k_msm_marginals_fs<C, 256, false>), grid, dim3(256), 0, st, __VA_ARGS__); } while (0)
// Entries per level-1 chunk of the fixed-shape pipeline.  Throughput callers keep 16.  For a caller waiting on ONE mid-size MSM the
// accumulate is a single round of workgroups whose duration is (entries per lane) x (latency of a mixed addition at the fill of
// the lane's SIMD: 6.7 us alone, ~5.6 us per resident wave from two up — tools/ubench_coop.hip): 2^16 terms at 16 entries per chunk
// make ~100 K lanes = 1.5 waves per SIMD, i.e. every lane waits for the SIMDs that hold two.  The cap is chosen so that the lanes
// fill a whole number of waves per SIMD with as few entries per lane as that allows (measured at 2^16 terms, secq256k1, accumulate /
// wall: cap 11 0.200 / 0.520 ms, 12 0.157 / 0.468, 13 0.169 / 0.479, 14 0.181 / 0.487, 16 0.205 / 0.506: profiles/r03_msm_chunk_cap.txt).
// binned: entries / buckets of the binned windows (bucket populations ~ Poisson); top: the narrow top window (a few full buckets).
static u32 fs_chunk_cap(const bp_ctx* ctx, double binned_entries, double binned_buckets, double top_entries, double top_buckets) {
    static const int cap_env = getenv("ARKBP_MSM_FS_CAP") ? atoi(getenv("ARKBP_MSM_FS_CAP")) : 0;   // experiments
    if (cap_env >= 8 && cap_env <= 64) return (u32)cap_env;
    if (ctx->tune_msm_chunk_cap) return (u32)ctx->tune_msm_chunk_cap;
    if (!msm_use_quad(ctx) || binned_buckets < 1.0) return 1u << MSM_CHL_BINNED;
    const double lanes_per_fill = 256.0 * 4 * 64;   // one wave on every SIMD of the chip
    const double mu = binned_entries / binned_buckets, sd = std::sqrt(std::max(mu, 1e-9));
    u32 best_cap = 1u << MSM_CHL_BINNED; double best_t = 1e300;
    for (u32 cap = 8; cap <= 32; cap++) {
        double e_chunks = 0, e_work = 0, wsum = 0;   // E ceil(X / cap) and E X / ceil(X / cap) for X ~ N(mu, mu)
        for (int q = -30; q <= 30; q++) {
            const double z = q / 10.0, x = std::max(1.0, mu + sd * z), w = std::exp(-0.5 * z * z), k = std::ceil(x / cap);
            e_chunks += w * k; e_work += w * x / k; wsum += w;
        }
        const double chunks = binned_buckets * e_chunks / wsum + top_entries / cap + 0.5 * top_buckets;
        const double fill = std::ceil(chunks / lanes_per_fill - 0.004);
        if (fill > 4.0) continue;   // throughput-bound anyway: the default
        const double t = (e_work / wsum) * (fill <= 1.0 ? 6.7 : 5.6 * fill) + 0.02 * (chunks / 1000.0);   // (+ the partials the next kernel has to add)
        if (t < best_t) { best_t = t; best_cap = cap; }
    }
    return best_cap;
}
// ---- the fixed-shape pipeline over GLV-split scalars (msm.cuh "GLV split"): 2n half-terms of 128 bits ------------------------------
// Same five launches as the fixed-shape pipeline in msm_run; what changes is the plan — half the windows, hence half the buckets
// for the reduction / aggregation trees and half the doublings (and marginal sums) of the host's Horner tail.  *done = false when it
// does not apply or a region overflowed (skewed scalars): the caller then runs the ordinary schedule, which recomputes everything.
template <class C> static int msm_run_fs_glv(bp_ctx* ctx, const BaseSegs& segs_in, const ScalSegs& d_scalars, size_t n, int scalars_mont, J4& result, bool& done) {
    typedef host::Grp<C> G;
    done = false;
    if constexpr (!C::HAS_GLV) { (void)ctx; (void)segs_in; (void)d_scalars; (void)n; (void)scalars_mont; (void)result; return BP_OK; }
    else {
    hipStream_t st = ctx->stream;
    constexpr int BITS = 128;                       // magnitudes of the halves (glv_split guarantees < 2^128 or flags the MSM)
    const size_t ne = 2 * n;                        // half-terms
    MsmPlan pl = msm_plan(ne, BITS);
    pl.n = (u32)n;                                  // scalars the partition kernel reads
    if (pl.W > MSM_MAXW || pl.B >= ctx->tune_msm_wsum_min) return BP_OK;
    const int bits_last = BITS - pl.c * (pl.W - 1);
    const int wbn = bits_last < pl.c - 1 ? pl.W - 1 : pl.W;
    if (wbn <= 0) return BP_OK;
    BinPlan bp; memset(&bp, 0, sizeof bp);
    {
        u32 nbin = 1, lg = 0;
        while ((size_t)nbin * 8192 < ne && nbin < (u32)pl.NB) { nbin <<= 1; lg++; }
        int LB = pl.c - 1 - (int)lg;
        while (LB > 11) { nbin <<= 1; LB--; }
        const double mu = (double)ne / nbin;
        const size_t cap = std::min<size_t>(ne, (size_t)(mu + 8.0 * std::sqrt(mu) + 64.0));
        const size_t lds_sort = (((size_t)1 << LB) + 8 + cap) * 4;
        if (!(ne < ((size_t)1 << (30 - LB)) && lds_sort <= 64 * 1024 && (size_t)wbn * nbin * cap < ((size_t)1 << 31))) return BP_OK;
        bp.LB = (u32)LB; bp.NBIN = nbin; bp.cap = (u32)cap; bp.wb = (u32)wbn; bp.glv = 1;
        bp.tpt = (u32)std::min<size_t>(16, std::max<size_t>(1, n / (256 * 512)));
        if (wbn < pl.W) {
            const size_t nb_top = std::min<size_t>((size_t)pl.NB, ((size_t)1 << std::max(bits_last, 0)) + 1);
            if (nb_top > 2049) return BP_OK;   // (k_msm_marginals_fs holds top_nb + 1 prefix counts in 2052 words)
            bp.top_nb = (u32)nb_top;
        }
    }
    if (!(bp.wb == (u32)pl.W || bp.top_nb > 0) || (size_t)bp.wb * bp.NBIN + 1 > MSM_FS_MAXBINS) return BP_OK;
    // slots of the narrow top window: the halves are not uniform below 2^128 (|k2| reaches 1.27 * 2^127, |k1| 1.08 * 2^127), so its
    // low buckets are fuller than a uniform spread: four times the mean
    SlotPlan sp; memset(&sp, 0, sizeof sp);
    size_t nslots = (size_t)bp.wb * bp.NBIN * bp.cap;
    for (int w = (int)bp.wb; w < pl.W; w++) {
        const size_t cap = std::min<size_t>(ne, 4 * ((ne + bp.top_nb - 1) / bp.top_nb) + 64);
        sp.base[w] = (u32)nslots; sp.cap[w] = (u32)cap;
        nslots += (size_t)bp.top_nb * cap;
    }
    const u32 chl_fs = fs_chunk_cap(ctx, (double)ne * bp.wb * (1.0 - std::ldexp(1.0, -pl.c)), (double)bp.wb * pl.NB, bp.wb < (u32)pl.W ? (double)ne : 0.0, (double)bp.top_nb);   // (entries per chunk; any value from 8)
    FsPlan fp; memset(&fp, 0, sizeof fp);
    fp.has_top = bp.wb < (u32)pl.W ? 1u : 0u;
    fp.nbins = bp.wb * bp.NBIN + fp.has_top;
    if (fp.has_top) { u32 t = bp.top_nb; while (t) { fp.top_bits++; t >>= 1; } }
    const size_t nwin = (size_t)pl.W;
    const u32 red_g = (size_t)bp.wb * pl.NB > 49152 ? 1u : 4u;
    const size_t maxch = (ne * nwin) / chl_fs + std::min<size_t>(ne * nwin, nwin * (size_t)pl.NB) + 64;
    fp.max_chunks = (u32)maxch;
    fp.top_parts = (u32)std::min<size_t>(MSM_TOP_PARTS_MAX, std::max<size_t>(4, ne >> 15));
    const size_t tc = (size_t)bp.wb * pl.c + (size_t)fp.top_bits * fp.top_parts;
    const size_t tb = tc * 96 + 64;
    if (nslots >= ((size_t)1 << 32) || maxch >= ((size_t)1 << 31)) return BP_OK;
    constexpr int NL = MSM_NLMAX;
    BPCHK(ctx->canon.ensure(n * MSM_GLV_WORDS * 4));
    BPCHK(ctx->hist.ensure_zeroed((size_t)pl.B * 4, st));
    BPCHK(ctx->totals.ensure_zeroed((NL + 2) * 4 + 4096, st));
    BPCHK(ctx->slots.ensure(nslots * 4));
    BPCHK(ctx->bin_cur.ensure_zeroed((size_t)pl.W * bp.NBIN * 4, st));
    BPCHK(ctx->boff.ensure((size_t)pl.B * 4));
    BPCHK(ctx->fs_bcnt.ensure((size_t)pl.B * 4));
    BPCHK(ctx->fs_loff.ensure((size_t)pl.B * 4));
    BPCHK(ctx->fs_binch.ensure((MSM_FS_MAXBINS + 1) * 4));
    BPCHK(ctx->fs_sums.ensure((size_t)bp.wb * pl.NB * 96));
    BPCHK(ctx->lvA.ensure(maxch * 96));
    BPCHK(ctx->Tbuf.ensure(tb));
    if (ctx->h_T_cap < tb) {
        if (ctx->h_T) HIPCHK(hipHostFree(ctx->h_T));
        HIPCHK(hipHostMalloc((void**)&ctx->h_T, tb + 4096));
        ctx->h_T_cap = tb + 4096;
    }
    BaseSegs segs = segs_in;
    segs.glv = 1;
    ScopedK total(ctx, BP_K_MSM_TOTAL);
    u32* d_over = ctx->totals.as<u32>() + (NL + 1);
    u32* d_info = ctx->Tbuf.as<u32>() + tc * 24;
    const int wg = std::max(1, (int)(12288 / bp.NBIN));
    const u32 gp = (u32)((n + (size_t)256 * bp.tpt - 1) / ((size_t)256 * bp.tpt));
    for (int wa = 0; wa < (int)bp.wb; wa += wg) {
        const int we = std::min<int>((int)bp.wb, wa + wg);
        hipLaunchKernelGGL(k_msm_bin_partition<C>, dim3(gp), dim3(256), ((size_t)(we - wa) * bp.NBIN + (wa == 0 ? bp.top_nb : 0)) * 4, st, d_scalars, ctx->canon.as<u32>(),
                           ctx->hist.as<u32>(), pl, scalars_mont, bp, sp, ctx->bin_cur.as<u32>(), ctx->slots.as<u32>(), d_over, wa, we, wa == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL(k_msm_bin_sort_fs, dim3(bp.NBIN, bp.wb + fp.has_top), dim3(256), (((size_t)1 << bp.LB) + 8 + bp.cap) * 4, st, ctx->slots.as<u32>(),
                       ctx->bin_cur.as<u32>(), ctx->hist.as<u32>(), ctx->boff.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(), ctx->fs_binch.as<u32>(),
                       d_over, pl, bp, sp, chl_fs);
    {
        ScopedK acc(ctx, BP_K_MSM_ACCUM_FS);
        hipLaunchKernelGGL(k_msm_accum_fs<C>, dim3((u32)((maxch + 255) / 256)), dim3(256), 0, st, segs, ctx->slots.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                           ctx->boff.as<u32>(), ctx->fs_binch.as<u32>(), ctx->lvA.as<u32>(), pl, bp, fp, chl_fs, d_info);
    }
    {
        ScopedK agg(ctx, BP_K_MSM_AGG);
        MSM_LAUNCH_REDUCE_FS(red_g, dim3((u32)(((size_t)bp.wb * pl.NB * red_g + 255) / 256)), ctx->lvA.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                           ctx->fs_binch.as<u32>(), ctx->fs_sums.as<u32>(), pl, bp, fp, chl_fs, red_g);
        MSM_LAUNCH_MARGINALS_FS(dim3((u32)tc), ctx->fs_sums.as<u32>(), ctx->lvA.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                           ctx->Tbuf.as<u32>(), pl, bp, fp, chl_fs, d_info, d_over);
    }
    HIPCHK(hipMemcpyAsync(ctx->h_T, ctx->Tbuf.p, tb, hipMemcpyDeviceToHost, st));
    total.stop();
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    const u32* info = (const u32*)((const uint8_t*)ctx->h_T + tc * 96);
    static const bool mtrace = getenv("ARKBP_MSM_TRACE") != nullptr;
    if (info[2] != 0) {
        if (mtrace) fprintf(stderr, "[msm-glv] n=%zu: overflow, the ordinary schedule takes over\n", n);
        return BP_OK;
    }
    J4 acc = G::inf();
    const u64* T = (const u64*)ctx->h_T;
    auto add_T = [&](size_t idx) {
        const u64* t = T + idx * 12;
        J4 p; memcpy(p.X.v, t, 32); memcpy(p.Y.v, t + 4, 32); memcpy(p.Z.v, t + 8, 32);
        if (!p.Z.is_zero()) acc = G::add(acc, p);
    };
    const int ngen = (int)bp.wb * pl.c;
    for (int j = ngen + (int)fp.top_bits - 1; j >= 0; j--) {
        acc = G::dbl(acc);
        if (j >= ngen) { for (u32 q = 0; q < fp.top_parts; q++) add_T((size_t)ngen + (size_t)(j - ngen) * fp.top_parts + q); }
        else add_T((size_t)j);
    }
    result = acc;
    done = true;
    if (mtrace) fprintf(stderr, "[msm-glv] n=%zu c=%d W=%d bins=%u chunks=%u/%u  marginals=%zu\n", n, pl.c, pl.W, fp.nbins, info[0], fp.max_chunks, tc);
    return BP_OK;
    }
}

// A rank's partial point -> the sum over the ranks of a sharded ctx (one small exchange): RCCL all-gather of the Jacobian partials
// inside the library, or the host's point-reduce callback.
template <class C> static int shard_point_reduce(bp_ctx* ctx, J4& part) {
    typedef host::Grp<C> G;
    if (ctx->nccl) {
        // RCCL point-reduce: group addition is not an RCCL reduce op -> all-gather of the partials + world-1 host additions.  The
        // partial travels as it is — Jacobian, 96 bytes — so no rank pays a field inversion per MSM just to ship it;
        // every rank adds the same points in the same order and normalises its own copy of the sum when the caller asks.
        uint64_t xyz[12];
        memcpy(xyz, part.X.v, 32); memcpy(xyz + 4, part.Y.v, 32); memcpy(xyz + 8, part.Z.v, 32);
        std::vector<uint64_t> all((size_t)ctx->shard_world * 12);
        BPCHK(ctx_native_allgather(ctx, xyz, 96, all.data()));
        J4 sum = G::inf();
        for (int r = 0; r < ctx->shard_world; r++) {
            J4 q; memcpy(q.X.v, &all[(size_t)r * 12], 32); memcpy(q.Y.v, &all[(size_t)r * 12 + 4], 32); memcpy(q.Z.v, &all[(size_t)r * 12 + 8], 32);
            sum = G::add(sum, q);
        }
        part = sum;
        return BP_OK;
    }
    if (!ctx->shard_cb) { g_err = "sharded mode: no point-reduce installed"; return BP_E_ARG; }
    A4 a = G::to_aff(part);      // (the callback's contract is an affine point: include/arkbp.h bp_point_reduce_cb)
    uint64_t xy[8]; memcpy(xy, a.x.v, 32); memcpy(xy + 4, a.y.v, 32);
    const int rc = ctx->shard_cb(ctx->shard_user, xy);
    if (rc) { g_err = "msm: the point-reduce callback failed"; return rc < 0 ? rc : BP_E_ARG; }
    memcpy(a.x.v, xy, 32); memcpy(a.y.v, xy + 4, 32);
    part = G::from_aff(a);
    return BP_OK;
}

template <class C> static int msm_run(bp_ctx* ctx, const BaseSegs& segs, const u32* d_scalars_one, size_t n, int scalars_mont, J4& result,
                                      int w_lo = 0, int w_hi = -1 /* window range for multi-GPU window sharding; default all */,
                                      int shard_mode = -1 /* -1: the ctx's mode (window partition + reduce when world > 1); 0: none (replicated);
                                                             2: the terms are this rank's own share: all windows, then the point-reduce */,
                                      const ScalSegs* sseg = nullptr /* the scalars as runs read in place (then d_scalars_one is unused) */) {
    ScalSegs d_scalars; memset(&d_scalars, 0, sizeof d_scalars);
    if (sseg) d_scalars = *sseg; else { d_scalars.nseg = 1; d_scalars.ptr[0] = d_scalars_one; d_scalars.start[0] = 0; d_scalars.start[1] = (u32)n; }
    typedef host::Grp<C> G;
    typedef host::Fld<typename C::Fq> F;
    result = G::inf();
    if (n == 0) return BP_OK;
    if (n >= (1u << 31)) { g_err = "msm: n too large"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    MsmPlan pl = msm_plan(n, C::Fr::BITS);
    const bool sharded = w_hi < 0 && ctx->shard_world > 1 && shard_mode != 0;
    if (sharded && shard_mode != 2) {   // contiguous block partition of the windows (the same rule as parallel.shard_range)
        const int base = pl.W / ctx->shard_world, rem = pl.W % ctx->shard_world, r = ctx->shard_rank;
        w_lo = r * base + std::min(r, rem);
        w_hi = w_lo + base + (r < rem ? 1 : 0);
    }
    auto finish_sharded = [&](J4& part) -> int {   // partial point -> sum over the ranks (one small exchange)
        if (!sharded) return BP_OK;
        return shard_point_reduce<C>(ctx, part);
    };
    if constexpr (C::HAS_GLV) {
        // mid-size MSMs on a curve with the endomorphism: the fixed-shape pipeline over GLV-split scalars (half the windows)
        static const bool no_glv = getenv("ARKBP_MSM_NOGLV") != nullptr, no_fs0 = getenv("ARKBP_MSM_NOFS") != nullptr;
        if (!no_glv && !no_fs0 && msm_use_quad(ctx) && !sharded && w_hi < 0 && ctx->shard_world == 1 && !segs.fixed_c4 && n >= std::max<size_t>(ctx->tune_msm_bin_min, ctx->tune_msm_glv_min) && n < ((size_t)1 << 27)) {
            bool done = false;
            BPCHK(msm_run_fs_glv<C>(ctx, segs, d_scalars, n, scalars_mont, result, done));
            if (done) return BP_OK;
        }
    }
    if (w_hi >= 0) { pl.w_lo = std::max(0, w_lo); pl.w_hi = std::min(pl.W, w_hi); }
    if (pl.w_lo >= pl.w_hi) return finish_sharded(result);  // this rank owns no window: the identity
    constexpr int NL = MSM_NLMAX;
    int nl = 2;  // levels 0..nl-1 can be needed: 16^(nl-1) >= n
    { u64 cap = MSM_CH; while (cap < n && nl < NL) { cap *= MSM_CH; nl++; } }   // sized for the smaller fan-in
    if (nl < 4) nl = 4;   // the special buckets of the top window meet one level after spl <= 2
    const int spl = n > ((size_t)1 << 17) ? 2 : 1;   // hot bucket ~ n/2 entries: keep the special kernel at <= 32 partials per lane
    const size_t Bp1 = (size_t)pl.B + 1;
    const u32 ntiles = (pl.B + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE;
    BPCHK(ctx->canon.ensure(n * 32));
    // hist, the bin cursors and the overflow flag are all-zero between MSMs: their last readers (k_msm_scan_apply, k_msm_bin_sort,
    // the host) restore that, so no memset is enqueued per MSM
    BPCHK(ctx->hist.ensure_zeroed(pl.B * 4, st));
    BPCHK(ctx->lvl_off.ensure(Bp1 * NL * 4));
    BPCHK(ctx->totals.ensure_zeroed((NL + 2) * 4 + (size_t)ntiles * (NL + 1) * 4, st));
    BPCHK(ctx->cursor.ensure(pl.B * 4));
    static const bool force_marginals = getenv("ARKBP_MSM_MARGINALS") != nullptr;   // A/B: the bit-marginal bucket aggregation everywhere
    const bool use_marginals = force_marginals || pl.B < ctx->tune_msm_wsum_min;
    const u32 nblk_ws = (u32)((pl.NB + 256 * MSM_SEG - 1) / (256 * MSM_SEG));   // workgroups per window of k_msm_window_sums
    const size_t tcount = use_marginals ? (size_t)pl.W * pl.c : (size_t)pl.W * nblk_ws;
    BPCHK(ctx->Tbuf.ensure(tcount * 96));
    if (!ctx->h_totals) HIPCHK(hipHostMalloc((void**)&ctx->h_totals, 64));
    const size_t tbytes = tcount * 96;
    if (ctx->h_T_cap < tbytes) {
        if (ctx->h_T) HIPCHK(hipHostFree(ctx->h_T));
        HIPCHK(hipHostMalloc((void**)&ctx->h_T, tbytes + 4096));
        ctx->h_T_cap = tbytes + 4096;
    }
    // slot plan of the one-pass sort; two-level (binned) sort for large MSMs: the full windows go to bin regions, the short
    // top window keeps its slots behind them
    if (pl.W > MSM_MAXW) { g_err = "msm: too many windows"; return BP_E_ARG; }
    BinPlan bp; memset(&bp, 0, sizeof bp);
    bool binned = false;
    {
        const int bits_last = C::Fr::BITS - pl.c * (pl.W - 1);
        const int wbn = bits_last < pl.c - 1 ? pl.W - 1 : pl.W;
        static const bool no_bins = getenv("ARKBP_MSM_NOBIN") != nullptr;
        if (n >= ctx->tune_msm_bin_min && wbn > 0 && !no_bins) {
            u32 nbin = 1, lg = 0;
            while ((size_t)nbin * 8192 < n && nbin < (u32)pl.NB) { nbin <<= 1; lg++; }
            // bins of up to 12 K entries (43 KB of LDS in the bin sort) where that keeps the MSM inside the fixed-shape pipeline's bin
            // limit: 2^20 < n <= 1.5 * 2^20, e.g. the 1.25 M-term MSM of a 4096-proof batch verification
            if ((size_t)wbn * nbin + 1 > MSM_FS_MAXBINS && nbin >= 2 && (size_t)(nbin / 2) * 12288 >= n && (size_t)wbn * (nbin / 2) + 1 <= MSM_FS_MAXBINS) { nbin >>= 1; lg--; }
            int LB = pl.c - 1 - (int)lg;
            while (LB > 11) { nbin <<= 1; LB--; }
            const double mu = (double)n / nbin;
            const size_t cap = std::min<size_t>(n, (size_t)(mu + 8.0 * std::sqrt(mu) + 64.0));
            const size_t lds_sort = (((size_t)1 << LB) + 4 + cap) * 4;
            if (n < ((size_t)1 << (31 - LB)) && lds_sort <= 64 * 1024 && (size_t)wbn * nbin * cap < ((size_t)1 << 31)) {
                bp.LB = (u32)LB; bp.NBIN = nbin; bp.cap = (u32)cap; bp.wb = (u32)wbn;
                static const int tpt_env = getenv("ARKBP_MSM_TPT") ? atoi(getenv("ARKBP_MSM_TPT")) : 0;
                bp.tpt = tpt_env > 0 ? (u32)tpt_env : (u32)std::min<size_t>(16, std::max<size_t>(1, n / (256 * 512)));
                if (wbn < pl.W) {   // same-address device atomics serialise (~170 ns each): count a narrow top window per workgroup
                    const size_t nb_top = std::min<size_t>((size_t)pl.NB, ((size_t)1 << std::max(bits_last, 0)) + 1);
                    if (nb_top <= 2048) bp.top_nb = (u32)nb_top;
                }
                binned = true;
            }
        }
    }
    auto make_slots = [&](SlotPlan& sp, int w_first, size_t first_slot) -> size_t {
        memset(&sp, 0, sizeof sp);
        size_t nslots = first_slot;
        for (int w = w_first; w < pl.W; w++) {
            const int bits_left = C::Fr::BITS - pl.c * w;  // scalar bits at or above this window's base
            size_t nb_eff = (size_t)pl.NB;
            if (bits_left < pl.c - 1) nb_eff = std::min<size_t>(nb_eff, ((size_t)1 << std::max(bits_left, 0)) + 1);
            size_t cap = std::min<size_t>(n, 2 * ((n + nb_eff - 1) / nb_eff) + 32);
            sp.base[w] = (u32)nslots; sp.cap[w] = (u32)cap;
            nslots += nb_eff * cap;
        }
        return nslots;
    };
    SlotPlan sp;
    u32* lvl = ctx->lvl_off.as<u32>();
    u32* d_tot = ctx->totals.as<u32>();
    u32* d_tiles = d_tot + (NL + 2);
    u32* d_over = d_tot + (NL + 1);
    const int TB = 256;
    const u32 gb = (u32)((n + TB - 1) / TB);
    // front end: digits, histogram, placement, scans; leaves the totals (and the overflow flag) in h_totals
    int chl = MSM_CHL;
    u32 b_gen = pl.B;   // buckets below b_gen go through the generic reduction tree; the rest (narrow top window) are special
    auto front_end = [&](bool bins) -> int {
        static const int chl_env = getenv("ARKBP_MSM_CHL") ? atoi(getenv("ARKBP_MSM_CHL")) : 0;
        chl = bins ? (chl_env >= 2 && chl_env <= 6 ? chl_env : MSM_CHL_BINNED) : MSM_CHL;
        static const bool no_special = getenv("ARKBP_MSM_NOSPECIAL") != nullptr;
        b_gen = (bins && bp.top_nb && bp.top_nb <= 16 && !no_special) ? (u32)bp.wb * (u32)pl.NB : pl.B;   // few, huge buckets only: a wider top window fits the generic tree
        const size_t nslots = make_slots(sp, bins ? (int)bp.wb : 0, bins ? (size_t)bp.wb * bp.NBIN * bp.cap : 0);
        if (nslots >= ((size_t)1 << 32)) { g_err = "msm: slot array too large"; return BP_E_ARG; }
        BPCHK(ctx->slots.ensure(nslots * 4));
        if (bins) {
            BPCHK(ctx->bin_cur.ensure_zeroed((size_t)pl.W * bp.NBIN * 4, st));
            BPCHK(ctx->boff.ensure((size_t)pl.B * 4));
            const int wg = std::max(1, (int)(12288 / bp.NBIN));   // windows per launch: 48 KiB of LDS counters
            const u32 gp = (u32)((n + (size_t)256 * bp.tpt - 1) / ((size_t)256 * bp.tpt));
            for (int wa = 0; wa < (int)bp.wb; wa += wg) {
                const int we = std::min<int>((int)bp.wb, wa + wg);
                hipLaunchKernelGGL(k_msm_bin_partition<C>, dim3(gp), dim3(256), ((size_t)(we - wa) * bp.NBIN + (wa == 0 ? bp.top_nb : 0)) * 4, st, d_scalars, ctx->canon.as<u32>(),
                                   ctx->hist.as<u32>(), pl, scalars_mont, bp, sp, ctx->bin_cur.as<u32>(), ctx->slots.as<u32>(), d_over, wa, we, wa == 0 ? 1 : 0);
            }
            hipLaunchKernelGGL(k_msm_bin_sort, dim3(bp.NBIN, pl.W), dim3(256), (((size_t)1 << bp.LB) + 4 + bp.cap) * 4, st, ctx->slots.as<u32>(),
                               ctx->bin_cur.as<u32>(), ctx->hist.as<u32>(), ctx->boff.as<u32>(), pl, bp, sp);
        } else {
            const u32 lds_hist = gb == 1 && (size_t)pl.B <= 12288 ? (u32)pl.B : 0u;   // one workgroup: count in LDS (48 KiB of counters at most)
            hipLaunchKernelGGL(k_msm_digits<C>, dim3(gb), dim3(TB), (size_t)lds_hist * 4, st, d_scalars, ctx->canon.as<u32>(), ctx->hist.as<u32>(), pl, scalars_mont, sp,
                               ctx->slots.as<u32>(), d_over, lds_hist);
        }
        hipLaunchKernelGGL(k_msm_scan_tiles, dim3(ntiles), dim3(256), 0, st, ctx->hist.as<u32>(), d_tiles, pl.B, nl, chl, b_gen, spl);
        hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(64), 0, st, d_tiles, ntiles, d_tot, lvl, pl.B);
        hipLaunchKernelGGL(k_msm_scan_apply, dim3(ntiles), dim3(256), 0, st, ctx->hist.as<u32>(), d_tiles, lvl, pl.B, nl, chl, b_gen, spl);
        HIPCHK(hipMemcpyAsync(ctx->h_totals, d_tot, (NL + 2) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx_stream_wait(ctx));
        HIPCHK(hipGetLastError());
        if (ctx->h_totals[NL + 1] != 0) HIPCHK(hipMemsetAsync(d_over, 0, 4, st));   // the overflow flag was raised: lower it for the next pass / MSM
        return BP_OK;
    };
    static const bool mtrace = getenv("ARKBP_MSM_TRACE") != nullptr;   // host-side phase times of every MSM on stderr
    // ---- the fixed-shape pipeline (msm.cuh 7): no host wait before the last kernel; falls through to the general path on overflow ----
    static const bool no_fs = getenv("ARKBP_MSM_NOFS") != nullptr;
    if (binned && use_marginals && !no_fs && (bp.wb == (u32)pl.W || bp.top_nb > 0) && (size_t)bp.wb * bp.NBIN + 1 <= MSM_FS_MAXBINS && !segs.fixed_c4) {
        const auto tfs = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_f0 = mtrace ? tfs() : 0;
        const u32 chl_fs = fs_chunk_cap(ctx, (double)n * bp.wb * (1.0 - std::ldexp(1.0, -pl.c)), (double)bp.wb * pl.NB, bp.wb < (u32)pl.W ? (double)n : 0.0, (double)bp.top_nb);   // (entries per chunk; any value from 8)
        FsPlan fp; memset(&fp, 0, sizeof fp);
        fp.has_top = bp.wb < (u32)pl.W ? 1u : 0u;
        fp.nbins = bp.wb * bp.NBIN + fp.has_top;
        if (fp.has_top) { u32 t = bp.top_nb; while (t) { fp.top_bits++; t >>= 1; } }
        const size_t nwin = (size_t)(pl.w_hi - pl.w_lo);
        const u32 red_g = (size_t)bp.wb * pl.NB > 49152 ? 1u : 4u;   // lanes per bucket of k_msm_reduce_fs: groups while the lanes fit the chip at once, one lane per bucket beyond
        const size_t maxch = (n * nwin) / chl_fs + std::min<size_t>(n * nwin, nwin * (size_t)pl.NB) + 64;
        fp.max_chunks = (u32)maxch;
        const size_t nslots = make_slots(sp, (int)bp.wb, (size_t)bp.wb * bp.NBIN * bp.cap);
        fp.top_parts = (u32)std::min<size_t>(MSM_TOP_PARTS_MAX, std::max<size_t>(4, n >> 15));
        const size_t tc = (size_t)bp.wb * pl.c + (size_t)fp.top_bits * fp.top_parts;
        const size_t tb = tc * 96 + 64;
        if (nslots < ((size_t)1 << 32) && maxch < ((size_t)1 << 31)) {
            BPCHK(ctx->slots.ensure(nslots * 4));
            BPCHK(ctx->bin_cur.ensure_zeroed((size_t)pl.W * bp.NBIN * 4, st));
            BPCHK(ctx->boff.ensure((size_t)pl.B * 4));
            BPCHK(ctx->fs_bcnt.ensure((size_t)pl.B * 4));
            BPCHK(ctx->fs_loff.ensure((size_t)pl.B * 4));
            BPCHK(ctx->fs_binch.ensure((MSM_FS_MAXBINS + 1) * 4));
            BPCHK(ctx->fs_sums.ensure((size_t)bp.wb * pl.NB * 96));
            BPCHK(ctx->lvA.ensure(maxch * 96));
            BPCHK(ctx->Tbuf.ensure(tb));
            if (ctx->h_T_cap < tb) {
                if (ctx->h_T) HIPCHK(hipHostFree(ctx->h_T));
                HIPCHK(hipHostMalloc((void**)&ctx->h_T, tb + 4096));
                ctx->h_T_cap = tb + 4096;
            }
            ScopedK total(ctx, BP_K_MSM_TOTAL);
            u32* d_over = ctx->totals.as<u32>() + (MSM_NLMAX + 1);
            u32* d_info = ctx->Tbuf.as<u32>() + tc * 24;
            const int wg = std::max(1, (int)(12288 / bp.NBIN));
            const u32 gp = (u32)((n + (size_t)256 * bp.tpt - 1) / ((size_t)256 * bp.tpt));
            for (int wa = 0; wa < (int)bp.wb; wa += wg) {
                const int we = std::min<int>((int)bp.wb, wa + wg);
                hipLaunchKernelGGL(k_msm_bin_partition<C>, dim3(gp), dim3(256), ((size_t)(we - wa) * bp.NBIN + (wa == 0 ? bp.top_nb : 0)) * 4, st, d_scalars, ctx->canon.as<u32>(),
                                   ctx->hist.as<u32>(), pl, scalars_mont, bp, sp, ctx->bin_cur.as<u32>(), ctx->slots.as<u32>(), d_over, wa, we, wa == 0 ? 1 : 0);
            }
            hipLaunchKernelGGL(k_msm_bin_sort_fs, dim3(bp.NBIN, bp.wb + fp.has_top), dim3(256), (((size_t)1 << bp.LB) + 8 + bp.cap) * 4, st, ctx->slots.as<u32>(),
                               ctx->bin_cur.as<u32>(), ctx->hist.as<u32>(), ctx->boff.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(), ctx->fs_binch.as<u32>(),
                               d_over, pl, bp, sp, chl_fs);
            {
                ScopedK acc(ctx, BP_K_MSM_ACCUM_FS);
                hipLaunchKernelGGL(k_msm_accum_fs<C>, dim3((u32)((maxch + 255) / 256)), dim3(256), 0, st, segs, ctx->slots.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                                   ctx->boff.as<u32>(), ctx->fs_binch.as<u32>(), ctx->lvA.as<u32>(), pl, bp, fp, chl_fs, d_info);
            }
            {
            ScopedK agg(ctx, BP_K_MSM_AGG);
            MSM_LAUNCH_REDUCE_FS(red_g, dim3((u32)(((size_t)bp.wb * pl.NB * red_g + 255) / 256)), ctx->lvA.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                               ctx->fs_binch.as<u32>(), ctx->fs_sums.as<u32>(), pl, bp, fp, chl_fs, red_g);
            MSM_LAUNCH_MARGINALS_FS(dim3((u32)tc), ctx->fs_sums.as<u32>(), ctx->lvA.as<u32>(), ctx->fs_bcnt.as<u32>(), ctx->fs_loff.as<u32>(),
                           ctx->Tbuf.as<u32>(), pl, bp, fp, chl_fs, d_info, d_over);
            }
            HIPCHK(hipMemcpyAsync(ctx->h_T, ctx->Tbuf.p, tb, hipMemcpyDeviceToHost, st));
            total.stop();
            const double t_f1 = mtrace ? tfs() : 0;
            HIPCHK(ctx_stream_wait(ctx));
            HIPCHK(hipGetLastError());
            const double t_f2 = mtrace ? tfs() : 0;
            const u32* info = (const u32*)((const uint8_t*)ctx->h_T + tc * 96);
            if (info[2] == 0) {
                J4 acc = G::inf();
                const u64* T = (const u64*)ctx->h_T;
                auto add_T = [&](size_t idx) {
                    const u64* t = T + idx * 12;
                    J4 p; memcpy(p.X.v, t, 32); memcpy(p.Y.v, t + 4, 32); memcpy(p.Z.v, t + 8, 32);
                    if (!p.Z.is_zero()) acc = G::add(acc, p);
                };
                const int ngen = (int)bp.wb * pl.c;
                for (int j = ngen + (int)fp.top_bits - 1; j >= 0; j--) {
                    acc = G::dbl(acc);
                    if (j >= ngen) { for (u32 q = 0; q < fp.top_parts; q++) add_T((size_t)ngen + (size_t)(j - ngen) * fp.top_parts + q); }
                    else add_T((size_t)j);
                }
                result = acc;
                if (mtrace) fprintf(stderr, "[msm-fs] n=%zu c=%d W=%d bins=%u chunks=%u/%u  enqueue %.1f us  wait %.1f us  host tail %.1f us\n", n, pl.c, pl.W, fp.nbins, info[0],
                                    fp.max_chunks, (t_f1 - t_f0) * 1e6, (t_f2 - t_f1) * 1e6, (tfs() - t_f2) * 1e6);
                return finish_sharded(result);
            }
            if (mtrace) fprintf(stderr, "[msm-fs] n=%zu: overflow, the general path takes over\n", n);
        }
    }
    ScopedK total(ctx, BP_K_MSM_TOTAL);
    const auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_m0 = mtrace ? tnow() : 0;
    BPCHK(front_end(binned));
    const double t_m1 = mtrace ? tnow() : 0;
    if (binned && ctx->h_totals[NL + 1] != 0) {   // a bin region overflowed (skewed scalars): one-pass slots for every window
        binned = false;
        BPCHK(front_end(false));
    }
    const bool slotted = ctx->h_totals[NL + 1] == 0;
    const u32* entries_ptr = ctx->slots.as<u32>();
    if (!slotted) {  // some bucket outgrew its slots (skewed scalars): exact counting-sort scatter
        BPCHK(ctx->entries.ensure(n * pl.W * 4));
        HIPCHK(hipMemcpyAsync(ctx->cursor.p, lvl, pl.B * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_msm_scatter, dim3(gb), dim3(TB), 0, st, ctx->canon.as<u32>(), ctx->cursor.as<u32>(), ctx->entries.as<u32>(), pl);
        entries_ptr = ctx->entries.as<u32>();
    }
    const u32* tot = ctx->h_totals;
    const u32 maxcnt = tot[NL];
    if (tot[0] == 0) { total.stop(); return finish_sharded(result); }  // every digit zero: the identity
    // levels: 1 = chunks of entries; k >= 2 = chunks of level k-1 partials; stop when a bucket holds <= 1
    int K = 1;
    { u64 cap = (u64)1 << chl; while (cap < maxcnt) { cap <<= chl; K++; } }
    const bool special = b_gen < pl.B;
    if (special && K < spl + 1) K = spl + 1;   // the special buckets' partials meet in level spl + 1
    if (K >= nl) { g_err = "msm: bucket population exceeds the reduction depth"; return BP_E_ARG; }
    BPCHK(ctx->lvA.ensure((size_t)tot[1] * 96));
    if (K >= 2) BPCHK(ctx->lvB.ensure((size_t)tot[2] * 96));
    static const bool accum_wave = getenv("ARKBP_MSM_ACCUM") && !strcmp(getenv("ARKBP_MSM_ACCUM"), "wave");   // A/B: one wavefront per bucket
    if (accum_wave && !special) {
        // the bucket sums go straight to the last level's slots (<= 1 per bucket): no tree above
        ScopedK acc(ctx, BP_K_MSM_ACCUM);
        hipLaunchKernelGGL(k_msm_accum_wave<C>, dim3((pl.B + 3) / 4), dim3(256), 0, st, segs, entries_ptr, lvl, lvl + Bp1 * K, ctx->lvA.as<u32>(), pl.B,
                           slotted ? (binned ? 2 : 1) : 0, sp, (u32)pl.NB, ctx->boff.as<u32>());
        acc.stop();
        const u32* last_off = lvl + Bp1 * K;
        if (use_marginals) hipLaunchKernelGGL(k_msm_marginals<C>, dim3(pl.W, pl.c), dim3(256), 0, st, ctx->lvA.as<u32>(), last_off, ctx->Tbuf.as<u32>(), pl);
        else hipLaunchKernelGGL(k_msm_window_sums<C>, dim3(nblk_ws, pl.W), dim3(256), 0, st, ctx->lvA.as<u32>(), last_off, ctx->Tbuf.as<u32>(), pl, nblk_ws);
    } else {
    {
        ScopedK acc(ctx, BP_K_MSM_ACCUM);
        hipLaunchKernelGGL(k_msm_accum<C>, dim3((tot[1] + TB - 1) / TB), dim3(TB), 0, st, segs, entries_ptr, lvl, lvl + Bp1, ctx->lvA.as<u32>(), pl.B,
                           tot[1], slotted ? (binned ? 2 : 1) : 0, sp, (u32)pl.NB, ctx->boff.as<u32>(), chl);
    }
    u32* cur = ctx->lvA.as<u32>();
    u32* nxt = ctx->lvB.as<u32>();
    ScopedK agg(ctx, BP_K_MSM_AGG);
    for (int k = 2; k <= K; k++) {
        hipLaunchKernelGGL(k_msm_reduce<C>, dim3((tot[k] + TB - 1) / TB), dim3(TB), 0, st, cur, lvl + Bp1 * (k - 1), lvl + Bp1 * k, nxt, pl.B, tot[k], chl,
                           (k == spl + 1 && special) ? b_gen : pl.B);
        if (k == spl + 1 && special)
            hipLaunchKernelGGL(k_msm_reduce_special<C>, dim3(bp.top_nb), dim3(256), 0, st, cur, lvl + Bp1 * spl, lvl + Bp1 * (spl + 1), nxt, b_gen, pl.B);
        u32* t = cur; cur = nxt; nxt = t;
    }
    if (use_marginals) hipLaunchKernelGGL(k_msm_marginals<C>, dim3(pl.W, pl.c), dim3(256), 0, st, cur, lvl + Bp1 * K, ctx->Tbuf.as<u32>(), pl);
    else hipLaunchKernelGGL(k_msm_window_sums<C>, dim3(nblk_ws, pl.W), dim3(256), 0, st, cur, lvl + Bp1 * K, ctx->Tbuf.as<u32>(), pl, nblk_ws);
    agg.stop();
    }
    HIPCHK(hipMemcpyAsync(ctx->h_T, ctx->Tbuf.p, tbytes, hipMemcpyDeviceToHost, st));
    total.stop();
    const double t_m2 = mtrace ? tnow() : 0;
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    const double t_m3 = mtrace ? tnow() : 0;
    J4 acc = G::inf();
    const u64* T = (const u64*)ctx->h_T;
    auto add_T = [&](size_t idx) {
        const u64* t = T + idx * 12;
        J4 p; memcpy(p.X.v, t, 32); memcpy(p.Y.v, t + 4, 32); memcpy(p.Z.v, t + 8, 32);
        if (!p.Z.is_zero()) acc = G::add(acc, p);
    };
    if (use_marginals) {
        // host tail: sum_{w,k} 2^(c*w+k) T[w][k], Horner from the top bit down
        for (int j = pl.W * pl.c - 1; j >= 0; j--) { acc = G::dbl(acc); add_T((size_t)j); }
    } else {
        // host tail: sum_w 2^(c*w) * (sum_j T[w][j]), Horner over the windows (c doublings of ONE point between them)
        for (int w = pl.W - 1; w >= 0; w--) {
            for (int d = 0; d < pl.c; d++) acc = G::dbl(acc);
            for (u32 j = 0; j < nblk_ws; j++) add_T((size_t)w * nblk_ws + j);
        }
    }
    (void)sizeof(F);
    result = acc;
    if (mtrace) fprintf(stderr, "[msm] n=%zu c=%d W=%d K=%d  front(enqueue+sync) %.1f us  back enqueue %.1f us  back wait %.1f us  host tail %.1f us\n", n, pl.c, pl.W, K,
                        (t_m1 - t_m0) * 1e6, (t_m2 - t_m1) * 1e6, (t_m3 - t_m2) * 1e6, (tnow() - t_m3) * 1e6);
    return finish_sharded(result);
}

// ---- fixed-base MSM over the generator tables (msm.cuh 1c) -----------------------------------------------------------------
static constexpr u32 FB_ROWS = 65;   // rows at bit positions 0, 4, 8, .., 256: any window width c that is a multiple of 4 finds its rows (the last one serves the carry window)
// A run of generator-table bases of a fixed-base MSM: which table (0 = G, 1 = H, 2 = PedersenGens {B, B_blinding}), first index, count
struct FbRun { int table; size_t first, count; size_t stride = 1; };   // stride: base j of the run is table[first + j * stride] (an index-cyclic slice)
template <class C> static int fb_tables_build(bp_ctx* ctx, size_t cap) {
    hipStream_t st = ctx->stream;
    if (cap == 0 || cap > ctx->gens_cap) { g_err = "msm tables: more bases than installed generators"; return BP_E_GENS_LENGTH; }
    if (!ctx->d_G.owned) { g_err = "msm tables: build them on the ctx that owns the generator tables, then bp_gens_share"; return BP_E_ARG; }
    ctx->fb_cap = 0;
    BPCHK(ctx->fb_G.ensure_exact((size_t)FB_ROWS * cap * 64)); BPCHK(ctx->fb_H.ensure_exact((size_t)FB_ROWS * cap * 64)); BPCHK(ctx->fb_pc.ensure((size_t)FB_ROWS * 2 * 64));
    const size_t slab = std::min<size_t>(cap, (size_t)1 << 18);   // bounds the Jacobian scratch: 65 rows x 2^18 x 96 B = 1.6 GB
    DevBuf tmp, pref, outb;
    BPCHK(tmp.ensure(FB_ROWS * slab * 96)); BPCHK(pref.ensure(FB_ROWS * slab * 32)); BPCHK(outb.ensure(FB_ROWS * slab * 64));
    auto build = [&](const u32* gens, u32* table, size_t count, size_t table_cap) -> int {
        for (size_t lo = 0; lo < count; lo += slab) {
            const size_t m = std::min(slab, count - lo);
            const u32 gb = (u32)((m + 255) / 256);
            hipLaunchKernelGGL(k_msm_fb_rows<C>, dim3(gb), dim3(256), 0, st, gens + lo * 16, tmp.as<u32>(), (u32)m, FB_ROWS);
            hipLaunchKernelGGL(k_ftab_normalize<C>, dim3(gb), dim3(256), 0, st, tmp.as<u32>(), pref.as<u32>(), outb.as<u32>(), (u32)m, FB_ROWS);
            HIPCHK(hipGetLastError());
            for (u32 r = 0; r < FB_ROWS; r++)   // [r][i] of the slab -> [r][lo + i] of the table
                HIPCHK(hipMemcpyAsync(table + ((size_t)r * table_cap + lo) * 16, outb.as<u32>() + (size_t)r * m * 16, m * 64, hipMemcpyDeviceToDevice, st));
        }
        return BP_OK;
    };
    BPCHK(build(ctx->d_G.as<u32>(), ctx->fb_G.as<u32>(), cap, cap));
    BPCHK(build(ctx->d_H.as<u32>(), ctx->fb_H.as<u32>(), cap, cap));
    BPCHK(build(ctx->d_pc.as<u32>(), ctx->fb_pc.as<u32>(), 2, 2));
    HIPCHK(ctx_stream_wait(ctx));
    tmp.release(); pref.release(); outb.release();
    ctx->fb_cap = cap;
    return BP_OK;
}
// sum_t s_t * Base(t) for bases given as runs of the generator tables; *done = false when the fixed-base path does not apply
// (no tables, a run beyond them, skewed scalars overflowing a bin region, window-sharded ctx): the caller then runs the
// ordinary MSM over the same bases.
template <class C>
static int msm_fixed_run_one(bp_ctx* ctx, const FbRun* runs, int nruns, const ScalSegs& ssegs, size_t n, int scalars_mont, J4& result, bool& done, int shard_mode) {
    typedef host::Grp<C> G;
    done = false;
    static const bool fbtrace = getenv("ARKBP_FB_TRACE") != nullptr;
#define FBX(k) do { if (fbtrace) fprintf(stderr, "[fixed-base] declined at exit %d (n = %zu, runs = %d, world = %d, mode = %d)\n", k, n, nruns, ctx->shard_world, shard_mode); return BP_OK; } while (0)
    const bool reduce_after = ctx->shard_world > 1 && shard_mode == 2;
    if (ctx->shard_world > 1 && !reduce_after) FBX(1);
    static const bool off = getenv("ARKBP_MSM_NOFIXED") != nullptr;   // A/B switch
    // pays from ~2^20 terms (2^21: all kernels 5.6 -> 4.5 ms; at 2^19 and below the single latency-bound aggregation pass costs more
    // than the saved additions: tools/exp_msm_gens.py)
    if (off || !ctx->fb_cap || n < std::max<size_t>(ctx->tune_msm_fixed_min, 4096) || n >= ((size_t)1 << 24) || nruns > MSM_MAXSEG) FBX(2);
    BaseSegs segs; memset(&segs, 0, sizeof segs);
    u32 at = 0;
    for (int k = 0; k < nruns; k++) {
        const size_t cap = runs[k].table == 2 ? 2 : ctx->fb_cap;
        if (runs[k].count && runs[k].first + (runs[k].count - 1) * runs[k].stride + 1 > cap) FBX(3);
        const DevBuf& tb = runs[k].table == 0 ? ctx->fb_G : runs[k].table == 1 ? ctx->fb_H : ctx->fb_pc;
        segs.ptr[k] = (const u32*)tb.p + runs[k].first * 16;
        segs.row_words[k] = (u64)cap * 16;
        segs.stride[k] = (u32)runs[k].stride;
        segs.start[k] = at; at += (u32)runs[k].count;
    }
    segs.start[nruns] = at; segs.nseg = nruns;
    if (at != n) { g_err = "msm_fixed: runs and term count differ"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    // window width: a multiple of 4 (the table rows); cost = mixed adds (n * W) + bucket aggregation (~6 add-equivalents per bucket)
    const int bits = C::Fr::BITS;
    int bc = 8; double best = 1e300;
    for (int c = 8; c <= 24; c += 4) {
        const double W = bits / c + 1, NBk = std::ldexp(1.0, c - 1);
        const double cost = n * W * 1.1 + NBk * 6.0;
        if (cost < best) { best = cost; bc = c; }
    }
    static const int c_env = getenv("ARKBP_MSM_FIXED_C") ? atoi(getenv("ARKBP_MSM_FIXED_C")) : 0;
    if (c_env >= 8 && c_env <= 24 && c_env % 4 == 0) bc = c_env;
    MsmPlan pl; pl.c = bc; pl.W = bits / bc + 1; pl.NB = 1 << (bc - 1); pl.B = (u32)pl.NB; pl.n = (u32)n; pl.w_lo = 0; pl.w_hi = pl.W;
    if ((u32)(pl.W - 1) * (u32)(bc / 4) >= FB_ROWS) FBX(4);
    u32 tbits = 1; while (((size_t)1 << tbits) < n) tbits++;   // (term indices are < n: n itself need not fit)
    u32 wbits = 1; while ((1 << wbits) < pl.W) wbits++;
    const u32 vbits = tbits + wbits;
    // bins: ~6 K entries each; the entry word must hold vterm | fine bucket | sign
    const double entries = (double)n * (pl.W - 1) + (double)n * 0.5;
    u32 nbin = 1, lg = 0;
    while ((double)nbin * 6000.0 < entries && nbin < (u32)pl.NB) { nbin <<= 1; lg++; }
    int LB = bc - 1 - (int)lg;
    while (LB > 0 && (LB > 11 || vbits + (u32)LB + 1 > 32)) { nbin <<= 1; LB--; }
    if (vbits + (u32)LB + 1 > 32 || (size_t)nbin * 4 > 64 * 1024) FBX(5);
    // expected load of the fullest bin: every full window spreads n(1 - 2^-c) digits uniformly; the top window (tb bits) only reaches
    // the lowest 2^(tb-1) buckets
    const int tb = bits - bc * (pl.W - 1);
    double mu = (double)n * (pl.W - 1) / nbin;
    if (tb > 0) { const double reach = std::max(1.0, std::ldexp(1.0, tb - 1) / (double)((u32)1 << LB)); mu += (double)n / std::min<double>(reach, nbin); }
    const size_t cap = (size_t)(mu + 8.0 * std::sqrt(mu) + 64.0);
    if ((((size_t)1 << LB) + 4 + cap) * 4 > 64 * 1024 || (size_t)nbin * cap >= ((size_t)1 << 31)) FBX(6);
    BinPlan bp; memset(&bp, 0, sizeof bp);
    bp.LB = (u32)LB; bp.NBIN = nbin; bp.cap = (u32)cap; bp.wb = 1; bp.tpt = (u32)std::min<size_t>(16, std::max<size_t>(1, n / (256 * 512)));
    SlotPlan sp; memset(&sp, 0, sizeof sp);
    constexpr int NL = MSM_NLMAX;
    int nl = 2;
    { u64 capl = MSM_CH; while (capl < entries + 1 && nl < NL) { capl *= MSM_CH; nl++; } }
    if (nl < 4) nl = 4;
    const size_t Bp1 = (size_t)pl.B + 1;
    const u32 ntiles = (pl.B + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE;
    const u32 nblk_ws = (u32)((pl.NB + 256 * MSM_SEG - 1) / (256 * MSM_SEG));
    BPCHK(ctx->canon.ensure(n * 32)); BPCHK(ctx->hist.ensure_zeroed((size_t)pl.B * 4, st)); BPCHK(ctx->lvl_off.ensure(Bp1 * NL * 4));
    BPCHK(ctx->totals.ensure_zeroed((NL + 2) * 4 + (size_t)ntiles * (NL + 1) * 4, st)); BPCHK(ctx->bin_cur.ensure_zeroed((size_t)nbin * 4, st)); BPCHK(ctx->boff.ensure((size_t)pl.B * 4));
    BPCHK(ctx->slots.ensure((size_t)nbin * cap * 4)); BPCHK(ctx->Tbuf.ensure(((size_t)nblk_ws + 1) * 96));
    if (!ctx->h_totals) HIPCHK(hipHostMalloc((void**)&ctx->h_totals, 64));
    if (ctx->h_T_cap < 96) { if (ctx->h_T) HIPCHK(hipHostFree(ctx->h_T)); HIPCHK(hipHostMalloc((void**)&ctx->h_T, 4096)); ctx->h_T_cap = 4096; }
    ScopedK total(ctx, BP_K_MSM_TOTAL);
    u32* lvl = ctx->lvl_off.as<u32>();
    u32* d_tot = ctx->totals.as<u32>();
    u32* d_tiles = d_tot + (NL + 2);
    u32* d_over = d_tot + (NL + 1);
    const int chl = MSM_CHL_BINNED;
    const u32 gp = (u32)((n + (size_t)256 * bp.tpt - 1) / ((size_t)256 * bp.tpt));
    hipLaunchKernelGGL(k_msm_fb_partition<C>, dim3(gp), dim3(256), (size_t)nbin * 4, st, ssegs, ctx->canon.as<u32>(), pl, scalars_mont, bp, tbits, ctx->bin_cur.as<u32>(),
                       ctx->slots.as<u32>(), d_over);
    MsmPlan pl1 = pl; pl1.W = 1;   // one bucket set from here on
    hipLaunchKernelGGL(k_msm_bin_sort, dim3(nbin, 1), dim3(256), (((size_t)1 << LB) + 4 + cap) * 4, st, ctx->slots.as<u32>(), ctx->bin_cur.as<u32>(), ctx->hist.as<u32>(),
                       ctx->boff.as<u32>(), pl1, bp, sp);
    hipLaunchKernelGGL(k_msm_scan_tiles, dim3(ntiles), dim3(256), 0, st, ctx->hist.as<u32>(), d_tiles, pl.B, nl, chl, pl.B, 1);
    hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(64), 0, st, d_tiles, ntiles, d_tot, lvl, pl.B);
    hipLaunchKernelGGL(k_msm_scan_apply, dim3(ntiles), dim3(256), 0, st, ctx->hist.as<u32>(), d_tiles, lvl, pl.B, nl, chl, pl.B, 1);
    HIPCHK(hipMemcpyAsync(ctx->h_totals, d_tot, (NL + 2) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    const u32* tot = ctx->h_totals;
    if (tot[NL + 1] != 0) {   // a bin region overflowed (skewed scalars): the ordinary MSM handles those
        HIPCHK(hipMemsetAsync(d_over, 0, 4, st));
        total.stop();
        FBX(7);
    }
    result = G::inf();
    done = true;
    if (tot[0] == 0) { total.stop(); return BP_OK; }
    const u32 maxcnt = tot[NL];
    int K = 1;
    { u64 capl = (u64)1 << chl; while (capl < maxcnt) { capl <<= chl; K++; } }
    if (K >= nl) { done = false; total.stop(); return BP_OK; }
    BPCHK(ctx->lvA.ensure((size_t)tot[1] * 96));
    if (K >= 2) BPCHK(ctx->lvB.ensure((size_t)tot[2] * 96));
    segs.fixed_c4 = (u32)(bc / 4); segs.tbits = tbits;
    {
        ScopedK acc(ctx, BP_K_MSM_ACCUM);
        hipLaunchKernelGGL(k_msm_accum<C>, dim3((tot[1] + 255) / 256), dim3(256), 0, st, segs, ctx->slots.as<u32>(), lvl, lvl + Bp1, ctx->lvA.as<u32>(), pl.B, tot[1], 2, sp,
                           (u32)pl.NB, ctx->boff.as<u32>(), chl);
    }
    u32* cur = ctx->lvA.as<u32>();
    u32* nxt = ctx->lvB.as<u32>();
    {
    ScopedK agg(ctx, BP_K_MSM_AGG);
    for (int k = 2; k <= K; k++) {
        hipLaunchKernelGGL(k_msm_reduce<C>, dim3((tot[k] + 255) / 256), dim3(256), 0, st, cur, lvl + Bp1 * (k - 1), lvl + Bp1 * k, nxt, pl.B, tot[k], chl, pl.B);
        u32* t2 = cur; cur = nxt; nxt = t2;
    }
    hipLaunchKernelGGL(k_msm_window_sums<C>, dim3(nblk_ws, 1), dim3(256), 0, st, cur, lvl + Bp1 * K, ctx->Tbuf.as<u32>(), pl1, nblk_ws);
    hipLaunchKernelGGL(k_msm_sum_partials<C>, dim3(1), dim3(256), 0, st, ctx->Tbuf.as<u32>(), nblk_ws, ctx->Tbuf.as<u32>() + (size_t)nblk_ws * 24);
    }
    HIPCHK(hipMemcpyAsync(ctx->h_T, ctx->Tbuf.as<u32>() + (size_t)nblk_ws * 24, 96, hipMemcpyDeviceToHost, st));
    total.stop();
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    const u64* T = (const u64*)ctx->h_T;
    J4 pnt; memcpy(pnt.X.v, T, 32); memcpy(pnt.Y.v, T + 4, 32); memcpy(pnt.Z.v, T + 8, 32);
    if (!pnt.Z.is_zero()) result = pnt;   // no Horner tail: the rows already carry the powers of two
    return BP_OK;
}
#undef FBX
// The entry word of the fixed-base sort packs (window, term, fine bucket, sign) into 32 bits: one call holds up to 2^22 terms.  Longer
// MSMs — the commitments of a 2^22-constraint proof have 2^23 + 1 terms — go through in pieces of the logical term range (runs and
// scalar segments cut alike), the partial points added on the host: the schedule's 13 mixed additions per term instead of the
// ordinary 17-18 at every size.  On a sharded ctx (shard_mode 2: the runs are THIS RANK'S share of the terms — all windows here, the
// single-GPU schedule at 1/world of the terms) ONE point-reduce follows, whatever the number of pieces; any other mode steps aside.
static constexpr size_t FB_MAX_TERMS = (size_t)1 << 22;
template <class C>
static int msm_fixed_run(bp_ctx* ctx, const FbRun* runs, int nruns, const ScalSegs& ssegs, size_t n, int scalars_mont, J4& result, bool& done, int shard_mode = -1) {
    typedef host::Grp<C> G;
    done = false;
    const bool reduce_after = ctx->shard_world > 1 && shard_mode == 2;
    if (n <= FB_MAX_TERMS) {
        BPCHK(msm_fixed_run_one<C>(ctx, runs, nruns, ssegs, n, scalars_mont, result, done, shard_mode));
    } else {
        if (ctx->shard_world > 1 && !reduce_after) return BP_OK;
        const size_t pieces = (n + FB_MAX_TERMS - 1) / FB_MAX_TERMS, per = (n + pieces - 1) / pieces;
        J4 acc = G::inf();
        bool all = true;
        for (size_t t0 = 0; t0 < n && all; t0 += per) {
            const size_t t1 = std::min(n, t0 + per);
            FbRun rs[MSM_MAXSEG]; int nr = 0;
            ScalSegs ss; memset(&ss, 0, sizeof ss);
            size_t at = 0;
            for (int k = 0; k < nruns; k++) {          // the runs' share of [t0, t1)
                const size_t lo = std::max(at, t0), hi = std::min(at + runs[k].count, t1);
                if (lo < hi) { rs[nr] = runs[k]; rs[nr].first = runs[k].first + (lo - at) * runs[k].stride; rs[nr].count = hi - lo; nr++; }
                at += runs[k].count;
            }
            u32 sat = 0;
            for (int k = 0; k < ssegs.nseg; k++) {      // the scalar segments' share
                const size_t lo = std::max<size_t>(ssegs.start[k], t0), hi = std::min<size_t>(ssegs.start[k + 1], t1);
                if (lo < hi) { ss.ptr[ss.nseg] = ssegs.ptr[k] + (lo - ssegs.start[k]) * 8; ss.start[ss.nseg] = sat; sat += (u32)(hi - lo); ss.nseg++; }
            }
            ss.start[ss.nseg] = sat;
            J4 part; bool ok = false;
            BPCHK(msm_fixed_run_one<C>(ctx, rs, nr, ss, t1 - t0, scalars_mont, part, ok, shard_mode));
            if (!ok) all = false; else acc = G::add(acc, part);
        }
        if (all) { result = acc; done = true; }
    }
    if (!done) return BP_OK;
    ctx->fb_runs++;
    if (reduce_after) { ctx->fb_runs_sharded++; return shard_point_reduce<C>(ctx, result); }
    return BP_OK;
}

// MSM over the resident generator tables (no base upload): bases = G[off..off+n) (if use_G) || H[off..off+n) (if use_H) || extras
struct MsmLatencyScope { bp_ctx* c; bool prev; MsmLatencyScope(bp_ctx* c_) : c(c_), prev(c_->msm_latency_first) { c->msm_latency_first = true; } ~MsmLatencyScope() { c->msm_latency_first = prev; } };
template <class C> static int msm_gens_entry(bp_ctx* c, int use_G, int use_H, size_t off, size_t n, const uint64_t* extra_xy, size_t n_extra,
                                             const uint64_t* scalars, int canonical, uint64_t out_xy[8]) {
    MsmLatencyScope latency(c);
    const size_t ng = use_G ? n : 0, nh = use_H ? n : 0, total = ng + nh + n_extra;
    if (total == 0) { memset(out_xy, 0, 64); return BP_OK; }
    BPCHK(c->io_scal.ensure(total * 32));
    HIPCHK(hipMemcpyAsync(c->io_scal.p, scalars, total * 32, hipMemcpyHostToDevice, c->stream));
    if (!n_extra && (ng || nh)) {   // only generator-table bases: the fixed-base schedule when the rows are installed (bp_gens_msm_tables)
        FbRun runs[2]; int nr = 0;
        if (ng) runs[nr++] = FbRun{0, off, ng};
        if (nh) runs[nr++] = FbRun{1, off, nh};
        ScalSegs ss; memset(&ss, 0, sizeof ss);
        ss.nseg = 1; ss.ptr[0] = c->io_scal.as<u32>(); ss.start[0] = 0; ss.start[1] = (u32)total;
        J4 r; bool done = false;
        BPCHK(msm_fixed_run<C>(c, runs, nr, ss, total, canonical ? 0 : 1, r, done));
        if (done) {
            A4 a = host::Grp<C>::to_aff(r);
            memcpy(out_xy, a.x.v, 32); memcpy(out_xy + 4, a.y.v, 32);
            if (c->profiling) collect_timers(c);
            return BP_OK;
        }
    }
    if (n_extra) {
        BPCHK(c->io_pts.ensure(n_extra * 64));
        HIPCHK(hipMemcpyAsync(c->io_pts.p, extra_xy, n_extra * 64, hipMemcpyHostToDevice, c->stream));
        BPCHK(bp_points_import(c, c->io_pts.p, c->io_pts.p, n_extra));
    }
    BaseSegs sg; memset(&sg, 0, sizeof sg);
    int k = 0; u32 at = 0;
    if (ng) { sg.ptr[k] = c->d_G.as<u32>() + off * 16; sg.start[k] = at; at += (u32)ng; k++; }
    if (nh) { sg.ptr[k] = c->d_H.as<u32>() + off * 16; sg.start[k] = at; at += (u32)nh; k++; }
    if (n_extra) { sg.ptr[k] = c->io_pts.as<u32>(); sg.start[k] = at; at += (u32)n_extra; k++; }
    sg.start[k] = at; sg.nseg = k;
    J4 r;
    BPCHK(msm_run<C>(c, sg, c->io_scal.as<u32>(), total, canonical ? 0 : 1, r));
    A4 a = host::Grp<C>::to_aff(r);
    memcpy(out_xy, a.x.v, 32); memcpy(out_xy + 4, a.y.v, 32);
    if (c->profiling) collect_timers(c);
    return BP_OK;
}
template <class C> static void aff_out(uint64_t out[8], const A4& a) { memcpy(out, a.x.v, 32); memcpy(out + 4, a.y.v, 32); }

template <class C> static int msm_dev_entry(bp_ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, int canonical, uint64_t out_xy[8],
                                            int w_lo = 0, int w_hi = -1) {
    MsmLatencyScope latency(ctx);
    BaseSegs segs; memset(&segs, 0, sizeof segs);
    segs.nseg = 1; segs.ptr[0] = (const u32*)d_bases; segs.start[0] = 0; segs.start[1] = (u32)n;
    J4 r;
    BPCHK(msm_run<C>(ctx, segs, (const u32*)d_scalars, n, canonical ? 0 : 1, r, w_lo, w_hi));
    A4 a = host::Grp<C>::to_aff(r);
    memcpy(out_xy, a.x.v, 32); memcpy(out_xy + 4, a.y.v, 32);
    if (ctx->profiling) collect_timers(ctx);
    return BP_OK;
}


// ---- InnerProductProof::create orchestration (src/inner_product_proof.rs:37-239) ------------------------
// All vectors are device-resident in the engine's layouts and are consumed (folded in place), like the
// reference's by-value Vec arguments.  `challenge` is the Fiat-Shamir step (:132-137): it receives affine
// L, R (ark layout) and returns u (ark Montgomery words).
typedef std::function<int(const uint64_t* L_xy, const uint64_t* R_xy, uint64_t* u_out)> ChallengeFn;

template <class F> static Words8 words_of(const F4& x) { Words8 w; memcpy(w.w, x.v, 32); return w; }
// non-adjacent form of the canonical value of x: sum_i (plus_i - minus_i) 2^i, no two adjacent non-zero digits
template <class S> static Naf naf_of(const F4& x) {
    Naf r; memset(&r, 0, sizeof r);
    uint64_t k[5] = {0, 0, 0, 0, 0};
    S::to_canon(k, x);
    for (int i = 0; i < 258; i++) {
        if (!(k[0] | k[1] | k[2] | k[3] | k[4])) break;
        if (k[0] & 1) {
            if ((k[0] & 3) == 1) { r.plus[i >> 5] |= 1u << (i & 31); k[0] &= ~(uint64_t)1; }
            else { r.minus[i >> 5] |= 1u << (i & 31); for (int j = 0; j < 5; j++) { if (++k[j]) break; } }  // k += 1
        }
        for (int j = 0; j < 4; j++) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    return r;
}

// ---- GLV decomposition on the host (once per round: the fold scalar is uniform) ---------------------------------------
// round(num / r) for a 512-bit num (8 limbs) and the 256-bit modulus r: binary long division, then round half up
static void div_round_512(const uint64_t num[8], const uint64_t r[4], uint64_t quot[8]) {
    uint64_t rem[5] = {0, 0, 0, 0, 0};
    memset(quot, 0, 64);
    auto geq = [&](const uint64_t* a /*5*/) { if (a[4]) return true; for (int i = 3; i >= 0; i--) { if (a[i] != r[i]) return a[i] > r[i]; } return true; };
    auto sub = [&](uint64_t* a) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)a[i] - r[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } a[4] -= (uint64_t)br; };
    for (int bit = 511; bit >= 0; bit--) {
        for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] = (rem[0] << 1) | ((num[bit >> 6] >> (bit & 63)) & 1);
        if (geq(rem)) { sub(rem); quot[bit >> 6] |= (uint64_t)1 << (bit & 63); }
    }
    // round: if 2*rem >= r, quot += 1
    uint64_t dbl[5];
    for (int i = 4; i > 0; i--) dbl[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
    dbl[0] = rem[0] << 1;
    if (geq(dbl)) { for (int i = 0; i < 8; i++) if (++quot[i]) break; }
}
static void mul_256x192(const uint64_t a[4], const uint64_t b[3], uint64_t out[8]) {
    memset(out, 0, 64);
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 3; j++) { c += (unsigned __int128)a[i] * b[j] + out[i + j]; out[i + j] = (uint64_t)c; c >>= 64; }
        for (int j = i + 3; j < 8 && c; j++) { c += out[j]; out[j] = (uint64_t)c; c >>= 64; }
    }
}
// NAF digit masks of a non-negative integer < 2^160 given as 3 limbs; `negate` swaps the +1 / -1 masks
static void naf_masks(const uint64_t k_in[3], bool negate, u32 plus[5], u32 minus[5]) {
    uint64_t k[3] = {k_in[0], k_in[1], k_in[2]};
    memset(plus, 0, 20); memset(minus, 0, 20);
    for (int i = 0; i < 160; i++) {
        if (!(k[0] | k[1] | k[2])) break;
        if (k[0] & 1) {
            const bool is_plus = (k[0] & 3) == 1;
            if (is_plus) k[0] &= ~(uint64_t)1; else { for (int j = 0; j < 3; j++) if (++k[j]) break; }
            u32* dst = (is_plus != negate) ? plus : minus;
            dst[i >> 5] |= 1u << (i & 31);
        }
        k[0] = (k[0] >> 1) | (k[1] << 63); k[1] = (k[1] >> 1) | (k[2] << 63); k[2] >>= 1;
    }
}
// t (field element) -> t1 + t2*lambda with short t1, t2; fills the four masks of one vector.  false if a half exceeds 129 bits.
struct GlvRaw { uint64_t mg1[3], mg2[3]; bool n1, n2; };   // t = (n1 ? -mg1 : mg1) + lambda * (n2 ? -mg2 : mg2)
template <class C> static bool glv_decompose(const F4& t, u32 p1[5], u32 m1[5], u32 p2[5], u32 m2[5], GlvRaw* raw = nullptr) {
    if constexpr (C::HAS_GLV) {
        typedef host::Fld<typename C::Fr> S;
        uint64_t tc[4]; S::to_canon(tc, t);
        uint64_t prod[8], c1[8], c2[8];
        mul_256x192(tc, C::GLV_B2, prod); div_round_512(prod, C::Fr::P64, c1);    // c1 = round(b2 * t / r)
        mul_256x192(tc, C::GLV_NB1, prod); div_round_512(prod, C::Fr::P64, c2);   // c2 = round(-b1 * t / r)
        auto small = [&](const uint64_t v[3]) { uint64_t c[4] = {v[0], v[1], v[2], 0}; return S::from_canon(c); };
        const F4 fc1 = S::from_canon(c1), fc2 = S::from_canon(c2);   // c1, c2 < 2^130: the low 4 limbs hold them
        // k2 = -c1*b1 - c2*b2 = c1*|b1| - c2*b2 ;  k1 = t - k2*lambda      (mod r; both are short up to sign)
        F4 lam; memcpy(lam.v, C::LAMBDA64, 32);
        const F4 k2 = S::sub(S::mul(fc1, small(C::GLV_NB1)), S::mul(fc2, small(C::GLV_B2)));
        const F4 k1 = S::sub(t, S::mul(k2, lam));
        auto split = [&](const F4& k, uint64_t mag[3], bool& neg) {
            uint64_t c[4]; S::to_canon(c, k);
            neg = false;
            if (c[3] | (c[2] >> 1)) {   // not short: it is r - |k|
                F4 nk = S::neg(k); S::to_canon(c, nk); neg = true;
                if (c[3] | (c[2] >> 1)) return false;
            }
            mag[0] = c[0]; mag[1] = c[1]; mag[2] = c[2];
            return true;
        };
        uint64_t mg1[3], mg2[3]; bool n1, n2;
        if (!split(k1, mg1, n1) || !split(k2, mg2, n2)) return false;
        if (raw) { memcpy(raw->mg1, mg1, 24); memcpy(raw->mg2, mg2, 24); raw->n1 = n1; raw->n2 = n2; return true; }
        naf_masks(mg1, n1, p1, m1);
        naf_masks(mg2, n2, p2, m2);
        return true;
    } else {
        (void)t; (void)p1; (void)m1; (void)p2; (void)m2; (void)raw;
        return false;
    }
}
template <class C> static bool glv_pair(const F4& tG, const F4& tH, Naf2& g, Naf2& h) {
    return glv_decompose<C>(tG, g.p1, g.m1, g.p2, g.m2) && glv_decompose<C>(tH, h.p1, h.m1, h.p2, h.m2);
}
// Rounds with at least 2^16 output points convert to affine through k_ipa_fold_finish (one inversion per m points); smaller
// rounds are launch/latency bound and keep the in-lane inversion.
struct FoldFinish { u32* jac = nullptr; u32* pref = nullptr; u32 m = 0; };
static int fold_finish_plan(bp_ctx* ctx, size_t lanes, FoldFinish& ff) {
    static const bool off = getenv("ARKBP_FOLD_NOBATCH") != nullptr;
    ff = FoldFinish();
    if (off || lanes < ctx->tune_fold_batch_min) return BP_OK;
    BPCHK(ctx->ipa_jac.ensure(lanes * 96));
    BPCHK(ctx->ipa_pref.ensure(lanes * 32));
    ff.jac = ctx->ipa_jac.as<u32>(); ff.pref = ctx->ipa_pref.as<u32>();
    ff.m = (u32)std::min<size_t>(8, std::max<size_t>(2, lanes / 32768));   // keep >= 2^15 lanes busy
    return BP_OK;
}
template <class C> static void fold_finish_launch(bp_ctx* ctx, const FoldFinish& ff, u32* d_G, u32* d_H, size_t n, int which, size_t lanes) {
    if (!ff.jac) return;
    ScopedK tk(ctx, BP_K_FOLD_FINISH);
    const u32 threads = (u32)((lanes + ff.m - 1) / ff.m);
    hipLaunchKernelGGL(k_ipa_fold_finish<C>, dim3((threads + 255) / 256), dim3(256), 0, ctx->stream, ff.jac, ff.pref, d_G, d_H, (u32)n, which, (u32)lanes, ff.m);
}
// launches the uniform fold for multipliers (tG, tH): GLV ladder where the curve has the endomorphism, plain NAF ladder otherwise
template <class C> static int launch_uniform_fold(bp_ctx* ctx, u32* d_G, u32* d_H, size_t n, const F4& tG, const F4& tH, int which) {
    typedef host::Fld<typename C::Fr> S;
    hipStream_t st = ctx->stream;
    const u32 lanes = (u32)(which == 3 ? 2 * n : n);
    FoldFinish ff;
    BPCHK(fold_finish_plan(ctx, lanes, ff));
    bool done = false;
    // rounds whose points leave most of the chip idle: four lanes per point (ecq.cuh) — ~1.6x less latency per round for 2.5x the
    // VALU work (profiles/r03_fold_quad_ab.txt); BP_TUNE_FOLD_QUAD_MAX = the largest round (in points) that takes this form
    const bool quad = lanes >= 64 && lanes <= ctx->tune_fold_quad_max;
    const u32 grid = quad ? (lanes * 4 + 255) / 256 : (lanes + 255) / 256;
    {
    ScopedK tk(ctx, BP_K_FOLD_LADDER);
    if constexpr (C::HAS_GLV) {
        Naf2 g, h;
        if (glv_pair<C>(tG, tH, g, h)) {
            if (quad) hipLaunchKernelGGL((k_ipa_fold_glv<C, true>), dim3(grid), dim3(256), 0, st, d_G, d_H, (u32)n, g, h, which, ff.jac);
            else hipLaunchKernelGGL((k_ipa_fold_glv<C, false>), dim3(grid), dim3(256), 0, st, d_G, d_H, (u32)n, g, h, which, ff.jac);
            done = true;
        }
    }
    if (!done) {
        Naf a = naf_of<S>(tG), b = naf_of<S>(tH);
        if (quad) hipLaunchKernelGGL((k_ipa_fold_uniform<C, true>), dim3(grid), dim3(256), 0, st, d_G, d_H, (u32)n, a, b, which, ff.jac);
        else hipLaunchKernelGGL((k_ipa_fold_uniform<C, false>), dim3(grid), dim3(256), 0, st, d_G, d_H, (u32)n, a, b, which, ff.jac);
    }
    }
    fold_finish_launch<C>(ctx, ff, d_G, d_H, n, which, lanes);
    return BP_OK;
}

// ---- fixed-base tables for the first fold round (ipa.cuh: k_ftab_window, k_ftab_normalize, k_ipa_fold_tab) --------------------
// signed w-bit digits of a non-negative integer (limbs little-endian, `nl` of them), least significant window first
static void ftab_recode(const uint64_t* mag, int nl, bool negate, int w, int nwin, unsigned short* e, unsigned long long& negmask) {
    uint64_t k[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < nl; i++) k[i] = mag[i];
    const uint64_t full = (uint64_t)1 << w, half = full >> 1;
    negmask = 0;
    for (int j = 0; j < nwin; j++) {
        uint64_t d = k[0] & (full - 1);
        for (int i = 0; i < 4; i++) k[i] = (k[i] >> w) | (k[i + 1] << (64 - w));
        k[4] >>= w;
        bool neg = false;
        if (d > half) { d = full - d; neg = true; for (int i = 0; i < 5; i++) if (++k[i]) break; }   // digit d - 2^w, carry 1
        e[j] = (unsigned short)d;
        if (d && (neg != negate)) negmask |= 1ull << j;
    }
}
template <class C> static int ftab_nwin_for(int w) { return (C::HAS_GLV ? 130 : 256) / w + 1; }
// element j of a rank's index-cyclic slice of a vector: in[rank + j * world] (32-byte scalars, 64-byte points)
template <class W8> __global__ void k_cyclic_gather(const u32* __restrict__ in, u32* __restrict__ out, u32 n_loc, u32 rank, u32 world) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_loc) return;
    const W8* src = (const W8*)in + ((size_t)rank + (size_t)j * world);
    ((W8*)out)[j] = *src;
}
struct Blk32 { uint4 a, b; };
struct Blk64 { uint4 a, b, c, d; };
template <class C> static bool ftab_digits(const bp_ctx* ctx, const F4& t, FtabDigits& d) {
    typedef host::Fld<typename C::Fr> S;
    memset(&d, 0, sizeof d);
    d.nwin = (u32)ctx->ftab_nwin;
    if (d.nwin > 44) return false;
    if constexpr (C::HAS_GLV) {
        GlvRaw raw; u32 dummy[5];
        if (!glv_decompose<C>(t, dummy, dummy, dummy, dummy, &raw)) return false;
        ftab_recode(raw.mg1, 3, raw.n1, ctx->ftab_w, ctx->ftab_nwin, d.e1, d.neg1);
        ftab_recode(raw.mg2, 3, raw.n2, ctx->ftab_w, ctx->ftab_nwin, d.e2, d.neg2);
    } else {
        uint64_t c[4]; S::to_canon(c, t);
        ftab_recode(c, 4, false, ctx->ftab_w, ctx->ftab_nwin, d.e1, d.neg1);
    }
    return true;
}
// Builds the tables for G[0..n), H[0..n) of the installed generators.  w = 0: the widest window (<= 8 bits) whose tables fit in
// `budget_bytes` (0 = 3/4 of the free device memory).
// first / stride: the tables of the generators first + i * stride, i < n — 1/world of the memory for a rank of a sharded prover whose
// inner-product argument works on that index-cyclic slice (these are the ctx's own tables then, whoever owns the generators).
template <class C> static int ftab_build(bp_ctx* ctx, size_t n, int w, size_t budget_bytes, u32 first = 0, u32 stride = 1) {
    hipStream_t st = ctx->stream;
    if (n == 0 || stride == 0 || (size_t)first + (n - 1) * (size_t)stride >= ctx->gens_cap) { g_err = "fold tables: more bases than installed generators"; return BP_E_GENS_LENGTH; }
    if (!ctx->d_G.owned && stride == 1) { g_err = "fold tables: build them on the ctx that owns the generator tables, then bp_gens_share"; return BP_E_ARG; }
    for (DevBuf* b : {&ctx->ftab_G, &ctx->ftab_H}) { if (!b->owned) { b->p = nullptr; b->cap = 0; b->owned = true; } else b->release(); }   // (a shared view is let go, not freed)
    ctx->ftab_n = 0; ctx->ftab_first = 0; ctx->ftab_stride = 1;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (!budget_bytes) budget_bytes = free_b / 4 * 3;
    auto bytes_for = [&](int ww) { const size_t E = (size_t)1 << (ww - 1); return 2 * (size_t)ftab_nwin_for<C>(ww) * E * n * 64 + E * n * (96 + 32) + n * 96; };
    if (w == 0) { w = 8; while (w > 2 && bytes_for(w) > budget_bytes) w--; }
    if (w < 2 || w > 8 || bytes_for(w) > free_b) { g_err = "fold tables: not enough device memory"; return BP_E_ARG; }
    const size_t E = (size_t)1 << (w - 1);
    const int nwin = ftab_nwin_for<C>(w);
    const size_t per_vec = (size_t)nwin * E * n * 64;
    BPCHK(ctx->ftab_G.ensure_exact(per_vec)); BPCHK(ctx->ftab_H.ensure_exact(per_vec));
    DevBuf tmp, pref, state, gsl;
    BPCHK(tmp.ensure_exact(E * n * 96)); BPCHK(pref.ensure_exact(E * n * 32)); BPCHK(state.ensure_exact(n * 96));
    const u32 gb = (u32)((n + 255) / 256);
    if (stride > 1 || first) BPCHK(gsl.ensure_exact(n * 64));
    for (int v = 0; v < 2; v++) {
        const u32* gens = v ? ctx->d_H.as<u32>() : ctx->d_G.as<u32>();
        if (gsl.p) {   // the slice as a compact vector: the builder below then runs unchanged
            hipLaunchKernelGGL(k_cyclic_gather<Blk64>, dim3(gb), dim3(256), 0, st, gens, gsl.as<u32>(), (u32)n, first, stride);
            gens = gsl.as<u32>();
        }
        u32* T = v ? ctx->ftab_H.as<u32>() : ctx->ftab_G.as<u32>();
        for (int j = 0; j < nwin; j++) {
            hipLaunchKernelGGL(k_ftab_window<C>, dim3(gb), dim3(256), 0, st, gens, state.as<u32>(), tmp.as<u32>(), (u32)n, (u32)E, j == 0 ? 1 : 0);
            hipLaunchKernelGGL(k_ftab_normalize<C>, dim3(gb), dim3(256), 0, st, tmp.as<u32>(), pref.as<u32>(), T + (size_t)j * E * n * 16, (u32)n, (u32)E);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(ctx_stream_wait(ctx));
    tmp.release(); pref.release(); state.release(); gsl.release();
    ctx->ftab_n = n; ctx->ftab_w = w; ctx->ftab_nwin = nwin; ctx->ftab_first = first; ctx->ftab_stride = stride;
    return BP_OK;
}
// Walks every entry of the fold tables and of the fixed-base MSM rows (ipa.cuh "integrity check"); counts the entries that break
// the chain rule.  ~0.3 s for 146 GB of fold tables.
template <class C> static int tables_check(bp_ctx* ctx, uint64_t* bad_fold, uint64_t* bad_rows) {
    hipStream_t st = ctx->stream;
    BPCHK(ctx->io_out.ensure(64));
    unsigned long long* d_bad = ctx->io_out.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(d_bad, 0, 16, st));
    if (ctx->ftab_n) {
        const u32 n = (u32)ctx->ftab_n, E = 1u << (ctx->ftab_w - 1), gb = (n + 255) / 256;
        hipLaunchKernelGGL(k_ftab_check<C>, dim3(gb), dim3(256), 0, st, ctx->d_G.as<u32>(), ctx->ftab_G.as<u32>(), n, E, (u32)ctx->ftab_nwin, d_bad, ctx->ftab_first, ctx->ftab_stride);
        hipLaunchKernelGGL(k_ftab_check<C>, dim3(gb), dim3(256), 0, st, ctx->d_H.as<u32>(), ctx->ftab_H.as<u32>(), n, E, (u32)ctx->ftab_nwin, d_bad, ctx->ftab_first, ctx->ftab_stride);
    }
    if (ctx->fb_cap) {
        const u32 n = (u32)ctx->fb_cap, gb = (n + 255) / 256;
        hipLaunchKernelGGL(k_fb_rows_check<C>, dim3(gb), dim3(256), 0, st, ctx->d_G.as<u32>(), ctx->fb_G.as<u32>(), n, (size_t)n, FB_ROWS, d_bad + 1);
        hipLaunchKernelGGL(k_fb_rows_check<C>, dim3(gb), dim3(256), 0, st, ctx->d_H.as<u32>(), ctx->fb_H.as<u32>(), n, (size_t)n, FB_ROWS, d_bad + 1);
        hipLaunchKernelGGL(k_fb_rows_check<C>, dim3(1), dim3(256), 0, st, ctx->d_pc.as<u32>(), ctx->fb_pc.as<u32>(), 2u, (size_t)2, FB_ROWS, d_bad + 1);
    }
    HIPCHK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(ctx));
    if (bad_fold) *bad_fold = h[0];
    if (bad_rows) *bad_rows = h[1];
    return BP_OK;
}
// Table columns of a state's local elements: element j (generator gens_first + j * gens_stride) is column c0 + j * cs of the ctx's
// fold tables — whole tables (column = generator index) or a rank's slice of them; false when the elements do not lie on the columns
static bool ftab_cols(const bp_ctx* ctx, const IpaState& s, u32& c0, u32& cs) {
    const u32 ts = ctx->ftab_stride ? ctx->ftab_stride : 1u, tf = ctx->ftab_first;
    if (!s.gens_stride || s.gens_stride % ts || s.gens_first < tf || (s.gens_first - tf) % ts) return false;
    c0 = (s.gens_first - tf) / ts; cs = s.gens_stride / ts;
    return true;
}
// the first-round uniform fold through the tables; false when they do not apply (then the ladder kernels run)
template <class C> static int launch_tab_fold(bp_ctx* ctx, const IpaState& s, u32* d_G, u32* d_H, size_t n, const F4& tG, const F4& tH, bool& done) {
    done = false;
    static const bool off = getenv("ARKBP_FOLD_NOTAB") != nullptr;   // A/B switch
    u32 c0 = 0, cs = 1;
    if (off || !ctx->ftab_n || n < 64 || !ftab_cols(ctx, s, c0, cs)) return BP_OK;
    if ((size_t)c0 + (n - 1) * (size_t)cs >= ctx->ftab_n) return BP_OK;   // left half reaches past the tabled bases
    FtabDigits dG, dH;
    if (!ftab_digits<C>(ctx, tG, dG) || !ftab_digits<C>(ctx, tH, dH)) return BP_OK;
    const u32 lanes = (u32)(2 * n);
    FoldFinish ff;
    BPCHK(fold_finish_plan(ctx, lanes, ff));
    {
    ScopedK tk(ctx, BP_K_FOLD_TAB);
    hipLaunchKernelGGL(k_ipa_fold_tab<C>, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, ctx->ftab_G.as<u32>(), ctx->ftab_H.as<u32>(), (u32)ctx->ftab_n,
                       1u << (ctx->ftab_w - 1), d_G, d_H, (u32)n, dG, dH, 3, c0, cs, ff.jac, s.d_G_in, s.d_H_in);
    }
    fold_finish_launch<C>(ctx, ff, d_G, d_H, n, 3, lanes);
    done = true;
    return BP_OK;
}

// whether the first fold of a prover round can be deferred (ipa.cuh "TWO fold rounds from the tables"): the vectors are the ctx's
// generator tables read in place, the fold tables cover the bases [0, n + n/2), both multipliers have table digits, and neither this
// round nor the next reaches the frozen-tail length
template <class C> static bool fold_can_defer(bp_ctx* ctx, const IpaState& s, size_t n, const F4& tG1, const F4& tH1) {
    static const bool off = getenv("ARKBP_FOLD_NODEFER") != nullptr || getenv("ARKBP_FOLD_NOTAB") != nullptr;   // A/B switches
    u32 c0 = 0, cs = 1;
    if (off || !ctx->ftab_n || !s.d_G_in || !s.d_H_in || !ftab_cols(ctx, s, c0, cs)) return false;
    // either the whole vectors on one GPU, or this rank's index-cyclic slice (local element j = table base first + j * stride; its
    // MSMs are partial sums followed by the point-reduce) with one more LOCAL round to come
    const bool whole = s.gens_stride == 1 && s.gens_first == 0 && ctx->shard_world == 1 && s.msm_mode == -1;
    const bool slice = s.msm_mode == 2 && ctx->shard_world > 1 && s.first && n > s.min_len;
    if (!whole && !slice) return false;
    if (n < 256 || (n & 1) || (size_t)c0 + (n + n / 2 - 1) * (size_t)cs >= ctx->ftab_n) return false;
    if (s.allow_freeze && n / 2 <= std::max<size_t>(ctx->tune_ipa_freeze_len, 2)) return false;
    if (!s.have_rho) return false;
    FtabDigits d;
    return ftab_digits<C>(ctx, tG1, d) && ftab_digits<C>(ctx, tH1, d);
}
// the second fold after a deferred first one, straight from the tables: m outputs per vector; false when a multiplier has no digits
template <class C> static int launch_tab_fold2(bp_ctx* ctx, const IpaState& s, u32* d_G, u32* d_H, size_t m, const F4& t2G, const F4& t2H, bool& done) {
    typedef host::Fld<typename C::Fr> S;
    done = false;
    FtabDigits3 dG, dH;
    u32 c0 = 0, cs = 1;
    if (!ftab_cols(ctx, s, c0, cs)) return BP_OK;
    if (!ftab_digits<C>(ctx, s.def_tG, dG.d[0]) || !ftab_digits<C>(ctx, t2G, dG.d[1]) || !ftab_digits<C>(ctx, S::mul(s.def_tG, t2G), dG.d[2])) return BP_OK;
    if (!ftab_digits<C>(ctx, s.def_tH, dH.d[0]) || !ftab_digits<C>(ctx, t2H, dH.d[1]) || !ftab_digits<C>(ctx, S::mul(s.def_tH, t2H), dH.d[2])) return BP_OK;
    const u32 lanes = (u32)(2 * m);
    FoldFinish ff;
    BPCHK(fold_finish_plan(ctx, lanes, ff));
    {
        ScopedK tk(ctx, BP_K_FOLD_TAB);
        hipLaunchKernelGGL(k_ipa_fold_tab2<C>, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, ctx->ftab_G.as<u32>(), ctx->ftab_H.as<u32>(), (u32)ctx->ftab_n,
                           1u << (ctx->ftab_w - 1), d_G, d_H, (u32)m, dG, dH, ff.jac, s.d_G_in, s.d_H_in, c0, cs);
    }
    fold_finish_launch<C>(ctx, ff, d_G, d_H, m, 3, lanes);
    done = true;
    return BP_OK;
}

// ---- direct window tables of the first generators (small.cuh): the small-statement path --------------------------------------------
static inline u32 dt_base_pc(int which /* 0 = B, 1 = B_blinding */) { return (u32)which; }
static inline u32 dt_base_G(const bp_ctx* ctx, size_t i) { (void)ctx; return (u32)(2 + i); }
static inline u32 dt_base_H(const bp_ctx* ctx, size_t i) { return (u32)(2 + ctx->dt_cap + i); }
// true when statements of padded size N go over the direct tables on this ctx (building them on first use)
template <class C> static int dt_ensure(bp_ctx* ctx, size_t N, bool& ready) {
    ready = false;
    if (!ctx->tune_direct_max || N == 0 || N > ctx->tune_direct_max || ctx->shard_world > 1 || ctx->gens_cap < N) return BP_OK;
    if (ctx->dt_cap >= N) { ready = true; return BP_OK; }
    if (!ctx->dt_tab.owned && ctx->dt_tab.p) { ctx->dt_tab.p = nullptr; ctx->dt_tab.cap = 0; ctx->dt_tab.owned = true; }   // (a shared table that is too short: build our own)
    hipStream_t st = ctx->stream;
    const size_t cap = std::min(ctx->gens_cap, ctx->tune_direct_max), nb = 2 + 2 * cap;
    ctx->dt_cap = 0;
    BPCHK(ctx->dt_tab.ensure_exact(nb * DT_BASE_BYTES));
    DevBuf ws;
    BPCHK(ws.ensure(nb * DT_WINDOWS * 96));
    auto part = [&](const u32* bases, size_t first, size_t count) -> int {
        hipLaunchKernelGGL(k_dt_window_bases<C>, dim3((u32)((count + 63) / 64)), dim3(64), 0, st, bases, (u32)count, ws.as<u32>() + first * DT_WINDOWS * 24);
        return BP_OK;
    };
    BPCHK(part(ctx->d_pc.as<u32>(), 0, 2));
    BPCHK(part(ctx->d_G.as<u32>(), 2, cap));
    BPCHK(part(ctx->d_H.as<u32>(), 2 + cap, cap));
    const size_t entries = nb * DT_PER_BASE;
    hipLaunchKernelGGL(k_dt_entries<C>, dim3((u32)((entries + 255) / 256)), dim3(256), 0, st, ws.as<u32>(), (u32)nb, ctx->dt_tab.as<u32>());
    HIPCHK(hipGetLastError());
    HIPCHK(ctx_stream_wait(ctx));
    ws.release();
    ctx->dt_cap = cap;
    ready = true;
    return BP_OK;
}
static constexpr u32 DT_HOST_SUM_MAX = 16;
// nout MSMs over the direct tables in one launch (+ one finishing launch when more than one workgroup per MSM is worth it): the
// launch half leaves the results on their way to pinned memory, the collect half reads them after the caller's stream wait
template <class C> static int msm_direct_launch(bp_ctx* ctx, const DtJobs& jobs, int nout) {
    hipStream_t st = ctx->stream;
    u32 maxterms = 0;
    for (int o = 0; o < nout; o++) maxterms = std::max(maxterms, jobs.job[o].terms);
    // A quad-cooperative mixed addition is ~4.5 us with a SIMD to itself, the workgroup's tree six levels of the same: one unit (four
    // additions) per quad while that leaves at most one workgroup per CU, two units beyond (the waves then share their SIMDs)
    const size_t units = (size_t)maxterms * DT_UNITS_PER_TERM;
    u32 nblk = (u32)std::max<size_t>(1, (units + 63) / 64);
    if ((size_t)nblk * nout > 256) nblk = (u32)std::min<size_t>(1024, (units + 127) / 128);
    BPCHK(ctx->dt_part.ensure((size_t)DT_MAXOUT * (nblk + 1) * 96));
    if (!ctx->h_dt) HIPCHK(hipHostMalloc((void**)&ctx->h_dt, (size_t)DT_MAXOUT * DT_HOST_SUM_MAX * 96));
    u32* part = ctx->dt_part.as<u32>();
    u32* res = part + (size_t)DT_MAXOUT * nblk * 24;
    // up to DT_HOST_SUM_MAX partial points per MSM are added by the HOST (0.4 us per addition there) instead of a second launch whose
    // six-level tree costs ~35 us however few points it adds
    const bool host_sum = nblk > 1 && nblk <= DT_HOST_SUM_MAX;
    {
        ScopedK tk(ctx, BP_K_MSM_ACCUM);
        hipLaunchKernelGGL(k_dt_accum<C>, dim3(nblk, (u32)nout), dim3(256), 0, st, ctx->dt_tab.as<u32>(), jobs, nblk == 1 ? res : part);
        if (nblk > 1 && !host_sum) hipLaunchKernelGGL(k_dt_finish<C>, dim3((u32)nout), dim3(256), 0, st, part, nblk, res);
    }
    ctx->dt_pending_parts = host_sum ? nblk : 1u;
    HIPCHK(hipMemcpyAsync(ctx->h_dt, host_sum ? part : res, (size_t)nout * ctx->dt_pending_parts * 96, hipMemcpyDeviceToHost, st));
    return BP_OK;
}
template <class C> static void msm_direct_collect(bp_ctx* ctx, int nout, J4* results) {
    typedef host::Grp<C> G;
    const u64* T = (const u64*)ctx->h_dt;
    const u32 np = ctx->dt_pending_parts;
    for (int o = 0; o < nout; o++) {
        J4 acc = G::inf();
        for (u32 j = 0; j < np; j++) {
            const u64* P = T + 12 * ((size_t)o * np + j);
            J4 pnt; memcpy(pnt.X.v, P, 32); memcpy(pnt.Y.v, P + 4, 32); memcpy(pnt.Z.v, P + 8, 32);
            if (pnt.Z.is_zero()) continue;
            acc = np == 1 ? pnt : G::add(acc, pnt);
        }
        results[o] = acc;
    }
    ctx->dt_runs += (uint64_t)nout;
}
template <class C> static int msm_direct(bp_ctx* ctx, const DtJobs& jobs, int nout, J4* results) {
    BPCHK(msm_direct_launch<C>(ctx, jobs, nout));
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    msm_direct_collect<C>(ctx, nout, results);
    return BP_OK;
}
// affine forms of several Jacobian points with ONE field inversion (Montgomery's trick; the identity stays (0, 0))
template <class C> static void to_aff_batch(const J4* in, int count, A4* out) {
    typedef host::Grp<C> G;
    typedef host::Fld<typename C::Fq> F;
    F4 pref[DT_MAXOUT];
    F4 run = F::one();
    for (int i = 0; i < count; i++) { pref[i] = run; if (!G::is_inf(in[i])) run = F::mul(run, in[i].Z); }
    F4 inv = F::inv(run);
    for (int i = count - 1; i >= 0; i--) {
        if (G::is_inf(in[i])) { out[i] = G::aff_inf(); continue; }
        const F4 zi = F::mul(inv, pref[i]), zi2 = F::sqr(zi);
        inv = F::mul(inv, in[i].Z);
        out[i] = A4{F::mul(in[i].X, zi2), F::mul(in[i].Y, F::mul(zi2, zi))};
    }
}

// State of one InnerProductProof::create in flight (the loop body of src/inner_product_proof.rs:70-237 cut at the Fiat-Shamir
// step): ipa_round_lr computes L, R of the current round, ipa_round_fold consumes the challenge.  ipa_create_dev drives it with a
// callback; bp_ipa_begin / bp_ipa_round_LR / bp_ipa_round_fold / bp_ipa_finish expose the same steps for hosts that keep the
// transcript on their side of the boundary, and for the index-cyclic multi-GPU partition (parallel.py), where L and R of a
// round are sums of per-rank partials.
static inline int ipa_lg2(size_t x) { int k = 0; while (((size_t)1 << k) < x) k++; return k; }

template <class C> static int ipa_begin_dev(bp_ctx* ctx, IpaState& s, const u32* d_Q, const u32* d_Gf, const u32* d_Hf, u32* d_G, u32* d_H, u32* d_a, u32* d_b,
                                            size_t n, const F4* gf_halves, const F4* rho_pw, const u32* d_rho_pow, bool allow_freeze = false) {
    typedef host::Fld<typename C::Fr> S;
    if (n == 0 || (n & (n - 1))) { g_err = "ipa_create: n must be a power of two (reference asserts, src/inner_product_proof.rs:66)"; return BP_E_ARG; }
    s = IpaState();
    s.d_Q = d_Q; s.d_Gf = d_Gf; s.d_Hf = d_Hf; s.d_G = d_G; s.d_H = d_H; s.d_a = d_a; s.d_b = d_b; s.n = n;
    s.gamma_G = S::one(); s.gamma_H = S::one();
    if (gf_halves) { s.have_gf = true; s.gf_halves[0] = gf_halves[0]; s.gf_halves[1] = gf_halves[1]; }
    if (rho_pw && d_rho_pow) { s.have_rho = true; s.rho_pw = rho_pw; s.d_rho_pow = d_rho_pow; }
    s.allow_freeze = allow_freeze;
    const size_t fz = std::max<size_t>(ctx->tune_ipa_freeze_len, 2);
    BPCHK(ctx->ipa_sL.ensure((std::max(n, 2 * fz) + 2) * 32));
    BPCHK(ctx->ipa_sR.ensure((std::max(n, 2 * fz) + 2) * 32));
    BPCHK(ctx->ipa_part.ensure(((std::max(n / 2, fz) + 255) / 256 + 1) * 64));
    return BP_OK;
}
// L, R of the current round (affine, ark layout): src/inner_product_proof.rs:78-131 (first round) / :166-213
template <class C> static int ipa_round_lr(bp_ctx* ctx, IpaState& s, uint64_t Lw[8], uint64_t Rw[8]) {
    typedef host::Fld<typename C::Fr> S;
    typedef host::Grp<C> G;
    if (s.n <= 1 || s.lr_done) { g_err = "ipa: round_LR out of sequence"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    const size_t n = s.n / 2;
    const u32 gb = (u32)((n + 255) / 256);
    u32* sL = ctx->ipa_sL.as<u32>();
    u32* sR = ctx->ipa_sR.as<u32>();
    if (s.frozen) {
        const size_t n0 = s.n0;
        const u32 gf = (u32)((n0 + 255) / 256);
        if (s.direct) {
            // one launch: the fold the previous challenge asked for (into the other buffer pair), the scalars, the inner products
            ScopedK tk(ctx, BP_K_IPA_SCALARS);
            hipLaunchKernelGGL(k_dt_round<C>, dim3(gf), dim3(256), 0, st, s.d_a, s.d_b, s.d_a_alt, s.d_b_alt, s.d_cG, s.d_cH, (u32)n, (u32)n0, s.fold_pending ? 1 : 0,
                               words_of<S>(s.pend_u), words_of<S>(s.pend_ui), sL, sR, ctx->ipa_part.as<u32>(), ctx->dt_ticket.as<u32>(), words_of<S>(s.qw));
            if (s.fold_pending) { std::swap(s.d_a, s.d_a_alt); std::swap(s.d_b, s.d_b_alt); s.fold_pending = false; }
        } else {
            ScopedK tk(ctx, BP_K_IPA_SCALARS);
            hipLaunchKernelGGL(k_ipa_frozen_scalars<C>, dim3(gf), dim3(256), 0, st, s.d_a, s.d_b, s.d_cG, s.d_cH, (u32)n, (u32)n0, sL, sR, ctx->ipa_part.as<u32>());
            hipLaunchKernelGGL(k_ipa_ip_finish<C>, dim3(1), dim3(256), 0, st, ctx->ipa_part.as<u32>(), gf, sL + 2 * n0 * 8, sR + 2 * n0 * 8, words_of<S>(s.qw), 0);
        }
        if (s.direct) {
            // L and R as sums over the direct window tables, one launch for both: [G[0..n0) | H[0..n0) | B] with c * Q = (c * qw) * B
            DtJobs jobs; memset(&jobs, 0, sizeof jobs);
            for (int o = 0; o < 2; o++) {
                const u32* sc = o ? sR : sL;
                DtJob& jb = jobs.job[o];
                // every base belongs to exactly one of L, R (k_ipa_frozen_scalars): L takes G where the element sits in the upper half
                // of its period of the current length and H in the lower half, R the other way round — only those are visited
                jb.nseg = 3; jb.terms = (u32)(n0 + 1);
                jb.seg[0] = DtSeg{sc, dt_base_G(ctx, 0), (u32)(n0 / 2), 0, (u32)n, o == 0 ? 1u : 0u};
                jb.seg[1] = DtSeg{sc + n0 * 8, dt_base_H(ctx, 0), (u32)(n0 / 2), 0, (u32)n, o == 0 ? 0u : 1u};
                jb.seg[2] = DtSeg{sc + (2 * n0 + 1) * 8, dt_base_pc(0), 1, 0, 0, 0};
            }
            J4 LR[2];
            BPCHK(msm_direct<C>(ctx, jobs, 2, LR));
            A4 LRa[2];
            to_aff_batch<C>(LR, 2, LRa);
            const A4 &La = LRa[0], &Ra = LRa[1];
            memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
            memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
            s.lr_done = true;
            return BP_OK;
        }
        BaseSegs sg; memset(&sg, 0, sizeof sg);
        sg.nseg = 3; sg.start[0] = 0; sg.start[1] = (u32)n0; sg.start[2] = (u32)(2 * n0); sg.start[3] = (u32)(2 * n0 + 1);
        sg.ptr[0] = s.d_G; sg.ptr[1] = s.d_H; sg.ptr[2] = s.d_Q;
        J4 Lj, Rj;
        BPCHK(msm_run<C>(ctx, sg, sL, 2 * n0 + 1, 0, Lj, 0, -1, s.msm_mode));
        BPCHK(msm_run<C>(ctx, sg, sR, 2 * n0 + 1, 0, Rj, 0, -1, s.msm_mode));
        A4 La = G::to_aff(Lj), Ra = G::to_aff(Rj);
        memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
        memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
        s.lr_done = true;
        return BP_OK;
    }
    if (s.deferred) {
        // the round after a deferred first fold: L and R over the generator tables themselves with split scalars (4n + 1 terms each)
        const size_t n1 = s.n;   // half length of round 1 = current length of a, b
        {
            ScopedK tk(ctx, BP_K_IPA_SCALARS);
            hipLaunchKernelGGL(k_ipa_scalars_deferred<C>, dim3(gb), dim3(256), 0, st, s.d_a, s.d_b, (u32)n, sL, sR, ctx->ipa_part.as<u32>(), s.pending ? (s.h_geo ? 2 : 1) : 0,
                               words_of<S>(s.gamma_G), words_of<S>(s.gamma_H), s.d_rho_pow, words_of<S>(s.def_tG), words_of<S>(s.def_tH));
            hipLaunchKernelGGL(k_ipa_ip_finish<C>, dim3(1), dim3(256), 0, st, ctx->ipa_part.as<u32>(), gb, sL + 4 * n * 8, sR + 4 * n * 8, words_of<S>(s.qw), s.have_qw ? 1 : 0);
        }
        J4 Lj, Rj; bool dl = false, dr = false;
        if (s.msm_mode == 2) {
            // this rank's index-cyclic slice (the working vectors are its compact copy, local element j = table base first + j * stride):
            // the same four runs read the fixed-base rows with a stride; every MSM ends in exactly one point-reduce, whichever way it ran
            BaseSegs sg; memset(&sg, 0, sizeof sg);
            sg.nseg = 5;
            for (int k = 0; k <= 4; k++) sg.start[k] = (u32)(k * n);
            sg.start[5] = (u32)(4 * n + 1);
            const size_t f0 = s.gens_first, sd = s.gens_stride;
            const bool rows = s.have_qw && ctx->fb_cap;
            ScalSegs ss; memset(&ss, 0, sizeof ss);
            ss.nseg = 2; ss.ptr[0] = sL; ss.start[0] = 0; ss.start[1] = (u32)(4 * n); ss.ptr[1] = sL + (4 * n + 1) * 8; ss.start[2] = (u32)(4 * n + 1);
            if (rows) {
                FbRun rl[5] = {{0, f0 + (n1 + n) * sd, n, sd}, {0, f0 + n * sd, n, sd}, {1, f0 + n1 * sd, n, sd}, {1, f0, n, sd}, {2, 0, 1}};
                BPCHK(msm_fixed_run<C>(ctx, rl, 5, ss, 4 * n + 1, 0, Lj, dl, 2));
            }
            if (!dl) {
                sg.ptr[0] = s.d_G_in + (n1 + n) * 16; sg.ptr[1] = s.d_G_in + n * 16; sg.ptr[2] = s.d_H_in + n1 * 16; sg.ptr[3] = s.d_H_in; sg.ptr[4] = s.d_Q;
                BPCHK(msm_run<C>(ctx, sg, sL, 4 * n + 1, 0, Lj, 0, -1, 2));
            }
            if (rows) {
                ss.ptr[0] = sR; ss.ptr[1] = sR + (4 * n + 1) * 8;
                FbRun rr[5] = {{0, f0 + n1 * sd, n, sd}, {0, f0, n, sd}, {1, f0 + (n1 + n) * sd, n, sd}, {1, f0 + n * sd, n, sd}, {2, 0, 1}};
                BPCHK(msm_fixed_run<C>(ctx, rr, 5, ss, 4 * n + 1, 0, Rj, dr, 2));
            }
            if (!dr) {
                sg.ptr[0] = s.d_G_in + n1 * 16; sg.ptr[1] = s.d_G_in; sg.ptr[2] = s.d_H_in + (n1 + n) * 16; sg.ptr[3] = s.d_H_in + n * 16; sg.ptr[4] = s.d_Q;
                BPCHK(msm_run<C>(ctx, sg, sR, 4 * n + 1, 0, Rj, 0, -1, 2));
            }
            A4 La = G::to_aff(Lj), Ra = G::to_aff(Rj);
            memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
            memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
            s.lr_done = true;
            return BP_OK;
        }
        const size_t gofs = (size_t)(s.d_G_in - ctx->d_G.as<u32>()) / 16, hofs = (size_t)(s.d_H_in - ctx->d_H.as<u32>()) / 16;
        if (s.have_qw && s.msm_mode == -1) {
            ScalSegs ss; memset(&ss, 0, sizeof ss);
            ss.nseg = 2; ss.ptr[0] = sL; ss.start[0] = 0; ss.start[1] = (u32)(4 * n); ss.ptr[1] = sL + (4 * n + 1) * 8; ss.start[2] = (u32)(4 * n + 1);
            FbRun rl[5] = {{0, gofs + n1 + n, n}, {0, gofs + n, n}, {1, hofs + n1, n}, {1, hofs, n}, {2, 0, 1}};
            BPCHK(msm_fixed_run<C>(ctx, rl, 5, ss, 4 * n + 1, 0, Lj, dl));
            if (dl) {
                ss.ptr[0] = sR; ss.ptr[1] = sR + (4 * n + 1) * 8;
                FbRun rr[5] = {{0, gofs + n1, n}, {0, gofs, n}, {1, hofs + n1 + n, n}, {1, hofs + n, n}, {2, 0, 1}};
                BPCHK(msm_fixed_run<C>(ctx, rr, 5, ss, 4 * n + 1, 0, Rj, dr));
            }
        }
        if (!(dl && dr)) {
            BaseSegs sg; memset(&sg, 0, sizeof sg);
            sg.nseg = 5;
            for (int k = 0; k <= 4; k++) sg.start[k] = (u32)(k * n);
            sg.start[5] = (u32)(4 * n + 1);
            sg.ptr[0] = s.d_G_in + (n1 + n) * 16; sg.ptr[1] = s.d_G_in + n * 16; sg.ptr[2] = s.d_H_in + n1 * 16; sg.ptr[3] = s.d_H_in; sg.ptr[4] = s.d_Q;
            BPCHK(msm_run<C>(ctx, sg, sL, 4 * n + 1, 0, Lj, 0, -1, s.msm_mode));
            sg.ptr[0] = s.d_G_in + n1 * 16; sg.ptr[1] = s.d_G_in; sg.ptr[2] = s.d_H_in + (n1 + n) * 16; sg.ptr[3] = s.d_H_in + n * 16;
            BPCHK(msm_run<C>(ctx, sg, sR, 4 * n + 1, 0, Rj, 0, -1, s.msm_mode));
        }
        A4 La = G::to_aff(Lj), Ra = G::to_aff(Rj);
        memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
        memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
        s.lr_done = true;
        return BP_OK;
    }
    {
        ScopedK tk(ctx, BP_K_IPA_SCALARS);
        hipLaunchKernelGGL(k_ipa_scalars<C>, dim3(gb), dim3(256), 0, st, s.d_a, s.d_b, s.d_Gf, s.d_Hf, s.first ? 1 : 0, (u32)n, sL, sR, ctx->ipa_part.as<u32>(),
                           s.pending ? (s.h_geo ? 2 : 1) : 0, words_of<S>(s.gamma_G), words_of<S>(s.gamma_H), s.d_rho_pow);
        hipLaunchKernelGGL(k_ipa_ip_finish<C>, dim3(1), dim3(256), 0, st, ctx->ipa_part.as<u32>(), gb, sL + 2 * n * 8, sR + 2 * n * 8, words_of<S>(s.qw), s.have_qw ? 1 : 0);
    }
    if (s.d_G_in && s.have_qw && s.msm_mode == -1) {
        // round 1 over the resident generator tables with Q = qw * B: fixed-base MSMs (c_L * Q = (c_L * qw) * B, the scaled
        // inner products are in slot 2n + 1).  L = <xl, G[n..2n)> + <yl, H[0..n)> + c_L Q,  R = <xr, G[0..n)> + <yr, H[n..2n)> + c_R Q.
        const size_t gofs = (size_t)(s.d_G_in - ctx->d_G.as<u32>()) / 16, hofs = (size_t)(s.d_H_in - ctx->d_H.as<u32>()) / 16;
        J4 Lj, Rj; bool dl = false, dr = false;
        ScalSegs ss; memset(&ss, 0, sizeof ss);
        ss.nseg = 2; ss.ptr[0] = sL; ss.start[0] = 0; ss.start[1] = (u32)(2 * n); ss.ptr[1] = sL + (2 * n + 1) * 8; ss.start[2] = (u32)(2 * n + 1);
        FbRun rl[3] = {{0, gofs + n, n}, {1, hofs, n}, {2, 0, 1}};
        BPCHK(msm_fixed_run<C>(ctx, rl, 3, ss, 2 * n + 1, 0, Lj, dl));
        if (dl) {
            ss.ptr[0] = sR; ss.ptr[1] = sR + (2 * n + 1) * 8;
            FbRun rr[3] = {{0, gofs, n}, {1, hofs + n, n}, {2, 0, 1}};
            BPCHK(msm_fixed_run<C>(ctx, rr, 3, ss, 2 * n + 1, 0, Rj, dr));
        }
        if (dl && dr) {
            A4 La = G::to_aff(Lj), Ra = G::to_aff(Rj);
            memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
            memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
            s.lr_done = true;
            return BP_OK;
        }
    }
    BaseSegs sg; memset(&sg, 0, sizeof sg);
    sg.nseg = 3; sg.start[0] = 0; sg.start[1] = (u32)n; sg.start[2] = (u32)(2 * n); sg.start[3] = (u32)(2 * n + 1);
    const u32* Gb = s.d_G_in ? s.d_G_in : s.d_G;
    const u32* Hb = s.d_H_in ? s.d_H_in : s.d_H;
    sg.ptr[0] = Gb + n * 16; sg.ptr[1] = Hb; sg.ptr[2] = s.d_Q;
    J4 Lj, Rj;
    // Index-cyclic sharded prover, round 1: this rank's terms are ITS slice of the generator tables (bases first + j * world) — with
    // fixed-base rows installed they keep the single-GPU schedule, reading the rows with a stride (Q = qw * B rides along as a row of
    // the Pedersen table with this rank's partial inner product).  A rank whose scalars do not go through it runs the ordinary MSM
    // over its compact copy: either way one point-reduce per L and per R.
    const bool cyc_fixed = s.first && s.msm_mode == 2 && s.gens_stride > 1 && s.have_qw && ctx->fb_cap;
    bool dl = false, dr = false;
    if (cyc_fixed) {
        ScalSegs ss; memset(&ss, 0, sizeof ss);
        ss.nseg = 2; ss.ptr[0] = sL; ss.start[0] = 0; ss.start[1] = (u32)(2 * n); ss.ptr[1] = sL + (2 * n + 1) * 8; ss.start[2] = (u32)(2 * n + 1);
        const size_t f0 = s.gens_first, sd = s.gens_stride;
        FbRun rl[3] = {{0, f0 + n * sd, n, sd}, {1, f0, n, sd}, {2, 0, 1}};
        BPCHK(msm_fixed_run<C>(ctx, rl, 3, ss, 2 * n + 1, 0, Lj, dl, 2));
    }
    if (!dl) BPCHK(msm_run<C>(ctx, sg, sL, 2 * n + 1, 0, Lj, 0, -1, s.msm_mode));
    sg.ptr[0] = Gb; sg.ptr[1] = Hb + n * 16;
    if (cyc_fixed) {
        ScalSegs ss; memset(&ss, 0, sizeof ss);
        ss.nseg = 2; ss.ptr[0] = sR; ss.start[0] = 0; ss.start[1] = (u32)(2 * n); ss.ptr[1] = sR + (2 * n + 1) * 8; ss.start[2] = (u32)(2 * n + 1);
        const size_t f0 = s.gens_first, sd = s.gens_stride;
        FbRun rr[3] = {{0, f0, n, sd}, {1, f0 + n * sd, n, sd}, {2, 0, 1}};
        BPCHK(msm_fixed_run<C>(ctx, rr, 3, ss, 2 * n + 1, 0, Rj, dr, 2));
    }
    if (!dr) BPCHK(msm_run<C>(ctx, sg, sR, 2 * n + 1, 0, Rj, 0, -1, s.msm_mode));
    A4 La = G::to_aff(Lj), Ra = G::to_aff(Rj);
    memcpy(Lw, La.x.v, 32); memcpy(Lw + 4, La.y.v, 32);
    memcpy(Rw, Ra.x.v, 32); memcpy(Rw + 4, Ra.y.v, 32);
    s.lr_done = true;
    return BP_OK;
}
// the folds of the round for challenge u (ark words): :137-155 (first round) / :214-224
template <class C> static int ipa_round_fold(bp_ctx* ctx, IpaState& s, const uint64_t uw[4]) {
    typedef host::Fld<typename C::Fr> S;
    if (!s.lr_done) { g_err = "ipa: round_fold before round_LR"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    const size_t n = s.n / 2;
    const u32 gb = (u32)((n + 255) / 256);
    u32 *d_G = s.d_G, *d_H = s.d_H;
    const bool first = s.first;
    F4 u; memcpy(u.v, uw, 32);
    F4 ui = S::inv(u);
    // the ladder kernels fold in place: they need the round-1 inputs in the working vectors (the table fold reads them from the
    // resident generator tables and needs no copy)
    auto working_copy = [&]() -> int {
        if (!s.d_G_in) return BP_OK;
        if (s.d_G_in == s.d_G) { s.d_G_in = s.d_H_in = nullptr; return BP_OK; }   // (an index-cyclic slice: the working vectors ARE the compact copy)
        HIPCHK(hipMemcpyAsync(s.d_G, s.d_G_in, 2 * n * 64, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(s.d_H, s.d_H_in, 2 * n * 64, hipMemcpyDeviceToDevice, st));
        s.d_G_in = s.d_H_in = nullptr;
        return BP_OK;
    };
    if (s.direct) {
        // the next round's k_dt_round folds (a, b and the coefficients); after the last round only a[0], b[0] are left to form
        if (n == 1) {
            ScopedK tk(ctx, BP_K_IPA_FOLD);
            hipLaunchKernelGGL(k_ipa_fold_ab<C>, dim3(1), dim3(256), 0, st, s.d_a, s.d_b, 1u, words_of<S>(u), words_of<S>(ui));
            tk.stop();
            HIPCHK(hipGetLastError());
        } else {
            s.fold_pending = true; s.pend_u = u; s.pend_ui = ui;
        }
        s.round++; s.n = n; s.lr_done = false;
        return BP_OK;
    }
    if (s.frozen) {
        ScopedK tk(ctx, BP_K_IPA_FOLD);
        hipLaunchKernelGGL(k_ipa_fold_ab<C>, dim3(gb), dim3(256), 0, st, s.d_a, s.d_b, (u32)n, words_of<S>(u), words_of<S>(ui));
        hipLaunchKernelGGL(k_ipa_frozen_fold<C>, dim3((u32)((s.n0 + 255) / 256)), dim3(256), 0, st, s.d_cG, s.d_cH, (u32)n, (u32)s.n0, words_of<S>(u), words_of<S>(ui));
        tk.stop();
        HIPCHK(hipGetLastError());
        s.round++; s.n = n; s.lr_done = false;
        return BP_OK;
    }
    {
        ScopedK tk(ctx, BP_K_IPA_FOLD);
        hipLaunchKernelGGL(k_ipa_fold_ab<C>, dim3(gb), dim3(256), 0, st, s.d_a, s.d_b, (u32)n, words_of<S>(u), words_of<S>(ui));
        const bool gf_ok = first && s.have_gf && !s.gf_halves[0].is_zero() && !s.gf_halves[1].is_zero();
        if (gf_ok && s.have_rho) {
            // both halves uniform.  G as below.  H: u*gL*rho^i*H_L + u^-1*gR*rho^(n+i)*H_R = (u^-1*gR*rho^n) * rho^i * (H_R + t*H_L),
            // t = u^2 * (gL/gR) * rho^-n: the pending factor of H stays geometric, K * rho^i.
            const int k = ipa_lg2(n);
            const F4 s2 = S::mul(u, s.gf_halves[1]);
            const F4 ginv = S::inv(s.gf_halves[1]);
            const F4 tG1 = S::mul(S::mul(ui, s.gf_halves[0]), S::inv(s2)), tH1 = S::mul(S::mul(S::sqr(u), S::mul(s.gf_halves[0], ginv)), s.rho_pw[k]);
            bool tab_done = false;
            if (fold_can_defer<C>(ctx, s, n, tG1, tH1)) {
                // TWO rounds from the tables: nothing is materialised now; the next round's L / R run over the tables with split scalars
                // and its fold (below, `s.deferred`) computes Ghat'' / Hhat'' directly
                s.deferred = true; s.def_tG = tG1; s.def_tH = tH1;
                tab_done = true;
                ctx->folds_deferred++;
            } else
            BPCHK(launch_tab_fold<C>(ctx, s, d_G, d_H, n, tG1, tH1, tab_done));   // the bases are the generator tables themselves: fixed-base look-ups
            if (!tab_done) { BPCHK(working_copy()); BPCHK(launch_uniform_fold<C>(ctx, d_G, d_H, n, tG1, tH1, 3)); }
            s.gamma_G = S::mul(s.gamma_G, s2);
            s.gamma_H = S::mul(S::mul(ui, s.gf_halves[1]), s.rho_pw[32 + k]);
            if (s.have_k0) s.gamma_H = S::mul(s.gamma_H, s.geo_k0);
            s.pending = true; s.h_geo = true;
        } else if (gf_ok) {
            BPCHK(working_copy());
            // G: u^-1*gL*G_L + u*gR*G_R = (u*gR) * (G_R + t*G_L), t = u^-1*gL / (u*gR): uniform, one NAF ladder; H: per-lane factors
            const F4 s2 = S::mul(u, s.gf_halves[1]);
            const F4 tG = S::mul(S::mul(ui, s.gf_halves[0]), S::inv(s2));
            BPCHK(launch_uniform_fold<C>(ctx, d_G, d_H, n, tG, tG, 1));
            FoldFinish ff;
            BPCHK(fold_finish_plan(ctx, n, ff));
            hipLaunchKernelGGL(k_ipa_fold_pts<C>, dim3(gb), dim3(256), 0, st, d_G, d_H, s.d_Gf, s.d_Hf, 1, (u32)n, words_of<S>(u), words_of<S>(ui), 2, ff.jac);
            fold_finish_launch<C>(ctx, ff, d_G, d_H, n, 2, n);
            s.gamma_G = S::mul(s.gamma_G, s2);
            s.pending = true;
        } else if (first) {
            BPCHK(working_copy());
            FoldFinish ff;
            BPCHK(fold_finish_plan(ctx, 2 * n, ff));
            hipLaunchKernelGGL(k_ipa_fold_pts<C>, dim3((u32)((2 * n + 255) / 256)), dim3(256), 0, st, d_G, d_H, s.d_Gf, s.d_Hf, 1, (u32)n, words_of<S>(u),
                               words_of<S>(ui), 3, ff.jac);
            fold_finish_launch<C>(ctx, ff, d_G, d_H, n, 3, 2 * n);
        } else {
            // Ghat' = G_R + u^-2 * G_L, gamma_G *= u;   Hhat' = H_R + u^2 * H_L, gamma_H *= u^-1
            // (geometric pending factor: c[i]/c[n+i] = rho^-n joins t, and K picks up rho^n)
            const int k = ipa_lg2(n);
            const F4 t2G = S::sqr(ui), t2H = s.h_geo ? S::mul(S::sqr(u), s.rho_pw[k]) : S::sqr(u);
            if (s.deferred) {
                bool done2 = false;
                BPCHK(launch_tab_fold2<C>(ctx, s, d_G, d_H, n, t2G, t2H, done2));
                if (done2) ctx->folds_tab2++;
                if (!done2) {   // (a multiplier whose digits do not fit: materialise round 1 after all, then the ladder)
                    bool done1 = false;
                    BPCHK(launch_tab_fold<C>(ctx, s, d_G, d_H, 2 * n, s.def_tG, s.def_tH, done1));
                    if (!done1) {
                        if (s.d_G_in != d_G) {
                        HIPCHK(hipMemcpyAsync(d_G, s.d_G_in, 4 * n * 64, hipMemcpyDeviceToDevice, st));
                        HIPCHK(hipMemcpyAsync(d_H, s.d_H_in, 4 * n * 64, hipMemcpyDeviceToDevice, st));
                        }
                        BPCHK(launch_uniform_fold<C>(ctx, d_G, d_H, 2 * n, s.def_tG, s.def_tH, 3));
                    }
                    BPCHK(launch_uniform_fold<C>(ctx, d_G, d_H, n, t2G, t2H, 3));
                }
                s.deferred = false;
            } else
            BPCHK(launch_uniform_fold<C>(ctx, d_G, d_H, n, S::sqr(ui), s.h_geo ? S::mul(S::sqr(u), s.rho_pw[k]) : S::sqr(u), 3));
            s.gamma_G = S::mul(s.gamma_G, u);
            s.gamma_H = s.h_geo ? S::mul(S::mul(s.gamma_H, ui), s.rho_pw[32 + k]) : S::mul(s.gamma_H, ui);
            s.pending = true;
        }
    }
    HIPCHK(hipGetLastError());
    s.first = false;
    if (!s.deferred) s.d_G_in = s.d_H_in = nullptr;   // (a deferred fold keeps reading the generator tables for one more round)
    s.round++;
    s.n = n;
    s.lr_done = false;
    if (s.allow_freeze && n >= 2 && n <= ctx->tune_ipa_freeze_len) {   // the vectors just reached the freeze length: coefficients from here on
        BPCHK(ctx->ipa_cG.ensure(n * 32)); BPCHK(ctx->ipa_cH.ensure(n * 32));
        s.d_cG = ctx->ipa_cG.as<u32>(); s.d_cH = ctx->ipa_cH.as<u32>();
        hipLaunchKernelGGL(k_ipa_freeze_init<C>, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, s.d_cG, s.d_cH, (u32)n, s.pending ? (s.h_geo ? 2 : 1) : 0,
                           words_of<S>(s.gamma_G), words_of<S>(s.gamma_H), s.d_rho_pow);
        HIPCHK(hipGetLastError());
        s.frozen = true; s.n0 = n;
    }
    return BP_OK;
}
// a[0], b[0] -> ark layout on the host (:226-231)
template <class C> static int ipa_finish_dev(bp_ctx* ctx, IpaState& s, uint64_t a_out[4], uint64_t b_out[4]) {
    if (s.n != 1) { g_err = "ipa: finish before the last round"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    BPCHK(ctx->io_out.ensure(64));
    hipLaunchKernelGGL(k_scalars_export<typename C::Fr>, dim3(1), dim3(64), 0, st, s.d_a, ctx->io_out.as<u32>(), 1u);
    hipLaunchKernelGGL(k_scalars_export<typename C::Fr>, dim3(1), dim3(64), 0, st, s.d_b, ctx->io_out.as<u32>() + 8, 1u);
    uint64_t ab[8];
    HIPCHK(hipMemcpyAsync(ab, ctx->io_out.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(ctx));
    HIPCHK(hipGetLastError());
    memcpy(a_out, ab, 32); memcpy(b_out, ab + 4, 32);
    if (ctx->profiling) collect_timers(ctx);
    return BP_OK;
}

// ---- InnerProductProof::create orchestration (src/inner_product_proof.rs:37-239): all vectors device-resident in the engine's
// layouts and consumed (folded in place), like the reference's by-value Vec arguments.
template <class C>
static int ipa_create_dev(bp_ctx* ctx, const u32* d_Q, const u32* d_Gf, const u32* d_Hf, u32* d_G, u32* d_H, u32* d_a, u32* d_b, size_t n,
                          const ChallengeFn& challenge, uint64_t* L_out, uint64_t* R_out, uint64_t a_out[4], uint64_t b_out[4],
                          const F4* gf_halves = nullptr /* optional hint: G_factors == gf_halves[0] on [0,n/2) and gf_halves[1] on [n/2,n) */,
                          const F4* rho_pw = nullptr /* optional hint (with gf_halves): H_factors[i] = rho^i * G_factors[i];
                                                        rho_pw[k] = rho^-(2^k), rho_pw[32+k] = rho^(2^k), k < 32 */,
                          const u32* d_rho_pow = nullptr /* device table of rho^(2^k), resident words */,
                          bool gens_are_tables = false /* d_G, d_H stand for the ctx's generator tables G[0..n), H[0..n) */,
                          bool gens_in_place = false /* ... and were NOT copied: round 1 reads ctx->d_G / d_H, d_G / d_H receive its output */,
                          const F4* q_scalar = nullptr /* Q = q_scalar * B (PedersenGens::B) */,
                          bool direct = false /* (with gens_are_tables and q_scalar, n within the ctx's direct window tables) never fold G and H:
                                                 the coefficients start as the factor vectors, every round's L and R are sums over the tables */) {
    IpaState s;
    BPCHK(ipa_begin_dev<C>(ctx, s, d_Q, d_Gf, d_Hf, d_G, d_H, d_a, d_b, n, gf_halves, rho_pw, d_rho_pow, true));
    if (gens_are_tables) { s.gens_first = 0; s.gens_stride = 1; }
    if (gens_in_place) { s.d_G_in = ctx->d_G.as<u32>(); s.d_H_in = ctx->d_H.as<u32>(); }
    if (n == 1) s.d_G_in = s.d_H_in = nullptr;
    if (q_scalar) { s.have_qw = true; s.qw = *q_scalar; }
    if (direct && gens_are_tables && q_scalar && n >= 2 && ctx->dt_cap >= n) {
        BPCHK(ctx->ipa_cG.ensure(n * 32)); BPCHK(ctx->ipa_cH.ensure(n * 32));
        BPCHK(ctx->ipa_sL.ensure((2 * n + 2) * 32)); BPCHK(ctx->ipa_sR.ensure((2 * n + 2) * 32));
        BPCHK(ctx->ipa_part.ensure(((n + 255) / 256 + 1) * 64));
        HIPCHK(hipMemcpyAsync(ctx->ipa_cG.p, d_Gf, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->ipa_cH.p, d_Hf, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
        BPCHK(ctx->dt_a2.ensure(n * 16)); BPCHK(ctx->dt_b2.ensure(n * 16));   // (the folded vectors: at most n / 2 elements)
        BPCHK(ctx->dt_ticket.ensure(64));
        HIPCHK(hipMemsetAsync(ctx->dt_ticket.p, 0, 64, ctx->stream));   // (the last workgroup of k_dt_round leaves it at zero; an aborted launch would not)
        s.d_cG = ctx->ipa_cG.as<u32>(); s.d_cH = ctx->ipa_cH.as<u32>();
        s.d_a_alt = ctx->dt_a2.as<u32>(); s.d_b_alt = ctx->dt_b2.as<u32>();
        s.d_G_in = s.d_H_in = nullptr;
        s.frozen = true; s.direct = true; s.n0 = n;
    }
    while (s.n != 1) {
        uint64_t Lw[8], Rw[8], uw[4];
        BPCHK(ipa_round_lr<C>(ctx, s, Lw, Rw));
        memcpy(L_out + 8 * s.round, Lw, 64); memcpy(R_out + 8 * s.round, Rw, 64);
        int rc = challenge(Lw, Rw, uw);
        if (rc) { g_err = "ipa_create: challenge callback failed"; return rc < 0 ? rc : BP_E_ARG; }
        BPCHK(ipa_round_fold<C>(ctx, s, uw));
    }
    return ipa_finish_dev<C>(ctx, s, a_out, b_out);
}

// ---- index-cyclic InnerProductProof::create inside a sharded prover -------------------------------------------
// Rank r of `world` keeps the elements i = r + j*world of a, b, G, H and of the factor vectors: element i and its fold partner
// i + n/2 live on the same rank while n >= 2*world, so every fold is local and 1/world of the single-GPU work.  A round's L and R
// are sums of per-rank partial MSMs (the <a_L, b_R> * Q term is linear too): the ctx's point-reduce callback.  When the global
// length reaches the frozen-tail length the ranks all-gather what is left (a few hundred elements) and every rank finishes the
// remaining rounds over the frozen generators, replicated.  L, R, a, b are bit-identical to the single-GPU result.
template <class C>
static int ipa_create_cyclic(bp_ctx* ctx, const u32* d_Q, const u32* d_Gf, const u32* d_Hf, const u32* d_Gtab, const u32* d_Htab, const u32* d_a, const u32* d_b,
                             size_t N, const ChallengeFn& challenge, uint64_t* L_out, uint64_t* R_out, uint64_t a_out[4], uint64_t b_out[4],
                             const F4* gf_halves, const F4* rho_pw, const u32* d_rho_pow, const F4* qw = nullptr /* Q = qw * B (the R1CS prover's Q) */) {
    typedef typename C::Fr FrP;
    typedef host::Fld<FrP> S;
    hipStream_t st = ctx->stream;
    const size_t W = (size_t)ctx->shard_world, r = (size_t)ctx->shard_rank;
    const int w = ipa_lg2(W);
    const size_t n_loc = N / W;
    size_t S_glob = std::max<size_t>(std::max<size_t>(ctx->tune_ipa_freeze_len, W), 2);   // global length at which the ranks gather
    { size_t p = 1; while (p < S_glob) p <<= 1; S_glob = p; }
    const size_t S_loc = S_glob / W;
    // this rank's slices (the generator tables are read in place: no full working copy)
    BPCHK(ctx->cyc_a.ensure(std::max(n_loc, S_glob) * 32)); BPCHK(ctx->cyc_b.ensure(std::max(n_loc, S_glob) * 32));
    BPCHK(ctx->cyc_Gf.ensure(n_loc * 32)); BPCHK(ctx->cyc_Hf.ensure(n_loc * 32));
    BPCHK(ctx->ipa_G.ensure(std::max(n_loc, S_glob) * 64)); BPCHK(ctx->ipa_H.ensure(std::max(n_loc, S_glob) * 64));
    const u32 gb = (u32)((n_loc + 255) / 256);
    hipLaunchKernelGGL(k_cyclic_gather<Blk32>, dim3(gb), dim3(256), 0, st, d_a, ctx->cyc_a.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    hipLaunchKernelGGL(k_cyclic_gather<Blk32>, dim3(gb), dim3(256), 0, st, d_b, ctx->cyc_b.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    hipLaunchKernelGGL(k_cyclic_gather<Blk32>, dim3(gb), dim3(256), 0, st, d_Gf, ctx->cyc_Gf.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    hipLaunchKernelGGL(k_cyclic_gather<Blk32>, dim3(gb), dim3(256), 0, st, d_Hf, ctx->cyc_Hf.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    hipLaunchKernelGGL(k_cyclic_gather<Blk64>, dim3(gb), dim3(256), 0, st, d_Gtab, ctx->ipa_G.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    hipLaunchKernelGGL(k_cyclic_gather<Blk64>, dim3(gb), dim3(256), 0, st, d_Htab, ctx->ipa_H.as<u32>(), (u32)n_loc, (u32)r, (u32)W);
    HIPCHK(hipGetLastError());
    // local geometric hint: H_factors[r + j*W] = rho^r * (rho^W)^j * G_factors: tables of rho^W = the global tables shifted by lg W
    F4 rho_loc[64];
    const bool geo = gf_halves && rho_pw && d_rho_pow;
    if (geo) for (int k = 0; k < 32; k++) { rho_loc[k] = k + w < 32 ? rho_pw[k + w] : S::one(); rho_loc[32 + k] = k + w < 32 ? rho_pw[32 + k + w] : S::one(); }
    IpaState s;
    BPCHK(ipa_begin_dev<C>(ctx, s, d_Q, ctx->cyc_Gf.as<u32>(), ctx->cyc_Hf.as<u32>(), ctx->ipa_G.as<u32>(), ctx->ipa_H.as<u32>(), ctx->cyc_a.as<u32>(),
                           ctx->cyc_b.as<u32>(), n_loc, gf_halves, geo ? rho_loc : nullptr, geo ? d_rho_pow + (size_t)w * 8 : nullptr, false));
    s.msm_mode = 2;
    s.gens_first = (u32)r; s.gens_stride = (u32)W;
    s.d_G_in = s.d_G; s.d_H_in = s.d_H;   // round 1 reads the slice where it lies (and may defer its fold: two rounds from the tables, as on one GPU)
    s.min_len = S_loc;
    if (qw) { s.qw = *qw; s.have_qw = true; }
    F4 rho_r = S::one(), rho_mr = S::one();   // rho^rank, rho^-rank
    if (geo) for (int k = 0; k < w; k++) if ((r >> k) & 1) { rho_r = S::mul(rho_r, rho_pw[32 + k]); rho_mr = S::mul(rho_mr, rho_pw[k]); }
    s.geo_k0 = rho_r; s.have_k0 = geo;
    while (s.n > S_loc) {
        uint64_t Lw[8], Rw[8], uw[4];
        BPCHK(ipa_round_lr<C>(ctx, s, Lw, Rw));
        memcpy(L_out + 8 * s.round, Lw, 64); memcpy(R_out + 8 * s.round, Rw, 64);
        int rc = challenge(Lw, Rw, uw);
        if (rc) { g_err = "ipa_create: challenge callback failed"; return rc < 0 ? rc : BP_E_ARG; }
        BPCHK(ipa_round_fold<C>(ctx, s, uw));
    }
    // gather: [a | b | G | H] of the S_loc local elements, ark layouts, from every rank; global index = rank + j * W
    const size_t m = s.n;   // == S_loc
    const size_t per = m * (32 + 32 + 64 + 64);
    std::vector<host::u8> send(per), recv(per * W);
    {
        const u32 g2 = (u32)((m + 255) / 256);
        BPCHK(ctx->io_out.ensure(per));
        u32* o = ctx->io_out.as<u32>();
        hipLaunchKernelGGL(k_scalars_export<FrP>, dim3(g2), dim3(256), 0, st, s.d_a, o, (u32)m);
        hipLaunchKernelGGL(k_scalars_export<FrP>, dim3(g2), dim3(256), 0, st, s.d_b, o + m * 8, (u32)m);
        hipLaunchKernelGGL(k_points_dev_to_ark<C>, dim3(g2), dim3(256), 0, st, s.d_G, o + m * 16, (u32)m);
        hipLaunchKernelGGL(k_points_dev_to_ark<C>, dim3(g2), dim3(256), 0, st, s.d_H, o + m * 32, (u32)m);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(send.data(), o, per, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx_stream_wait(ctx));
    }
    BPCHK(shard_allgather(ctx, send.data(), per, recv.data()));
    std::vector<F4> ga(S_glob), gbv(S_glob);
    std::vector<A4> gG(S_glob), gH(S_glob);
    for (size_t rr = 0; rr < W; rr++) {
        const host::u8* blk = recv.data() + rr * per;
        for (size_t j = 0; j < m; j++) {
            const size_t i = rr + j * W;
            memcpy(&ga[i], blk + j * 32, 32); memcpy(&gbv[i], blk + m * 32 + j * 32, 32);
            memcpy(&gG[i], blk + m * 64 + j * 64, 64); memcpy(&gH[i], blk + m * 128 + j * 64, 64);
        }
    }
    BPCHK(upload_scalars<C>(ctx, ctx->cyc_a.as<u32>(), ga.data(), S_glob));
    BPCHK(upload_scalars<C>(ctx, ctx->cyc_b.as<u32>(), gbv.data(), S_glob));
    HIPCHK(hipMemcpyAsync(ctx->ipa_G.p, gG.data(), S_glob * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_H.p, gH.data(), S_glob * 64, hipMemcpyHostToDevice, st));
    BPCHK(bp_points_import(ctx, ctx->ipa_G.p, ctx->ipa_G.p, S_glob));
    BPCHK(bp_points_import(ctx, ctx->ipa_H.p, ctx->ipa_H.p, S_glob));
    // the tail: generators frozen as gathered, their pending factors as per-element coefficients.  In the global index the
    // geometric factor is K_common * rho^i with K_common = (this rank's K) * rho^-rank — the same on every rank.
    IpaState t;
    BPCHK(ipa_begin_dev<C>(ctx, t, d_Q, nullptr, nullptr, ctx->ipa_G.as<u32>(), ctx->ipa_H.as<u32>(), ctx->cyc_a.as<u32>(), ctx->cyc_b.as<u32>(), S_glob, nullptr, nullptr,
                           nullptr, false));
    BPCHK(ctx->ipa_sL.ensure((2 * S_glob + 1) * 32)); BPCHK(ctx->ipa_sR.ensure((2 * S_glob + 1) * 32));
    BPCHK(ctx->ipa_part.ensure(((S_glob + 255) / 256 + 1) * 64));
    t.round = s.round; t.first = false; t.msm_mode = 0;
    BPCHK(ctx->ipa_cG.ensure(S_glob * 32)); BPCHK(ctx->ipa_cH.ensure(S_glob * 32));
    t.d_cG = ctx->ipa_cG.as<u32>(); t.d_cH = ctx->ipa_cH.as<u32>();
    const F4 kH = s.h_geo ? S::mul(s.gamma_H, rho_mr) : s.gamma_H;
    hipLaunchKernelGGL(k_ipa_freeze_init<C>, dim3((u32)((S_glob + 255) / 256)), dim3(256), 0, st, t.d_cG, t.d_cH, (u32)S_glob, s.pending ? (s.h_geo ? 2 : 1) : 0,
                       words_of<S>(s.gamma_G), words_of<S>(kH), d_rho_pow);
    HIPCHK(hipGetLastError());
    HIPCHK(ctx_stream_wait(ctx));   // ga .. gH are locals
    t.frozen = true; t.n0 = S_glob;
    while (t.n != 1) {
        uint64_t Lw[8], Rw[8], uw[4];
        BPCHK(ipa_round_lr<C>(ctx, t, Lw, Rw));
        memcpy(L_out + 8 * t.round, Lw, 64); memcpy(R_out + 8 * t.round, Rw, 64);
        int rc = challenge(Lw, Rw, uw);
        if (rc) { g_err = "ipa_create: challenge callback failed"; return rc < 0 ? rc : BP_E_ARG; }
        BPCHK(ipa_round_fold<C>(ctx, t, uw));
    }
    return ipa_finish_dev<C>(ctx, t, a_out, b_out);
}
// whether a sharded prover can partition an IPA of padded size N this way
static inline bool ipa_cyclic_applies(const bp_ctx* ctx, size_t N) {
    const size_t W = (size_t)ctx->shard_world;
    if (W <= 1 || !shard_has_allgather(ctx) || (W & (W - 1)) || N < ctx->tune_cyclic_min) return false;
    size_t S_glob = std::max<size_t>(std::max<size_t>(ctx->tune_ipa_freeze_len, W), 2);
    { size_t p = 1; while (p < S_glob) p <<= 1; S_glob = p; }
    return N >= 2 * S_glob && N / W >= 2;   // at least one partitioned round
}

// uploads the host vectors of an IPA instance into the ctx's working buffers (engine layouts)
template <class C>
static int ipa_upload_host(bp_ctx* ctx, const uint64_t* Q, const uint64_t* Gf, const uint64_t* Hf, const uint64_t* Gv, const uint64_t* Hv, const uint64_t* a,
                           const uint64_t* b, size_t n) {
    typedef typename C::Fr Fr;
    hipStream_t st = ctx->stream;
    BPCHK(ctx->ipa_G.ensure(n * 64)); BPCHK(ctx->ipa_H.ensure(n * 64));
    BPCHK(ctx->ipa_a.ensure(n * 32)); BPCHK(ctx->ipa_b.ensure(n * 32));
    BPCHK(ctx->ipa_Gf.ensure(n * 32)); BPCHK(ctx->ipa_Hf.ensure(n * 32));
    BPCHK(ctx->ipa_Q.ensure(64));
    HIPCHK(hipMemcpyAsync(ctx->ipa_G.p, Gv, n * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_H.p, Hv, n * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_Q.p, Q, 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_a.p, a, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_b.p, b, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_Gf.p, Gf, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->ipa_Hf.p, Hf, n * 32, hipMemcpyHostToDevice, st));
    BPCHK(bp_points_import(ctx, ctx->ipa_G.p, ctx->ipa_G.p, n));
    BPCHK(bp_points_import(ctx, ctx->ipa_H.p, ctx->ipa_H.p, n));
    BPCHK(bp_points_import(ctx, ctx->ipa_Q.p, ctx->ipa_Q.p, 1));
    const u32 gb = (u32)((n + 255) / 256);
    DevBuf* sc[] = {&ctx->ipa_a, &ctx->ipa_b, &ctx->ipa_Gf, &ctx->ipa_Hf};
    for (auto s : sc) hipLaunchKernelGGL(k_scalars_import<Fr>, dim3(gb), dim3(256), 0, st, s->as<u32>(), s->as<u32>(), (u32)n);
    HIPCHK(hipGetLastError());
    HIPCHK(ctx_stream_wait(ctx));   // the host buffers are the caller's: done with them before returning
    return BP_OK;
}
template <class C>
static int ipa_create_host_entry(bp_ctx* ctx, const uint64_t* Q, const uint64_t* Gf, const uint64_t* Hf, const uint64_t* Gv, const uint64_t* Hv,
                                 const uint64_t* a, const uint64_t* b, size_t n, const ChallengeFn& fn, uint64_t* L_out, uint64_t* R_out,
                                 uint64_t* a_out, uint64_t* b_out) {
    BPCHK(ipa_upload_host<C>(ctx, Q, Gf, Hf, Gv, Hv, a, b, n));
    return ipa_create_dev<C>(ctx, ctx->ipa_Q.as<u32>(), ctx->ipa_Gf.as<u32>(), ctx->ipa_Hf.as<u32>(), ctx->ipa_G.as<u32>(), ctx->ipa_H.as<u32>(),
                             ctx->ipa_a.as<u32>(), ctx->ipa_b.as<u32>(), n, fn, L_out, R_out, a_out, b_out);
}
template <class C>
static int ipa_begin_host_entry(bp_ctx* ctx, const uint64_t* Q, const uint64_t* Gf, const uint64_t* Hf, const uint64_t* Gv, const uint64_t* Hv,
                                const uint64_t* a, const uint64_t* b, size_t n) {
    ctx->ipa_step_active = false;
    if (n == 0 || (n & (n - 1))) { g_err = "ipa_begin: n must be a power of two (reference asserts, src/inner_product_proof.rs:66)"; return BP_E_ARG; }
    BPCHK(ipa_upload_host<C>(ctx, Q, Gf, Hf, Gv, Hv, a, b, n));
    BPCHK(ipa_begin_dev<C>(ctx, ctx->ipa_step, ctx->ipa_Q.as<u32>(), ctx->ipa_Gf.as<u32>(), ctx->ipa_Hf.as<u32>(), ctx->ipa_G.as<u32>(), ctx->ipa_H.as<u32>(),
                           ctx->ipa_a.as<u32>(), ctx->ipa_b.as<u32>(), n, nullptr, nullptr, nullptr));
    ctx->ipa_step_active = true;
    return BP_OK;
}
// current a, b, G, H (first n_cur elements, ark layout) and the pending factors: G_true[i] = gamma_G * G[i], H_true[i] = gamma_H * H[i]
template <class C>
static int ipa_export_host(bp_ctx* ctx, uint64_t* a, uint64_t* b, uint64_t* G_xy, uint64_t* H_xy, uint64_t gG[4], uint64_t gH[4], size_t* n_cur) {
    typedef typename C::Fr Fr;
    IpaState& s = ctx->ipa_step;
    if (s.lr_done) { g_err = "ipa_export: between round_LR and round_fold"; return BP_E_ARG; }
    hipStream_t st = ctx->stream;
    const size_t n = s.n;
    const u32 gb = (u32)((n + 255) / 256);
    BPCHK(ctx->io_out.ensure(n * 64 * 2 + n * 32 * 2));
    u32* o = ctx->io_out.as<u32>();
    hipLaunchKernelGGL(k_points_dev_to_ark<C>, dim3(gb), dim3(256), 0, st, s.d_G, o, (u32)n);
    hipLaunchKernelGGL(k_points_dev_to_ark<C>, dim3(gb), dim3(256), 0, st, s.d_H, o + n * 16, (u32)n);
    hipLaunchKernelGGL(k_scalars_export<Fr>, dim3(gb), dim3(256), 0, st, s.d_a, o + n * 32, (u32)n);
    hipLaunchKernelGGL(k_scalars_export<Fr>, dim3(gb), dim3(256), 0, st, s.d_b, o + n * 40, (u32)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(G_xy, o, n * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H_xy, o + n * 16, n * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a, o + n * 32, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b, o + n * 40, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(ctx));
    memcpy(gG, s.gamma_G.v, 32); memcpy(gH, s.gamma_H.v, 32);
    *n_cur = n;
    return BP_OK;
}

#include "r1cs_host.inc"

// ---- unit-test kernels -----------------------------------------------------------------------------
template <class F> __global__ void k_dbg_field(int op, const u32* a, const u32* b, u32* out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wa[8], wb[8];
    load_words8(wa, a + (size_t)i * 8); load_words8(wb, b + (size_t)i * 8);
    Fe x = fe_load_ark<F>(wa), y = fe_load_ark<F>(wb), r;
    switch (op) {
        case 0: r = fe_mul<F>(x, y); break;
        case 1: r = fe_norm(fe_add(x, y)); break;
        case 2: r = fe_sub<F, 2>(x, y); break;
        case 3: r = fe_sqr<F>(x); break;
        default: r = fe_inv<F>(x); break;
    }
    fe_store_ark<F>(wa, r);
    store_words8(out + (size_t)i * 8, wa);
}
template <class C> __global__ void k_dbg_point(int op, const u32* p, const u32* q, const u32* k, u32* out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[16];
    load_words8(w, p + (size_t)i * 16); load_words8(w + 8, p + (size_t)i * 16 + 8);
    Aff P = aff_load_ark<C>(w);
    load_words8(w, q + (size_t)i * 16); load_words8(w + 8, q + (size_t)i * 16 + 8);
    Aff Q = aff_load_ark<C>(w);
    Jac r;
    if (op == 0) r = jac_add<C>(jac_from_aff<C>(P), jac_from_aff<C>(Q));
    else if (op == 1) r = jac_madd<C>(jac_from_aff<C>(P), Q);
    else if (op == 2) r = jac_dbl<C>(jac_from_aff<C>(P));
    else {
        u32 kk[8];
        load_words8(kk, k + (size_t)i * 8);
        r = jac_inf<C>();
        for (int b = 255; b >= 0; b--) {
            r = jac_dbl<C>(r);
            if ((kk[b >> 5] >> (b & 31)) & 1) r = jac_madd<C>(r, P);
        }
    }
    Aff a = jac_to_aff<C>(r);
    aff_store_ark<C>(w, a);
    store_words8(out + (size_t)i * 16, w); store_words8(out + (size_t)i * 16 + 8, w + 8);
}

template <class C> static int dbg_rng_draws(void* transcript, const uint64_t* witness, size_t nw, const uint8_t* seeds, int lanes, size_t count, uint64_t* out) {
    typedef typename C::Fr FrP; typedef host::Fld<FrP> S;
    std::vector<std::unique_ptr<host::TranscriptRng>> rngs;
    for (int j = 0; j < lanes; j++) {
        rngs.emplace_back(new host::TranscriptRng(*(host::Transcript*)transcript));
        for (size_t i = 0; i < nw; i++) { F4 w; memcpy(w.v, witness + 4 * i, 32); host::u8 b[32]; S::to_bytes(b, w); rngs[j]->rekey("v_blinding", b, 32); }
        host::ChaChaRng ext(seeds + 32 * j);
        rngs[j]->finalize(ext);
    }
    if (lanes == 1) { for (size_t i = 0; i < count; i++) { F4 x = host::rand_fe<FrP>(*rngs[0]); memcpy(out + 4 * i, x.v, 32); } return BP_OK; }
#if defined(__x86_64__)
    if (lanes != 8 || !host::cpu_has_avx512() || count < 3) return BP_E_ARG;
    host::TranscriptRng* rp[8];
    for (int j = 0; j < 8; j++) {
        for (size_t i = 0; i < 3; i++) { F4 x = host::rand_fe<FrP>(*rngs[j]); memcpy(out + 4 * (j * count + i), x.v, 32); }
        rp[j] = rngs[j].get();
        if (rp[j]->s.pos != 8 || rp[j]->s.pos_begin != 0) return BP_E_ARG;
    }
    std::vector<u64> words((count - 3) * 4 * 8);
    host::transcript_rng_x8_words(rp, words.data(), (count - 3) * 4);
    for (int j = 0; j < 8; j++)
        for (size_t i = 3; i < count; i++)
            for (int w = 0; w < 4; w++) {
                u64 x = words[(4 * (i - 3) + w) * 8 + j];
                if (w == 3 && FrP::BITS < 256) x &= (~(u64)0) >> (256 - FrP::BITS);  // Fp::rand masks the top limb
                out[4 * (j * count + i) + w] = x;
            }
    // the scalar continuation from the x8 state must also agree: one more draw each, returned in place of the last one
    for (int j = 0; j < 8; j++) { F4 x = host::rand_fe<FrP>(*rngs[j]); (void)x; }
    return BP_OK;
#else
    return BP_E_ARG;
#endif
}

// ---- recorder handles (see include/arkbp.h): a Prover<G, T> or Verifier<G, T> of the reference as seen through the C ABI ----
struct bp_cs {
    int curve = 0;
    bool proving = false;
    host::Transcript* tr = nullptr;                 // the caller's transcript handle (borrowed, like `T: BorrowMut<Transcript>`)
    std::unique_ptr<host::Transcript> owned_tr;     // statements own theirs
    std::unique_ptr<host::ConstraintSystem<Secq>> cs0;
    std::unique_ptr<host::ConstraintSystem<Zorro>> cs1;
    ProvePre<Secq> pre0;
    ProvePre<Zorro> pre1;
    std::unique_ptr<HostCsc> csc;                   // constraint index of a single-phase statement (built on first need, outside prove())
    bool csc_tried = false;
    host::u8 rng32[32] = {0};                       // what prove() draws from the external rng
    bool have_rng = false;
    bool consumed = false;                          // prove(self) / verify(self)
    bool running = false;                           // inside prove / verify: the randomized-phase callbacks may still record
    std::vector<A4> commitments;                    // V points in commit order (both modes)
    virtual ~bp_cs() {}
    template <class C> host::ConstraintSystem<C>* cs();
    size_t num_vars() const { return curve == 0 ? cs0->num_vars : cs1->num_vars; }
    size_t num_constraints() const { return curve == 0 ? cs0->num_constraints() : cs1->num_constraints(); }
};
template <> host::ConstraintSystem<Secq>* bp_cs::cs<Secq>() { return cs0.get(); }
template <> host::ConstraintSystem<Zorro>* bp_cs::cs<Zorro>() { return cs1.get(); }
// scenario statements: a prover handle plus what the scenario made public
struct bp_stmt : bp_cs {
    int scenario = 0;
    host::StatementIO io;
};

template <class C> static void cs_handle_init(bp_cs* h, bool proving, host::Transcript* tr) {
    h->curve = C::ID; h->proving = proving; h->tr = tr;
    auto* cs = new host::ConstraintSystem<C>();
    cs->tr = tr; cs->proving = proving; cs->owner = h;
    if (C::ID == 0) h->cs0.reset((host::ConstraintSystem<Secq>*)cs); else h->cs1.reset((host::ConstraintSystem<Zorro>*)cs);
    host::TP<C>::r1cs_domain_sep(*tr);   // Prover::new (prover.rs:291-308) / Verifier::new (verifier.rs:252-263)
}
template <class C> static int stmt_build(bp_stmt* s, const uint64_t* params, const uint8_t* seed, bp_ctx* ctx = nullptr) {
    s->owned_tr.reset(new host::Transcript(host::scenario_label(s->scenario)));
    cs_handle_init<C>(s, true, s->owned_tr.get());
    host::ChaChaRng prng(seed);
    host::PedersenGens<C> pc = host::PedersenGens<C>::make_default();
    if (ctx) pedersen_attach<C>(ctx, pc);
    int rc = host::scenario_prover<C>(*s->cs<C>(), pc, prng, s->scenario, params, s->io);
    if (rc) return rc;
    s->commitments = s->io.commitments;
    prng.fill_bytes(s->rng32, 32); s->have_rng = true;   // the scenario's external rng is consumed by prove() for exactly these bytes
    s->csc.reset(new HostCsc());
    if (!build_host_csc<C>(*s->cs<C>(), *s->csc)) s->csc.reset();
    s->csc_tried = true;
    return BP_OK;
}
template <class C> static int cs_prove(bp_ctx* c, bp_cs* s, uint8_t* proof_out, size_t* proof_len, double* timing) {
    host::ProofData pf;
    StageTimes tm;
    auto* pre = C::ID == 0 ? (ProvePre<C>*)&s->pre0 : (ProvePre<C>*)&s->pre1;
    if (!s->csc_tried) {   // single-phase circuits: the constraint index is statement construction, not prove()
        s->csc.reset(new HostCsc());
        if (!build_host_csc<C>(*s->cs<C>(), *s->csc)) s->csc.reset();
        s->csc_tried = true;
    }
    int rc = r1cs_prove<C>(c, *s->cs<C>(), s->rng32, pf, tm, pre, s->csc.get());
    if (rc) return rc;
    std::vector<host::u8> bytes = host::proof_to_bytes<C>(pf);
    if (bytes.size() > *proof_len) { g_err = "prove: output buffer too small"; return BP_E_ARG; }
    memcpy(proof_out, bytes.data(), bytes.size()); *proof_len = bytes.size();
    if (timing) { timing[0] = tm.total; timing[1] = 0; timing[2] = tm.rng; timing[3] = tm.upload; timing[4] = tm.commit_msm; timing[5] = tm.flatten; timing[6] = tm.poly; timing[7] = tm.ipa; }
    return BP_OK;
}
template <class C> static int cs_precompute_batch(bp_cs** hs, size_t count) {
    for (size_t g = 0; g < count; g += 8) {
        const size_t c = std::min<size_t>(8, count - g);
        host::ConstraintSystem<C>* css[8]; ProvePre<C>* pres[8]; const host::u8* rngs[8];
        for (size_t j = 0; j < c; j++) {
            css[j] = hs[g + j]->cs<C>(); rngs[j] = hs[g + j]->rng32;
            pres[j] = C::ID == 0 ? (ProvePre<C>*)&hs[g + j]->pre0 : (ProvePre<C>*)&hs[g + j]->pre1;
        }
        prove_precompute_batch<C>(css, rngs, pres, c);
    }
    return BP_OK;
}
// batch_verify over recorded verifier handles (each consumed, like `verify(self)`)
template <class C>
static int cs_batch_verify(bp_ctx* c, size_t count, bp_cs* const* vs, const uint8_t* proofs, const size_t* proof_lens, const uint64_t* alphas, double* timing,
                           uint64_t* point_out) {
    typedef host::Fld<typename C::Fr> S;
    std::vector<size_t> poff(count + 1, 0);
    for (size_t k = 0; k < count; k++) poff[k + 1] = poff[k] + proof_lens[k];
    std::vector<F4> al(count);
    for (size_t k = 0; k < count; k++) { if (alphas) memcpy(al[k].v, alphas + 4 * k, 32); else al[k] = S::one(); }
    VfyProvider<C> prov;
    prov.m_of = [&](size_t k) { return vs[k]->cs<C>()->V.size(); };
    prov.get = [&](size_t k, VfyInstance<C>& out) -> int { out.cs = vs[k]->cs<C>(); return BP_OK; };
    // like-instances of one single-phase recording (bp_verifier_new_like): the device front end continues their transcripts from
    // where Verifier::commit left them
    prov.dev_batch = [&](VfyDevBatch<C>& db) -> bool {
        const host::ConstraintSystem<C>* c0 = vs[0]->cs<C>();
        if (!c0->base || !c0->tr) return false;
        for (size_t k = 0; k < count; k++) {
            const host::ConstraintSystem<C>* ck = vs[k]->cs<C>();
            if (ck->base != c0->base || ck->cs_off.size() != 1 || !ck->deferred.empty() || ck->phase2 || ck->num_vars != c0->num_vars || ck->V.size() != c0->V.size() || !ck->tr) return false;
        }
        db.src = c0; db.m = c0->V.size(); db.absorb_commitments = false; db.shared_state = false;
        db.state_of = [&](size_t k) { return (const host::Strobe*)&vs[k]->cs<C>()->tr->s; };
        db.commit_xy = [&](size_t k) { return (const uint64_t*)vs[k]->cs<C>()->V.data(); };
        return true;
    };
    for (size_t k = 0; k < count; k++) { vs[k]->consumed = true; vs[k]->running = true; }
    const int rc = batch_verify_core<C>(c, count, prov, proofs, poff.data(), al.data(), timing, point_out);
    for (size_t k = 0; k < count; k++) vs[k]->running = false;
    return rc;
}

// ---- debug hooks for the reference-held constants (src/util.rs:147-166, src/inner_product_proof.rs:556-562) ----
// out[i] = x^i for i < n, through pow_table — the device function with which the prover / verifier kernels form y^i, y^-i and z^q
// (the reference's exp_iter, util.rs:55-58).  xtab: x^(2^k), k < 32, resident words.
// host only: the challenge sequence of up to eight scenario verifications, from the per-proof live transcript (verify_prepare_t over
// LiveTr) or from the lockstep replay (replay_challenges_x8) — both must follow ONE Fiat-Shamir schedule
template <class C> struct LogLiveTr {
    LiveTr<C> in;
    std::vector<F4>* log;
    void append_u64(const char* l, uint64_t x) { in.append_u64(l, x); }
    bool vpoint(const char* l, const A4& p) { return in.vpoint(l, p); }
    void point(const char* l, const A4& p) { in.point(l, p); }
    void scalar(const char* l, const F4& x) { in.scalar(l, x); }
    F4 chal(const char* l) { const F4 c = in.chal(l); log->push_back(c); return c; }
    void ipp(uint64_t n) { in.ipp(n); }
    F4 chal_r() { const F4 c = in.chal_r(); log->push_back(c); return c; }
    bool inverse(F4&) { return false; }
};
template <class C>
static int dbg_verify_challenges(size_t count, const int* scenarios, const uint64_t* params, const uint8_t* proofs, const size_t* proof_lens, const uint64_t* commit_xy,
                                 const size_t* ms, const uint64_t* publics, const size_t* npubs, int use_x8, uint64_t* out, size_t* nchal) {
    std::vector<std::unique_ptr<host::Transcript>> trs(count);
    std::vector<std::unique_ptr<host::ConstraintSystem<C>>> css(count);
    std::vector<host::ProofData> pfs(count);
    size_t po = 0, co = 0, uo = 0;
    for (size_t k = 0; k < count; k++) {
        host::StatementIO io;
        io.commitments.resize(ms[k]); io.publics.resize(npubs[k]);
        if (ms[k]) memcpy(io.commitments.data(), commit_xy + 8 * co, ms[k] * 64);
        if (npubs[k]) memcpy(io.publics.data(), publics + 4 * uo, npubs[k] * 32);
        trs[k].reset(new host::Transcript(host::scenario_label(scenarios[k])));
        css[k].reset(new host::ConstraintSystem<C>());
        BPCHK(build_verifier_cs<C>(*css[k], *trs[k], scenarios[k], params + 8 * k, io));
        BPCHK(host::proof_from_bytes<C>(pfs[k], proofs + po, proof_lens[k]));
        po += proof_lens[k]; co += ms[k]; uo += npubs[k];
    }
    if (use_x8) {
#if defined(__x86_64__)
        if (count != 8 || !host::cpu_has_avx512()) return 1;
        host::ConstraintSystem<C>* csp[8]; const host::ProofData* pfp[8];
        for (int j = 0; j < 8; j++) { csp[j] = css[j].get(); pfp[j] = &pfs[j]; }
        std::vector<F4> ch[8];
        if (!replay_challenges_x8<C>(csp, pfp, ch)) return 1;
        for (int j = 0; j < 8; j++) { nchal[j] = ch[j].size(); if (ch[j].size() > 40) return BP_E_ARG; if (!ch[j].empty()) memcpy(out + (size_t)j * 40 * 4, ch[j].data(), ch[j].size() * 32); }
        return BP_OK;
#else
        return 1;
#endif
    }
    for (size_t k = 0; k < count; k++) {
        std::vector<F4> log;
        LogLiveTr<C> T{LiveTr<C>{*trs[k]}, &log};
        VerifyPrep<C> vp;
        const int rc = verify_prepare_t<C>(*css[k], pfs[k], (size_t)-1, vp, T);
        if (rc) return rc;
        if (log.size() > 40) return BP_E_ARG;
        nchal[k] = log.size();
        if (!log.empty()) memcpy(out + k * 40 * 4, log.data(), log.size() * 32);
    }
    return BP_OK;
}
template <class C> __global__ void k_dbg_scalars_to_ark(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fe_store_ark<typename C::Fr>(w, load_fe_dev<typename C::Fr>(in + (size_t)i * 8));
    store_words8(out + (size_t)i * 8, w);
}
template <class F> __global__ void k_dbg_exp_iter(const u32* __restrict__ xtab, u32 n, u32* __restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fe_store_ark<F>(w, pow_table<F>(xtab, i));
    store_words8(out + (size_t)i * 8, w);
}
template <class C> static int dbg_exp_iter(bp_ctx* ctx, const uint64_t* x, size_t n, uint64_t* out) {
    typedef host::Fld<typename C::Fr> S;
    F4 tab[32]; memcpy(tab[0].v, x, 32);
    for (int k = 1; k < 32; k++) tab[k] = S::sqr(tab[k - 1]);
    BPCHK(ctx->r_ypow.ensure(64 * 32)); BPCHK(ctx->io_out.ensure(std::max<size_t>(n, 1) * 32));
    BPCHK(upload_scalars<C>(ctx, ctx->r_ypow.as<u32>(), tab, 32));
    hipLaunchKernelGGL(k_dbg_exp_iter<typename C::Fr>, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->r_ypow.as<u32>(), (u32)n, ctx->io_out.as<u32>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, ctx->io_out.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx_stream_wait(ctx));
    return BP_OK;
}
// <a, b> over the scalar field with the kernels InnerProductProof::create uses for c_L = <a_L, b_R> (k_ipa_scalars +
// k_ipa_ip_finish; inner_product_proof.rs:83-84, 390-399): the inputs are laid out as a || 0 and 0 || b, so c_L = <a, b>, c_R = 0.
template <class C> static int dbg_inner_product(bp_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    typedef host::Fld<typename C::Fr> S;
    hipStream_t st = ctx->stream;
    std::vector<F4> av(2 * n, S::zero()), bv(2 * n, S::zero()), ones(2 * n, S::one());
    memcpy(av.data(), a, n * 32); memcpy(bv.data() + n, b, n * 32);
    BPCHK(ctx->ipa_a.ensure(2 * n * 32)); BPCHK(ctx->ipa_b.ensure(2 * n * 32)); BPCHK(ctx->ipa_Gf.ensure(2 * n * 32)); BPCHK(ctx->ipa_Hf.ensure(2 * n * 32));
    BPCHK(upload_scalars<C>(ctx, ctx->ipa_a.as<u32>(), av.data(), 2 * n)); BPCHK(upload_scalars<C>(ctx, ctx->ipa_b.as<u32>(), bv.data(), 2 * n));
    BPCHK(upload_scalars<C>(ctx, ctx->ipa_Gf.as<u32>(), ones.data(), 2 * n)); BPCHK(upload_scalars<C>(ctx, ctx->ipa_Hf.as<u32>(), ones.data(), 2 * n));
    const u32 gb = (u32)((n + 255) / 256);
    BPCHK(ctx->ipa_sL.ensure((2 * n + 1) * 32)); BPCHK(ctx->ipa_sR.ensure((2 * n + 1) * 32)); BPCHK(ctx->ipa_part.ensure((size_t)gb * 64));
    u32* sL = ctx->ipa_sL.as<u32>();
    u32* sR = ctx->ipa_sR.as<u32>();
    Words8 zero; memset(&zero, 0, sizeof zero);
    hipLaunchKernelGGL(k_ipa_scalars<C>, dim3(gb), dim3(256), 0, st, ctx->ipa_a.as<u32>(), ctx->ipa_b.as<u32>(), ctx->ipa_Gf.as<u32>(), ctx->ipa_Hf.as<u32>(), 0, (u32)n, sL, sR,
                       ctx->ipa_part.as<u32>(), 0, zero, zero, (const u32*)nullptr);
    hipLaunchKernelGGL(k_ipa_ip_finish<C>, dim3(1), dim3(256), 0, st, ctx->ipa_part.as<u32>(), gb, sL + 2 * n * 8, sR + 2 * n * 8, zero, 0);
    HIPCHK(hipGetLastError());
    uint64_t canon[4];
    HIPCHK(hipMemcpyAsync(canon, sL + 2 * n * 8, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(ctx));
    F4 r = S::from_canon(canon);
    memcpy(out, r.v, 32);
    return BP_OK;
}

// ---- helpers of the bp_cs entry points (templates cannot live inside extern "C") ----
#define CS_DISPATCH(h, expr0, expr1) ((h)->curve == 0 ? (expr0) : (expr1))
// ---- r1cs::Prover / Verifier / ConstraintSystem for a caller's own gadgets -----------------------------------------------------
static inline bool cs_live(bp_cs* h) { return h && (!h->consumed || h->running); }
static inline void terms_in(std::vector<host::Term>& out, const bp_var* vars, const uint64_t* coefs, size_t n) {
    out.resize(n);
    for (size_t i = 0; i < n; i++) { out[i].v.k = (host::VKind)vars[i].kind; out[i].v.i = vars[i].index; memcpy(out[i].c.v, coefs + 4 * i, 32); }
}
template <class C> static bool terms_valid(bp_cs* h, const bp_var* vars, size_t n) {
    const host::ConstraintSystem<C>& cs = *h->cs<C>();
    const size_t m = cs.proving ? cs.v.size() : cs.V.size();
    for (size_t i = 0; i < n; i++) {
        if (vars[i].kind > BP_VAR_ONE) return false;
        if (vars[i].kind == BP_VAR_COMMITTED ? vars[i].index >= m : vars[i].kind != BP_VAR_ONE && vars[i].index >= cs.num_vars) return false;
    }
    return true;
}
static inline bp_var var_out(const host::Var& v) { bp_var o; o.kind = (uint32_t)v.k; o.index = v.i; return o; }
template <class C> static int verifier_new_like(bp_cs* of, host::Transcript* tr, bp_cs** out) {
    host::ConstraintSystem<C>& src = *of->cs<C>();
    BPCHK(src.freeze());
    bp_cs* h = new bp_cs();
    cs_handle_init<C>(h, false, tr);
    h->cs<C>()->init_like(src);
    *out = h;
    return BP_OK;
}

template <class C> static int prover_commit(bp_cs* h, bp_ctx* ctx, const uint64_t* v, const uint64_t* blind, size_t count, uint64_t* V_xy, bp_var* vars) {
    host::ConstraintSystem<C>& cs = *h->cs<C>();
    if (cs.phase2) return BP_E_ARG;
    host::PedersenGens<C> pc = host::PedersenGens<C>::make_default();
    if (ctx) pedersen_attach<C>(ctx, pc);
    std::vector<A4> pts(count);
    BPCHK(pc.commit_many((const F4*)v, (const F4*)blind, count, pts.data()));
    for (size_t i = 0; i < count; i++) {   // Prover::commit (prover.rs:327-341), in order
        const u32 idx = (u32)cs.v.size();
        F4 a, b; memcpy(a.v, v + 4 * i, 32); memcpy(b.v, blind + 4 * i, 32);
        cs.v.push_back(a); cs.v_blinding.push_back(b);
        host::TP<C>::append_point(*cs.tr, "V", pts[i]);
        h->commitments.push_back(pts[i]);
        if (V_xy) memcpy(V_xy + 8 * i, &pts[i], 64);
        if (vars) { vars[i].kind = BP_VAR_COMMITTED; vars[i].index = idx; }
    }
    return BP_OK;
}
template <class C> static int verifier_commit(bp_cs* h, const uint64_t* V_xy, size_t count, bp_var* vars) {
    host::ConstraintSystem<C>& cs = *h->cs<C>();
    if (cs.phase2) return BP_E_ARG;
    for (size_t i = 0; i < count; i++) {   // Verifier::commit (verifier.rs:279-287)
        A4 p; memcpy(&p, V_xy + 8 * i, 64);
        const u32 idx = (u32)cs.V.size();
        cs.V.push_back(p);
        host::TP<C>::append_point(*cs.tr, "V", p);
        h->commitments.push_back(p);
        if (vars) { vars[i].kind = BP_VAR_COMMITTED; vars[i].index = idx; }
    }
    return BP_OK;
}
// a like-instance shares its source's phase-1 constraints: recording more of them there would fork the shared structure
template <class C> static bool cs_can_record(bp_cs* h) { const auto& cs = *h->cs<C>(); return cs.phase2 || !cs.base; }

template <class C> static int cs_multiply_t(bp_cs* h, const bp_var* lv, const uint64_t* lc, size_t nl, const bp_var* rv, const uint64_t* rc, size_t nr, bp_var out[3]) {
    if (!terms_valid<C>(h, lv, nl) || !terms_valid<C>(h, rv, nr)) return BP_E_ARG;
    std::vector<host::Term> l, r; terms_in(l, lv, lc, nl); terms_in(r, rv, rc, nr);
    host::Var o[3];
    h->cs<C>()->multiply(l.data(), nl, r.data(), nr, o);
    for (int i = 0; i < 3; i++) out[i] = var_out(o[i]);
    h->csc_tried = false; h->csc.reset();
    return BP_OK;
}
template <class C> static int cs_allocate_t(bp_cs* h, const uint64_t* assignment, bp_var* out) {
    F4 a; if (assignment) memcpy(a.v, assignment, 32);
    host::Var o;
    int rc = h->cs<C>()->allocate(assignment ? &a : nullptr, o);
    if (rc) return rc;
    *out = var_out(o);
    return BP_OK;
}
template <class C> static int cs_allocate_multiplier_t(bp_cs* h, const uint64_t* l, const uint64_t* r, bp_var out[3]) {
    F4 a, b; if (l) memcpy(a.v, l, 32); if (r) memcpy(b.v, r, 32);
    host::Var o[3];
    int rc = h->cs<C>()->allocate_multiplier(l ? &a : nullptr, r ? &b : nullptr, o);
    if (rc) return rc;
    for (int i = 0; i < 3; i++) out[i] = var_out(o[i]);
    return BP_OK;
}
template <class C> static int cs_constrain_t(bp_cs* h, const bp_var* vars, const uint64_t* coefs, size_t n) {
    if (!terms_valid<C>(h, vars, n)) return BP_E_ARG;
    std::vector<host::Term> t; terms_in(t, vars, coefs, n);
    h->cs<C>()->constrain(t.data(), n);
    h->csc_tried = false; h->csc.reset();
    return BP_OK;
}
// bulk forms: one call per gadget instead of one per gate
template <class C> static int cs_allocate_multipliers_t(bp_cs* h, const uint64_t* l, const uint64_t* r, size_t count, uint32_t* first) {
    host::ConstraintSystem<C>& cs = *h->cs<C>();
    if (cs.proving && (!l || !r)) return BP_E_MISSING;
    if (first) *first = (uint32_t)cs.num_vars;
    cs.reserve(count, 0, 0);
    for (size_t i = 0; i < count; i++) {
        host::Var o[3];
        if (cs.proving) { F4 a, b; memcpy(a.v, l + 4 * i, 32); memcpy(b.v, r + 4 * i, 32); cs.allocate_multiplier(&a, &b, o); }
        else cs.allocate_multiplier(nullptr, nullptr, o);
    }
    return BP_OK;
}
template <class C> static int cs_constrain_many_t(bp_cs* h, const bp_var* vars, const uint64_t* coefs, const size_t* offsets, size_t nc) {
    host::ConstraintSystem<C>& cs = *h->cs<C>();
    const size_t nt = offsets[nc];
    if (!terms_valid<C>(h, vars, nt)) return BP_E_ARG;
    for (size_t q = 0; q < nc; q++) if (offsets[q] > offsets[q + 1]) return BP_E_ARG;
    cs.reserve(0, nc, nt);
    std::vector<host::Term> t; terms_in(t, vars, coefs, nt);
    for (size_t q = 0; q < nc; q++) cs.constrain(t.data() + offsets[q], offsets[q + 1] - offsets[q]);
    h->csc_tried = false; h->csc.reset();
    return BP_OK;
}
template <class C> static int cs_specify_t(bp_cs* h, bp_randomize_cb cb, void* user) {
    // the closure receives the handle the randomized phase runs on: the recorder passes itself, and a like-instance's copy of the
    // closure must reach ITS handle — so the handle is found through the recorder, not captured
    h->cs<C>()->specify_randomized_constraints([cb, user](host::ConstraintSystem<C>& cs) -> int { return cb(user, (bp_cs*)cs.owner); });
    return BP_OK;
}
#define CS_RECORD_GUARD(h) do { if (!cs_live(h)) return BP_E_ARG; if (!CS_DISPATCH(h, cs_can_record<Secq>(h), cs_can_record<Zorro>(h))) { g_err = "this verifier shares its phase-1 constraints (bp_verifier_new_like): only commits and randomized constraints can be added"; return BP_E_ARG; } } while (0)

// ---- C ABI ----------------------------------------------------------------------------------------
#if defined(__x86_64__)
template <class C> __attribute__((target("avx512f"))) static void dbg_append_points_x8(void* const* trs, int lanes, const char* label, const uint64_t* xy, size_t npts) {
    host::StrobeX8 sx;
    sx.broadcast(((host::Transcript*)trs[0])->s);
    uint8_t ser[8][72];
    const uint8_t* ptr[8];
    for (int l = 0; l < 8; l++) { ptr[l] = ser[l]; memset(ser[l], 0, 72); }
    for (size_t v = 0; v < npts; v++) {
        for (int l = 0; l < lanes; l++) {
            A4 p; memcpy(p.x.v, xy + ((size_t)l * npts + v) * 8, 32); memcpy(p.y.v, xy + ((size_t)l * npts + v) * 8 + 4, 32);
            host::Grp<C>::ser_uncompressed(ser[l], p);
        }
        sx.append_message_each(label, ptr, 65);
    }
    host::Strobe* outs[8];
    for (int l = 0; l < lanes; l++) outs[l] = &((host::Transcript*)trs[l])->s;
    sx.scatter(outs, lanes);
}
#endif

#if defined(__x86_64__)
__attribute__((target("avx512f"))) static int dbg_challenge_x8(void* const* trs, const char* msg_label, const uint8_t* msg, size_t msg_len, const char* chal_label, size_t nbytes,
                                                                uint8_t* out) {
    host::StrobeX8 sx;
    const host::Strobe* in[8];
    for (int l = 0; l < 8; l++) in[l] = &((host::Transcript*)trs[l])->s;
    if (!sx.gather(in)) return BP_E_ARG;
    sx.append_message_same(msg_label, msg, msg_len);
    uint8_t buf[8][64];
    uint8_t* wp[8];
    for (int l = 0; l < 8; l++) wp[l] = buf[l];
    sx.challenge_bytes_each(chal_label, wp, nbytes);
    for (int l = 0; l < 8; l++) memcpy(out + (size_t)l * nbytes, buf[l], nbytes);
    host::Strobe* outs[8];
    for (int l = 0; l < 8; l++) outs[l] = &((host::Transcript*)trs[l])->s;
    sx.scatter(outs, 8);
    return BP_OK;
}
#endif

extern "C" {

int bp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bp_ctx_create(int curve, int device, bp_ctx** out) {
    if (!out || (curve != BP_CURVE_SECQ256K1 && curve != BP_CURVE_ZORRO)) { g_err = "bp_ctx_create: bad argument"; return BP_E_ARG; }
    int n = bp_device_count();
    if (n <= 0 || device < 0 || device >= n) { g_err = "bp_ctx_create: no HIP device (the engine has no CPU fallback)"; return BP_E_NO_DEVICE; }
    HIPCHK(hipSetDevice(device));
    bp_ctx* c = new bp_ctx();
    c->curve = curve; c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; g_err = hipGetErrorString(e); return BP_E_HIP; }
    *out = c;
    return BP_OK;
}
void bp_ctx_destroy(bp_ctx* c) {
    if (!c) return;
    g_dry = c->host_only;
    c->pool.reset();             // (the pool's threads end before the buffers they may have touched go away)
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    collect_timers(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    if (c->sync_ev_aux) (void)hipEventDestroy(c->sync_ev_aux);
    DevBuf* bufs[] = {&c->canon, &c->hist, &c->lvl_off, &c->totals, &c->cursor, &c->entries, &c->slots, &c->bin_cur, &c->boff, &c->lvA, &c->lvB, &c->Tbuf, &c->io_pts, &c->io_scal, &c->io_out,
                      &c->ipa_G, &c->ipa_H, &c->ipa_a, &c->ipa_b, &c->ipa_Gf, &c->ipa_Hf, &c->ipa_sL, &c->ipa_sR, &c->ipa_part, &c->ipa_Q, &c->ipa_jac, &c->ipa_pref, &c->ipa_cG, &c->ipa_cH,
                      &c->d_G, &c->d_H, &c->d_pc, &c->pc_table, &c->r_aL, &c->r_aR, &c->r_aO, &c->r_sL, &c->r_sR, &c->r_wL, &c->r_wR, &c->r_wO, &c->r_msmsc,
                      &c->r_ypow, &c->r_part, &c->r_small, &c->r_g, &c->r_h, &c->r_chal, &c->r_tail, &c->v_params, &c->v_gpart, &c->v_hpart, &c->v_alpha, &c->v_tables, &c->v_dec, &c->cyc_a, &c->cyc_b, &c->cyc_Gf, &c->cyc_Hf, &c->ftab_G, &c->ftab_H, &c->fb_G, &c->fb_H, &c->fb_pc, &c->p_moff, &c->p_ment, &c->p_mc, &c->p_coefs, &c->p_ztab, &c->fs_bcnt, &c->fs_loff, &c->fs_binch, &c->fs_sums};
    c->templates.clear();
    c->vfe_classes.clear();
    for (auto b : bufs) b->release();
    DevBuf* vbufs[] = {&c->vfe_in, &c->vfe_msg, &c->vfe_chal, &c->vfe_ws, &c->vfe_small};
    for (auto b : vbufs) b->release();
    if (c->h_vfe) (void)hipHostFree(c->h_vfe);
    c->dt_tab.release(); c->dt_part.release(); c->pc_dt.release(); c->dt_a2.release(); c->dt_b2.release(); c->dt_ticket.release();
    if (c->h_dt) (void)hipHostFree(c->h_dt);
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    if (c->h_T) (void)hipHostFree(c->h_T);
    for (int i = 0; i < 2; i++) { if (c->h_vstage[i]) (void)hipHostFree(c->h_vstage[i]); if (c->vstage_ev[i]) (void)hipEventDestroy(c->vstage_ev[i]); if (c->vtail_ev[i]) (void)hipEventDestroy(c->vtail_ev[i]); if (c->dec_ev[i]) (void)hipEventDestroy(c->dec_ev[i]); }
    for (int i = 0; i < 2; i++) if (c->h_vaux[i]) (void)hipHostFree(c->h_vaux[i]);
    if (c->h_upload) (void)hipHostFree(c->h_upload);
    if (c->h_csc) (void)hipHostFree(c->h_csc);
    if (c->h_dec) (void)hipHostFree(c->h_dec);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->nccl) { (void)rccl_api().CommDestroy(c->nccl); c->nccl = nullptr; }
    c->coll_send.release(); c->coll_recv.release();
    if (c->h_coll) (void)hipHostFree(c->h_coll);
    (void)hipStreamDestroy(c->stream);
    delete c;
    g_dry = false;
}
int bp_debug_ctx_create_hostonly(int curve, size_t gens_capacity, bp_ctx** out) {
    if (!out || (curve != BP_CURVE_SECQ256K1 && curve != BP_CURVE_ZORRO)) return BP_E_ARG;
    bp_ctx* c = new bp_ctx();
    c->curve = curve; c->device = -1; c->host_only = true; c->gens_cap = gens_capacity;
    c->stream = (hipStream_t)&dry::g_token;
    *out = c;
    return BP_OK;
}
int bp_ctx_set_window_shard(bp_ctx* c, int rank, int world, bp_point_reduce_cb cb, void* user) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !cb)) return BP_E_ARG;
    c->shard_rank = rank; c->shard_world = world; c->shard_cb = cb; c->shard_user = user;
    return BP_OK;
}
int bp_rccl_unique_id(uint8_t out[128]) {
    if (!out) return BP_E_ARG;
    RcclApi& api = rccl_api();
    if (!api.ok) { g_err = "RCCL unavailable: " + api.why; return BP_E_HIP; }
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) { g_err = std::string("ncclGetUniqueId: ") + api.GetErrorString(r); return BP_E_HIP; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, 128);
    return BP_OK;
}
int bp_ctx_rccl_init(bp_ctx* c, const uint8_t unique_id[128], int rank, int world) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return BP_E_ARG;
    RcclApi& api = rccl_api();
    if (!api.ok) { g_err = "RCCL unavailable: " + api.why; return BP_E_HIP; }
    HIPCHK(hipSetDevice(c->device));
    if (c->nccl) { (void)api.CommDestroy(c->nccl); c->nccl = nullptr; }
    ncclUniqueId id; memcpy(&id, unique_id, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { g_err = std::string("ncclCommInitRank: ") + api.GetErrorString(r); return BP_E_HIP; }
    c->nccl = comm;
    c->shard_rank = rank; c->shard_world = world;
    c->shard_cb = nullptr; c->shard_user = nullptr; c->gather_cb = nullptr; c->gather_user = nullptr;
    c->coll_count = 0; c->coll_seconds = 0;
    return BP_OK;
}
int bp_ctx_rccl_shutdown(bp_ctx* c) {
    if (!c) return BP_E_ARG;
    if (c->nccl) { HIPCHK(hipSetDevice(c->device)); HIPCHK(ctx_stream_wait(c)); (void)rccl_api().CommDestroy(c->nccl); c->nccl = nullptr; }
    c->shard_rank = 0; c->shard_world = 1;
    return BP_OK;
}
int bp_ctx_collective_stats(bp_ctx* c, uint64_t* count, double* seconds) {
    if (!c) return BP_E_ARG;
    if (count) *count = c->coll_count;
    if (seconds) *seconds = c->coll_seconds;
    return BP_OK;
}
int bp_debug_rccl_allgather(bp_ctx* c, const uint8_t* send, size_t bytes, uint8_t* recv) {
    if (!c || !send || !recv || !bytes) return BP_E_ARG;
    if (!c->nccl) { g_err = "no RCCL communicator on this ctx (bp_ctx_rccl_init)"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return ctx_native_allgather(c, send, bytes, recv);
}
int bp_ctx_set_shard_allgather(bp_ctx* c, bp_allgather_cb cb, void* user) {
    if (!c) return BP_E_ARG;
    c->gather_cb = cb; c->gather_user = user;
    return BP_OK;
}
int bp_ctx_set_tuning(bp_ctx* c, int knob, uint64_t value) {
    if (!c) return BP_E_ARG;
    switch (knob) {
        case BP_TUNE_CYCLIC_MIN: c->tune_cyclic_min = (size_t)value; return BP_OK;
        case BP_TUNE_MSM_FIXED_MIN: c->tune_msm_fixed_min = (size_t)value; return BP_OK;
        case BP_TUNE_HOST_THREADS: if (value > 256) return BP_E_ARG; c->tune_host_threads = (size_t)value; c->pool.reset(); return BP_OK;
        case BP_TUNE_FOLD_BATCH_MIN: c->tune_fold_batch_min = (size_t)value; return BP_OK;
        case BP_TUNE_MSM_BIN_MIN: c->tune_msm_bin_min = (size_t)value; return BP_OK;
        case BP_TUNE_IPA_FREEZE_LEN: c->tune_ipa_freeze_len = (size_t)value; return BP_OK;
        case BP_TUNE_MSM_WSUM_MIN: c->tune_msm_wsum_min = (size_t)value; return BP_OK;
        case BP_TUNE_MSM_GLV_MIN: c->tune_msm_glv_min = (size_t)value; return BP_OK;
        case BP_TUNE_FOLD_QUAD_MAX: c->tune_fold_quad_max = (size_t)value; return BP_OK;
        case BP_TUNE_WAIT_SLEEP: if (value > 1000) return BP_E_ARG; c->tune_wait_sleep = value == 1 ? 30u : (unsigned)value; return BP_OK;
        case BP_TUNE_MSM_CHUNK_CAP: if (value && (value < 8 || value > 64)) return BP_E_ARG; c->tune_msm_chunk_cap = (size_t)value; return BP_OK;
        case BP_TUNE_VFY_DEVICE: c->tune_vfy_device = value != 0; return BP_OK;
        case BP_TUNE_DIRECT_MAX: if (value > ((uint64_t)1 << 16)) return BP_E_ARG; c->tune_direct_max = (size_t)value; return BP_OK;
    }
    return BP_E_ARG;
}
int bp_ctx_sync(bp_ctx* c) {
    if (!c) return BP_E_ARG;
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}

int bp_dev_alloc(bp_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 1));
    return BP_OK;
}
int bp_dev_free(bp_ctx* c, void* dptr) {
    if (!c) return BP_E_ARG;
    HIPCHK(ctx_stream_wait(c));
    HIPCHK(hipFree(dptr));
    return BP_OK;
}
int bp_dev_upload(bp_ctx* c, void* dptr, const void* hostp, size_t bytes) {
    if (!c || (!dptr && bytes) || (!hostp && bytes)) return BP_E_ARG;
    HIPCHK(hipMemcpyAsync(dptr, hostp, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}
int bp_dev_download(bp_ctx* c, void* hostp, const void* dptr, size_t bytes) {
    if (!c || (!dptr && bytes) || (!hostp && bytes)) return BP_E_ARG;
    HIPCHK(hipMemcpyAsync(hostp, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}
int bp_points_import(bp_ctx* c, const void* d_in, void* d_out, size_t n) {
    if (!c || ((!d_in || !d_out) && n)) return BP_E_ARG;
    if (!n) return BP_OK;
    const u32 gb = (u32)((n + 255) / 256);
    if (c->curve == 0) hipLaunchKernelGGL(k_points_ark_to_dev<Secq>, dim3(gb), dim3(256), 0, c->stream, (const u32*)d_in, (u32*)d_out, (u32)n);
    else hipLaunchKernelGGL(k_points_ark_to_dev<Zorro>, dim3(gb), dim3(256), 0, c->stream, (const u32*)d_in, (u32*)d_out, (u32)n);
    HIPCHK(hipGetLastError());
    return BP_OK;
}
int bp_points_export(bp_ctx* c, const void* d_in, void* d_out, size_t n) {
    if (!c || ((!d_in || !d_out) && n)) return BP_E_ARG;
    if (!n) return BP_OK;
    const u32 gb = (u32)((n + 255) / 256);
    if (c->curve == 0) hipLaunchKernelGGL(k_points_dev_to_ark<Secq>, dim3(gb), dim3(256), 0, c->stream, (const u32*)d_in, (u32*)d_out, (u32)n);
    else hipLaunchKernelGGL(k_points_dev_to_ark<Zorro>, dim3(gb), dim3(256), 0, c->stream, (const u32*)d_in, (u32*)d_out, (u32)n);
    HIPCHK(hipGetLastError());
    return BP_OK;
}

int bp_msm_dev(bp_ctx* c, const void* d_bases, const void* d_scalars, size_t n, int canonical, uint64_t out_xy[8]) {
    if (!c || !out_xy || ((!d_bases || !d_scalars) && n)) { g_err = "bp_msm_dev: bad argument"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? msm_dev_entry<Secq>(c, d_bases, d_scalars, n, canonical, out_xy) : msm_dev_entry<Zorro>(c, d_bases, d_scalars, n, canonical, out_xy);
}
int bp_msm_gens(bp_ctx* c, int use_G, int use_H, size_t off, size_t n, const uint64_t* extra_bases_xy, size_t n_extra, const uint64_t* scalars,
                int canonical, uint64_t out_xy[8]) {
    if (!c || !out_xy || (n_extra && !extra_bases_xy)) { g_err = "bp_msm_gens: bad argument"; return BP_E_ARG; }
    const size_t total = (use_G ? n : 0) + (use_H ? n : 0) + n_extra;
    if (total && !scalars) { g_err = "bp_msm_gens: bad argument"; return BP_E_ARG; }
    if ((use_G || use_H) && n && off + n > c->gens_cap) { g_err = "bp_msm_gens: range exceeds the installed generator tables"; return BP_E_GENS_LENGTH; }
    if (total >= ((size_t)1 << 31)) { g_err = "bp_msm_gens: too many terms"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? msm_gens_entry<Secq>(c, use_G, use_H, off, n, extra_bases_xy, n_extra, scalars, canonical, out_xy)
                         : msm_gens_entry<Zorro>(c, use_G, use_H, off, n, extra_bases_xy, n_extra, scalars, canonical, out_xy);
}
int bp_msm_window_count(int curve, size_t n, int* windows, int* window_bits) {
    if ((curve != 0 && curve != 1) || !windows) return BP_E_ARG;
    MsmPlan pl = msm_plan(n ? n : 1, curve == 0 ? Secq::Fr::BITS : Zorro::Fr::BITS);
    *windows = pl.W;
    if (window_bits) *window_bits = pl.c;
    return BP_OK;
}
int bp_msm_dev_windows(bp_ctx* c, const void* d_bases, const void* d_scalars, size_t n, int canonical, int w_lo, int w_hi, uint64_t out_xy[8]) {
    if (!c || !out_xy || ((!d_bases || !d_scalars) && n) || w_lo < 0 || w_hi < w_lo) { g_err = "bp_msm_dev_windows: bad argument"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? msm_dev_entry<Secq>(c, d_bases, d_scalars, n, canonical, out_xy, w_lo, w_hi)
                         : msm_dev_entry<Zorro>(c, d_bases, d_scalars, n, canonical, out_xy, w_lo, w_hi);
}
int bp_msm(bp_ctx* c, const uint64_t* bases_xy, const uint64_t* scalars, size_t n, int canonical, uint64_t out_xy[8]) {
    if (!c || !out_xy || ((!bases_xy || !scalars) && n)) { g_err = "bp_msm: bad argument"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) { memset(out_xy, 0, 64); return BP_OK; }
    BPCHK(c->io_pts.ensure(n * 64));
    BPCHK(c->io_scal.ensure(n * 32));
    HIPCHK(hipMemcpyAsync(c->io_pts.p, bases_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->io_scal.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    BPCHK(bp_points_import(c, c->io_pts.p, c->io_pts.p, n));
    return bp_msm_dev(c, c->io_pts.p, c->io_scal.p, n, canonical, out_xy);
}


int bp_ipa_create(bp_ctx* c, const uint64_t Q_xy[8], const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t* G_xy, const uint64_t* H_xy,
                  const uint64_t* a, const uint64_t* b, size_t n, bp_challenge_cb cb, void* user, uint64_t* L_out_xy, uint64_t* R_out_xy, uint64_t a_out[4],
                  uint64_t b_out[4]) {
    if (!c || !Q_xy || !G_factors || !H_factors || !G_xy || !H_xy || !a || !b || !cb || !a_out || !b_out || (n > 1 && (!L_out_xy || !R_out_xy))) {
        g_err = "bp_ipa_create: bad argument"; return BP_E_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    ChallengeFn fn = [cb, user](const uint64_t* L, const uint64_t* R, uint64_t* u) { return cb(user, L, R, u); };
    return c->curve == 0 ? ipa_create_host_entry<Secq>(c, Q_xy, G_factors, H_factors, G_xy, H_xy, a, b, n, fn, L_out_xy, R_out_xy, a_out, b_out)
                         : ipa_create_host_entry<Zorro>(c, Q_xy, G_factors, H_factors, G_xy, H_xy, a, b, n, fn, L_out_xy, R_out_xy, a_out, b_out);
}

int bp_ipa_begin(bp_ctx* c, const uint64_t Q_xy[8], const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t* G_xy, const uint64_t* H_xy,
                 const uint64_t* a, const uint64_t* b, size_t n) {
    if (!c || !Q_xy || !G_factors || !H_factors || !G_xy || !H_xy || !a || !b) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? ipa_begin_host_entry<Secq>(c, Q_xy, G_factors, H_factors, G_xy, H_xy, a, b, n)
                         : ipa_begin_host_entry<Zorro>(c, Q_xy, G_factors, H_factors, G_xy, H_xy, a, b, n);
}
int bp_ipa_round_LR(bp_ctx* c, uint64_t L_xy[8], uint64_t R_xy[8]) {
    if (!c || !L_xy || !R_xy) return BP_E_ARG;
    if (!c->ipa_step_active) { g_err = "bp_ipa_round_LR: no bp_ipa_begin"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? ipa_round_lr<Secq>(c, c->ipa_step, L_xy, R_xy) : ipa_round_lr<Zorro>(c, c->ipa_step, L_xy, R_xy);
}
int bp_ipa_round_fold(bp_ctx* c, const uint64_t u[4]) {
    if (!c || !u) return BP_E_ARG;
    if (!c->ipa_step_active) { g_err = "bp_ipa_round_fold: no bp_ipa_begin"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? ipa_round_fold<Secq>(c, c->ipa_step, u) : ipa_round_fold<Zorro>(c, c->ipa_step, u);
}
int bp_ipa_finish(bp_ctx* c, uint64_t a[4], uint64_t b[4]) {
    if (!c || !a || !b) return BP_E_ARG;
    if (!c->ipa_step_active) { g_err = "bp_ipa_finish: no bp_ipa_begin"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    int rc = c->curve == 0 ? ipa_finish_dev<Secq>(c, c->ipa_step, a, b) : ipa_finish_dev<Zorro>(c, c->ipa_step, a, b);
    if (rc == BP_OK) c->ipa_step_active = false;
    return rc;
}
int bp_ipa_export(bp_ctx* c, uint64_t* a, uint64_t* b, uint64_t* G_xy, uint64_t* H_xy, uint64_t gamma_G[4], uint64_t gamma_H[4], size_t* n_cur) {
    if (!c || !a || !b || !G_xy || !H_xy || !gamma_G || !gamma_H || !n_cur) return BP_E_ARG;
    if (!c->ipa_step_active) { g_err = "bp_ipa_export: no bp_ipa_begin"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? ipa_export_host<Secq>(c, a, b, G_xy, H_xy, gamma_G, gamma_H, n_cur) : ipa_export_host<Zorro>(c, a, b, G_xy, H_xy, gamma_G, gamma_H, n_cur);
}

// ---- generators -------------------------------------------------------------------------------------
int bp_gens_derive(bp_ctx* c, size_t cap) {
    if (!c || !cap) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? gens_derive<Secq>(c, cap) : gens_derive<Zorro>(c, cap);
}
int bp_gens_upload(bp_ctx* c, const uint64_t* G_xy, const uint64_t* H_xy, size_t cap) {
    if (!c || !G_xy || !H_xy || !cap) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? gens_install<Secq>(c, G_xy, H_xy, cap) : gens_install<Zorro>(c, G_xy, H_xy, cap);
}
int bp_gens_msm_tables(bp_ctx* c, size_t count, size_t* bytes_out) {
    if (!c) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (count == 0) { if (c->fb_G.owned) { c->fb_G.release(); c->fb_H.release(); c->fb_pc.release(); } c->fb_cap = 0; return BP_OK; }
    const int rc = c->curve == 0 ? fb_tables_build<Secq>(c, count) : fb_tables_build<Zorro>(c, count);
    if (rc) return rc;
    if (bytes_out) *bytes_out = 2 * (size_t)FB_ROWS * count * 64;
    return BP_OK;
}
int bp_gens_direct_tables(bp_ctx* c, size_t count, size_t* bytes_out) {
    if (!c) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (bytes_out) *bytes_out = 0;
    if (count == 0) { if (c->dt_tab.owned) c->dt_tab.release(); else { c->dt_tab.p = nullptr; c->dt_tab.cap = 0; c->dt_tab.owned = true; } c->dt_cap = 0; return BP_OK; }
    if (!c->gens_cap) { g_err = "direct tables: no generators installed"; return BP_E_GENS_LENGTH; }
    if (c->shard_world > 1 || !c->tune_direct_max) { g_err = "direct tables: the small-statement path is off on this ctx (sharded, or BP_TUNE_DIRECT_MAX = 0)"; return BP_E_ARG; }
    const size_t want = std::min(std::min(count, c->gens_cap), c->tune_direct_max);
    // dt_ensure builds for min(generators, BP_TUNE_DIRECT_MAX): narrow the knob for the build so that `count` is what is built
    const size_t keep = c->tune_direct_max;
    c->tune_direct_max = want;
    if (c->dt_cap != want) c->dt_cap = 0;
    bool ready = false;
    const int rc = c->curve == 0 ? dt_ensure<Secq>(c, want, ready) : dt_ensure<Zorro>(c, want, ready);
    c->tune_direct_max = keep;
    if (rc) return rc;
    if (bytes_out) *bytes_out = (2 + 2 * c->dt_cap) * DT_BASE_BYTES;
    return BP_OK;
}
int bp_gens_fold_tables(bp_ctx* c, size_t count, int window_bits, size_t budget_bytes, int* window_bits_out, size_t* bytes_out) {
    if (!c || window_bits < 0 || window_bits == 1 || window_bits > 8) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (count == 0) { if (c->ftab_G.owned) { c->ftab_G.release(); c->ftab_H.release(); } c->ftab_n = 0; c->ftab_first = 0; c->ftab_stride = 1; return BP_OK; }
    const int rc = c->curve == 0 ? ftab_build<Secq>(c, count, window_bits, budget_bytes) : ftab_build<Zorro>(c, count, window_bits, budget_bytes);
    if (rc) return rc;
    if (window_bits_out) *window_bits_out = c->ftab_w;
    if (bytes_out) *bytes_out = 2 * (size_t)c->ftab_nwin * ((size_t)1 << (c->ftab_w - 1)) * c->ftab_n * 64;
    return BP_OK;
}
int bp_gens_fold_tables_slice(bp_ctx* c, size_t count, int window_bits, size_t budget_bytes, int rank, int world, int* window_bits_out, size_t* bytes_out) {
    if (!c || window_bits < 0 || window_bits == 1 || window_bits > 8 || world < 1 || rank < 0 || rank >= world || count == 0 || count % (size_t)world) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    const size_t cols = count / (size_t)world;
    const int rc = c->curve == 0 ? ftab_build<Secq>(c, cols, window_bits, budget_bytes, (u32)rank, (u32)world) : ftab_build<Zorro>(c, cols, window_bits, budget_bytes, (u32)rank, (u32)world);
    if (rc) return rc;
    if (window_bits_out) *window_bits_out = c->ftab_w;
    if (bytes_out) *bytes_out = 2 * (size_t)c->ftab_nwin * ((size_t)1 << (c->ftab_w - 1)) * c->ftab_n * 64;
    return BP_OK;
}
int bp_gens_tables_check(bp_ctx* c, uint64_t* bad_fold_entries, uint64_t* bad_msm_rows) {
    if (!c) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (!c->gens_cap) { g_err = "tables_check: generators not installed"; return BP_E_GENS_LENGTH; }
    return c->curve == 0 ? tables_check<Secq>(c, bad_fold_entries, bad_msm_rows) : tables_check<Zorro>(c, bad_fold_entries, bad_msm_rows);
}
int bp_debug_tables_ptr(bp_ctx* c, int which, void** dptr, size_t* nbytes) {
    if (!c || !dptr || !nbytes || which < 0 || which > 3) return BP_E_ARG;
    DevBuf& b = which == 0 ? c->ftab_G : which == 1 ? c->ftab_H : which == 2 ? c->fb_G : c->fb_H;
    const size_t used = which < 2 ? (c->ftab_n ? (size_t)c->ftab_nwin * ((size_t)1 << (c->ftab_w - 1)) * c->ftab_n * 64 : 0) : (size_t)FB_ROWS * c->fb_cap * 64;
    *dptr = used ? b.p : nullptr; *nbytes = used;
    return BP_OK;
}
int bp_gens_download(bp_ctx* c, uint64_t* G_xy, uint64_t* H_xy, size_t n) {
    if (!c || !G_xy || !H_xy || n > c->gens_cap) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    BPCHK(c->io_pts.ensure(n * 64));
    for (int k = 0; k < 2; k++) {
        BPCHK(bp_points_export(c, k ? c->d_H.p : c->d_G.p, c->io_pts.p, n));
        HIPCHK(hipMemcpyAsync(k ? H_xy : G_xy, c->io_pts.p, n * 64, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(ctx_stream_wait(c));
    }
    return BP_OK;
}
int bp_pedersen_gens(int curve, uint64_t B_xy[8], uint64_t B_blinding_xy[8]) {
    if (!B_xy || !B_blinding_xy) return BP_E_ARG;
    A4 B, Bb;
    if (curve == 0) { auto g = host::PedersenGens<Secq>::make_default(); B = g.B; Bb = g.B_blinding; }
    else if (curve == 1) { auto g = host::PedersenGens<Zorro>::make_default(); B = g.B; Bb = g.B_blinding; }
    else return BP_E_ARG;
    memcpy(B_xy, B.x.v, 32); memcpy(B_xy + 4, B.y.v, 32); memcpy(B_blinding_xy, Bb.x.v, 32); memcpy(B_blinding_xy + 4, Bb.y.v, 32);
    return BP_OK;
}
int bp_host_derive_generators(int curve, int which_H, uint32_t party, size_t count, uint64_t* out_xy) {
    if (!out_xy) return BP_E_ARG;
    std::vector<A4> v;
    if (curve == 0) host::derive_generators<Secq>(v, which_H ? 'H' : 'G', party, count);
    else if (curve == 1) host::derive_generators<Zorro>(v, which_H ? 'H' : 'G', party, count);
    else return BP_E_ARG;
    if (count) memcpy(out_xy, v.data(), count * 64);
    return BP_OK;
}

// ---- host transcript (the product's own merlin restatement), exposed for CPU-side tests and for shims ----
void* bp_transcript_new(const uint8_t* label, size_t n) { return new host::Transcript(label, n); }
void bp_transcript_free(void* t) { delete (host::Transcript*)t; }
void bp_transcript_append_message(void* t, const char* label, const uint8_t* m, size_t n) { ((host::Transcript*)t)->append_message(label, m, n); }
void bp_transcript_challenge_bytes(void* t, const char* label, uint8_t* out, size_t n) { ((host::Transcript*)t)->challenge_bytes(label, out, n); }
int bp_transcript_append_point(int curve, void* t, const char* label, const uint64_t xy[8]) {
    A4 p; memcpy(p.x.v, xy, 32); memcpy(p.y.v, xy + 4, 32);
    if (curve == 0) host::TP<Secq>::append_point(*(host::Transcript*)t, label, p);
    else if (curve == 1) host::TP<Zorro>::append_point(*(host::Transcript*)t, label, p);
    else return BP_E_ARG;
    return BP_OK;
}
int bp_debug_append_points_x8(int curve, void* const* transcripts, int lanes, const char* label, const uint64_t* points_xy, size_t npts) {
    if (!transcripts || lanes < 2 || lanes > 8 || !label || (npts && !points_xy) || (curve != 0 && curve != 1)) return BP_E_ARG;
#if defined(__x86_64__)
    if (!host::cpu_has_avx512()) return BP_E_ARG;
    if (curve == 0) dbg_append_points_x8<Secq>(transcripts, lanes, label, points_xy, npts); else dbg_append_points_x8<Zorro>(transcripts, lanes, label, points_xy, npts);
    return BP_OK;
#else
    return BP_E_ARG;
#endif
}
// test hook of the rest of the lockstep replay (host::StrobeX8::gather / append_message_same / challenge_bytes_each / scatter): the
// eight transcripts (any states at the same STROBE position) take `msg` under `msg_label` and then give `nbytes` (<= 64) challenge
// bytes each under `chal_label`: out = [8][nbytes].  Must equal bp_transcript_append_message + challenge bytes lane by lane.
int bp_debug_challenge_x8(void* const* transcripts, const char* msg_label, const uint8_t* msg, size_t msg_len, const char* chal_label, size_t nbytes, uint8_t* out) {
    if (!transcripts || !msg_label || !chal_label || !out || nbytes == 0 || nbytes > 64 || (msg_len && !msg)) return BP_E_ARG;
#if defined(__x86_64__)
    if (!host::cpu_has_avx512()) return BP_E_ARG;
    return dbg_challenge_x8(transcripts, msg_label, msg, msg_len, chal_label, nbytes, out);
#else
    return BP_E_ARG;
#endif
}
int bp_transcript_challenge_scalar(int curve, void* t, const char* label, uint64_t out[4]) {
    F4 r;
    if (curve == 0) r = host::TP<Secq>::challenge_scalar(*(host::Transcript*)t, label);
    else if (curve == 1) r = host::TP<Zorro>::challenge_scalar(*(host::Transcript*)t, label);
    else return BP_E_ARG;
    memcpy(out, r.v, 32);
    return BP_OK;
}
int bp_host_sha3_512(const uint8_t* m, size_t n, uint8_t out[64]) { host::sha3_512(out, m, n); return BP_OK; }
// test hook (host only): the prover's TranscriptRng (build_rng, rekey "v_blinding" per witness scalar, finalize with
// ChaCha20(seed)), then `count` Fr::rand draws.  lanes = 1: scalar path; lanes = 8: eight rngs (seeds[8][32]) advanced by the
// AVX-512 x8 stream after 3 scalar draws each, out = 8 x count scalars.  Returns BP_E_ARG when lanes = 8 is unavailable.
int bp_debug_rng_draws(int curve, void* transcript, const uint64_t* witness, size_t nw, const uint8_t* seeds, int lanes, size_t count, uint64_t* out) {
    if (!transcript || !seeds || !out || (nw && !witness)) return BP_E_ARG;
    return curve == 0 ? dbg_rng_draws<Secq>(transcript, witness, nw, seeds, lanes, count, out)
         : curve == 1 ? dbg_rng_draws<Zorro>(transcript, witness, nw, seeds, lanes, count, out) : BP_E_ARG;
}
// sum of `count` affine points on the host: the "point-reduce" after an all-gather of per-GPU partial results
int bp_host_points_sum(int curve, const uint64_t* pts_xy, size_t count, uint64_t out_xy[8]) {
    if ((!pts_xy && count) || !out_xy) return BP_E_ARG;
    A4 r;
    if (curve == 0) { J4 acc = host::Grp<Secq>::inf(); for (size_t i = 0; i < count; i++) { A4 p; memcpy(p.x.v, pts_xy + 8 * i, 32); memcpy(p.y.v, pts_xy + 8 * i + 4, 32); acc = host::Grp<Secq>::madd(acc, p); } r = host::Grp<Secq>::to_aff(acc); }
    else if (curve == 1) { J4 acc = host::Grp<Zorro>::inf(); for (size_t i = 0; i < count; i++) { A4 p; memcpy(p.x.v, pts_xy + 8 * i, 32); memcpy(p.y.v, pts_xy + 8 * i + 4, 32); acc = host::Grp<Zorro>::madd(acc, p); } r = host::Grp<Zorro>::to_aff(acc); }
    else return BP_E_ARG;
    memcpy(out_xy, r.x.v, 32); memcpy(out_xy + 4, r.y.v, 32);
    return BP_OK;
}

// ---- R1CS prove (scenario-level driver) ---------------------------------------------------------------
int bp_r1cs_prove_scenario(bp_ctx* c, int scenario, const uint64_t* params, const uint8_t seed[32], uint8_t* proof_out, size_t* proof_len,
                           uint64_t* commit_xy, size_t m_cap, size_t* m_out, uint64_t* publics, size_t* npub, double* timing) {
    if (!c || !params || !seed || !proof_out || !proof_len || !commit_xy || !m_out || !publics || !npub) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    std::vector<host::u8> bytes;
    host::StatementIO io;
    int rc = c->curve == 0 ? prove_scenario<Secq>(c, scenario, params, seed, bytes, io, timing) : prove_scenario<Zorro>(c, scenario, params, seed, bytes, io, timing);
    if (rc) return rc;
    if (bytes.size() > *proof_len || io.commitments.size() > m_cap || io.publics.size() > 8) { g_err = "prove: output buffer too small"; return BP_E_ARG; }
    memcpy(proof_out, bytes.data(), bytes.size()); *proof_len = bytes.size();
    if (!io.commitments.empty()) memcpy(commit_xy, io.commitments.data(), io.commitments.size() * 64);   // (memcpy's pointers must be valid even for 0 bytes: UBSan)
    *m_out = io.commitments.size();
    if (!io.publics.empty()) memcpy(publics, io.publics.data(), io.publics.size() * 32);
    *npub = io.publics.size();
    return BP_OK;
}

// ---- R1CS verify / batch verify (scenario-level drivers) ------------------------------------------------
int bp_r1cs_verify_scenario(bp_ctx* c, int scenario, const uint64_t* params, const uint8_t* proof, size_t proof_len, const uint64_t* commit_xy, size_t m,
                            const uint64_t* publics, size_t npub) {
    if (!c || !params || !proof || (m && !commit_xy) || (npub && !publics)) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (!c->gens_cap) { g_err = "verify: generators not installed"; return BP_E_GENS_LENGTH; }
    return c->curve == 0 ? verify_scenario<Secq>(c, scenario, params, proof, proof_len, commit_xy, m, publics, npub)
                         : verify_scenario<Zorro>(c, scenario, params, proof, proof_len, commit_xy, m, publics, npub);
}
int bp_r1cs_batch_verify_scenarios(bp_ctx* c, size_t count, const int* scenarios, const uint64_t* params, const uint8_t* proofs, const size_t* proof_lens,
                                   const uint64_t* commit_xy, const size_t* ms, const uint64_t* publics, const size_t* npubs, const uint8_t alpha_seed[32],
                                   double* timing, size_t alpha_skip, uint64_t* check_point_xy) {
    if (!c) return BP_E_ARG;
    if (count == 0) { if (check_point_xy) memset(check_point_xy, 0, 64); return BP_OK; }   // the reference's mega-check of nothing is the identity (verifier.rs:685-690)
    if (!scenarios || !params || !proofs || !proof_lens || !commit_xy || !ms || !npubs || !alpha_seed) return BP_E_ARG;
    g_dry = c->host_only;
    HIPCHK(hipSetDevice(c->device));
    if (!c->gens_cap) { g_err = "batch_verify: generators not installed"; g_dry = false; return BP_E_GENS_LENGTH; }
    const int rc = c->curve == 0 ? batch_verify_scenarios<Secq>(c, count, scenarios, params, proofs, proof_lens, commit_xy, ms, publics, npubs, alpha_seed, timing, alpha_skip, check_point_xy)
                                 : batch_verify_scenarios<Zorro>(c, count, scenarios, params, proofs, proof_lens, commit_xy, ms, publics, npubs, alpha_seed, timing, alpha_skip, check_point_xy);
    g_dry = false;
    return rc;
}

// ---- statements: Prover::new + commits + gadget, separated from prove() ----------------------------------
int bp_stmt_prover_create(int curve, int scenario, const uint64_t* params, const uint8_t seed[32], bp_stmt** out) {
    if (!params || !seed || !out || (curve != 0 && curve != 1)) return BP_E_ARG;
    bp_stmt* s = new bp_stmt();
    s->scenario = scenario;
    int rc = curve == 0 ? stmt_build<Secq>(s, params, seed) : stmt_build<Zorro>(s, params, seed);
    if (rc) { delete s; return rc; }
    *out = s;
    return BP_OK;
}
// as bp_stmt_prover_create, with the statement's Pedersen commitments computed on ctx's GPU in one batch
int bp_stmt_prover_create_dev(bp_ctx* c, int scenario, const uint64_t* params, const uint8_t seed[32], bp_stmt** out) {
    if (!c || !params || !seed || !out) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    bp_stmt* s = new bp_stmt();
    s->scenario = scenario;
    int rc = c->curve == 0 ? stmt_build<Secq>(s, params, seed, c) : stmt_build<Zorro>(s, params, seed, c);
    if (rc) { delete s; return rc; }
    *out = s;
    return BP_OK;
}
int bp_pedersen_commit_batch(bp_ctx* c, const uint64_t* v, const uint64_t* blind, size_t m, uint64_t* out_xy) {
    if (!c || (m && (!v || !blind || !out_xy))) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? pedersen_commit_batch<Secq>(c, v, blind, m, out_xy) : pedersen_commit_batch<Zorro>(c, v, blind, m, out_xy);
}
void bp_stmt_free(bp_stmt* s) { delete s; }
int bp_stmt_info(bp_stmt* s, uint64_t* commit_xy, size_t m_cap, size_t* m_out, uint64_t* publics, size_t* npub, size_t* multipliers, size_t* constraints) {
    if (!s || !m_out || !npub) return BP_E_ARG;
    if (s->io.commitments.size() > m_cap || s->io.publics.size() > 8) return BP_E_ARG;
    if (commit_xy && !s->io.commitments.empty()) memcpy(commit_xy, s->io.commitments.data(), s->io.commitments.size() * 64);   // (an empty vector's data() may be null: UB for memcpy even with 0 bytes — found by the UBSan build, tests/sanitize)
    if (publics && !s->io.publics.empty()) memcpy(publics, s->io.publics.data(), s->io.publics.size() * 32);
    *m_out = s->io.commitments.size(); *npub = s->io.publics.size();
    if (multipliers) *multipliers = s->num_vars();
    if (constraints) *constraints = s->num_constraints();
    return BP_OK;
}
bp_cs* bp_stmt_as_prover(bp_stmt* s) { return s; }
int bp_stmt_prove(bp_ctx* c, bp_stmt* s, uint8_t* proof_out, size_t* proof_len, double* timing) { return bp_prover_prove(c, s, nullptr, proof_out, proof_len, timing); }
int bp_stmt_precompute(bp_stmt* s) { return bp_prover_precompute(s, nullptr); }
int bp_stmt_precompute_batch(bp_stmt** stmts, size_t count) { return bp_prover_precompute_batch((bp_cs**)stmts, count); }


int bp_prover_new(int curve, void* transcript, bp_cs** out) {
    if (!transcript || !out || (curve != 0 && curve != 1)) return BP_E_ARG;
    bp_cs* h = new bp_cs();
    if (curve == 0) cs_handle_init<Secq>(h, true, (host::Transcript*)transcript); else cs_handle_init<Zorro>(h, true, (host::Transcript*)transcript);
    *out = h;
    return BP_OK;
}
int bp_verifier_new(int curve, void* transcript, bp_cs** out) {
    if (!transcript || !out || (curve != 0 && curve != 1)) return BP_E_ARG;
    bp_cs* h = new bp_cs();
    if (curve == 0) cs_handle_init<Secq>(h, false, (host::Transcript*)transcript); else cs_handle_init<Zorro>(h, false, (host::Transcript*)transcript);
    *out = h;
    return BP_OK;
}
int bp_verifier_new_like(bp_cs* of, void* transcript, bp_cs** out) {
    if (!cs_live(of) || of->proving || !transcript || !out) { g_err = "bp_verifier_new_like: needs a live verifier handle"; return BP_E_ARG; }
    return CS_DISPATCH(of, verifier_new_like<Secq>(of, (host::Transcript*)transcript, out), verifier_new_like<Zorro>(of, (host::Transcript*)transcript, out));
}
void bp_cs_free(bp_cs* h) { delete h; }
void* bp_cs_transcript(bp_cs* h) { return h ? h->tr : nullptr; }
int bp_cs_metrics(bp_cs* h, size_t* multipliers, size_t* constraints, size_t* commitments) {
    if (!h) return BP_E_ARG;
    if (multipliers) *multipliers = h->num_vars();
    if (constraints) *constraints = h->num_constraints();
    if (commitments) *commitments = h->commitments.size();
    return BP_OK;
}
int bp_prover_commit(bp_cs* h, bp_ctx* ctx, const uint64_t* v, const uint64_t* v_blinding, size_t count, uint64_t* V_xy_out, bp_var* vars_out) {
    if (!cs_live(h) || !h->proving || (count && (!v || !v_blinding))) return BP_E_ARG;
    if (ctx) { if (ctx->curve != h->curve) return BP_E_ARG; HIPCHK(hipSetDevice(ctx->device)); }
    return CS_DISPATCH(h, prover_commit<Secq>(h, ctx, v, v_blinding, count, V_xy_out, vars_out), prover_commit<Zorro>(h, ctx, v, v_blinding, count, V_xy_out, vars_out));
}
int bp_verifier_commit(bp_cs* h, const uint64_t* V_xy, size_t count, bp_var* vars_out) {
    if (!cs_live(h) || h->proving || (count && !V_xy)) return BP_E_ARG;
    return CS_DISPATCH(h, verifier_commit<Secq>(h, V_xy, count, vars_out), verifier_commit<Zorro>(h, V_xy, count, vars_out));
}
int bp_cs_multiply(bp_cs* h, const bp_var* lv, const uint64_t* lc, size_t nl, const bp_var* rv, const uint64_t* rc, size_t nr, bp_var out[3]) {
    CS_RECORD_GUARD(h);
    if (!out || (nl && (!lv || !lc)) || (nr && (!rv || !rc))) return BP_E_ARG;
    return CS_DISPATCH(h, cs_multiply_t<Secq>(h, lv, lc, nl, rv, rc, nr, out), cs_multiply_t<Zorro>(h, lv, lc, nl, rv, rc, nr, out));
}
int bp_cs_allocate(bp_cs* h, const uint64_t* assignment, bp_var* out) {
    CS_RECORD_GUARD(h);
    if (!out) return BP_E_ARG;
    return CS_DISPATCH(h, cs_allocate_t<Secq>(h, assignment, out), cs_allocate_t<Zorro>(h, assignment, out));
}
int bp_cs_allocate_multiplier(bp_cs* h, const uint64_t* left, const uint64_t* right, bp_var out[3]) {
    CS_RECORD_GUARD(h);
    if (!out) return BP_E_ARG;
    return CS_DISPATCH(h, cs_allocate_multiplier_t<Secq>(h, left, right, out), cs_allocate_multiplier_t<Zorro>(h, left, right, out));
}
int bp_cs_constrain(bp_cs* h, const bp_var* vars, const uint64_t* coefs, size_t n) {
    CS_RECORD_GUARD(h);
    if (n && (!vars || !coefs)) return BP_E_ARG;
    return CS_DISPATCH(h, cs_constrain_t<Secq>(h, vars, coefs, n), cs_constrain_t<Zorro>(h, vars, coefs, n));
}
int bp_cs_allocate_multipliers(bp_cs* h, const uint64_t* left, const uint64_t* right, size_t count, uint32_t* first_index) {
    CS_RECORD_GUARD(h);
    return CS_DISPATCH(h, cs_allocate_multipliers_t<Secq>(h, left, right, count, first_index), cs_allocate_multipliers_t<Zorro>(h, left, right, count, first_index));
}
int bp_cs_constrain_many(bp_cs* h, const bp_var* vars, const uint64_t* coefs, const size_t* offsets, size_t nconstraints) {
    CS_RECORD_GUARD(h);
    if (!offsets || (offsets[nconstraints] && (!vars || !coefs))) return BP_E_ARG;
    return CS_DISPATCH(h, cs_constrain_many_t<Secq>(h, vars, coefs, offsets, nconstraints), cs_constrain_many_t<Zorro>(h, vars, coefs, offsets, nconstraints));
}
int bp_cs_specify_randomized_constraints(bp_cs* h, bp_randomize_cb cb, void* user) {
    if (!cs_live(h) || !cb) return BP_E_ARG;
    if (CS_DISPATCH(h, h->cs0->phase2, h->cs1->phase2)) { g_err = "specify_randomized_constraints inside the randomized phase"; return BP_E_ARG; }
    CS_RECORD_GUARD(h);
    return CS_DISPATCH(h, cs_specify_t<Secq>(h, cb, user), cs_specify_t<Zorro>(h, cb, user));
}
int bp_cs_challenge_scalar(bp_cs* h, const char* label, uint64_t out[4]) {
    if (!cs_live(h) || !label || !out) return BP_E_ARG;
    if (!CS_DISPATCH(h, h->cs0->phase2, h->cs1->phase2)) { g_err = "challenge_scalar is only available to randomized constraints (RandomizedConstraintSystem)"; return BP_E_ARG; }
    F4 r = CS_DISPATCH(h, h->cs0->challenge_scalar(label), h->cs1->challenge_scalar(label));
    memcpy(out, r.v, 32);
    return BP_OK;
}
int bp_prover_precompute(bp_cs* h, const uint8_t rng_bytes[32]) {
    if (!cs_live(h) || !h->proving) return BP_E_ARG;
    if (rng_bytes) { memcpy(h->rng32, rng_bytes, 32); h->have_rng = true; }
    if (!h->have_rng) { g_err = "prove: the external rng bytes are missing"; return BP_E_ARG; }
    if (h->curve == 0) { if (!h->pre0.rng) prove_precompute<Secq>(*h->cs0, h->rng32, h->pre0); }
    else { if (!h->pre1.rng) prove_precompute<Zorro>(*h->cs1, h->rng32, h->pre1); }
    return BP_OK;
}
// the same for a batch: groups of 8 same-shaped statements share one AVX-512 Keccak-f x8 stream; others go one by one
int bp_prover_precompute_batch(bp_cs** hs, size_t count) {
    if (!hs) return BP_E_ARG;
    for (size_t i = 0; i < count; i++) if (!cs_live(hs[i]) || !hs[i]->proving || !hs[i]->have_rng || hs[i]->curve != hs[0]->curve) return BP_E_ARG;
    if (!count) return BP_OK;
    return hs[0]->curve == 0 ? cs_precompute_batch<Secq>(hs, count) : cs_precompute_batch<Zorro>(hs, count);
}
int bp_prover_set_rng(bp_cs* h, const uint8_t rng_bytes[32]) {
    if (!cs_live(h) || !h->proving || !rng_bytes) return BP_E_ARG;
    memcpy(h->rng32, rng_bytes, 32); h->have_rng = true;
    return BP_OK;
}
int bp_prover_prove(bp_ctx* c, bp_cs* h, const uint8_t rng_bytes[32], uint8_t* proof_out, size_t* proof_len, double* timing) {
    if (!c || !h || !h->proving || !proof_out || !proof_len || h->curve != c->curve) return BP_E_ARG;
    if (h->consumed) { g_err = "prove: the prover was already consumed (Prover::prove takes self)"; return BP_E_ARG; }
    if (rng_bytes) {
        if (h->have_rng && (h->pre0.rng || h->pre1.rng) && memcmp(h->rng32, rng_bytes, 32)) { g_err = "prove: rng bytes differ from the precomputed ones"; return BP_E_ARG; }
        memcpy(h->rng32, rng_bytes, 32); h->have_rng = true;
    }
    if (!h->have_rng) { g_err = "prove: the external rng bytes are missing"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    if (!c->gens_cap) { g_err = "prove: generators not installed (bp_gens_derive / bp_gens_upload / bp_gens_share)"; return BP_E_GENS_LENGTH; }
    h->consumed = true; h->running = true;
    const int rc = c->curve == 0 ? cs_prove<Secq>(c, h, proof_out, proof_len, timing) : cs_prove<Zorro>(c, h, proof_out, proof_len, timing);
    h->running = false;
    return rc;
}
int bp_r1cs_batch_verify(bp_ctx* c, size_t count, bp_cs* const* verifiers, const uint8_t* proofs, const size_t* proof_lens, const uint64_t* alphas, double* timing,
                         uint64_t* check_point_xy) {
    if (!c) return BP_E_ARG;
    if (count == 0) { if (check_point_xy) memset(check_point_xy, 0, 64); return BP_OK; }   // the reference's mega-check of nothing is the identity (verifier.rs:685-690)
    if (!verifiers || !proofs || !proof_lens) return BP_E_ARG;
    // The reference weights every instance with a fresh random scalar (verifier.rs:649): with equal weights the errors of two
    // invalid proofs can cancel in the single mega-check.  Only Verifier::verify (one instance) may leave the weight at 1.
    if (count > 1 && !alphas) { g_err = "batch_verify: per-instance weights (alphas) are required for more than one instance"; return BP_E_ARG; }
    for (size_t k = 0; k < count; k++) {
        if (!verifiers[k] || verifiers[k]->consumed || verifiers[k]->proving || verifiers[k]->curve != c->curve) { g_err = "batch_verify: every instance needs a live verifier of the ctx's curve"; return BP_E_ARG; }
    }
    if (count > 1) {   // a verifier is consumed by ONE instance (two pool threads on one recorder would race): any count
        std::vector<const bp_cs*> sorted(verifiers, verifiers + count);
        std::sort(sorted.begin(), sorted.end(), std::less<const bp_cs*>());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) { g_err = "batch_verify: a verifier is consumed by ONE instance"; return BP_E_ARG; }
    }
    g_dry = c->host_only;
    HIPCHK(hipSetDevice(c->device));
    if (!c->gens_cap) { g_err = "batch_verify: generators not installed"; g_dry = false; return BP_E_GENS_LENGTH; }
    const int rc = c->curve == 0 ? cs_batch_verify<Secq>(c, count, verifiers, proofs, proof_lens, alphas, timing, check_point_xy)
                                 : cs_batch_verify<Zorro>(c, count, verifiers, proofs, proof_lens, alphas, timing, check_point_xy);
    g_dry = false;
    return rc;
}
int bp_verifier_verify(bp_ctx* c, bp_cs* v, const uint8_t* proof, size_t proof_len) {
    return bp_r1cs_batch_verify(c, 1, &v, proof, &proof_len, nullptr, nullptr, nullptr);
}
// merlin::Transcript state for hosts that keep their own merlin: 200 bytes of Keccak state, then pos, pos_begin, cur_flags
int bp_transcript_export_state(const void* t, uint8_t out[203]) {
    if (!t || !out) return BP_E_ARG;
    const host::Transcript* tr = (const host::Transcript*)t;
    memcpy(out, tr->s.st.b, 200); out[200] = tr->s.pos; out[201] = tr->s.pos_begin; out[202] = tr->s.cur;
    return BP_OK;
}
int bp_transcript_import_state(void* t, const uint8_t in[203]) {
    if (!t || !in || in[200] >= host::Strobe::RATE || in[201] > host::Strobe::RATE) return BP_E_ARG;
    host::Transcript* tr = (host::Transcript*)t;
    memcpy(tr->s.st.b, in, 200); tr->s.pos = in[200]; tr->s.pos_begin = in[201]; tr->s.cur = in[202];
    return BP_OK;
}
void* bp_transcript_clone(const void* t) { return t ? new host::Transcript(*(const host::Transcript*)t) : nullptr; }

// dst uses src's resident generator tables (same device, same curve) without copying; src must outlive dst
int bp_gens_share(bp_ctx* dst, bp_ctx* src) {
    if (!dst || !src || dst == src || dst->curve != src->curve || dst->device != src->device || !src->gens_cap) return BP_E_ARG;
    DevBuf* d[] = {&dst->d_G, &dst->d_H, &dst->d_pc};
    DevBuf* sr[] = {&src->d_G, &src->d_H, &src->d_pc};
    for (int i = 0; i < 3; i++) { d[i]->release(); d[i]->p = sr[i]->p; d[i]->cap = sr[i]->cap; d[i]->owned = false; }
    dst->gens_cap = src->gens_cap; dst->pc_B = src->pc_B; dst->pc_Bb = src->pc_Bb;
    DevBuf* dt[] = {&dst->ftab_G, &dst->ftab_H, &dst->fb_G, &dst->fb_H, &dst->fb_pc};
    DevBuf* stb[] = {&src->ftab_G, &src->ftab_H, &src->fb_G, &src->fb_H, &src->fb_pc};
    for (int i = 0; i < 5; i++) { dt[i]->release(); dt[i]->p = stb[i]->p; dt[i]->cap = stb[i]->cap; dt[i]->owned = false; }
    dst->dt_tab.release(); dst->dt_cap = 0;
    if (src->dt_cap) { dst->dt_tab.p = src->dt_tab.p; dst->dt_tab.cap = src->dt_tab.cap; dst->dt_tab.owned = false; dst->dt_cap = src->dt_cap; }
    dst->fb_cap = src->fb_cap;
    dst->ftab_n = src->ftab_n; dst->ftab_w = src->ftab_w; dst->ftab_nwin = src->ftab_nwin; dst->ftab_first = src->ftab_first; dst->ftab_stride = src->ftab_stride;
    return BP_OK;
}

int bp_r1cs_verification_gh(bp_ctx* c, size_t n, size_t n1, const uint64_t* wL, const uint64_t* wR, const uint64_t* wO, const uint64_t y[4],
                            const uint64_t x[4], const uint64_t u[4], const uint64_t a[4], const uint64_t b[4], const uint64_t* ipa_challenges, size_t k,
                            uint64_t* g_out, uint64_t* h_out) {
    if (!c || !y || !x || !u || !a || !b || !g_out || !h_out || (n && (!wL || !wR || !wO)) || (k && !ipa_challenges) || k >= 32) { g_err = "bp_r1cs_verification_gh: bad argument"; return BP_E_ARG; }
    if (n > ((size_t)1 << k) || n1 > n) { g_err = "bp_r1cs_verification_gh: n exceeds the padded size 2^k"; return BP_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? verification_gh_host_entry<Secq>(c, n, n1, wL, wR, wO, y, x, u, a, b, ipa_challenges, k, g_out, h_out)
                         : verification_gh_host_entry<Zorro>(c, n, n1, wL, wR, wO, y, x, u, a, b, ipa_challenges, k, g_out, h_out);
}
int bp_ipa_verify(bp_ctx* c, size_t n, const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t P_xy[8], const uint64_t Q_xy[8],
                  const uint64_t* G_xy, const uint64_t* H_xy, const uint64_t* L_xy, const uint64_t* R_xy, size_t lg_n, const uint64_t* challenges,
                  const uint64_t a[4], const uint64_t b[4]) {
    if (!c || !n || !G_factors || !H_factors || !P_xy || !Q_xy || !G_xy || !H_xy || !a || !b || (lg_n && (!L_xy || !R_xy || !challenges))) {
        g_err = "bp_ipa_verify: bad argument"; return BP_E_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? ipa_verify_host_entry<Secq>(c, n, G_factors, H_factors, P_xy, Q_xy, G_xy, H_xy, L_xy, R_xy, lg_n, challenges, a, b)
                         : ipa_verify_host_entry<Zorro>(c, n, G_factors, H_factors, P_xy, Q_xy, G_xy, H_xy, L_xy, R_xy, lg_n, challenges, a, b);
}

// test hook (host only): the GLV split of the uniform fold multiplier t (ark words): masks = p1[5] m1[5] p2[5] m2[5],
// t = sum (p1-m1)_i 2^i + lambda * sum (p2-m2)_i 2^i (mod r).  BP_E_ARG for a curve without the endomorphism.
int bp_debug_glv_decompose(int curve, const uint64_t t[4], uint32_t masks[20], uint64_t lambda_out[4]) {
    if (curve != 0 || !t || !masks || !lambda_out) return BP_E_ARG;
    F4 x; memcpy(x.v, t, 32);
    if (!glv_decompose<Secq>(x, masks, masks + 5, masks + 10, masks + 15)) { g_err = "glv_decompose: half longer than 129 bits"; return BP_E_ARG; }
    F4 lam; memcpy(lam.v, Secq::LAMBDA64, 32);
    host::Fld<Secq::Fr>::to_canon(lambda_out, lam);
    return BP_OK;
}

int bp_debug_decompress(bp_ctx* c, const uint8_t* compressed33, size_t n, uint64_t* out_xy, uint32_t* out_ok) {
    if (!c || (n && (!compressed33 || !out_xy || !out_ok))) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    std::vector<F4> xs(n); std::vector<u32> fl(n);
    for (size_t i = 0; i < n; i++) {
        const uint8_t* d = compressed33 + 33 * i;
        fl[i] = d[32];
        bool ok = !(fl[i] & 0x3f) && (fl[i] & 0xc0) != 0xc0 &&
                  (c->curve == 0 ? host::Fld<Secq::Fq>::from_bytes(xs[i], d) : host::Fld<Zorro::Fq>::from_bytes(xs[i], d));
        if (ok && (fl[i] & 0x40) && !xs[i].is_zero()) ok = false;
        if (!ok) { xs[i] = F4{{0, 0, 0, 0}}; fl[i] = 0xFFu; }   // malformed framing: reported as not ok below
    }
    std::vector<A4> pts; std::vector<u32> okv;
    BPCHK(c->curve == 0 ? decompress_points<Secq>(c, xs.data(), fl.data(), n, pts, okv) : decompress_points<Zorro>(c, xs.data(), fl.data(), n, pts, okv));
    for (size_t i = 0; i < n; i++) {
        out_ok[i] = fl[i] == 0xFFu ? 0 : okv[i];
        if (out_ok[i]) { memcpy(out_xy + 8 * i, pts[i].x.v, 32); memcpy(out_xy + 8 * i + 4, pts[i].y.v, 32); } else memset(out_xy + 8 * i, 0, 64);
    }
    return BP_OK;
}

// ---- verifier front end: test hooks (include/arkbp.h) ------------------------------------------------------------------------
int bp_debug_vfe_schedule_replay(const uint8_t state203[203], int absorb_commitments, uint64_t m, uint32_t k, uint64_t n, const uint8_t* items, uint8_t* seeds_out,
                                 uint32_t* nblocks_out) {
    if (!state203 || !items || !seeds_out || k >= 32 || state203[200] >= host::Strobe::RATE || m > 65535) return BP_E_ARG;
    vfe::Schedule sc;
    if (!vfe::build_verifier_schedule(sc, state203[200], state203[201], absorb_commitments != 0, m, k, n)) { g_err = "vfe schedule: unsupported shape"; return BP_E_ARG; }
    uint64_t st[25];
    memcpy(st, state203, 200);
    vfe::run_schedule_cpu(sc, st, items, seeds_out, [](uint64_t* s) { host::keccakf((host::u64*)s); });
    if (nblocks_out) *nblocks_out = (uint32_t)sc.blocks.size();
    return BP_OK;
}
int bp_debug_vfe_challenges(bp_ctx* c, size_t count, const uint8_t* proofs, size_t proof_len, const uint64_t* commit_xy, size_t m, const uint8_t* states203,
                            int shared_state, int absorb_commitments, uint8_t* seeds_out, uint64_t* chal_out, uint32_t* status_out) {
    if (!c || !count || !proofs || !states203 || !seeds_out || !chal_out || !status_out || (m && !commit_xy)) return BP_E_ARG;
    if (proof_len < 539 || (proof_len - 539) % 66 || count > 65535 || m > 65535) return BP_E_ARG;
    const size_t k = (proof_len - 539) / 66;
    if (k >= 32) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nst = shared_state ? 1 : count;
    for (size_t i = 0; i < nst; i++) if (states203[203 * i + 200] != states203[200] || states203[203 * i + 201] != states203[201]) { g_err = "vfe: transcript positions differ"; return BP_E_ARG; }
    vfe::Schedule sc;
    if (!vfe::build_verifier_schedule(sc, states203[200], states203[201], absorb_commitments != 0, m, (uint32_t)k, (uint64_t)1 << k)) return BP_E_ARG;
    const std::vector<uint32_t> enc = sc.encode();
    vfe::Shape sh;
    sh.P = (uint32_t)count; sh.m = (uint32_t)m; sh.nV = absorb_commitments ? (uint32_t)m : 0u; sh.k = (uint32_t)k; sh.plen = (uint32_t)proof_len;
    sh.tail = (uint32_t)(6 + m + 5 + 2 * k); sh.nitems = sh.nV + 11 + 2 * (uint32_t)k + 3;
    const size_t nch = 6 + k;
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_proofs = up256(count * proof_len), b_V = up256(count * m * 64), b_st = up256(nst * 200), b_sched = up256(enc.size() * 4);
    BPCHK(c->vfe_in.ensure(b_proofs + b_V + b_st + b_sched));
    BPCHK(c->vfe_msg.ensure((size_t)sh.nitems * vfe::ITEM_WORDS * count * 8));
    BPCHK(c->vfe_chal.ensure(count * nch * 32));
    BPCHK(c->vfe_ws.ensure(count * nch * 32 + count * nch * 32));
    BPCHK(c->vfe_small.ensure(4096));
    BPCHK(c->r_tail.ensure(count * sh.tail * 64));
    uint8_t* d_in = (uint8_t*)c->vfe_in.p;
    std::vector<uint8_t> states(nst * 200);
    for (size_t i = 0; i < nst; i++) memcpy(&states[200 * i], states203 + 203 * i, 200);
    HIPCHK(hipMemcpyAsync(d_in, proofs, count * proof_len, hipMemcpyHostToDevice, st));
    if (m) HIPCHK(hipMemcpyAsync(d_in + b_proofs, commit_xy, count * m * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_in + b_proofs + b_V, states.data(), nst * 200, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_in + b_proofs + b_V + b_st, enc.data(), enc.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c->vfe_small.p, 0, 256, st));
    if (vfe::launch_points(c->curve, st, sh, d_in, (const u32*)(d_in + b_proofs), (uint64_t*)c->vfe_msg.p, c->r_tail.as<u32>(), c->vfe_small.as<u32>())) return BP_E_HIP;
    uint8_t* d_seeds = (uint8_t*)c->vfe_ws.p;
    u32* d_ark = c->vfe_ws.as<u32>() + count * nch * 8;
    if (vfe::launch_sponge(c->curve, st, sh, (const u32*)(d_in + b_proofs + b_V + b_st), (const uint64_t*)(d_in + b_proofs + b_V), shared_state ? 0u : 25u, (const uint64_t*)c->vfe_msg.p,
                           c->vfe_chal.as<u32>(), d_seeds)) return BP_E_HIP;
    if (c->curve == 0) hipLaunchKernelGGL(k_dbg_scalars_to_ark<Secq>, dim3((u32)((count * nch + 255) / 256)), dim3(256), 0, st, c->vfe_chal.as<u32>(), d_ark, (u32)(count * nch));
    else hipLaunchKernelGGL(k_dbg_scalars_to_ark<Zorro>, dim3((u32)((count * nch + 255) / 256)), dim3(256), 0, st, c->vfe_chal.as<u32>(), d_ark, (u32)(count * nch));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(seeds_out, d_seeds, count * nch * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(chal_out, d_ark, count * nch * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(status_out, c->vfe_small.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}
int bp_debug_verify_challenges(int curve, size_t count, const int* scenarios, const uint64_t* params, const uint8_t* proofs, const size_t* proof_lens,
                               const uint64_t* commit_xy, const size_t* ms, const uint64_t* publics, const size_t* npubs, int use_x8, uint64_t* out, size_t* nchal) {
    if (!count || count > 8 || !scenarios || !params || !proofs || !proof_lens || !commit_xy || !ms || !publics || !npubs || !out || !nchal) return BP_E_ARG;
    return curve == 0 ? dbg_verify_challenges<Secq>(count, scenarios, params, proofs, proof_lens, commit_xy, ms, publics, npubs, use_x8, out, nchal)
                      : dbg_verify_challenges<Zorro>(count, scenarios, params, proofs, proof_lens, commit_xy, ms, publics, npubs, use_x8, out, nchal);
}
int bp_ctx_fold_stats(bp_ctx* c, uint64_t* deferred_first_folds, uint64_t* second_folds_from_tables) {
    if (!c) return BP_E_ARG;
    if (deferred_first_folds) *deferred_first_folds = c->folds_deferred;
    if (second_folds_from_tables) *second_folds_from_tables = c->folds_tab2;
    return BP_OK;
}
int bp_ctx_direct_stats(bp_ctx* c, uint64_t* direct_msms, size_t* bases_per_vector) {
    if (!c) return BP_E_ARG;
    if (direct_msms) *direct_msms = c->dt_runs;
    if (bases_per_vector) *bases_per_vector = c->dt_cap;
    return BP_OK;
}
int bp_ctx_msm_stats(bp_ctx* c, uint64_t* fixed_base_runs, uint64_t* fixed_base_runs_sharded) {
    if (!c) return BP_E_ARG;
    if (fixed_base_runs) *fixed_base_runs = c->fb_runs;
    if (fixed_base_runs_sharded) *fixed_base_runs_sharded = c->fb_runs_sharded;
    return BP_OK;
}
int bp_ctx_vfe_stats(bp_ctx* c, uint64_t* device_batches, uint64_t* host_fallbacks) {
    if (!c) return BP_E_ARG;
    if (device_batches) *device_batches = c->vfe_batches;
    if (host_fallbacks) *host_fallbacks = c->vfe_fallbacks;
    return BP_OK;
}

int bp_ctx_set_profiling(bp_ctx* c, int enabled) { if (!c) return BP_E_ARG; c->profiling = enabled != 0; return BP_OK; }
int bp_ctx_kernel_time(bp_ctx* c, int which, double* ms_total, uint64_t* launches) {
    if (!c || which < 0 || which >= BP_K_COUNT) return BP_E_ARG;
    collect_timers(c);
    if (ms_total) *ms_total = c->timers[which].ms;
    if (launches) *launches = c->timers[which].launches;
    return BP_OK;
}
int bp_ctx_reset_profiling(bp_ctx* c) {
    if (!c) return BP_E_ARG;
    collect_timers(c);
    for (int k = 0; k < BP_K_COUNT; k++) { c->timers[k].ms = 0; c->timers[k].launches = 0; }
    return BP_OK;
}

int bp_debug_exp_iter(bp_ctx* c, const uint64_t x[4], size_t n, uint64_t* out) {
    if (!c || !x || (n && !out) || n >= ((size_t)1 << 31)) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? dbg_exp_iter<Secq>(c, x, n, out) : dbg_exp_iter<Zorro>(c, x, n, out);
}
int bp_debug_inner_product(bp_ctx* c, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
    if (!c || !a || !b || !out || !n || n >= ((size_t)1 << 30)) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    return c->curve == 0 ? dbg_inner_product<Secq>(c, a, b, n, out) : dbg_inner_product<Zorro>(c, a, b, n, out);
}
int bp_debug_field_op(bp_ctx* c, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!c || !a || !b || !out || field < 0 || field > 3) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    BPCHK(c->io_pts.ensure(n * 32)); BPCHK(c->io_scal.ensure(n * 32)); BPCHK(c->io_out.ensure(n * 32));
    HIPCHK(hipMemcpyAsync(c->io_pts.p, a, n * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->io_scal.p, b, n * 32, hipMemcpyHostToDevice, c->stream));
    const u32 gb = (u32)((n + 63) / 64);
    const u32 *pa = c->io_pts.as<u32>(), *pb = c->io_scal.as<u32>();
    u32* po = c->io_out.as<u32>();
    switch (field) {
        case 0: hipLaunchKernelGGL(k_dbg_field<SecqFq>, dim3(gb), dim3(64), 0, c->stream, op, pa, pb, po, (u32)n); break;
        case 1: hipLaunchKernelGGL(k_dbg_field<SecqFr>, dim3(gb), dim3(64), 0, c->stream, op, pa, pb, po, (u32)n); break;
        case 2: hipLaunchKernelGGL(k_dbg_field<ZorroFq>, dim3(gb), dim3(64), 0, c->stream, op, pa, pb, po, (u32)n); break;
        default: hipLaunchKernelGGL(k_dbg_field<ZorroFr>, dim3(gb), dim3(64), 0, c->stream, op, pa, pb, po, (u32)n); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, po, n * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}
int bp_debug_point_op(bp_ctx* c, int op, const uint64_t* p, const uint64_t* q, const uint64_t* k, uint64_t* out, size_t n) {
    if (!c || !p || !q || !k || !out) return BP_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    BPCHK(c->io_pts.ensure(n * 128)); BPCHK(c->io_scal.ensure(n * 32)); BPCHK(c->io_out.ensure(n * 64));
    u32* dp = c->io_pts.as<u32>();
    u32* dq = dp + n * 16;
    HIPCHK(hipMemcpyAsync(dp, p, n * 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dq, q, n * 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->io_scal.p, k, n * 32, hipMemcpyHostToDevice, c->stream));
    const u32 gb = (u32)((n + 63) / 64);
    if (c->curve == 0) hipLaunchKernelGGL(k_dbg_point<Secq>, dim3(gb), dim3(64), 0, c->stream, op, dp, dq, c->io_scal.as<u32>(), c->io_out.as<u32>(), (u32)n);
    else hipLaunchKernelGGL(k_dbg_point<Zorro>, dim3(gb), dim3(64), 0, c->stream, op, dp, dq, c->io_scal.as<u32>(), c->io_out.as<u32>(), (u32)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx_stream_wait(c));
    return BP_OK;
}

}  // extern "C"
