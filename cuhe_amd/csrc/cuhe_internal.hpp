// cuhe_internal.hpp -- what the three translation units of the C ABI share (context / transforms / key switch): the error
// convention, the state of the library (parameters, per-device tables, per-thread workspaces) and the functions one unit
// calls in another.  Nothing here is part of the boundary (include/cuhe_hip.h is).
#pragma once
// the library is built with -fvisibility=hidden: only the boundary is exported
#pragma GCC visibility push(default)
#include "../../include/cuhe_hip.h"
#pragma GCC visibility pop

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <vector>

#include "host_math.hpp"
#include "ntt_kernels.cuh"
#include "ntt_onewg.hpp"
#include "ops_kernels.cuh"
#include "icrt_mfma.cuh"

namespace cuhe_impl {
using namespace cuhe;
using cuhe::host::BigU;
using cuhe::host::Params;

// ------------------------------------------------------------------ errors
extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);
#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(CUHE_EHIP, "%s failed at %s:%d : %s", #call, __FILE__, __LINE__,     \
                        hipGetErrorString(e_));                                              \
    } while (0)
#define CHK(call) do { int r_ = (call); if (r_ != CUHE_OK) return r_; } while (0)

// ------------------------------------------------------------------ state
struct NttTab {
    u64 *T1w = nullptr, *T2 = nullptr, *T2inv = nullptr;    // T1w: inner twiddles of pass 1; T2 / T2inv: outer twiddles (x L^-1)
    u64 *tw = nullptr, *twinv = nullptr;                    // negacyclic twist psi^j and psi^-j, psi^2 = w_L (ensure_twist)
    u64 *Wn1 = nullptr;                                     // w_N1^e, e < N1: stage twiddles of the low-latency pass 1
    std::atomic<int> ready{0};                              // (ntt_chunk setting + 1) the tables and `chunk` below were prepared for
    NttTab() {}
    NttTab(const NttTab &o) : T1w(o.T1w), T2(o.T2), T2inv(o.T2inv), tw(o.tw), twinv(o.twinv), Wn1(o.Wn1), ready(o.ready.load()), chunk(o.chunk) {}
    NttTab &operator=(const NttTab &o) { T1w = o.T1w; T2 = o.T2; T2inv = o.T2inv; tw = o.tw; twinv = o.twinv; Wn1 = o.Wn1; ready.store(o.ready.load()); chunk = o.chunk; return *this; }
    int chunk = 0;                             // transforms per launch pair (slab size / transform size)
};
// Mutable scratch of ONE host thread on one device.  The reference keeps a single set per device and is therefore
// not re-entrant per device (cuhe/Operations.cu:171-209, Relinearization.cu:37-38); here every host thread that
// calls into the library gets its own set, so several threads can drive the same GPU on their own streams and the
// small kernels of independent ciphertext operations overlap on the device.
struct Workspace {
    u64 *slab[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // pass-1 -> pass-2 slabs per length
    size_t slab_bytes[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    size_t n_barrett = 0, n_alias = 0, n_relin = 0;     // rows / elements the buffers below currently hold
    // scratch of the batched multiply + relinearise (cuhe_hip_mul_relin_batch), sized by the largest batch seen
    u64 *bt_ntt = nullptr; u32 *bt_crt = nullptr; size_t n_bt = 0;
    u64 *mr_ntt = nullptr; u32 *mr_crt = nullptr; size_t n_mr = 0;        // cuhe_hip_mul_raw_batch
    u64 *b_ntt = nullptr;                // Barrett scratch (cuhe/Operations.cu:196-209)
    u32 *b_mq = nullptr, *b_crt = nullptr;  // q (at offset n) and (m - x^n) q
    u32 *hold = nullptr;                 // inttResult (Operations.cu:171-172)
    u32 *b_alias = nullptr;              // copy of the input when barrett() is asked to work in place
    u64 *relin = nullptr;                // NTT-domain windows of the ciphertext being relinearised
    u32 *win = nullptr;                  // u32[numEvalKey][crtLen] window rows
    // scratch of the CRT-prime-sharded multiply + relinearise (own operand rows, gathered CRT rows, raw, own result rows)
    u64 *sh_a = nullptr, *sh_b = nullptr; u32 *sh_rows = nullptr, *sh_raw = nullptr, *sh_out = nullptr; bool sh_ready = false;
    unsigned *pair_cnt = nullptr;        // rendezvous counters of the persistent one-workgroup transforms (one per pair of workgroups): per host
                                         // thread like every scratch, so that launches of different threads on one device never share them
    hipStream_t last = nullptr; bool used = false;
    hipEvent_t ev = nullptr;             // orders this thread's work when it moves to another stream
    // lanes of the batched relinearisation (relin_batch_core): a helper lane owns a stream; lane 0 marks "inputs ready"
    hipStream_t lane_stream = nullptr; hipEvent_t ev_lane = nullptr, ev_in = nullptr;
    u64 *rc_acc = nullptr;               // the sums of the single-ciphertext relinearisation chain (cuhe_hip_relin_crt)
    // gathered rows of the list transforms (cuhe_hip_ct_ntt_list / cuhe_hip_ct_intt_list) when a call does not take a one-workgroup kernel
    u32 *ls_crt = nullptr; u64 *ls_ntt = nullptr; size_t n_ls_crt = 0, n_ls_ntt = 0;
};
struct IcrtLevel {
    u32 *M = nullptr, *mi = nullptr, *bi = nullptr; double *rp = nullptr; int W = 0, np = 0;
    // the constants in the layout of the matrix-core form (icrt_mfma.cuh); dig == nullptr: that form does not apply (primes of 2^28 and more)
    unsigned char *dig = nullptr; IcrtPrimeConst *pc = nullptr; u32 *nm = nullptr; int tiles = 0, ksteps = 0;
};

// tables of the one-workgroup transforms (ntt_onewg.cuh) of Lh = 2^(13 + index) points
struct OwTab {
    u64 *TW1f = nullptr, *TW1i = nullptr, *TW1h = nullptr, *TW2 = nullptr;      // forward, inverse (x Lh^-1), both parities of the zero-padded form, stage 2
    u64 *TW1hi = nullptr;                                                       // TW1h / (2 Lh): the halves of a SPLIT inverse row of 2 Lh points
    u64 *TW1g = nullptr; u64 c128 = 0; int i4neg = 0;                           // halves of the negacyclic forward transform of 2 Lh points (ensure_onewg_twist)
    std::atomic<int> ready{0};
    OwTab() {}
    OwTab(const OwTab &o) : TW1f(o.TW1f), TW1i(o.TW1i), TW1h(o.TW1h), TW2(o.TW2), TW1hi(o.TW1hi), TW1g(o.TW1g), c128(o.c128), i4neg(o.i4neg), ready(o.ready.load()) {}
    OwTab &operator=(const OwTab &o) { TW1f = o.TW1f; TW1i = o.TW1i; TW1h = o.TW1h; TW2 = o.TW2; TW1hi = o.TW1hi; TW1g = o.TW1g; c128 = o.c128; i4neg = o.i4neg; ready.store(o.ready.load()); return *this; }
};
struct DevCtx {
    bool ready = false;
    NttTab ntt[4];                       // LG 13 (one-workgroup form only: twist tables), 14, 15, 16
    OwTab ow[4];                         // sub-transforms of 4K, 8K, 16K, 32K points
    int cus = 0;                         // compute units (policy of the one-workgroup transforms)
    // prime tables
    u32 *p = nullptr, *e64 = nullptr, *pow32 = nullptr, *invp = nullptr;
    u64 *pinv = nullptr;
    int maxW = 0;
    std::vector<IcrtLevel> icrt;
    // Barrett tables / scratch (cuhe/Operations.cu:193-209, Base.cu:181-223)
    u64 *u_ntt = nullptr, *m_ntt = nullptr;
    u64 *uh_ntt = nullptr, *mh_ntt = nullptr;      // folded reduction: half-length transforms of U and Phi mod (x^Lh - 1)
    FoldGeom fold{0, 0, 0, 0, 0}; bool fold_ok = false;
    u32 *m_crt = nullptr;
    // relinearisation (cuhe/Relinearization.cu:37-38) -- keys resident in HBM
    u64 *ek = nullptr;
    int ek_first = 0, ek_count = 0;      // CRT primes whose keys this device holds: row 0 of `ek` is prime ek_first (cuhe_hip_init_relin_range)
    unsigned char *ekd = nullptr; MacDigGeom ekg{0, 0, 0, 0, 0}; bool ekd_unavailable = false;      // signed base-256 digits of the keys in MFMA operand order (built on first use)
    std::vector<Workspace *> spaces;     // every workspace of this device (owned here)
    std::vector<Workspace *> idle;       // workspaces of finished threads, adopted by later ones
    // allocator (cuhe/DeviceManager.cu:98-138)
    // helper streams/events for the pass-1 / pass-2 software pipeline
    hipStream_t sh_stream = nullptr;     // in-process sharded multiply: this device's stream and its stage events
    hipEvent_t sh_e1 = nullptr, sh_e2 = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    hipEvent_t ev_start = nullptr, ev_p1[2] = {nullptr, nullptr}, ev_p2[2] = {nullptr, nullptr};
    std::multimap<size_t, void *> freeBlocks;
    std::map<void *, size_t> allocated;
    size_t cachedBytes = 0;              // bytes parked in freeBlocks and streamBlocks
    std::map<hipStream_t, std::multimap<size_t, void *>> streamBlocks;   // freed in stream order, not yet synchronised
    std::set<hipStream_t> ownStreams;    // streams made by cuhe_hip_stream_create (and not destroyed yet): the only ones, besides the null stream, the allocator
                                         // queries or records on by itself -- a client's own handle may have been destroyed behind the library's back
    std::vector<hipEvent_t> fenceEvents;                                  // hand-over of a parked block to another stream (cuhe_hip_malloc_stream)
};

struct Global {
    Params prm;
    Params prm_init; int nc_mode_init = -1;   // what the live context was initialised with (cuhe_hip_same_ring)
    bool params_set = false, inited = false, relin_ready = false;
    int ndev = 1, dev_base = 0;
    bool virtual_devices = false;  // tests: logical devices 0..ndev-1 all live on physical device dev_base
    std::vector<uint32_t> primes;
    std::vector<BigU> coeffModulus;
    std::vector<int32_t> modulus;
    int reduce_kind = 0;                 // 0 generic, 1 x^n+1, 2 prime m
    int nc_mode = -1;                    // -1: negacyclic ciphertext domain wherever it applies (default), 0: never (tests)
    bool nc = false;                     // ciphertext-domain transforms are NEGACYCLIC of length modLen (decided by init)
    bool force_generic = false;
    bool no_fold = false;          // tests: take the five-transform form of the generic reduction
    bool allocator_on = false;
    size_t cache_cap = (size_t)4 << 30;  // with the pooled allocator off, freed blocks are still kept up to this many bytes
    int ntt_chunk = 0;
    // one-workgroup transforms: 0 never; 1 where they exist and the call fills the chip; 2 wherever they exist (tests)
    int onewg = getenv("CUHE_ONEWG") ? atoi(getenv("CUHE_ONEWG")) : 1;
    // zero-padded rows of 64K points (32K-point halves, ONE workgroup per CU): 0 two-pass kernels, 1 one workgroup per half,
    // 2 (default) persistent workgroups with LDS-DMA prefetch of the samples and a rendezvous of the two halves of a row before
    // their stores, for calls that give every workgroup at least two halves (2.71 vs 2.56 M transforms/s,
    // profiles/r03_onewg_ab.txt); smaller calls and unaligned rows take the two-pass kernels
    int onewg64 = getenv("CUHE_ONEWG64") ? atoi(getenv("CUHE_ONEWG64")) : 2;
    // full-length negacyclic rows as two half-length sub-transforms (32K points: two workgroups per CU instead of one), calls that
    // fill the chip: 0 off, 1 (default) inverse rows of 32K points, 2 also forward rows of 32K and inverse rows of 64K points (tests, A/B)
    int onewg_split = getenv("CUHE_ONEWG_SPLIT") ? atoi(getenv("CUHE_ONEWG_SPLIT")) : 1;
    bool ntt_overlap = false;     // measured: concurrent pass-1/pass-2 streams do not help (profiles/r01_chunk_sweep.txt)
    std::vector<DevCtx> dev;
    std::mutex mu;
    uint64_t generation = 1;      // bumped by shutdown: thread-local workspace pointers of older generations are stale
};
extern Global G_;


inline int lg_index(int len) { return len == 8192 ? 0 : len == 16384 ? 1 : len == 32768 ? 2 : len == 65536 ? 3 : -1; }
inline hipStream_t S(void *s) { return (hipStream_t)s; }

inline int phys_dev(int dev) { return G_.virtual_devices ? G_.dev_base : G_.dev_base + dev; }

// hipFuncSetAttribute once per (kernel instantiation, device); host threads may race to be first
struct AttrOnce {
    std::mutex mu; std::atomic<uint64_t> done{0};
    template <typename K> int set(K kern, int bytes) {
        int cur = 0;
        HIPCHK(hipGetDevice(&cur));
        const uint64_t bit = 1ull << (cur & 63);
        if (done.load(std::memory_order_acquire) & bit) return CUHE_OK;        // the common case: no lock on the launch path
        std::lock_guard<std::mutex> lk(mu);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            done.fetch_or(bit, std::memory_order_release);
        }
        return CUHE_OK;
    }
};
// store epilogue of an inverse transform beyond "mod p": kind 1 = reversed quotient, 2 = final subtraction of the folded
// reduction (aux = the product rows f, aux_stride their row length); see ntt_kernels.cuh
struct Epilogue { int kind = 0; const u32 *aux = nullptr; long aux_stride = 0; FoldGeom fg{0, 0, 0, 0, 0}; };
constexpr int kFoldXn1 = -1;     // nstore sentinel: inverse transform fused with the reduction mod x^(L/2)+1
constexpr int kNcInverse = -2;   // nstore sentinel: inverse NEGACYCLIC transform (untwist, centred lift, mod p), all L outputs

struct EvTimer {                 // optional per-pass hipEvent timing (bench)
    std::vector<hipEvent_t> ev;
    bool on = false;
};

constexpr int kLanes = 4;
extern thread_local int tls_lane;
struct LaneReset { ~LaneReset() { tls_lane = 0; } };

// ---- context (cuhe_context.hip)
int set_dev(int dev);
int workspace_of_thread(int dev, Workspace **out);
int workspace(int dev, hipStream_t st, Workspace **out);
int ws_barrett(Workspace &w, int rows = 0);
int ws_relin(Workspace &w, int cts = 1);
int ws_slab(Workspace &w, int li, int which, size_t bytes, u64 **out);
void free_workspace(Workspace *w);
int need_init(int dev);
int level_of(int logq, int *lvl, int *np, int *W);
inline PrimeTab prime_tab(const DevCtx &D) { return PrimeTab{D.p, D.pinv, D.e64, D.pow32, D.maxW}; }
PrimeTab prime_tab_at(const DevCtx &D, int prime0);
// CRT of `batch` polynomials onto the primes prime0 .. prime0 + np - 1
int launch_crt(u32 *dst, const u32 *src, const DevCtx &D, int prime0, int np, int W, int batch, long src_ct_stride, long dst_ct_stride, hipStream_t st);
template <typename T>
int ws_buffer(T **ptr, size_t count) {           // lazily allocated, fixed-size workspace member
    if (!*ptr) HIPCHK(hipMalloc((void **)ptr, std::max<size_t>(count, 1) * sizeof(T)));
    return CUHE_OK;
}
// A scratch buffer that has been outgrown is RETIRED, not freed: hipFree waits for the whole device, and under the gate scheduler
// that wait sat in the middle of the first block of a process (481 hipFree calls, 146 ms: profiles/r05_sched_prince.txt).  Work already
// enqueued may still use a retired buffer; it is freed at the next cuhe_hip_device_sync / shutdown.  Growth is geometric, so the
// retired buffers of a workspace member add up to less than its final size.
void ws_retire(void *p);
template <typename T>
int ws_grow(T **ptr, size_t *have, size_t count) {           // grow-only, geometric
    if (*have >= count && *ptr) return CUHE_OK;
    if (*ptr) { ws_retire(*ptr); count = std::max(count, 2 * *have); }
    *ptr = nullptr; *have = 0;
    HIPCHK(hipMalloc((void **)ptr, std::max<size_t>(count, 1) * sizeof(T)));
    *have = count;
    return CUHE_OK;
}
template <typename T>
int upload(T **dptr, const std::vector<T> &h) {
    HIPCHK(hipMalloc((void **)dptr, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIPCHK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return CUHE_OK;
}

// ---- matrix-core inverse CRT (icrt_mfma.hip)
bool icrt_mfma_supported(const IcrtLevel &I);
int launch_icrt_mfma(u32 *dst, const u32 *src, const DevCtx &D, const IcrtLevel &I, int np, int W, int batch, long src_ct_stride, long dst_ct_stride,
                     hipStream_t st, const IcrtWindows &wo);
// ---- transforms (cuhe_transforms.hip)
int ensure_ntt(int dev, int len, int batch_hint);
int ensure_twist(int dev, int len);
int ensure_onewg(OwTab &tab, int lgh);            // tables of the one-workgroup transforms of 2^lgh points (the current device); made at init
int ensure_onewg_twist(OwTab &tab, int lgh);
int run_ntt(int len, int mode, void *dst, const void *src, int batch, long src_stride, long dst_stride, int nstore,
            int prime0, WindowArgs wa, int dev, hipStream_t st, EvTimer *tm = nullptr, const u64 *mul_tab = nullptr, int np_mod = 0,
            const Epilogue *ep = nullptr, const RowRebase *rb = nullptr);
// (internal) returned by the transforms when rows in separate blocks (RowRebase) were asked for and the call would not take a one-workgroup
// kernel: nothing has been enqueued, the caller gathers the rows and calls again without
constexpr int kNoListForm = -1000;
int barrett_impl(u32 *dst, const u32 *src, int prime0, int np, int dev, hipStream_t st, int np_mod);
inline int ct_len() { return G_.nc ? G_.prm.modLen : G_.prm.nttLen; }
int need_cyclic();
int ct_forward(u64 *X, const u32 *x, int rows, int dev, hipStream_t st, const u64 *mul_tab = nullptr, int np_mod = 0, const RowRebase *rb = nullptr);
bool fused_xn1();
int ct_inverse(u32 *dst, const u64 *X, int rows, int prime0, int np_mod, bool is_prod, int dev, hipStream_t st, const u64 *Y = nullptr, const RowRebase *rb = nullptr);

}  // namespace cuhe_impl
