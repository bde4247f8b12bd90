// cuhe_hip.hip -- C ABI (include/cuhe_hip.h) of the gfx950 backend: context,
// precomputation, launch sequencing.  Replaces the L2 "operation drivers" layer
// of the reference (cuhe/Operations.cu, cuhe/Relinearization.cu,
// cuhe/DeviceManager.cu, the upload half of cuhe/Base.cu) -- same entry points
// and argument meaning, different machine mapping: every driver is ONE batched
// launch sequence over all CRT primes, constants are HBM tables, evaluation keys
// are device resident.
#include "../../include/cuhe_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "comm.hpp"
#include "host_math.hpp"
#include "ntt_kernels.cuh"
#include "ntt_onewg.hpp"
#include "ops_kernels.cuh"

using namespace cuhe;
using cuhe::host::BigU;
using cuhe::host::Params;

namespace {

// ------------------------------------------------------------------ errors
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
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
    hipStream_t last = nullptr; bool used = false;
    hipEvent_t ev = nullptr;             // orders this thread's work when it moves to another stream
    // lanes of the batched relinearisation (relin_batch_core): a helper lane owns a stream; lane 0 marks "inputs ready"
    hipStream_t lane_stream = nullptr; hipEvent_t ev_lane = nullptr, ev_in = nullptr;
};
struct IcrtLevel { u32 *M = nullptr, *mi = nullptr, *bi = nullptr; double *rp = nullptr; int W = 0, np = 0; };

// tables of the one-workgroup transforms (ntt_onewg.cuh) of Lh = 2^(13 + index) points
struct OwTab {
    u64 *TW1f = nullptr, *TW1i = nullptr, *TW1h = nullptr, *TW2 = nullptr;      // forward, inverse (x Lh^-1), both parities of the zero-padded form, stage 2
    u64 *TW1g = nullptr; u64 c128 = 0; int i4neg = 0;                           // 32K points only: halves of the negacyclic 64K-point forward transform (ensure_onewg_twist64)
    std::atomic<int> ready{0};
    OwTab() {}
    OwTab(const OwTab &o) : TW1f(o.TW1f), TW1i(o.TW1i), TW1h(o.TW1h), TW2(o.TW2), TW1g(o.TW1g), c128(o.c128), i4neg(o.i4neg), ready(o.ready.load()) {}
    OwTab &operator=(const OwTab &o) { TW1f = o.TW1f; TW1i = o.TW1i; TW1h = o.TW1h; TW2 = o.TW2; TW1g = o.TW1g; c128 = o.c128; i4neg = o.i4neg; ready.store(o.ready.load()); return *this; }
};
struct DevCtx {
    bool ready = false;
    NttTab ntt[4];                       // LG 13 (one-workgroup form only: twist tables), 14, 15, 16
    OwTab ow[4];                         // sub-transforms of 4K, 8K, 16K, 32K points
    int cus = 0;                         // compute units (policy of the one-workgroup transforms)
    unsigned *pair_cnt = nullptr;        // rendezvous counters of the persistent one-workgroup transform (one per pair of workgroups)
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
};

struct Global {
    Params prm;
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
    bool ntt_overlap = false;     // measured: concurrent pass-1/pass-2 streams do not help (profiles/r01_chunk_sweep.txt)
    std::vector<DevCtx> dev;
    std::mutex mu;
    uint64_t generation = 1;      // bumped by shutdown: thread-local workspace pointers of older generations are stale
} G_;

inline int lg_index(int len) { return len == 8192 ? 0 : len == 16384 ? 1 : len == 32768 ? 2 : len == 65536 ? 3 : -1; }
inline hipStream_t S(void *s) { return (hipStream_t)s; }

inline int phys_dev(int dev) { return G_.virtual_devices ? G_.dev_base : G_.dev_base + dev; }
int set_dev(int dev) {
    if (dev < 0 || dev >= G_.ndev) return fail(CUHE_EINVAL, "device %d out of range (numGPUs=%d)", dev, G_.ndev);
    HIPCHK(hipSetDevice(phys_dev(dev)));
    if ((int)G_.dev.size() < G_.ndev) {            // (multi_gpus / init size it already; kept for callers that skip them)
        std::lock_guard<std::mutex> lk(G_.mu);
        if ((int)G_.dev.size() < G_.ndev) G_.dev.resize(G_.ndev);
    }
    return CUHE_OK;
}

// the calling thread's workspace on `dev`, ordered after whatever this thread last enqueued with it
// A thread has kLanes workspaces per device: lane 0 is the one every entry point uses; lanes 1.. exist only while a
// batched relinearisation spreads groups of ciphertexts over helper streams (tls_lane selects the lane for everything
// the group's stages fetch through workspace_of_thread).
constexpr int kLanes = 4;
thread_local int tls_lane = 0;
struct LaneReset { ~LaneReset() { tls_lane = 0; } };
struct TlsSpaces {
    uint64_t gen = 0;
    std::vector<Workspace *> lanes[kLanes];
    ~TlsSpaces();                                 // a finished thread hands its workspaces to later threads
};
TlsSpaces::~TlsSpaces() {
    std::lock_guard<std::mutex> lk(G_.mu);
    if (gen != G_.generation) return;             // the library was shut down since: already freed
    for (auto &per_dev : lanes)
        for (size_t d = 0; d < per_dev.size() && d < G_.dev.size(); ++d)
            if (per_dev[d]) G_.dev[d].idle.push_back(per_dev[d]);
}
thread_local TlsSpaces tls_spaces;
int workspace_of_thread(int dev, Workspace **out) {
    TlsSpaces &T = tls_spaces;
    if (T.gen != G_.generation) { for (auto &v : T.lanes) v.clear(); T.gen = G_.generation; }
    std::vector<Workspace *> &per_dev = T.lanes[tls_lane];
    if ((int)per_dev.size() <= dev) per_dev.resize(dev + 1, nullptr);
    Workspace *w = per_dev[dev];
    if (!w) {
        {
            std::lock_guard<std::mutex> lk(G_.mu);
            auto &idle = G_.dev[dev].idle;
            if (!idle.empty()) { w = idle.back(); idle.pop_back(); }
        }
        if (w) {                                  // adopted from a finished thread: its last work may still be in flight
            HIPCHK(hipDeviceSynchronize());
            w->used = false; w->last = nullptr;
        } else {
            w = new Workspace();
            HIPCHK(hipEventCreateWithFlags(&w->ev, hipEventDisableTiming));
            std::lock_guard<std::mutex> lk(G_.mu);
            G_.dev[dev].spaces.push_back(w);
        }
        per_dev[dev] = w;
    }
    *out = w;
    return CUHE_OK;
}
int workspace(int dev, hipStream_t st, Workspace **out) {
    Workspace *w = nullptr;
    CHK(workspace_of_thread(dev, &w));
    if (w->used && w->last != st) {               // same thread, other stream: keep the scratch hazards ordered
        if (hipEventRecord(w->ev, w->last) == hipSuccess) HIPCHK(hipStreamWaitEvent(st, w->ev, 0));
        else { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }     // the previous stream no longer exists
    }
    w->last = st; w->used = true;
    *out = w;
    return CUHE_OK;
}
template <typename T>
int ws_buffer(T **ptr, size_t count) {           // lazily allocated, fixed-size workspace member
    if (!*ptr) HIPCHK(hipMalloc((void **)ptr, std::max<size_t>(count, 1) * sizeof(T)));
    return CUHE_OK;
}
template <typename T>
int ws_grow(T **ptr, size_t *have, size_t count) {           // grow-only (re-allocation synchronises: sizes settle at once)
    if (*have >= count && *ptr) return CUHE_OK;
    if (*ptr) HIPCHK(hipFree(*ptr));
    *ptr = nullptr; *have = 0;
    HIPCHK(hipMalloc((void **)ptr, std::max<size_t>(count, 1) * sizeof(T)));
    *have = count;
    return CUHE_OK;
}
// Barrett / inttResult scratch for `rows` polynomial rows (at least one level-0 ciphertext)
int ws_barrett(Workspace &w, int rows = 0) {
    const Params &q = G_.prm;
    const size_t need = (size_t)std::max(rows, q.numCrtPrime);
    if (w.n_barrett >= need && w.hold) return CUHE_OK;
    size_t a = 0, b = 0, c = 0, d = 0;
    CHK(ws_grow(&w.b_mq, &a, need * q.nttLen)); CHK(ws_grow(&w.b_crt, &b, need * q.nttLen));
    CHK(ws_grow(&w.b_ntt, &c, need * q.nttLen)); CHK(ws_grow(&w.hold, &d, need * q.nttLen));
    w.n_barrett = need;
    return CUHE_OK;
}
// window rows and their transforms for `cts` ciphertexts
int ws_relin(Workspace &w, int cts = 1) {
    const Params &q = G_.prm;
    if (w.n_relin >= (size_t)cts && w.relin) return CUHE_OK;
    size_t a = 0, b = 0;
    if (w.relin) { HIPCHK(hipFree(w.relin)); w.relin = nullptr; }
    if (w.win) { HIPCHK(hipFree(w.win)); w.win = nullptr; }
    CHK(ws_grow(&w.relin, &a, (size_t)cts * q.numEvalKey * q.nttLen)); CHK(ws_grow(&w.win, &b, (size_t)cts * q.numEvalKey * q.crtLen));
    w.n_relin = cts;
    return CUHE_OK;
}
int ws_slab(Workspace &w, int li, int which, size_t bytes, u64 **out) {        // grow-only
    if (w.slab_bytes[li][which] < bytes) {
        if (w.slab[li][which]) HIPCHK(hipFree(w.slab[li][which]));            // (synchronises: rare, sizes settle at once)
        w.slab[li][which] = nullptr; w.slab_bytes[li][which] = 0;
        HIPCHK(hipMalloc((void **)&w.slab[li][which], bytes));
        w.slab_bytes[li][which] = bytes;
    }
    *out = w.slab[li][which];
    return CUHE_OK;
}
void free_workspace(Workspace *w) {
    for (auto &per_len : w->slab) for (auto &sl : per_len) if (sl) hipFree(sl);
    void *ptrs[] = {w->b_ntt, w->b_mq, w->b_crt, w->hold, w->b_alias, w->relin, w->win, w->bt_ntt, w->bt_crt, w->mr_ntt, w->mr_crt,
                    w->sh_a, w->sh_b, w->sh_rows, w->sh_raw, w->sh_out};
    for (void *p : ptrs) if (p) hipFree(p);
    if (w->ev) hipEventDestroy(w->ev);
    if (w->ev_lane) hipEventDestroy(w->ev_lane);
    if (w->ev_in) hipEventDestroy(w->ev_in);
    if (w->lane_stream) hipStreamDestroy(w->lane_stream);
    delete w;
}

template <typename T>
int upload(T **dptr, const std::vector<T> &h) {
    HIPCHK(hipMalloc((void **)dptr, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIPCHK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return CUHE_OK;
}

// ------------------------------------------------------------------ NTT tables + launch
template <int LG>
int make_ntt_tables(NttTab &tab) {
    constexpr int L = 1 << LG, N1 = L / 64, RA = N1 / 64;
    std::vector<u64> r(L);
    const u64 w = host::powP(host::G, 65536 / L);                 // cuhe/Base.cu:63-70
    r[0] = 1;
    for (int i = 1; i < L; ++i) r[i] = host::mulP(r[i - 1], w);
    std::vector<u64> t2(L), t2i(L);
    const u64 linv = host::powP((u64)L, host::P - 2);             // cuhe/Base.cu:489,656,841
    for (int j2 = 0; j2 < 64; ++j2)
        for (int k1 = 0; k1 < N1; ++k1) {
            const u64 v = r[((long)j2 * k1) % L];
            t2[(size_t)j2 * N1 + k1] = v;
            t2i[(size_t)j2 * N1 + k1] = host::mulP(v, linv);
        }
    // pass 1: N1 = RA x 64, t1w[c*64 + b] = w_N1^(b*c), b < 64, c < RA
    std::vector<u64> t1w((size_t)N1);
    for (int c = 0; c < RA; ++c)
        for (int b = 0; b < 64; ++b) t1w[(size_t)c * 64 + b] = r[(64L * b * c) % L];
    std::vector<u64> wn1((size_t)N1);
    for (int e = 0; e < N1; ++e) wn1[e] = r[64L * e];
    CHK(upload(&tab.Wn1, wn1));
    CHK(upload(&tab.T1w, t1w));
    CHK(upload(&tab.T2, t2));
    CHK(upload(&tab.T2inv, t2i));
    return CUHE_OK;
}

int ensure_ntt(int dev, int len, int batch_hint) {
    (void)batch_hint;
    const int li = lg_index(len);
    if (li < 0) return fail(CUHE_EINVAL, "unsupported transform length %d (8192/16384/32768/65536 only)", len);
    {
        const NttTab &t0 = G_.dev[dev].ntt[li];         // launch path: tables exist and the chunk setting is unchanged -> no lock
        if (t0.ready.load(std::memory_order_acquire) == (G_.ntt_chunk + 1)) return CUHE_OK;
    }
    std::lock_guard<std::mutex> lk(G_.mu);
    NttTab &tab = G_.dev[dev].ntt[li];
    if (!tab.T1w) {                               // (8192 points: the one-workgroup form only, its tables are made by ensure_onewg)
        if (li == 1) CHK(make_ntt_tables<14>(tab));
        else if (li == 2) CHK(make_ntt_tables<15>(tab));
        else if (li == 3) CHK(make_ntt_tables<16>(tab));
    }
    // transforms per launch pair
    int chunk = G_.ntt_chunk > 0 ? G_.ntt_chunk : (256 << 20) / (len * 8);     // slab of 256 MiB: profiles/r01_chunk_sweep.txt
    if (chunk < 8) chunk = 8;
    tab.chunk = (chunk + 7) & ~7;
    DevCtx &D = G_.dev[dev];
    if (!D.s1) {
        int cur = 0;
        HIPCHK(hipGetDevice(&cur));
        HIPCHK(hipDeviceGetAttribute(&D.cus, hipDeviceAttributeMultiprocessorCount, cur));
        HIPCHK(hipStreamCreateWithFlags(&D.s1, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&D.s2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&D.ev_start, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipEventCreateWithFlags(&D.ev_p1[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&D.ev_p2[i], hipEventDisableTiming));
        }
    }
    tab.ready.store(G_.ntt_chunk + 1, std::memory_order_release);
    return CUHE_OK;
}

// twist tables of the negacyclic transform of length `len` (psi^j, psi^-j; psi^2 = w_len)
int ensure_twist(int dev, int len) {
    CHK(ensure_ntt(dev, len, 1));
    std::lock_guard<std::mutex> lk(G_.mu);
    NttTab &tab = G_.dev[dev].ntt[lg_index(len)];
    if (tab.tw) return CUHE_OK;
    const u64 psi = host::root_2len(len);
    if (!psi) return fail(CUHE_EINVAL, "no primitive %d-th root of unity found", 2 * len);
    const u64 ipsi = host::powP(psi, host::P - 2);
    std::vector<u64> tw(len), twi(len);
    u64 a = 1, b = 1;
    for (int j = 0; j < len; ++j) { tw[j] = a; twi[j] = b; a = host::mulP(a, psi); b = host::mulP(b, ipsi); }
    CHK(upload(&tab.tw, tw));
    CHK(upload(&tab.twinv, twi));
    return CUHE_OK;
}


// ---- one-workgroup transforms: tables (the index formulas are those of tests/onewg_model.py)
int ensure_onewg(OwTab &tab, int lgh) {
    if (tab.ready.load(std::memory_order_acquire)) return CUHE_OK;
    std::lock_guard<std::mutex> lk(G_.mu);
    if (tab.ready.load(std::memory_order_relaxed)) return CUHE_OK;
    const int Lh = 1 << lgh, T = Lh / 32, R = T / 32;
    const u64 W = host::powP(host::G, 65536 / (2 * Lh));          // w_(2 Lh); w_Lh = W^2 (cuhe/Base.cu:63-70)
    std::vector<u64> r(2 * (size_t)Lh);
    r[0] = 1;
    for (size_t i = 1; i < r.size(); ++i) r[i] = host::mulP(r[i - 1], W);
    const u64 linv = host::powP((u64)Lh, host::P - 2);
    std::vector<u64> f(Lh), fi(Lh), fh(2 * (size_t)Lh), t2(T);
    for (int ka = 0; ka < 32; ++ka)
        for (int m = 0; m < T; ++m) {
            const size_t o = (size_t)ka * T + m;
            f[o] = r[(2L * m * ka) % (2L * Lh)];                 // w_Lh^(m ka)
            fi[o] = host::mulP(f[o], linv);
            fh[o] = f[o];
            fh[Lh + o] = r[((long)m * (2 * ka + 1)) % (2L * Lh)]; // W^(m (2 ka + 1)): the odd outputs of the zero-padded transform
        }
    for (int kb = 0; kb < R; ++kb)
        for (int c = 0; c < 32; ++c) t2[(size_t)kb * 32 + c] = r[(64L * c * kb) % (2L * Lh)];      // w_T^(c kb) = w_Lh^(32 c kb)
    CHK(upload(&tab.TW1f, f)); CHK(upload(&tab.TW1i, fi)); CHK(upload(&tab.TW1h, fh)); CHK(upload(&tab.TW2, t2));
    tab.ready.store(1, std::memory_order_release);
    return CUHE_OK;
}
// tables of the two 32K-point halves of the NEGACYCLIC forward transform of 64K points (ntt_onewg.cuh: StreamTwist):
// TW1g[h][ka 1024 + m] = psi^(m (1 + 2h + 4 ka)), c128 = psi^1024, i4 = psi^32768 = +-2^48; psi = root_2len(65536)
int ensure_onewg_twist64(OwTab &tab) {
    CHK(ensure_onewg(tab, 15));
    std::lock_guard<std::mutex> lk(G_.mu);
    if (tab.TW1g) return CUHE_OK;
    const u64 psi = host::root_2len(65536);
    if (!psi) return fail(CUHE_EINVAL, "no primitive 2^17-th root of unity found");
    std::vector<u64> r((size_t)1 << 17);
    r[0] = 1;
    for (size_t i = 1; i < r.size(); ++i) r[i] = host::mulP(r[i - 1], psi);
    const u64 i4 = r[32768], p48 = (u64)1 << 48;
    if (i4 != p48 && i4 != host::P - p48) return fail(CUHE_EINVAL, "psi^32768 is not +-2^48");
    std::vector<u64> g(2 * (size_t)32768);
    for (int h = 0; h < 2; ++h)
        for (int ka = 0; ka < 32; ++ka)
            for (int m = 0; m < 1024; ++m) g[(size_t)h * 32768 + (size_t)ka * 1024 + m] = r[((long)m * (1 + 2 * h + 4 * ka)) & ((1 << 17) - 1)];
    CHK(upload(&tab.TW1g, g));
    tab.c128 = r[1024]; tab.i4neg = i4 == p48 ? 0 : 1;
    return CUHE_OK;
}
int onewg_launch(int lgh, int mode, int out, bool half, const OwArgs &a, hipStream_t st) {
    hipError_t e = lgh == 12 ? ow_launch_12(mode, out, half, a, st) : lgh == 13 ? ow_launch_13(mode, out, half, a, st)
                 : lgh == 14 ? ow_launch_14(mode, out, half, a, st) : ow_launch_15(mode, out, half, a, st);
    if (e != hipSuccess) return fail(CUHE_EHIP, "one-workgroup transform (2^%d points, source %d, store %d%s): %s", lgh, mode, out, half ? ", half" : "", hipGetErrorString(e));
    return CUHE_OK;
}

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
// rows per call up to which the low-latency kernels (4 values per thread, ntt_kernels.cuh) replace the throughput ones
// measured crossover (profiles/r02_small_batch_latency.txt): the low-latency pair wins up to ~24 rows of 32K points, ~12 rows of
// 64K points -- the threshold is in units of 32K-point rows and scales with the transform length
int g_ll_rows = getenv("CUHE_LL_ROWS") ? atoi(getenv("CUHE_LL_ROWS")) : 24;     // (environment override: A/B runs of whole programs)
template <int LG, int MODE>
int launch_pass1(const void *src, u64 *scratch, const NttTab &tab, long src_stride, int nb, WindowArgs wa, hipStream_t st, bool ll, const u64 *second = nullptr) {
    const u64 *tw = MODE == kSrcU64NegMul ? second : (const u64 *)tab.tw;
    if (MODE == kSrcU64NegMul && !second) return fail(CUHE_EINVAL, "second operand missing");
    if (MODE == kSrcU32Twist && !tab.tw) return fail(CUHE_EINVAL, "negacyclic twist table missing");
    if (ll) {
        using Gl = P1llGeom<LG>;
        const int grid = ((nb + 7) / 8) * 8 * (64 / Gl::CW);
        hipLaunchKernelGGL((ntt_pass1_ll<LG, MODE>), dim3(grid), dim3(Gl::T), Gl::bytes, st, src, scratch, (const u64 *)tab.Wn1, src_stride, nb, wa, tw);
        HIPCHK(hipGetLastError());
        return CUHE_OK;
    }
    using Gw = P1wGeom<LG>;
    static AttrOnce once;
    auto kern = ntt_pass1w<LG, MODE>;
    CHK(once.set(kern, (int)Gw::bytes));
    const int grid = ((nb + 7) / 8) * 8 * (64 / Gw::NC);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kP1wThreads), Gw::bytes, st, src, scratch, tab.T1w, src_stride, nb, wa, tw);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// store epilogue of an inverse transform beyond "mod p": kind 1 = reversed quotient, 2 = final subtraction of the folded
// reduction (aux = the product rows f, aux_stride their row length); see ntt_kernels.cuh
struct Epilogue { int kind = 0; const u32 *aux = nullptr; long aux_stride = 0; FoldGeom fg{0, 0, 0, 0, 0}; };
template <int LG, int OUT>
int launch_pass2(void *dst, const u64 *scratch, const NttTab &tab, long dst_stride, int nb, int nstore, const u32 *primes,
                 const u64 *pinv, int prime0, hipStream_t st, bool ll, int np_mod = 0, const Epilogue *ep = nullptr, const u64 *xtab = nullptr) {
    constexpr int N1 = (1 << LG) / 64;
    if ((OUT == kOutU64Mul || OUT == kOutModPNc) && !xtab) return fail(CUHE_EINVAL, "pass 2: table missing");
    const Epilogue none;
    const Epilogue &e = ep ? *ep : none;
    if (ll)
        hipLaunchKernelGGL((ntt_pass2_ll<LG, OUT>), dim3(((nb + 7) / 8) * 8 * (N1 / kP2llCols)), dim3(256), 0, st, dst, scratch,
                           out_is_inverse(OUT) ? tab.T2inv : tab.T2, dst_stride, nb, nstore, primes, pinv, prime0, np_mod,
                           e.aux, e.aux_stride, e.fg, xtab);
    else
        hipLaunchKernelGGL((ntt_pass2w<LG, OUT>), dim3(((nb + 7) / 8) * 8 * (N1 / kP2wCols)), dim3(256), kP2wLdsBytes, st, dst, scratch,
                           out_is_inverse(OUT) ? tab.T2inv : tab.T2, dst_stride, nb, nstore, primes, pinv, prime0, np_mod,
                           e.aux, e.aux_stride, e.fg, xtab);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

int icrt_lds_attr(size_t lds) {                 // k_icrt needs the large-LDS attribute for many primes
    static AttrOnce once;
    return lds > 64 * 1024 ? once.set(k_icrt, 160 * 1024) : CUHE_OK;
}

constexpr int kFoldXn1 = -1;     // nstore sentinel: inverse transform fused with the reduction mod x^(L/2)+1
constexpr int kNcInverse = -2;   // nstore sentinel: inverse NEGACYCLIC transform (untwist, centred lift, mod p), all L outputs

struct EvTimer {                 // optional per-pass hipEvent timing (bench)
    std::vector<hipEvent_t> ev;
    bool on = false;
};

// one batched transform, chunked so that the pass-1 -> pass-2 slab stays cache resident
template <int LG>
int run_ntt_lg(int mode, void *dst, const void *src, int batch, long src_stride, long dst_stride, int nstore,
               int prime0, WindowArgs wa, DevCtx &D, Workspace &W, hipStream_t st, EvTimer *tm, const u64 *mul_tab, int np_mod,
               const Epilogue *ep) {
    constexpr int L = 1 << LG;
    NttTab &tab = D.ntt[LG - 13];
    const int chunk = tab.chunk;
    // Two-stage software pipeline over chunks: pass 1 (VALU/LDS bound) of chunk c+1 runs on stream s1 while
    // pass 2 (load/store heavy, 1 wave/SIMD fits beside pass 1's 2) of chunk c runs on s2.
    // ---- the one-workgroup form wherever it exists: ONE launch for the whole batch, no slab (ntt_onewg.cuh).  A zero-padded
    // source (the reference contract) is done as the two half-length transforms of its even and odd outputs.
    {
        const bool half = src_is_ext(mode);
        const int lgh = half ? LG - 1 : LG;
        // worth it once the call's workgroups (1 / 2 / 4 fit a CU at 32K / 16K / 8K points) fill the chip; below that the
        // two-pass kernels spread a row over 8 - 16 workgroups and finish sooner
        const long wgs = (long)batch * (half ? 2 : 1);
        // (8192-point transforms exist in this form only: whatever the row count)
        const bool fills = LG == 13 || G_.onewg == 2 || (lgh >= 13 && lgh <= 15 && wgs >= (long)D.cus * (1 << (15 - lgh)));
        const bool rows64 = half && lgh == 15;
        const int grid64 = D.cus & ~15;
        const bool stream_ok = rows64 && mode == kSrcU32Ext && grid64 >= 16 && wgs >= 2L * grid64 && ((uintptr_t)src & 15) == 0 && (src_stride & 3) == 0;
        const bool rows64_onewg = G_.onewg64 == 1 || (G_.onewg64 == 2 && stream_ok);
        // negacyclic forward transform of full 64K-point rows (the ciphertext domain of x^65536 + 1): the persistent form, two
        // 32K-point halves per row meeting before their interleaved stores, from two halves per workgroup on (or forced)
        if (LG == 16 && mode == kSrcU32Twist && !mul_tab && G_.onewg && G_.onewg64 == 2 && grid64 >= 16 &&
            (G_.onewg == 2 || 2L * batch >= 2L * grid64) && ((uintptr_t)src & 15) == 0 && (src_stride & 3) == 0) {
            OwTab &ot = D.ow[3];
            CHK(ensure_onewg_twist64(ot));
            OwArgs a{dst, src, ot.TW1g, ot.TW2, src_stride, dst_stride, batch, nstore, wa, nullptr, D.p, D.pinv, prime0, np_mod, nullptr, 0, FoldGeom{0, 0, 0, 0, 0}, nullptr};
            if (tm && tm->on) for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
            if (!D.pair_cnt) HIPCHK(hipMalloc((void **)&D.pair_cnt, 128 * sizeof(unsigned)));
            hipError_t he = ow_launch_stream(kSrcU32Twist, kOutU64, a, grid64, D.pair_cnt, ot.c128, ot.i4neg, st);
            if (he != hipSuccess) return fail(CUHE_EHIP, "persistent one-workgroup transform (negacyclic rows): %s", hipGetErrorString(he));
            if (tm && tm->on) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
            return CUHE_OK;
        }
        if ((G_.onewg || LG == 13) && lgh <= 15 && fills && (!rows64 || rows64_onewg)) {
            int out, nst = nstore; const u64 *xt = nullptr; Epilogue e;
            if (mode == kSrcU64Neg || mode == kSrcU64NegMul) {
                if (ep && ep->kind) { out = ep->kind == 1 ? kOutModPRevQ : kOutFoldFinal; e = *ep; }
                else if (nstore == kNcInverse) { out = kOutModPNc; nst = L; xt = tab.twinv; }
                else if (nstore == kFoldXn1) { out = kOutModPFoldXn1; nst = L / 2; }
                else out = kOutModP;
            } else { out = mul_tab ? kOutU64Mul : kOutU64; xt = mul_tab; }
            if (ow_supported(mode, out, half)) {
                OwTab &ot = D.ow[lgh - 12];
                CHK(ensure_onewg(ot, lgh));
                const u64 *tw = mode == kSrcU64NegMul ? mul_tab : mode == kSrcU32Twist ? (const u64 *)tab.tw : nullptr;
                if (mode == kSrcU64NegMul && !tw) return fail(CUHE_EINVAL, "second operand missing");
                if (mode == kSrcU32Twist && !tw) return fail(CUHE_EINVAL, "negacyclic twist table missing");
                if (out == kOutModPNc && !xt) return fail(CUHE_EINVAL, "negacyclic untwist table missing");
                const bool inv = out_is_inverse(out);
                OwArgs a{dst, src, half ? ot.TW1h : inv ? ot.TW1i : ot.TW1f, ot.TW2, mode == kSrcWindow ? 0 : src_stride, dst_stride, batch, nst, wa, tw,
                         D.p, D.pinv, prime0, np_mod, e.aux, e.aux_stride, e.fg, xt};
                if (tm && tm->on) for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                // 64K-point rows: one workgroup per CU walks over its share of the halves, the next half's samples arriving by
                // LDS-DMA beside stage 3 of the current one (16-byte aligned rows, at least two halves per workgroup)
                const int grid = grid64;
                const bool stream = G_.onewg64 == 2 && stream_ok;
                if (stream) {
                    if (!D.pair_cnt) HIPCHK(hipMalloc((void **)&D.pair_cnt, 128 * sizeof(unsigned)));
                    hipError_t he = ow_launch_stream(kSrcU32Ext, out, a, grid, D.pair_cnt, 0, 0, st);
                    if (he != hipSuccess) return fail(CUHE_EHIP, "persistent one-workgroup transform: %s", hipGetErrorString(he));
                } else CHK(onewg_launch(lgh, mode, out, half, a, st));
                if (tm && tm->on) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                return CUHE_OK;
            }
        }
    }
    if constexpr (LG == 13) {
        return fail(CUHE_EINVAL, "8192-point transforms exist in the one-workgroup form only (source %d, %s)", mode, ep && ep->kind ? "folded-reduction store" : "plain store");
    } else {
    const bool pipe = G_.ntt_overlap && !(tm && tm->on) && batch > chunk;
    hipStream_t q1 = pipe ? D.s1 : st, q2 = pipe ? D.s2 : st;
    const size_t slab_bytes = (size_t)((std::min(chunk, batch) + 7) & ~7) * L * sizeof(u64);
    u64 *slabs[2] = {nullptr, nullptr};
    CHK(ws_slab(W, LG - 13, 0, slab_bytes, &slabs[0]));
    if (pipe) CHK(ws_slab(W, LG - 13, 1, slab_bytes, &slabs[1]));
    if (pipe) {
        HIPCHK(hipEventRecord(D.ev_start, st));
        HIPCHK(hipStreamWaitEvent(D.s1, D.ev_start, 0));
        HIPCHK(hipStreamWaitEvent(D.s2, D.ev_start, 0));
    }
    const bool ll = (long)batch * L <= (long)g_ll_rows * 32768;      // few rows: the duration of one workgroup is what counts
    // pass 2 alone keeps its low-latency form up to twice that size (profiles/r02_small_batch_latency.txt: 9.8 vs 11.2 us at 48 rows of 32K)
    const bool ll2 = (long)batch * L <= 2L * g_ll_rows * 32768;
    int c = 0, last = 0;
    for (int b0 = 0; b0 < batch; b0 += chunk, ++c) {
        const int nb = std::min(chunk, batch - b0);
        const int sl = pipe ? (c & 1) : 0;
        u64 *slab = slabs[sl];
        if (pipe && c >= 2) HIPCHK(hipStreamWaitEvent(q1, D.ev_p2[sl], 0));       // slab free again
        if (tm && tm->on) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, st); tm->ev.push_back(e); }
        if (mode == kSrcU32Ext) {
            const u32 *s = (const u32 *)src + (long)b0 * src_stride;
            CHK((launch_pass1<LG, kSrcU32Ext>(s, slab, tab, src_stride, nb, wa, q1, ll)));
        } else if (mode == kSrcU32Twist) {
            const u32 *s = (const u32 *)src + (long)b0 * src_stride;
            CHK((launch_pass1<LG, kSrcU32Twist>(s, slab, tab, src_stride, nb, wa, q1, ll)));
        } else if (mode == kSrcWindow) {
            WindowArgs w2 = wa; w2.wid0 += b0;
            CHK((launch_pass1<LG, kSrcWindow>(src, slab, tab, 0, nb, w2, q1, ll)));
        } else if (mode == kSrcU64NegMul) {                 // inverse transform of a product: the second operand rides in `mul_tab`
            const u64 *s = (const u64 *)src + (long)b0 * src_stride;
            CHK((launch_pass1<LG, kSrcU64NegMul>(s, slab, tab, src_stride, nb, wa, q1, ll, mul_tab ? mul_tab + (long)b0 * src_stride : nullptr)));
        } else {
            const u64 *s = (const u64 *)src + (long)b0 * src_stride;
            CHK((launch_pass1<LG, kSrcU64Neg>(s, slab, tab, src_stride, nb, wa, q1, ll)));
        }
        if (pipe) { HIPCHK(hipEventRecord(D.ev_p1[sl], q1)); HIPCHK(hipStreamWaitEvent(q2, D.ev_p1[sl], 0)); }
        if (tm && tm->on) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, st); tm->ev.push_back(e); }
        if (mode == kSrcU64Neg || mode == kSrcU64NegMul) {
            u32 *d = (u32 *)dst + (long)b0 * dst_stride;
            if (ep && ep->kind) {
                Epilogue e = *ep;
                if (e.aux) e.aux += (long)b0 * e.aux_stride;
                if (e.kind == 1) CHK((launch_pass2<LG, kOutModPRevQ>(d, slab, tab, dst_stride, nb, nstore, D.p, D.pinv, prime0 + b0, q2, ll2, np_mod, &e)));
                else CHK((launch_pass2<LG, kOutFoldFinal>(d, slab, tab, dst_stride, nb, nstore, D.p, D.pinv, prime0 + b0, q2, ll2, np_mod, &e)));
            } else if (nstore == kNcInverse) CHK((launch_pass2<LG, kOutModPNc>(d, slab, tab, dst_stride, nb, L, D.p, D.pinv, prime0 + b0, q2, ll2, np_mod, nullptr, tab.twinv)));
            else if (nstore == kFoldXn1) CHK((launch_pass2<LG, kOutModPFoldXn1>(d, slab, tab, dst_stride, nb, L / 2, D.p, D.pinv, prime0 + b0, q2, ll2, np_mod)));
            else CHK((launch_pass2<LG, kOutModP>(d, slab, tab, dst_stride, nb, nstore, D.p, D.pinv, prime0 + b0, q2, ll2, np_mod)));
        } else {
            u64 *d = (u64 *)dst + (long)b0 * dst_stride;
            if (mul_tab) CHK((launch_pass2<LG, kOutU64Mul>(d, slab, tab, dst_stride, nb, nstore, nullptr, nullptr, prime0 + b0, q2, ll2, np_mod, nullptr, mul_tab)));
            else CHK((launch_pass2<LG, kOutU64>(d, slab, tab, dst_stride, nb, nstore, nullptr, nullptr, 0, q2, ll2)));
        }
        if (pipe) { HIPCHK(hipEventRecord(D.ev_p2[sl], q2)); last = sl; }
        if (tm && tm->on) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, st); tm->ev.push_back(e); }
    }
    if (pipe) HIPCHK(hipStreamWaitEvent(st, D.ev_p2[last], 0));
    return CUHE_OK;
    }
}

int run_ntt(int len, int mode, void *dst, const void *src, int batch, long src_stride, long dst_stride, int nstore,
            int prime0, WindowArgs wa, int dev, hipStream_t st, EvTimer *tm = nullptr, const u64 *mul_tab = nullptr, int np_mod = 0,
            const Epilogue *ep = nullptr) {
    if (batch <= 0) return CUHE_OK;
    CHK(ensure_ntt(dev, len, batch));
    DevCtx &D = G_.dev[dev];
    Workspace *W = nullptr;
    CHK(workspace(dev, st, &W));
    switch (len) {
        case 8192:  return run_ntt_lg<13>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep);
        case 16384: return run_ntt_lg<14>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep);
        case 32768: return run_ntt_lg<15>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep);
        default:    return run_ntt_lg<16>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep);
    }
}

// ------------------------------------------------------------------ helpers
int need_init(int dev) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "cuhe_hip_init has not been called");
    CHK(set_dev(dev));
    if (!G_.dev[dev].ready) return fail(CUHE_ENOTINIT, "device %d not initialised", dev);
    return CUHE_OK;
}
PrimeTab prime_tab(const DevCtx &D) { return PrimeTab{D.p, D.pinv, D.e64, D.pow32, D.maxW}; }

// ICRT of `batch` ciphertexts of level lvl (np primes, W words)
int launch_icrt(u32 *dst, const u32 *src, const DevCtx &D, int lvl, int np, int W, int batch, long src_ct_stride, long dst_ct_stride, hipStream_t st,
                IcrtWindows wo = IcrtWindows{nullptr, 0, 0, 0, 0}) {
    const Params &q = G_.prm;
    const IcrtLevel &I = D.icrt[lvl];
    IcrtTab it{I.M, I.mi, I.bi, I.rp};
    const dim3 grid((q.modLen + kIcrtCoef - 1) / kIcrtCoef, batch), block(kIcrtCoef * kIcrtGroups);
    const size_t lds = icrt_lds_bytes(np, W);
    CHK(icrt_lds_attr(lds));
    hipLaunchKernelGGL(k_icrt, grid, block, lds, st, dst, src, prime_tab(D), it, np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride, wo);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

int level_of(int logq, int *lvl, int *np, int *W) {
    const Params &q = G_.prm;
    *lvl = q.getLevel(logq);
    if (*lvl >= q.depth) return fail(CUHE_EINVAL, "logq %d maps to level %d >= depth %d", logq, *lvl, q.depth);
    *np = q.numCrtPrimeAt(*lvl);
    *W = q.wordsCoeff(*lvl);
    return CUHE_OK;
}

// PrimeTab whose row 0 is prime `prime0` (CRT-prime-sharded calls address their own rows from 0)
PrimeTab prime_tab_at(const DevCtx &D, int prime0) {
    return PrimeTab{D.p + prime0, D.pinv + prime0, D.e64 + prime0, D.pow32 + (size_t)prime0 * D.maxW, D.maxW};
}

// reduction modulo the polynomial modulus of rows belonging to primes [prime0, prime0+np)
// np_mod > 0: `np` rows = several ciphertexts of the same np_mod primes (prime0 must be 0)
int barrett_impl(u32 *dst, const u32 *src, int prime0, int np, int dev, hipStream_t st, int np_mod) {
    const Params &q = G_.prm;
    DevCtx &D = G_.dev[dev];
    const int n = q.modLen, L = q.nttLen, cl = q.crtLen;
    PrimeTab pt = prime_tab_at(D, prime0);
    const int kind = G_.force_generic ? 0 : G_.reduce_kind;
    if (kind == 1) {
        hipLaunchKernelGGL((k_reduce_special<0>), dim3((cl + 255) / 256, np), dim3(256), 0, st, dst, src, pt, n, cl, L, np_mod);
        HIPCHK(hipGetLastError());
        return CUHE_OK;
    }
    if (kind == 2) {
        hipLaunchKernelGGL((k_reduce_special<1>), dim3((cl + 255) / 256, np), dim3(256), 0, st, dst, src, pt, n, cl, L, np_mod);
        HIPCHK(hipGetLastError());
        return CUHE_OK;
    }
    // generic: the algorithm of cuhe/Operations.cu:460-501 (q = ((f >> (n-1)) * u) >> n, r = f - q x^n - (m - x^n) q)
    // with every per-prime loop batched and its elementwise steps fused into their neighbours:
    //   * the two pointwise products by the precomputed NTT-domain constants (u, m - x^n) ride on the forward
    //     transforms' output (kOutU64Mul), instead of two more passes over u64[np][L];
    //   * f is only read (no working copy), q stays in b_crt while the last inverse transform writes to b_mq;
    //   * one kernel forms r[0..n) = f - (m - x^n) q, applies the reference's "coefficient n is non-zero -> subtract m
    //     once more" correction (barrett_sub_mc, Base.cu:978-1001) from r[n] = f[n] - q[0] - ((m - x^n) q)[n], and
    //     writes the crtLen-strided result.  The q x^n term only touches coefficients >= n, which are not output.
    // 11 launches per call (5 transform pairs + 1) instead of 18.
    const u64 *u_ntt = D.u_ntt + (size_t)prime0 * L, *m_ntt = D.m_ntt + (size_t)prime0 * L;
    const u32 *m_crt = D.m_crt + (size_t)prime0 * cl;
    WindowArgs wa{0, 0, 0};
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    CHK(ws_barrett(*Wp, np));
    Workspace &Ws = *Wp;
    if (np_mod > 0 && prime0 != 0) return fail(CUHE_EINVAL, "batched reduction needs a whole level");
    const size_t rows = (size_t)np * L;
    if (dst < src + rows && src < dst + (size_t)np * cl) {      // result rows would overwrite input rows still to be read
        CHK(ws_grow(&Ws.b_alias, &Ws.n_alias, rows));
        HIPCHK(hipMemcpyAsync(Ws.b_alias, src, rows * sizeof(u32), hipMemcpyDeviceToDevice, st));
        src = Ws.b_alias;
    }
    if (D.fold_ok && !G_.no_fold) {
        // Folded form: Phi_m divides x^m - 1, so f is first folded to g = f mod (x^m - 1) (length D = min(m, 2n-1)); the
        // quotient q = floor(g / Phi) then has only Kq = D - n coefficients and comes from the top Kq coefficients of g
        // (reversed) times the inverse series of rev(Phi), a product that fits the HALF-length transform; and since
        // r = g - q Phi has degree < n <= Lh it can be formed modulo x^Lh - 1, i.e. with a half-length cyclic product.
        // 4 half-length transforms + 1 elementwise kernel instead of 4 full-length transforms + 1: the quotient reversal
        // and the final subtraction are done in the stores of the two inverse transforms (kOutModPRevQ, kOutFoldFinal).
        const FoldGeom &Gf = D.fold;
        const int Lh = Gf.Lh, hl = Lh / 2;
        if (cl > Lh) return fail(CUHE_EINVAL, "folded reduction: crtLen %d > %d", cl, Lh);
        u32 *A = Ws.b_crt, *Q = Ws.b_mq;
        const dim3 gh((hl + 255) / 256, np);
        hipLaunchKernelGGL(k_fold_top_rev, gh, dim3(256), 0, st, A, src, pt, Gf, L, np_mod);
        CHK(run_ntt(Lh, kSrcU32Ext, Ws.b_ntt, A, np, hl, Lh, Lh, 0, wa, dev, st, nullptr, D.uh_ntt + (size_t)prime0 * Lh, np_mod));
        Epilogue rev; rev.kind = 1; rev.fg = Gf;
        CHK(run_ntt(Lh, kSrcU64Neg, Q, Ws.b_ntt, np, Lh, hl, hl, prime0, wa, dev, st, nullptr, nullptr, np_mod, &rev));   // q = rev(first Kq of A * U), zero padded
        CHK(run_ntt(Lh, kSrcU32Ext, Ws.b_ntt, Q, np, hl, Lh, Lh, 0, wa, dev, st, nullptr, D.mh_ntt + (size_t)prime0 * Lh, np_mod));
        Epilogue fin; fin.kind = 2; fin.aux = src; fin.aux_stride = L; fin.fg = Gf;
        CHK(run_ntt(Lh, kSrcU64Neg, dst, Ws.b_ntt, np, Lh, cl, cl, prime0, wa, dev, st, nullptr, nullptr, np_mod, &fin)); // g - q * Phi mod (x^Lh - 1)
        HIPCHK(hipGetLastError());
        return CUHE_OK;
    }
    CHK(run_ntt(L, kSrcU32Ext, Ws.b_ntt, src + (n - 1), np, L, L, L, 0, wa, dev, st, nullptr, u_ntt, np_mod));   // (f >> (n-1)) * u
    CHK(run_ntt(L, kSrcU64Neg, Ws.b_crt, Ws.b_ntt, np, L, L, L, prime0, wa, dev, st, nullptr, nullptr, np_mod));    // q at [n, 2n-1)
    CHK(run_ntt(L, kSrcU32Ext, Ws.b_ntt, Ws.b_crt + n, np, L, L, L, 0, wa, dev, st, nullptr, m_ntt, np_mod));     // q * (m - x^n)
    CHK(run_ntt(L, kSrcU64Neg, Ws.b_mq, Ws.b_ntt, np, L, L, L, prime0, wa, dev, st, nullptr, nullptr, np_mod));
    hipLaunchKernelGGL(k_barrett_final, dim3((cl + 255) / 256, np), dim3(256), 0, st, dst, src, Ws.b_crt, Ws.b_mq, m_crt, pt, n, cl, L, np_mod);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

// ------------------------------------------------------------------ ciphertext-domain ("ct") transforms
// The NTT representation ciphertext operations work in.  On general rings it is the reference's: cyclic transforms of
// nttLen = 2 modLen2 points of the zero-padded residues, products reduced modulo Phi_m afterwards (Operations.cu:394-504).
// When the polynomial modulus is x^n + 1 with n a transform length (16384 / 32768 / 65536) and the primes obey
// 2 n p^2 < P, it is the NEGACYCLIC transform of n points: half the points per polynomial, half the bytes per
// evaluation key, and products are already reduced modulo x^n + 1.  Results in the CRT / raw domain are identical.
inline int ct_len() { return G_.nc ? G_.prm.modLen : G_.prm.nttLen; }
int need_cyclic() {
    if (G_.prm.ncOnly()) return fail(CUHE_EINVAL, "ring degree %d has only the negacyclic representation (cuhe_hip_ct_*): the cyclic transforms of the reference stop at 65536 points", G_.prm.modLen);
    return CUHE_OK;
}
// CRT rows u32[rows][crtLen] -> ct rows u64[rows][ct_len]; mul_tab: rows the outputs are multiplied by on the way out
int ct_forward(u64 *X, const u32 *x, int rows, int dev, hipStream_t st, const u64 *mul_tab = nullptr, int np_mod = 0) {
    const Params &q = G_.prm;
    if (G_.nc) return run_ntt(q.modLen, kSrcU32Twist, X, x, rows, q.crtLen, q.modLen, q.modLen, 0, WindowArgs{0, 0, 0}, dev, st, nullptr, mul_tab, np_mod);
    return run_ntt(q.nttLen, kSrcU32Ext, X, x, rows, q.crtLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0}, dev, st, nullptr, mul_tab, np_mod);
}
// Phi_m = x^n + 1 with n = L/2 on the cyclic representation: INTT, mod p_i and the reduction in one pass-2 epilogue
bool fused_xn1() {
    return !G_.force_generic && G_.reduce_kind == 1 && G_.prm.modLen * 2 == G_.prm.nttLen && G_.prm.crtLen == G_.prm.modLen;
}
// ct rows -> CRT rows u32[rows][crtLen]; row r is reduced modulo prime prime0 + r (row r mod np_mod when np_mod > 0, prime0
// = 0 then); is_prod: the rows are products of two reduced polynomials (cyclic representation: reduce modulo Phi_m)
// Y != null: the rows are the pointwise products X * Y, multiplied as pass 1 loads them (kSrcU64NegMul)
int ct_inverse(u32 *dst, const u64 *X, int rows, int prime0, int np_mod, bool is_prod, int dev, hipStream_t st, const u64 *Y = nullptr) {
    const Params &q = G_.prm;
    const int n = q.modLen, L = q.nttLen, cl = q.crtLen;
    const WindowArgs wa{0, 0, 0};
    const int mode = Y ? kSrcU64NegMul : kSrcU64Neg;
    if (G_.nc) return run_ntt(n, mode, dst, X, rows, n, cl, kNcInverse, prime0, wa, dev, st, nullptr, Y, np_mod);
    if (!is_prod) return run_ntt(L, mode, dst, X, rows, L, cl, cl, prime0, wa, dev, st, nullptr, Y, np_mod);
    if (fused_xn1()) return run_ntt(L, mode, dst, X, rows, L, cl, kFoldXn1, prime0, wa, dev, st, nullptr, Y, np_mod);
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    CHK(ws_barrett(*Wp, rows));
    CHK(run_ntt(L, mode, Wp->hold, X, rows, L, L, L, prime0, wa, dev, st, nullptr, Y, np_mod));
    return barrett_impl(dst, Wp->hold, prime0, rows, dev, st, np_mod);
}

int init_device(int dev) {
    CHK(set_dev(dev));
    DevCtx &D = G_.dev[dev];
    const Params &q = G_.prm;
    const int pnum = q.numCrtPrime, L = q.nttLen, n = q.modLen, cl = q.crtLen;
    // ---- prime tables (preload_crt_p / preload_crt_invp: cuhe/Base.cu:145-160)
    // rows padded to a multiple of 8 words and kCrtPB zero rows appended: k_crt reads the table in unguarded blocks
    D.maxW = (q.wordsCoeff(0) + 1 + 7) & ~7;
    std::vector<u32> hp(G_.primes), he(pnum), hpow((size_t)(pnum + kCrtPB) * D.maxW, 0), hinv((size_t)pnum * (pnum - 1) / 2 + 1, 0);
    std::vector<u64> hpi(pnum);
    for (int i = 0; i < pnum; ++i) {
        const u32 p = hp[i];
        hpi[i] = (u64)(((host::u128)1 << 64) / p);
        he[i] = (u32)((((host::u128)1) << 64) % p);
        u64 c = 1 % p;
        for (int k = 0; k <= q.wordsCoeff(0); ++k) { hpow[(size_t)i * D.maxW + k] = (u32)c; c = (c << 32) % p; }
    }
    for (int i = 1; i < pnum; ++i)                                 // cuhe/Operations.cu:91-99
        for (int j = 0; j < i; ++j) hinv[(size_t)i * (i - 1) / 2 + j] = host::invmod32(hp[i] % hp[j], hp[j]);
    CHK(upload(&D.p, hp)); CHK(upload(&D.pinv, hpi)); CHK(upload(&D.e64, he));
    CHK(upload(&D.pow32, hpow)); CHK(upload(&D.invp, hinv));
    // ---- ICRT constants for every level, all resident (cuhe/Operations.cu:107-156)
    D.icrt.resize(q.depth);
    for (int lvl = 0; lvl < q.depth; ++lvl) {
        IcrtLevel &I = D.icrt[lvl];
        I.np = pnum - lvl; I.W = q.wordsCoeff(lvl);
        const BigU &M = G_.coeffModulus[lvl];
        const int W4 = (I.W + 3) & ~3, np8 = (I.np + 7) & ~7;        // padded for k_icrt's unguarded scalar blocks
        std::vector<u32> hM(I.W), hmi((size_t)np8 * W4, 0), hbi(I.np);
        std::vector<double> hrp(I.np);
        M.to_words(hM.data(), I.W);
        for (int i = 0; i < I.np; ++i) {
            BigU mi = M.div_small(hp[i]);
            mi.to_words(&hmi[(size_t)i * W4], I.W);
            hbi[i] = host::invmod32(mi.mod_small(hp[i]), hp[i]);
            hrp[i] = 1.0 / (double)hp[i];
        }
        CHK(upload(&I.M, hM)); CHK(upload(&I.mi, hmi)); CHK(upload(&I.bi, hbi)); CHK(upload(&I.rp, hrp));
    }
    // ---- transforms + scratch (initNtt: cuhe/Operations.cu:173-184)
    CHK(ensure_ntt(dev, L, pnum));
    if (G_.nc) CHK(ensure_twist(dev, n));
    if (q.ncOnly()) {                    // no cyclic representation, hence no Barrett tables (the ring has x^n + 1 only)
        HIPCHK(hipDeviceSynchronize());
        D.ready = true;
        return CUHE_OK;
    }
    // ---- Barrett (initBarrett: cuhe/Operations.cu:196-238)
    HIPCHK(hipMalloc((void **)&D.u_ntt, (size_t)pnum * L * sizeof(u64)));
    HIPCHK(hipMalloc((void **)&D.m_ntt, (size_t)pnum * L * sizeof(u64)));
    std::vector<long long> u;
    if (!host::barrett_u(G_.modulus, u)) return fail(CUHE_EINVAL, "polynomial modulus has unbounded Barrett quotient");
    std::vector<u32> hu((size_t)pnum * cl, 0), hm((size_t)pnum * cl, 0);
    for (int i = 0; i < pnum; ++i)
        for (int k = 0; k < n; ++k) {
            hu[(size_t)i * cl + k] = host::smod(u[k], hp[i]);
            hm[(size_t)i * cl + k] = host::smod(G_.modulus[k], hp[i]);   // m - x^n: coefficient n dropped
        }
    CHK(upload(&D.m_crt, hm));
    u32 *tmp = nullptr;
    CHK(upload(&tmp, hu));
    WindowArgs wa{0, 0, 0};
    // folded form of the generic reduction (barrett_impl): applicable when the half-length transform exists
    // (Lh >= 16384) and the quotient fits its half-length input
    {
        const int m = q.mSize, Lh = L / 2, Dg = (m < 2 * n - 1) ? m : 2 * n - 1, Kq = Dg - n;
        D.fold_ok = G_.reduce_kind == 0 && lg_index(Lh) >= 0 && Kq >= 1 && Kq <= Lh / 2 && Kq <= n - 1 && n <= Lh;
        if (D.fold_ok) {
            D.fold = FoldGeom{n, m, Dg, Kq, Lh};
            std::vector<u64> huh((size_t)pnum * Lh), hmh((size_t)pnum * Lh);
            std::vector<uint64_t> a(Lh);
            for (int i = 0; i < pnum; ++i) {
                std::fill(a.begin(), a.end(), 0);
                for (int j = 0; j < Kq; ++j) a[j] = host::smod(u[n - 1 - j], hp[i]);          // inverse series of rev(Phi), Kq terms
                host::ntt_host(a, Lh);
                for (int t = 0; t < Lh; ++t) huh[(size_t)i * Lh + t] = (u64)a[t];
                std::fill(a.begin(), a.end(), 0);
                for (int k = 0; k <= n; ++k) {                                                // Phi mod (x^Lh - 1)
                    const uint32_t c = k < n ? host::smod(G_.modulus[k], hp[i]) : 1u;
                    a[k % Lh] = (a[k % Lh] + c) % hp[i];
                }
                host::ntt_host(a, Lh);
                for (int t = 0; t < Lh; ++t) hmh[(size_t)i * Lh + t] = (u64)a[t];
            }
            CHK(upload(&D.uh_ntt, huh)); CHK(upload(&D.mh_ntt, hmh));
            CHK(ensure_ntt(dev, Lh, pnum));
        }
    }
    CHK(run_ntt(L, kSrcU32Ext, D.u_ntt, tmp, pnum, cl, L, L, 0, wa, dev, 0));
    CHK(run_ntt(L, kSrcU32Ext, D.m_ntt, D.m_crt, pnum, cl, L, L, 0, wa, dev, 0));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(tmp));
    D.ready = true;
    return CUHE_OK;
}

template <int OP>
__global__ void k_modp_test(u64 *z, const u64 *x, const u64 *y, int l, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 a = x[i];
    if (OP == 0) z[i] = addp(canon(a), canon(y[i]));
    else if (OP == 1) z[i] = subp(canon(a), canon(y[i]));
    else if (OP == 2) z[i] = mulp(canon(a), canon(y[i]));
    else {
        // runtime shift amount: square-and-multiply on 2 (test hook only; kernels use compile-time shifts)
        u64 r = canon(a), b = 2; int e = l % 192;
        while (e) { if (e & 1) r = mulp(r, b); b = mulp(b, b); e >>= 1; }
        z[i] = r;
    }
}
template <int K>
__global__ void k_shl_const(u64 *z, const u64 *x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = (K >= 96) ? negp(shlp<(K >= 96 ? K - 96 : K)>(canon(x[i]))) : shlp<(K >= 96 ? K - 96 : K)>(canon(x[i]));
}

template <int K>
void shl_dispatch(int l, uint64_t *z, const uint64_t *x, size_t n, hipStream_t st, bool &done) {
    if constexpr (K < 192) {
        if (l == K) { hipLaunchKernelGGL((k_shl_const<K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u64 *)z, (const u64 *)x, n); done = true; }
        else shl_dispatch<K + 3>(l, z, x, n, st, done);
    }
}

// ---- key-switch inner product on the matrix cores: key digits, launch
// at most two K steps are instantiated, and the window tile of a workgroup (8 columns x 16 ciphertexts x max(keys, padded
// primes) words) has to fit the LDS: beyond that (levels with more than 144 primes) the VALU kernels serve
bool mac_mfma_supported(int K, int k_lvl = 0, int np_lvl = 0) {
    const int JS = (k_lvl + 2) & ~1, NPAD = ((np_lvl + 15) / 16) * 16;
    const size_t lds = (size_t)kMacMfmaCols * (kMacMfmaCts * std::max(JS, NPAD) + 1) * sizeof(u64);
    return K >= 1 && K <= 128 && lds <= 160 * 1024;
}
int need_all_keys(const DevCtx &D) {
    if (D.ek_first != 0 || D.ek_count != G_.prm.numCrtPrime)
        return fail(CUHE_EINVAL, "this device holds the keys of primes [%d, %d) only (cuhe_hip_init_relin_range): the call needs all %d", D.ek_first, D.ek_first + D.ek_count, G_.prm.numCrtPrime);
    return CUHE_OK;
}
int ensure_key_digits(int dev, hipStream_t st) {
    DevCtx &D = G_.dev[dev];
    CHK(need_all_keys(D));
    std::lock_guard<std::mutex> lk(G_.mu);
    if (D.ekd) return CUHE_OK;
    const Params &q = G_.prm;
    const int K = q.numEvalKey, np = q.numCrtPrime, L = ct_len();
    MacDigGeom g;
    g.nfull = K / 64;
    const int r = K % 64;
    g.tail = r == 0 ? 0 : r <= 32 ? 32 : 64;
    g.tail_groups = g.tail == 32 ? (r + 7) / 8 : g.tail == 64 ? (r + 15) / 16 : 0;
    g.lb_bytes = g.nfull * 1024 + g.tail_groups * (g.tail == 32 ? 128 : 256);
    g.npt = (np + 15) / 16;
    const size_t bytes = (size_t)L * g.npt * 8 * g.lb_bytes;
    if (D.ekd_unavailable) return CUHE_OK;
    if (hipMalloc((void **)&D.ekd, bytes) != hipSuccess) {       // no room for a second copy of the keys: the VALU kernel serves
        (void)hipGetLastError();
        D.ekd = nullptr; D.ekd_unavailable = true;
        return CUHE_OK;
    }
    D.ekg = g;
    hipLaunchKernelGGL(k_ek_digits, dim3((L + 255) / 256, g.npt * 16, g.nfull * 4 + g.tail_groups), dim3(256), 0, st,
                       D.ekd, (const u64 *)D.ek, K, np, L, (long)K * L, g);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));            // other host threads' streams may use the digits right after the lock is released
    return CUHE_OK;
}
template <int NFULL, int TAIL>
int launch_mac_mfma(u64 *dst, const u64 *c, const DevCtx &D, int k, int L, int np, long c_ct_stride, long dst_ct_stride, int ncts, hipStream_t st) {
    static AttrOnce once;
    const int JS = (k + 2) & ~1, NPAD = ((np + 15) / 16) * 16;
    const size_t lds = (size_t)kMacMfmaCols * (kMacMfmaCts * std::max(JS, NPAD) + 1) * sizeof(u64);
    if (lds > 160 * 1024) return fail(CUHE_EINVAL, "window tile of %zu bytes", lds);
    CHK(once.set(k_relin_mac_mfma<NFULL, TAIL>, 160 * 1024));
    const int ngroups = (ncts + kMacMfmaCts - 1) / kMacMfmaCts;
    hipLaunchKernelGGL((k_relin_mac_mfma<NFULL, TAIL>), dim3((L / kMacMfmaCols) * ngroups), dim3(kMacMfmaThreads), lds, st,
                       dst, c, (const unsigned char *)D.ekd, k, L, np, c_ct_stride, dst_ct_stride, ncts, D.ekg);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int run_mac_mfma(u64 *dst, const u64 *c, const DevCtx &D, int k, int L, int np, long c_ct_stride, long dst_ct_stride, int ncts, hipStream_t st) {
    const int key = D.ekg.nfull * 100 + D.ekg.tail;
    switch (key) {
    case 32: return launch_mac_mfma<0, 32>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    case 64: return launch_mac_mfma<0, 64>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    case 100: return launch_mac_mfma<1, 0>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    case 132: return launch_mac_mfma<1, 32>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    case 164: return launch_mac_mfma<1, 64>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    case 200: return launch_mac_mfma<2, 0>(dst, c, D, k, L, np, c_ct_stride, dst_ct_stride, ncts, st);
    }
    return fail(CUHE_EINVAL, "no matrix-core inner product for %d evaluation keys", G_.prm.numEvalKey);
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

const char *cuhe_hip_last_error(void) { return g_err.c_str(); }
const char *cuhe_hip_version(void) { return "cuhe_amd 0.1 (gfx950)"; }

int cuhe_hip_set_parameters(int d, int p, int w, int min, int cut, int m) {
    if (d < 1 || p < 2 || w < 0 || w > 31 || min < 1 || cut < 1 || m < 3)
        return fail(CUHE_EINVAL, "setParameters(%d,%d,%d,%d,%d,%d): invalid", d, p, w, min, cut, m);
    G_.prm.set(d, p, w, min, cut, m);
    if (lg_index(G_.prm.nttLen) < 0)
        return fail(CUHE_EINVAL, "ring degree %d needs nttLen %d (supported: 16384/32768/65536; degree 65536 only as m = 131072, x^65536 + 1)", G_.prm.modLen, G_.prm.nttLen);
    if (G_.prm.numCrtPrime > 103 * 4) return fail(CUHE_EINVAL, "too many CRT primes (%d)", G_.prm.numCrtPrime);
    G_.params_set = true;
    return CUHE_OK;
}
int cuhe_hip_reset_parameters(void) { G_.prm = Params(); G_.params_set = false; return CUHE_OK; }
int cuhe_hip_get_parameters(cuhe_params_t *o) {
    if (!o) return fail(CUHE_EINVAL, "null");
    const Params &q = G_.prm;
    *o = cuhe_params_t{q.mSize, q.modLen, q.modLen2, q.rawLen, q.crtLen, q.nttLen, q.logCoeffMax, q.logCoeffMin,
                       q.logCoeffCut, q.depth, q.modMsg, q.logMsg, q.wordsMsg, q.logRelin, q.numEvalKey,
                       q.logCrtPrime, q.numCrtPrime};
    return CUHE_OK;
}
int cuhe_hip_num_crt_prime(int lvl) { return G_.prm.numCrtPrimeAt(lvl); }
int cuhe_hip_log_coeff(int lvl) { return G_.prm.logCoeff(lvl); }
int cuhe_hip_words_coeff(int lvl) { return G_.prm.wordsCoeff(lvl); }
int cuhe_hip_num_eval_key(int lvl) { return G_.prm.numEvalKeyAt(lvl); }
int cuhe_hip_get_level(int logq) { return G_.prm.getLevel(logq); }

int cuhe_hip_multi_gpus(int num) {
    int cnt = 0;
    HIPCHK(hipGetDeviceCount(&cnt));
    if (num < 1 || (!G_.virtual_devices && G_.dev_base + num > cnt)) return fail(CUHE_EINVAL, "multiGPUs(%d): %d device(s) visible", num, cnt);
    if (G_.inited) return fail(CUHE_EINVAL, "multiGPUs must precede initCuHE (cuhe/DeviceManager.cu:38-41)");
    G_.ndev = num;
    G_.dev.resize(num);
    return CUHE_OK;
}
int cuhe_hip_num_gpus(void) { return G_.ndev; }
// test hook: with `on`, multi_gpus(n) accepts any n and every logical device is backed by the one physical device,
// each with its own context (tables, keys, allocator, workspaces) -- the in-process multi-device code paths
// (per-device indexing, moveTo / copyTo) can then be exercised on a single-GPU box
int cuhe_hip_set_virtual_devices(int on) {
    if (G_.inited) return fail(CUHE_EINVAL, "set_virtual_devices must precede init");
    G_.virtual_devices = on != 0;
    return CUHE_OK;
}
int cuhe_hip_set_device_base(int dev) {
    if (G_.inited) return fail(CUHE_EINVAL, "set_device_base must precede init");
    G_.dev_base = dev;
    return CUHE_OK;
}

int cuhe_hip_init(const int32_t *modulus, int ncoeffs) {
    if (!G_.params_set) return fail(CUHE_EINVAL, "setParameters must precede initCuHE");
    if (G_.inited) return fail(CUHE_EINVAL, "already initialised");
    const Params &q = G_.prm;
    if (modulus) {
        if (ncoeffs != q.modLen + 1 || modulus[q.modLen] != 1)
            return fail(CUHE_EINVAL, "modulus must be monic of degree modLen=%d", q.modLen);
        G_.modulus.assign(modulus, modulus + ncoeffs);
    } else {
        G_.modulus = host::cyclotomic(q.mSize);
        if ((int)G_.modulus.size() != q.modLen + 1) return fail(CUHE_EINVAL, "cyclotomic(%d) degree mismatch", q.mSize);
    }
    // which exact reduction applies
    {
        const int n = q.modLen;
        bool xn1 = (G_.modulus[0] == 1), ones = true;
        for (int i = 1; i < n; ++i) { if (G_.modulus[i] != 0) xn1 = false; }
        for (int i = 0; i <= n; ++i) { if (G_.modulus[i] != 1) ones = false; }
        G_.reduce_kind = xn1 ? 1 : (ones ? 2 : 0);
    }
    G_.primes = host::gen_crt_primes(q);                           // cuhe/Operations.cu:37-80
    // negacyclic ciphertext domain: modulus x^n + 1, n a transform length, and the centred lift must be unambiguous:
    // a product coefficient is a signed sum of n terms below p^2 (2 n p^2 < P), a key-switch sum one of k n terms below 2^w p
    {
        const int n = q.modLen;
        host::u128 pmax = 0;
        for (uint32_t p : G_.primes) pmax = std::max<host::u128>(pmax, p);
        const bool shape = G_.reduce_kind == 1 && lg_index(n) >= 0 && q.crtLen == n;
        const bool bound = 2 * (host::u128)n * (pmax - 1) * (pmax - 1) < host::P &&
                           (!q.logRelin || 2 * (host::u128)q.numEvalKey * n * (((host::u128)1 << q.logRelin) - 1) * (pmax - 1) < host::P);
        G_.nc = G_.nc_mode != 0 && shape && bound;
        if (q.ncOnly() && !G_.nc)
            return fail(CUHE_EINVAL, "ring degree %d needs the negacyclic representation: modulus x^n + 1, primes with 2 n p^2 < P%s", n,
                        G_.nc_mode == 0 ? " (and cuhe_hip_set_negacyclic(0) is in effect)" : "");
    }
    G_.coeffModulus.assign(q.depth, BigU(1));                      // cuhe/Operations.cu:81-90
    for (int i = 0; i < q.depth; ++i)
        for (int j = 0; j < q.numCrtPrime - i; ++j) G_.coeffModulus[i].mul_small(G_.primes[j]);
    G_.dev.resize(G_.ndev);
    G_.inited = true;
    for (int dev = 0; dev < G_.ndev; ++dev) {
        int r = init_device(dev);
        if (r != CUHE_OK) { G_.inited = false; return r; }
    }
    for (int i = 0; i < G_.ndev && !G_.virtual_devices; ++i) {      // cuhe/CuHE.cu:42-45 peer access
        hipSetDevice(G_.dev_base + i);
        for (int j = 0; j < G_.ndev; ++j)
            if (i != j) { int can = 0; hipDeviceCanAccessPeer(&can, G_.dev_base + i, G_.dev_base + j);
                          if (can) hipDeviceEnablePeerAccess(G_.dev_base + j, 0); }
    }
    (void)hipGetLastError();
    return CUHE_OK;
}

int cuhe_hip_is_initialised(void) { return G_.inited ? 1 : 0; }
int cuhe_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(G_.mu);
    for (int d = 0; d < (int)G_.dev.size(); ++d) {
        if (hipSetDevice(phys_dev(d)) != hipSuccess) { (void)hipGetLastError(); continue; }
        (void)hipDeviceSynchronize();
        DevCtx &D = G_.dev[d];
        for (auto &t : D.ntt) { hipFree(t.T1w); hipFree(t.T2); hipFree(t.T2inv); hipFree(t.tw); hipFree(t.twinv); hipFree(t.Wn1); t = NttTab(); }
        if (D.s1) { hipStreamDestroy(D.s1); hipStreamDestroy(D.s2); hipEventDestroy(D.ev_start); for (int i = 0; i < 2; ++i) { hipEventDestroy(D.ev_p1[i]); hipEventDestroy(D.ev_p2[i]); } }
        if (D.sh_stream) { hipStreamDestroy(D.sh_stream); hipEventDestroy(D.sh_e1); hipEventDestroy(D.sh_e2); }
        void *ptrs[] = {D.p, D.e64, D.pow32, D.invp, D.pinv, D.u_ntt, D.m_ntt, D.uh_ntt, D.mh_ntt, D.m_crt, D.ek, D.ekd};
        for (Workspace *w : D.spaces) free_workspace(w);
        for (void *p : ptrs) if (p) hipFree(p);
        for (auto &I : D.icrt) { hipFree(I.M); hipFree(I.mi); hipFree(I.bi); hipFree(I.rp); }
        for (auto &kv : D.freeBlocks) hipFree(kv.second);
        for (auto &sb : D.streamBlocks) for (auto &kv : sb.second) hipFree(kv.second);     // parked in stream order
        for (auto &kv : D.allocated) hipFree(kv.first);
        D = DevCtx();
    }
    G_.inited = false; G_.relin_ready = false; G_.allocator_on = false;
    ++G_.generation;
    return CUHE_OK;
}

int cuhe_hip_get_coeff_modulus(int lvl, uint8_t *le, size_t cap, size_t *len) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "not initialised");
    if (lvl < 0 || lvl >= G_.prm.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    const BigU &M = G_.coeffModulus[lvl];
    size_t nb = M.w.size() * 4;
    if (len) *len = nb;
    if (cap < nb) return fail(CUHE_EINVAL, "buffer too small");
    memcpy(le, M.w.data(), nb);
    return CUHE_OK;
}
int cuhe_hip_get_crt_primes(uint32_t *out, int cap) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "not initialised");
    if (cap < (int)G_.primes.size()) return fail(CUHE_EINVAL, "buffer too small");
    memcpy(out, G_.primes.data(), G_.primes.size() * 4);
    return CUHE_OK;
}
int cuhe_hip_reduce_kind(void) { return G_.force_generic ? 0 : G_.reduce_kind; }
int cuhe_hip_force_generic_reduce(int on) { G_.force_generic = on != 0; G_.no_fold = on == 2; return CUHE_OK; }

// ---------------------------------------------------------------- allocator
int cuhe_hip_start_allocator(void) { G_.allocator_on = true; return CUHE_OK; }   // no "grab all VRAM" 
static void drop_cached(DevCtx &D) {
    for (auto &kv : D.freeBlocks) hipFree(kv.second);            // (hipFree waits for the device: in-flight users are safe)
    D.freeBlocks.clear();
    for (auto &sb : D.streamBlocks) for (auto &kv : sb.second) hipFree(kv.second);
    D.streamBlocks.clear();
    D.cachedBytes = 0;
}
// blocks freed in stream order become ordinary free blocks once that stream has been synchronised
static void settle_stream_blocks(DevCtx &D, hipStream_t st) {
    std::lock_guard<std::mutex> lk(G_.mu);
    auto it = D.streamBlocks.find(st);
    if (it == D.streamBlocks.end()) return;
    for (auto &kv : it->second) D.freeBlocks.insert(kv);
    D.streamBlocks.erase(it);
}
int cuhe_hip_stop_allocator(void) {
    G_.allocator_on = false;
    std::lock_guard<std::mutex> lk(G_.mu);
    for (int d = 0; d < (int)G_.dev.size(); ++d) {
        if (hipSetDevice(phys_dev(d)) != hipSuccess) { (void)hipGetLastError(); continue; }
        drop_cached(G_.dev[d]);
    }
    return CUHE_OK;
}
int cuhe_hip_set_alloc_cache(size_t bytes) { G_.cache_cap = bytes; return CUHE_OK; }
// Size-keyed block cache.  hipMalloc/hipFree cost tens to hundreds of microseconds and hipFree synchronises the
// device, which is more than a whole CRT or NTT stage of a ciphertext takes, and the API allocates and frees a
// representation on every domain change (cuhe/CuHE.cu:356-408).  Freed blocks are therefore parked and handed out
// again for the same size: without limit while startAllocator() is in effect, up to cache_cap bytes otherwise.
void *cuhe_hip_malloc(int dev, size_t bytes) {
    if (set_dev(dev) != CUHE_OK) return nullptr;
    DevCtx &D = G_.dev[dev];
    std::lock_guard<std::mutex> lk(G_.mu);
    auto it = D.freeBlocks.find(bytes);
    if (it != D.freeBlocks.end()) {
        void *p = it->second;
        D.freeBlocks.erase(it); D.cachedBytes -= bytes; D.allocated[p] = bytes;
        return p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        drop_cached(D);                                   // give the parked blocks back and try once more
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(CUHE_EHIP, "hipMalloc(%zu) failed", bytes); return nullptr; }
    }
    D.allocated[p] = bytes;
    return p;
}
int cuhe_hip_free(int dev, void *ptr) {
    if (!ptr) return CUHE_OK;
    CHK(set_dev(dev));
    DevCtx &D = G_.dev[dev];
    std::lock_guard<std::mutex> lk(G_.mu);
    auto it = D.allocated.find(ptr);
    if (it == D.allocated.end()) return fail(CUHE_EINVAL, "free of unknown pointer");
    const size_t sz = it->second;
    D.allocated.erase(it);
    if (G_.allocator_on || D.cachedBytes + sz <= G_.cache_cap) { D.freeBlocks.insert({sz, ptr}); D.cachedBytes += sz; }
    else HIPCHK(hipFree(ptr));
    return CUHE_OK;
}
// Stream-ordered variants: a block freed with free_stream may still be in use by work already enqueued on `st`, so it
// is handed out again only to allocations made for the SAME stream (which run after that work) until the stream has
// been synchronised through cuhe_hip_stream_sync.  This is what lets a caller enqueue a whole chain of ciphertext
// operations without a host synchronisation after each (the C++ layer's setAsynchronous(true)).
void *cuhe_hip_malloc_stream(int dev, size_t bytes, void *st) {
    if (set_dev(dev) != CUHE_OK) return nullptr;
    DevCtx &D = G_.dev[dev];
    {
        std::lock_guard<std::mutex> lk(G_.mu);
        auto sb = D.streamBlocks.find(S(st));
        if (sb != D.streamBlocks.end()) {
            auto it = sb->second.find(bytes);
            if (it != sb->second.end()) {
                void *p = it->second;
                sb->second.erase(it); D.cachedBytes -= bytes; D.allocated[p] = bytes;
                return p;
            }
        }
    }
    return cuhe_hip_malloc(dev, bytes);
}
int cuhe_hip_free_stream(int dev, void *ptr, void *st) {
    if (!ptr) return CUHE_OK;
    CHK(set_dev(dev));
    DevCtx &D = G_.dev[dev];
    std::lock_guard<std::mutex> lk(G_.mu);
    auto it = D.allocated.find(ptr);
    if (it == D.allocated.end()) return fail(CUHE_EINVAL, "free of unknown pointer");
    const size_t sz = it->second;
    D.allocated.erase(it);
    if (G_.allocator_on || D.cachedBytes + sz <= G_.cache_cap) { D.streamBlocks[S(st)].insert({sz, ptr}); D.cachedBytes += sz; }
    else HIPCHK(hipFree(ptr));
    return CUHE_OK;
}
// pinned host staging memory for the ZZX <-> raw conversions of the C++ layer (cuhe/CuHE.cu:317-348 uses pageable)
void *cuhe_hip_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(CUHE_EHIP, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
int cuhe_hip_host_free(void *ptr) { if (ptr) HIPCHK(hipHostFree(ptr)); return CUHE_OK; }
int cuhe_hip_memset_async(int dev, void *p, int v, size_t n, void *st) { CHK(set_dev(dev)); HIPCHK(hipMemsetAsync(p, v, n, S(st))); return CUHE_OK; }
int cuhe_hip_memcpy_h2d(int dev, void *d, const void *s, size_t n, void *st) { CHK(set_dev(dev)); HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, S(st))); return CUHE_OK; }
int cuhe_hip_memcpy_d2h(int dev, void *d, const void *s, size_t n, void *st) { CHK(set_dev(dev)); HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, S(st))); return CUHE_OK; }
int cuhe_hip_memcpy_d2d(int dev, void *d, const void *s, size_t n, void *st) { CHK(set_dev(dev)); HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, S(st))); return CUHE_OK; }
int cuhe_hip_memcpy_peer(void *d, int dd, const void *s, int sd, size_t n, void *st) {
    CHK(set_dev(sd));
    if (dd < 0 || dd >= G_.ndev) return fail(CUHE_EINVAL, "device %d out of range (numGPUs=%d)", dd, G_.ndev);
    if (G_.virtual_devices) HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, S(st)));
    else HIPCHK(hipMemcpyPeerAsync(d, G_.dev_base + dd, s, G_.dev_base + sd, n, S(st)));
    return CUHE_OK;
}
int cuhe_hip_stream_create(int dev, void **out) {
    CHK(set_dev(dev));
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void *)s;
    return CUHE_OK;
}
int cuhe_hip_stream_destroy(int dev, void *st) {
    CHK(set_dev(dev));
    if (st) { HIPCHK(hipStreamSynchronize(S(st))); settle_stream_blocks(G_.dev[dev], S(st)); HIPCHK(hipStreamDestroy(S(st))); }
    return CUHE_OK;
}
int cuhe_hip_stream_sync(int dev, void *st) {
    CHK(set_dev(dev));
    HIPCHK(hipStreamSynchronize(S(st)));
    settle_stream_blocks(G_.dev[dev], S(st));
    return CUHE_OK;
}

// waits for everything enqueued on the device; every block freed in stream order becomes an ordinary free block
int cuhe_hip_device_sync(int dev) {
    CHK(set_dev(dev));
    HIPCHK(hipDeviceSynchronize());
    DevCtx &D = G_.dev[dev];
    std::lock_guard<std::mutex> lk(G_.mu);
    for (auto &sb : D.streamBlocks) for (auto &kv : sb.second) D.freeBlocks.insert(kv);
    D.streamBlocks.clear();
    return CUHE_OK;
}

// ---------------------------------------------------------------- drivers
int cuhe_hip_crt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    DevCtx &D = G_.dev[dev];
    if (W > D.maxW) return fail(CUHE_EINVAL, "coefficient words %d exceed table %d", W, D.maxW);
    const Params &q = G_.prm;
    hipLaunchKernelGGL(k_crt, dim3((q.modLen + kCrtCoef - 1) / kCrtCoef), dim3(kCrtCoef * kCrtGroups), (size_t)((W + 7) & ~7) * kCrtCoef * 4, S(st), dst, src, prime_tab(D),
                       np, W, q.modLen, q.crtLen, 0L, 0L);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_icrt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (lvl < 0) return fail(CUHE_EINVAL, "icrt below level 0");
    DevCtx &D = G_.dev[dev];
    const Params &q = G_.prm;
    return launch_icrt(dst, src, D, lvl, np, W, 1, 0L, 0L, S(st));
}
int cuhe_hip_crt_add(uint32_t *sum, const uint32_t *x, const uint32_t *y, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    hipLaunchKernelGGL(k_crt_add, dim3((q.modLen + 255) / 256, np), dim3(256), 0, S(st), sum, x, y, prime_tab(G_.dev[dev]),
                       q.modLen, q.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_crt_add_int(uint32_t *sum, const uint32_t *x, unsigned a, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    hipLaunchKernelGGL(k_crt_add_int, dim3((np + 63) / 64), dim3(64), 0, S(st), sum, x, a, prime_tab(G_.dev[dev]), np,
                       G_.prm.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_crt_add_nx1(uint32_t *sum, const uint32_t *x, const uint32_t *s, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    hipLaunchKernelGGL(k_crt_add_nx1, dim3((q.modLen + 255) / 256, np), dim3(256), 0, S(st), sum, x, s,
                       prime_tab(G_.dev[dev]), q.modLen, q.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_crt_mul_int(uint32_t *prod, const uint32_t *x, int a, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    hipLaunchKernelGGL(k_crt_mul_int, dim3((np + 63) / 64), dim3(64), 0, S(st), prod, x, a, prime_tab(G_.dev[dev]), np,
                       G_.prm.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_crt_mod_switch(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (np < 2) return fail(CUHE_EINVAL, "modSwitch needs >= 2 primes");
    const Params &q = G_.prm;
    DevCtx &D = G_.dev[dev];
    hipLaunchKernelGGL(k_modswitch, dim3((q.modLen + 255) / 256, np - 1), dim3(256), 0, S(st), dst, src, prime_tab(D),
                       D.invp, np, q.modLen, q.crtLen, q.modMsg, 0L, 0L);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

int cuhe_hip_ntt(uint64_t *X, const uint32_t *x, int logq, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    return run_ntt(q.nttLen, kSrcU32Ext, X, x, np, q.crtLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_nttw(uint64_t *X, const uint32_t *x, int logq, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    if (!q.logRelin) return fail(CUHE_EINVAL, "logRelin = 0");
    return run_ntt(q.nttLen, kSrcWindow, X, x, q.numEvalKeyAt(lvl), 0, q.nttLen, q.nttLen, 0,
                   WindowArgs{W, q.logRelin, 0}, dev, S(st));
}
int cuhe_hip_intt(uint32_t *x, const uint64_t *X, int logq, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    return run_ntt(q.nttLen, kSrcU64Neg, x, X, np, q.nttLen, q.crtLen, q.crtLen, 0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_intt_hold(const uint64_t *X, int logq, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    Workspace *Wp = nullptr;
    CHK(workspace(dev, S(st), &Wp));
    CHK(ws_barrett(*Wp));
    return run_ntt(q.nttLen, kSrcU64Neg, Wp->hold, X, np, q.nttLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0},
                   dev, S(st));
}
int cuhe_hip_intt_double_deg(uint32_t *x, const uint64_t *X, int logq, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    return run_ntt(q.nttLen, kSrcU64Neg, x, X, np, q.nttLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_barrett(uint32_t *dst, const uint32_t *src, int lvl, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    if (lvl < 0 || lvl >= G_.prm.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    return barrett_impl(dst, src, 0, G_.prm.numCrtPrimeAt(lvl), dev, S(st), 0);
}
int cuhe_hip_barrett_hold(uint32_t *dst, int lvl, int dev, void *st) {
    CHK(need_init(dev));
    Workspace *Wp = nullptr;
    CHK(workspace(dev, S(st), &Wp));
    CHK(ws_barrett(*Wp));
    return cuhe_hip_barrett(dst, Wp->hold, lvl, dev, st);
}
int cuhe_hip_intt_mod(uint32_t *x, const uint64_t *X, int logq, int dev, void *st) {
    CHK(need_cyclic());
    if (fused_xn1()) {          // Phi_m = x^n + 1 with n = L/2: INTT, mod p_i and the reduction in one pass-2 epilogue
        CHK(need_init(dev));
        int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
        if (lvl < 0) return fail(CUHE_EINVAL, "inttMod below level 0");
        const Params &q = G_.prm;
        return run_ntt(q.nttLen, kSrcU64Neg, x, X, np, q.nttLen, q.crtLen, kFoldXn1, 0, WindowArgs{0, 0, 0}, dev, S(st));
    }
    CHK(cuhe_hip_intt_hold(X, logq, dev, st));
    int lvl = G_.prm.getLevel(logq);
    if (lvl < 0) return fail(CUHE_EINVAL, "inttMod below level 0");
    Workspace *Wp = nullptr;
    CHK(workspace(dev, S(st), &Wp));
    return barrett_impl(x, Wp->hold, 0, G_.prm.numCrtPrimeAt(lvl), dev, S(st), 0);
}
uint32_t *cuhe_hip_intt_result(int dev) {
    if (!G_.inited || dev < 0 || dev >= (int)G_.dev.size() || set_dev(dev) != CUHE_OK) return nullptr;
    Workspace *Wp = nullptr;                      // the CALLING thread's buffer (every host thread has its own)
    if (workspace_of_thread(dev, &Wp) != CUHE_OK || ws_barrett(*Wp) != CUHE_OK) return nullptr;
    return Wp->hold;
}

static int binop(bool mul, bool nx1, uint64_t *z, const uint64_t *x, const uint64_t *y, int logq, int dev, void *st, bool ct = false) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const int L = ct ? ct_len() : G_.prm.nttLen;
    if (!nx1) {
        const long pairs = (long)np * L / 2;
        const int grid = (int)std::min<long>((pairs + 255) / 256, 8192);
        if (mul) hipLaunchKernelGGL((k_ntt_binop<true>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, pairs);
        else hipLaunchKernelGGL((k_ntt_binop<false>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, pairs);
    } else {
        dim3 grid((L / 2 + 255) / 256, np);
        if (mul) hipLaunchKernelGGL((k_ntt_binop_nx1<true>), grid, dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, np, L / 2);
        else hipLaunchKernelGGL((k_ntt_binop_nx1<false>), grid, dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, np, L / 2);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_ntt_mul(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *st) { return binop(true, false, z, y, x, logq, dev, st); }
int cuhe_hip_ntt_mul_nx1(uint64_t *z, const uint64_t *x, const uint64_t *s, int logq, int dev, void *st) { return binop(true, true, z, x, s, logq, dev, st); }
int cuhe_hip_ntt_add(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *st) { return binop(false, false, z, y, x, logq, dev, st); }
int cuhe_hip_ntt_add_nx1(uint64_t *z, const uint64_t *x, const uint64_t *s, int logq, int dev, void *st) { return binop(false, true, z, x, s, logq, dev, st); }

// ---------------------------------------------------------------- ciphertext-domain (ct) drivers: what CuCtxt runs on
int cuhe_hip_set_negacyclic(int mode) {
    if (G_.inited) return fail(CUHE_EINVAL, "set_negacyclic must precede init");
    if (mode != 0 && mode != -1) return fail(CUHE_EINVAL, "negacyclic mode %d (-1 = where it applies, 0 = never)", mode);
    G_.nc_mode = mode;
    return CUHE_OK;
}
// products of two reduced polynomials whose SUM the inverse ct transform still recovers exactly: the integer coefficients of
// a product are below n p^2 in magnitude and must stay below P (cyclic) / P/2 (negacyclic: centred lift)
int cuhe_hip_ct_prod_headroom(void) {
    if (!G_.inited) return 0;
    host::u128 pmax = 0;
    for (uint32_t p : G_.primes) pmax = std::max<host::u128>(pmax, p);
    const host::u128 one = (host::u128)G_.prm.modLen * (pmax - 1) * (pmax - 1) * (G_.nc ? 2 : 1);
    const host::u128 h = one ? (host::u128)host::P / one : 1;
    return h > 1000000 ? 1000000 : (int)h;
}
int cuhe_hip_ct_negacyclic(void) { return G_.inited && G_.nc ? 1 : 0; }
int cuhe_hip_ct_len(void) { return G_.params_set ? (G_.inited ? ct_len() : G_.prm.nttLen) : 0; }
int cuhe_hip_ct_ntt(uint64_t *X, const uint32_t *x, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    return ct_forward((u64 *)X, x, np, dev, S(st));
}
int cuhe_hip_ct_intt(uint32_t *x, const uint64_t *X, int logq, int is_prod, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (is_prod && lvl < 0) return fail(CUHE_EINVAL, "product below level 0");
    return ct_inverse(x, (const u64 *)X, np, 0, 0, is_prod != 0, dev, S(st));
}
int cuhe_hip_ct_mul(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *st) { return binop(true, false, z, y, x, logq, dev, st, true); }
int cuhe_hip_ct_mul_nx1(uint64_t *z, const uint64_t *x, const uint64_t *s, int logq, int dev, void *st) { return binop(true, true, z, x, s, logq, dev, st, true); }
int cuhe_hip_ct_add(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *st) { return binop(false, false, z, y, x, logq, dev, st, true); }
int cuhe_hip_ct_add_nx1(uint64_t *z, const uint64_t *x, const uint64_t *s, int logq, int dev, void *st) { return binop(false, true, z, x, s, logq, dev, st, true); }

int cuhe_hip_ntt_one(uint64_t *X, const uint32_t *x, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    const Params &q = G_.prm;
    return run_ntt(q.nttLen, kSrcU32Ext, X, x, 1, q.crtLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_nttw_one(uint64_t *X, const uint32_t *x, int coeffwords, int relinIdx, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    const Params &q = G_.prm;
    return run_ntt(q.nttLen, kSrcWindow, X, x, 1, 0, q.nttLen, q.nttLen, 0, WindowArgs{coeffwords, q.logRelin, relinIdx},
                   dev, S(st));
}
int cuhe_hip_intt_one(uint32_t *x, const uint64_t *X, int crtidx, int dev, void *st) {
    CHK(need_init(dev));
    CHK(need_cyclic());
    const Params &q = G_.prm;
    if (crtidx < 0 || crtidx >= q.numCrtPrime) return fail(CUHE_EINVAL, "crtidx %d", crtidx);
    return run_ntt(q.nttLen, kSrcU64Neg, x, X, 1, q.nttLen, q.nttLen, q.nttLen, crtidx, WindowArgs{0, 0, 0}, dev, S(st));
}

// ---------------------------------------------------------------- relinearisation
// keys of the primes [first[dev], first[dev] + count[dev]) on every device (count < 0: all primes)
static int init_relin_impl(const uint32_t *ek_host, const int *first, const int *count) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "not initialised");
    const Params &q = G_.prm;
    const int K = q.numEvalKey, np = q.numCrtPrime, L = ct_len(), W0 = q.wordsCoeff(0);     // keys live in the ct domain
    if (K <= 0) return fail(CUHE_EINVAL, "numEvalKey = 0");
    const size_t rawBytes = (size_t)q.rawLen * W0 * 4;
    for (int dev = 0; dev < G_.ndev; ++dev) {
        const int p0 = first ? first[dev] : 0, pc = count ? count[dev] : np;
        if (p0 < 0 || pc < 1 || p0 + pc > np) return fail(CUHE_EINVAL, "key range [%d, %d) of %d primes", p0, p0 + pc, np);
        CHK(set_dev(dev));
        DevCtx &D = G_.dev[dev];
        if (D.ek) { hipFree(D.ek); D.ek = nullptr; }
        if (D.ekd) { hipFree(D.ekd); D.ekd = nullptr; }
        D.ekd_unavailable = false;
        HIPCHK(hipMalloc((void **)&D.ek, (size_t)pc * K * L * sizeof(u64)));
        D.ek_first = p0; D.ek_count = pc;
        u32 *raw = nullptr, *crt = nullptr; u64 *ntt = nullptr;
        HIPCHK(hipMalloc((void **)&raw, rawBytes));
        HIPCHK(hipMalloc((void **)&crt, (size_t)np * q.crtLen * 4));
        HIPCHK(hipMalloc((void **)&ntt, (size_t)pc * L * 8));
        for (int j = 0; j < K; ++j) {                              // cuhe/Relinearization.cu:49-56
            HIPCHK(hipMemcpy(raw, ek_host + (size_t)j * q.rawLen * W0, rawBytes, hipMemcpyHostToDevice));
            HIPCHK(hipMemsetAsync(crt, 0, (size_t)np * q.crtLen * 4, 0));
            CHK(cuhe_hip_crt(crt, raw, q.logCoeff(0), dev, nullptr));
            CHK(ct_forward(ntt, crt + (size_t)p0 * q.crtLen, pc, dev, nullptr));
            // ek[prime i - p0][key j][L]
            HIPCHK(hipMemcpy2DAsync(D.ek + (size_t)j * L, (size_t)K * L * 8, ntt, (size_t)L * 8, (size_t)L * 8, pc,
                                    hipMemcpyDeviceToDevice, 0));
        }
        HIPCHK(hipDeviceSynchronize());
        hipFree(raw); hipFree(crt); hipFree(ntt);
    }
    G_.relin_ready = true;
    return CUHE_OK;
}
int cuhe_hip_init_relin(const uint32_t *ek_host) { return init_relin_impl(ek_host, nullptr, nullptr); }
// The keys of `count` CRT primes from `prime0` on only: what a participant of the CRT-prime-sharded multiply needs
// (cuhe_hip_key_range gives the range that covers its block at every level): key memory / number of participants.
int cuhe_hip_init_relin_range(const uint32_t *ek_host, int prime0, int count) {
    std::vector<int> f(std::max(G_.ndev, 1), prime0), c(std::max(G_.ndev, 1), count);
    return init_relin_impl(ek_host, f.data(), c.data());
}
// primes participant `rank` of `nranks` owns at ANY level (its contiguous block moves down as the levels drop primes)
int cuhe_hip_key_range(int nranks, int rank, int *first, int *count) {
    if (!G_.params_set || nranks < 1 || rank < 0 || rank >= nranks || !first || !count) return fail(CUHE_EINVAL, "key_range(nranks %d, rank %d)", nranks, rank);
    int lo = 1 << 30, hi = 0;
    for (int lvl = 0; lvl < G_.prm.depth; ++lvl) {
        int f = 0, c = 0;
        comm::shard_bounds(G_.prm.numCrtPrimeAt(lvl), nranks, rank, &f, &c);
        if (c > 0) { lo = std::min(lo, f); hi = std::max(hi, f + c); }
    }
    if (hi <= lo) { lo = 0; hi = 1; }
    *first = lo; *count = hi - lo;
    return CUHE_OK;
}
// in-process form: device d of multiGPUs(n) keeps the keys of the primes it owns in cuhe_hip_mul_relin_sharded_inproc
int cuhe_hip_init_relin_sharded(const uint32_t *ek_host) {
    std::vector<int> f(G_.ndev), c(G_.ndev);
    for (int d = 0; d < G_.ndev; ++d) CHK(cuhe_hip_key_range(G_.ndev, d, &f[d], &c[d]));
    return init_relin_impl(ek_host, f.data(), c.data());
}

// ---- binary evaluation-key cache.  initRelinearization costs numEvalKey * numCrtPrime forward
// transforms plus the upload of the raw keys; the NTT-domain keys it produces depend only on the parameter set,
// the CRT primes and the key polynomials, so a deployment computes them once and reloads this image.
//   header (96 bytes, little endian): magic "CUHEEK\0\1", u32 version (2), i32 d,p,w,min,cut,m, i32 numCrtPrime,
//   i32 numEvalKey, i32 row length, 2 x u32 0, u64 FNV-1a of the CRT primes, u64 payload bytes, u64 payload hash (a
//   position-dependent multiply-rotate hash over the payload words: swapped words and paired bit flips change it),
//   u64 FNV-1a of the polynomial modulus coefficients, u64 key representation (0 = cyclic rows of nttLen, 1 = negacyclic rows of modLen);
//   payload: u64[prime][key][row length], canonical residues mod P  (the HBM layout, cuhe/Relinearization.cu:45-55);
//   import also refuses any word >= P (the field arithmetic assumes canonical operands).
struct EkHeader {
    char magic[8]; uint32_t version; int32_t set[6]; int32_t np, k, L; uint32_t zero[2];
    uint64_t primes_fnv, payload_bytes, payload_hash, modulus_fnv, key_rep;
};
static_assert(sizeof(EkHeader) == 96, "cache header layout");
static const char kEkMagic[8] = {'C', 'U', 'H', 'E', 'E', 'K', 0, 1};
static uint64_t fnv1a(const void *p, size_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= ((const uint8_t *)p)[i]; h *= 1099511628211ULL; }
    return h;
}
// four independent lanes of (h ^ word) * odd, rotated: position dependent, ~10 GB/s; returns canonical = false if a word is >= P
static uint64_t hash_words(const uint64_t *p, size_t n, bool *canonical) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ULL, 0xC2B2AE3D27D4EB4FULL, 0x165667B19E3779F9ULL, 0x27D4EB2F165667C5ULL};
    bool ok = true;
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int l = 0; l < 4; ++l) {
            const uint64_t w = p[i + l];
            ok &= w < host::P;
            uint64_t x = (h[l] ^ w) * 0x9FB21C651E98DF25ULL;
            h[l] = (x << 29) | (x >> 35);
        }
    for (; i < n; ++i) { const uint64_t w = p[i]; ok &= w < host::P; uint64_t x = (h[0] ^ w) * 0x9FB21C651E98DF25ULL; h[0] = (x << 29) | (x >> 35); }
    if (canonical) *canonical = ok;
    uint64_t r = n;
    for (int l = 0; l < 4; ++l) { r = (r ^ h[l]) * 0xD6E8FEB86659FD93ULL; r ^= r >> 32; }
    return r;
}
static EkHeader ek_header_now() {
    const Params &q = G_.prm;
    EkHeader h; memset(&h, 0, sizeof h);
    memcpy(h.magic, kEkMagic, 8); h.version = 2;
    const int set[6] = {q.depth, q.modMsg, q.logRelin, q.logCoeffMin, q.logCoeffCut, q.mSize};
    memcpy(h.set, set, sizeof set);
    h.np = q.numCrtPrime; h.k = q.numEvalKey; h.L = ct_len(); h.key_rep = G_.nc ? 1 : 0;
    h.primes_fnv = fnv1a(G_.primes.data(), G_.primes.size() * sizeof(uint32_t));
    h.modulus_fnv = fnv1a(G_.modulus.data(), G_.modulus.size() * sizeof(int32_t));
    h.payload_bytes = (uint64_t)q.numCrtPrime * q.numEvalKey * ct_len() * sizeof(u64);
    return h;
}
size_t cuhe_hip_relin_cache_size(void) {
    if (!G_.inited || G_.prm.numEvalKey <= 0) return 0;
    return sizeof(EkHeader) + (size_t)ek_header_now().payload_bytes;
}
int cuhe_hip_relin_export(void *dst, size_t cap, int dev) {
    CHK(need_init(dev));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    EkHeader h = ek_header_now();
    if (!dst || cap < sizeof h + h.payload_bytes) return fail(CUHE_EINVAL, "export buffer too small: %zu < %zu", cap, sizeof h + (size_t)h.payload_bytes);
    uint8_t *out = (uint8_t *)dst;
    CHK(need_all_keys(G_.dev[dev]));
    HIPCHK(hipMemcpy(out + sizeof h, G_.dev[dev].ek, h.payload_bytes, hipMemcpyDeviceToHost));
    h.payload_hash = hash_words((const uint64_t *)(out + sizeof h), h.payload_bytes / 8, nullptr);
    memcpy(out, &h, sizeof h);
    return CUHE_OK;
}
int cuhe_hip_relin_import(const void *src, size_t bytes) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "not initialised");
    if (!src || bytes < sizeof(EkHeader)) return fail(CUHE_EINVAL, "evaluation-key cache: truncated header");
    EkHeader h; memcpy(&h, src, sizeof h);
    const EkHeader want = ek_header_now();
    if (memcmp(h.magic, kEkMagic, 8) != 0 || h.version != 2) return fail(CUHE_EINVAL, "evaluation-key cache: bad magic / version");
    if (memcmp(h.set, want.set, sizeof h.set) != 0 || h.np != want.np || h.k != want.k || h.L != want.L)
        return fail(CUHE_EINVAL, "evaluation-key cache was made for other parameters");
    if (h.primes_fnv != want.primes_fnv) return fail(CUHE_EINVAL, "evaluation-key cache was made for other CRT primes");
    if (h.modulus_fnv != want.modulus_fnv || h.key_rep != want.key_rep) return fail(CUHE_EINVAL, "evaluation-key cache was made for another polynomial modulus / key representation");
    if (h.payload_bytes != want.payload_bytes || bytes < sizeof h + h.payload_bytes) return fail(CUHE_EINVAL, "evaluation-key cache: truncated payload");
    const uint8_t *payload = (const uint8_t *)src + sizeof h;
    bool canonical = true;
    if (hash_words((const uint64_t *)payload, h.payload_bytes / 8, &canonical) != h.payload_hash) return fail(CUHE_EINVAL, "evaluation-key cache: payload checksum mismatch");
    if (!canonical) return fail(CUHE_EINVAL, "evaluation-key cache: payload holds a word >= P");
    const Params &q = G_.prm;
    for (int dev = 0; dev < G_.ndev; ++dev) {
        CHK(set_dev(dev));
        DevCtx &D = G_.dev[dev];
        if (D.ek && (D.ek_first != 0 || D.ek_count != G_.prm.numCrtPrime)) { hipFree(D.ek); D.ek = nullptr; }     // a partial set: re-allocate
        if (!D.ek) HIPCHK(hipMalloc((void **)&D.ek, h.payload_bytes));
        D.ek_first = 0; D.ek_count = G_.prm.numCrtPrime;
        if (D.ekd) { hipFree(D.ekd); D.ekd = nullptr; }
        D.ekd_unavailable = false;
        HIPCHK(hipMemcpy(D.ek, payload, h.payload_bytes, hipMemcpyHostToDevice));
    }
    G_.relin_ready = true;
    return CUHE_OK;
}
static int relin_range(uint64_t *dst, const uint32_t *src, int lvl, int prime0, int count, int dev, void *st) {
    CHK(need_init(dev));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    const int k = q.numEvalKeyAt(lvl), np = q.numCrtPrimeAt(lvl), L = ct_len();
    if (prime0 < 0 || count < 1 || prime0 + count > np) return fail(CUHE_EINVAL, "prime range [%d,%d) at level %d", prime0, prime0 + count, lvl);
    DevCtx &D = G_.dev[dev];
    Workspace *Wp = nullptr;
    CHK(workspace(dev, S(st), &Wp));
    CHK(ws_relin(*Wp));
    // window rows once (coalesced), then k plain zero-padded transforms (replaces k strided window loads)
    const int W = q.wordsCoeff(lvl);
    hipLaunchKernelGGL(k_extract_windows, dim3((q.crtLen + kWinCoef - 1) / kWinCoef), dim3(kWinCoef * kWinGroups),
                       (size_t)W * kWinCoef * 4, S(st), Wp->win, src, W, q.logRelin, k, q.crtLen, q.crtLen, 0L, 0L);
    HIPCHK(hipGetLastError());
    CHK(ct_forward(Wp->relin, Wp->win, k, dev, S(st)));
    // primes per workgroup (each window value fetched from cache serves PB key streams): as many as still leave ~6
    // workgroups per CU -- the kernel streams the keys from HBM and needs that many loads in flight (12 waves per CU reach
    // 4.3 TB/s, 24 reach 6 TB/s: profiles/r02_experiments_log.txt)
    if (prime0 < D.ek_first || prime0 + count > D.ek_first + D.ek_count)
        return fail(CUHE_EINVAL, "keys of primes [%d, %d) wanted, device %d holds [%d, %d)", prime0, prime0 + count, dev, D.ek_first, D.ek_first + D.ek_count);
    const u64 *ekp = D.ek + (size_t)(prime0 - D.ek_first) * q.numEvalKey * L;
    const long target = 6L * 256;
    auto blocks = [&](int pb) { return (long)(L / 512) * ((count + pb - 1) / pb); };
    if (blocks(4) >= target || count <= 1)
        hipLaunchKernelGGL((k_relin_mac<4, 1>), dim3((unsigned)blocks(4)), dim3(256), 0, S(st), (u64 *)dst, Wp->relin, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    else if (blocks(2) >= target || count <= 2)
        hipLaunchKernelGGL((k_relin_mac<2, 1>), dim3((unsigned)blocks(2)), dim3(256), 0, S(st), (u64 *)dst, Wp->relin, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    else
        hipLaunchKernelGGL((k_relin_mac<1, 1>), dim3((unsigned)blocks(1)), dim3(256), 0, S(st), (u64 *)dst, Wp->relin, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_relinearization(uint64_t *dst, const uint32_t *src, int lvl, int dev, void *st) {
    return relin_range(dst, src, lvl, 0, G_.prm.numCrtPrimeAt(lvl < 0 ? 0 : lvl), dev, st);
}

// ---- key digits for the inner product on the matrix cores (k_relin_mac_mfma): built on the first batched call that
// wants them, from the ct-domain keys; as large as the keys themselves.
// measured crossover at config 4: 2 and 4 ciphertexts are a little faster on the VALU kernel (0.190 / 0.127 vs 0.197 / 0.134 ms per
// ciphertext), 6 already on the matrix cores (0.110 vs 0.141: one half-filled tile instead of two VALU groups)
static int g_mac_mfma_min = getenv("CUHE_MAC_MFMA_MIN") ? atoi(getenv("CUHE_MAC_MFMA_MIN")) : 5;     // smallest batch that takes the MFMA kernel; 0 = never
int cuhe_hip_set_relin_mfma(int min_batch) {
    if (min_batch < 0) return fail(CUHE_EINVAL, "min_batch %d", min_batch);
    g_mac_mfma_min = min_batch;
    return CUHE_OK;
}
// ---------------------------------------------------------------- batched multiply + relinearise
// `batch` independent (cAnd ; relin) chains of one level in a single call: NTT-domain operands a, b as
// u64[batch][np][L], reduced CRT-domain results as u32[batch][np][crtLen].  Same arithmetic as `batch` calls of
// ntt_mul, intt_mod, icrt, relinearization, intt_mod; what changes is the shape of the work: every stage runs once
// over batch*np (or batch*k) rows -- several hundred workgroups instead of a few dozen, so the transforms leave their
// latency floor (profiles/r01_small_batch_latency.txt) -- and the inner product fetches each key value once for
// four ciphertexts.  The reference has no batched form: its circuits issue ciphertext operations one at a time.
// core of the batched calls: a, b != null -> products of NTT-domain operands first (cAnd ; relin);
// crt_in != null -> relinearisation of CRT-domain ciphertexts (CuCtxt::relin on a reduced ciphertext)
static int relin_batch_run(uint32_t *dst, const uint64_t *a, const uint64_t *b, const uint32_t *crt_in, int lvl, int batch, int dev, void *st_) {
    CHK(need_init(dev));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (batch < 1) return fail(CUHE_EINVAL, "batch %d", batch);
    hipStream_t st = S(st_);
    DevCtx &D = G_.dev[dev];
    CHK(need_all_keys(D));
    const int np = q.numCrtPrimeAt(lvl), k = q.numEvalKeyAt(lvl), W = q.wordsCoeff(lvl), L = ct_len(), cl = q.crtLen;
    const int rows = batch * np;
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    Workspace &Ws = *Wp;
    if (Ws.n_bt < (size_t)batch) {
        size_t x = 0, y = 0;
        if (Ws.bt_ntt) { HIPCHK(hipFree(Ws.bt_ntt)); Ws.bt_ntt = nullptr; }
        if (Ws.bt_crt) { HIPCHK(hipFree(Ws.bt_crt)); Ws.bt_crt = nullptr; }
        CHK(ws_grow(&Ws.bt_ntt, &x, (size_t)batch * q.numCrtPrime * L));
        CHK(ws_grow(&Ws.bt_crt, &y, (size_t)batch * q.numCrtPrime * cl));
        Ws.n_bt = batch;
    }
    CHK(ws_relin(Ws, batch));
    // reduction of `rows` ct-domain product rows to CRT rows (n2c with isProd, CuHE.cu:398-408)
    auto reduce_rows = [&](u32 *out, const u64 *in) -> int { return ct_inverse(out, in, rows, 0, np, true, dev, st); };
    const u32 *crt_rows = crt_in;
    if (!crt_in) {
        // 1.-2. x2r of the pointwise products: INTT + reduction; the products are formed as the first pass loads its samples
        CHK(ct_inverse(Ws.bt_crt, (const u64 *)a, rows, 0, np, true, dev, st, (const u64 *)b));
        crt_rows = Ws.bt_crt;
    }
    // ICRT of every ciphertext, 3. straight into the relinearisation windows (batch*k rows): the raw form is never stored
    if (q.modLen < cl)                                                // coefficients modLen .. crtLen of every window row are zero
        HIPCHK(hipMemset2DAsync(Ws.win + q.modLen, (size_t)cl * sizeof(u32), 0, (size_t)(cl - q.modLen) * sizeof(u32), (size_t)batch * k, st));
    CHK(launch_icrt(nullptr, crt_rows, D, lvl, np, W, batch, (long)np * cl, 0L, st, IcrtWindows{Ws.win, (long)k * cl, q.logRelin, k, cl}));
    CHK(ct_forward(Ws.relin, Ws.win, batch * k, dev, st));
    // 4. key-switch inner products: a key value fetched once serves four ciphertexts
    // window tiles of 4 ciphertexts resident in LDS, every key value fetched once per 4 ciphertexts (k_relin_mac_lds);
    // PB (primes per thread and pass) is the one of 2, 3, 4 that wastes the fewest of the 8 x PB prime slots per pass.
    // Falls back to the register-blocked kernel (2 primes x 4 ciphertexts per workgroup) when the tile exceeds LDS.
    // Batches of >= g_mac_mfma_min ciphertexts: the products run on the matrix cores in groups of 16 ciphertexts
    // (k_relin_mac_mfma); a remainder below that size and small batches take the VALU kernel below.
    int done = 0;
    if (g_mac_mfma_min > 0 && batch >= g_mac_mfma_min && mac_mfma_supported(q.numEvalKey, k, np) && (L % 64) == 0) {
        CHK(ensure_key_digits(dev, st));
        if (D.ekd) {
            const int rem = batch % kMacMfmaCts;
            done = (rem >= g_mac_mfma_min || batch < kMacMfmaCts) ? batch : batch - rem;
            CHK(run_mac_mfma(Ws.bt_ntt, Ws.relin, D, k, L, np, (long)k * L, (long)np * L, done, st));
        }
    }
    if (done < batch) {                                               // the VALU kernels take the remaining `rest` ciphertexts
        const int rest = batch - done;
        u64 *const out_rows = Ws.bt_ntt + (size_t)done * np * L;
        const u64 *const win_rows = Ws.relin + (size_t)done * k * L;
        constexpr int BB = 4;
        constexpr int CBr = 32, NGr = kMacLdsThreads / CBr;      // 16-column tiles (3 workgroups per CU) measured the same
        const size_t lds = (size_t)BB * k * CBr * sizeof(u64);
        if (lds <= 150 * 1024) {
            int best = 2; double eff = 0;
            for (int pb = 2; pb <= 4; ++pb) {
                const int slots = ((np + NGr * pb - 1) / (NGr * pb)) * NGr * pb;
                const double f = (double)np / slots;
                if (f >= eff) { eff = f; best = pb; }
            }
            const dim3 grid((L / CBr) * ((rest + BB - 1) / BB)), block(kMacLdsThreads);        // (tile, group) pairs, see the kernel
#define MACL(PB_, CB_) do { \
                if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_relin_mac_lds<PB_, BB, CB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                hipLaunchKernelGGL((k_relin_mac_lds<PB_, BB, CB_>), grid, block, lds, st, out_rows, win_rows, D.ek, k, (long)q.numEvalKey * L, L, np, \
                                   (long)k * L, (long)np * L, rest); } while (0)
            if (best == 2) MACL(2, CBr); else if (best == 3) MACL(3, CBr); else MACL(4, CBr);
#undef MACL
        } else {
            constexpr int PB = 2;
            hipLaunchKernelGGL((k_relin_mac<PB, BB, 1>), dim3((L / 512) * ((np + PB - 1) / PB), 1, (rest + BB - 1) / BB), dim3(256), 0, st,
                               out_rows, win_rows, D.ek, k, (long)q.numEvalKey * L, L, np, (long)k * L, (long)np * L, rest);
        }
    }
    HIPCHK(hipGetLastError());
    // 5. n2c of the sums
    return reduce_rows(dst, Ws.bt_ntt);
}

// Optional: groups of four ciphertexts go round-robin to `lanes` streams (the caller's and helper streams of the calling
// thread, each with its own scratch), so that the inner product of one group streams keys while the transforms of
// another keep the vector units busy.  This was the default on rings with >= 1 GiB of keys per level in round 1; since
// the inner-product kernel places the ciphertext groups of one column tile next to each other on one XCD (a key value
// then leaves HBM once per BATCH, not once per group) one launch sequence over the whole batch is faster on every ring
// measured (profiles/r02_relin_lanes_ab.txt), so the default is 1 lane; cuhe_hip_set_relin_lanes(n) still selects more.
static int g_relin_lanes = getenv("CUHE_RELIN_LANES") ? atoi(getenv("CUHE_RELIN_LANES")) : 1;
static bool g_relin_lanes_any_size = getenv("CUHE_RELIN_LANES") != nullptr;          // -n: n lanes whatever the ring size (tests)
int cuhe_hip_set_relin_lanes(int n) {
    const int m = n < 0 ? -n : n;
    if (m < 1 || m > kLanes) return fail(CUHE_EINVAL, "lanes %d (1..%d)", n, kLanes);
    g_relin_lanes = m; g_relin_lanes_any_size = n < 0;
    return CUHE_OK;
}
static int relin_batch_core(uint32_t *dst, const uint64_t *a, const uint64_t *b, const uint32_t *crt_in, int lvl, int batch, int dev, void *st_) {
    // a group is what one launch sequence handles: 16 ciphertexts (one tile of the matrix-core inner product) when that
    // kernel will run, 4 (one window tile of the VALU kernel) otherwise
    const Params &q = G_.prm;
    const bool mfma = g_mac_mfma_min > 0 && batch >= 2 * kMacMfmaCts && G_.inited && lvl >= 0 && lvl < q.depth &&
                      mac_mfma_supported(q.numEvalKey, q.numEvalKeyAt(lvl), q.numCrtPrimeAt(lvl));
    const int GB = mfma ? kMacMfmaCts : 4;
    const int groups = (batch + GB - 1) / GB, lanes = std::min(g_relin_lanes, groups);
    if (lanes <= 1 || !G_.inited || lvl < 0 || lvl >= q.depth ||
        (!g_relin_lanes_any_size && (size_t)q.numEvalKeyAt(lvl) * q.numCrtPrimeAt(lvl) * ct_len() * sizeof(u64) < ((size_t)1 << 30)))
        return relin_batch_run(dst, a, b, crt_in, lvl, batch, dev, st_);
    CHK(need_init(dev));
    const size_t np = q.numCrtPrimeAt(lvl), L = ct_len(), cl = q.crtLen;
    hipStream_t st = S(st_);
    Workspace *W0 = nullptr, *LW[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    CHK(workspace(dev, st, &W0));
    if (!W0->ev_in) HIPCHK(hipEventCreateWithFlags(&W0->ev_in, hipEventDisableTiming));
    if (mfma) CHK(ensure_key_digits(dev, st));                     // built once, on the caller's stream, before any lane can want it
    HIPCHK(hipEventRecord(W0->ev_in, st));                         // whatever produced the operands on `st` is before this
    LaneReset reset;
    for (int g = 0; g < groups; ++g) {
        const int lane = g % lanes, b0 = g * GB, nb = std::min(GB, batch - b0);
        tls_lane = lane;
        hipStream_t s = st;
        if (lane) {
            Workspace *w = nullptr;
            CHK(workspace_of_thread(dev, &w));
            if (!w->lane_stream) {
                HIPCHK(hipStreamCreateWithFlags(&w->lane_stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&w->ev_lane, hipEventDisableTiming));
            }
            if (!LW[lane]) { LW[lane] = w; HIPCHK(hipStreamWaitEvent(w->lane_stream, W0->ev_in, 0)); }
            s = w->lane_stream;
        }
        CHK(relin_batch_run(dst + (size_t)b0 * np * cl, a ? a + (size_t)b0 * np * L : nullptr, b ? b + (size_t)b0 * np * L : nullptr,
                            crt_in ? crt_in + (size_t)b0 * np * cl : nullptr, lvl, nb, dev, (void *)s));
    }
    tls_lane = 0;
    for (int lane = 1; lane < kLanes; ++lane) if (LW[lane]) {
        HIPCHK(hipEventRecord(LW[lane]->ev_lane, LW[lane]->lane_stream));
        HIPCHK(hipStreamWaitEvent(st, LW[lane]->ev_lane, 0));
    }
    return CUHE_OK;
}

int cuhe_hip_mul_relin_batch(uint32_t *dst, const uint64_t *a, const uint64_t *b, int lvl, int batch, int dev, void *st) {
    if (!a || !b) return fail(CUHE_EINVAL, "null operand");
    return relin_batch_core(dst, a, b, nullptr, lvl, batch, dev, st);
}
// CuCtxt::relin (CuHE.cu:570-581) for `batch` reduced CRT-domain ciphertexts u32[batch][np][crtLen] of one level:
// ICRT, windows, window transforms, key-switch inner products, INTT + reduction, in one call
int cuhe_hip_relin_batch(uint32_t *dst, const uint32_t *src, int lvl, int batch, int dev, void *st) {
    if (!src) return fail(CUHE_EINVAL, "null operand");
    return relin_batch_core(dst, nullptr, nullptr, src, lvl, batch, dev, st);
}

// ---------------------------------------------------------------- gates on arrays of ciphertexts
// The C++ gates (cAnd, cXor, cNot, modSwitch: CuHE.cu:101-215,545-568) act on one ciphertext per call; a circuit
// layer (the 16 S-boxes of a PRINCE round) is hundreds of them.  These entry points apply one kind of gate to a whole
// array u32[count][np][crtLen] / u64[count][np][nttLen] of ciphertexts of one level in a single launch sequence.
int cuhe_hip_intt_mod_batch(uint32_t *dst, const uint64_t *src, int lvl, int batch, int dev, void *st_) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (batch < 1) return fail(CUHE_EINVAL, "batch %d", batch);
    hipStream_t st = S(st_);
    const int np = q.numCrtPrimeAt(lvl);
    return ct_inverse(dst, (const u64 *)src, batch * np, 0, np, true, dev, st);
}
// modSwitch of `batch` ciphertexts of level lvl: src u32[batch][np][crtLen] -> dst u32[batch][np-1][crtLen] (packed)
int cuhe_hip_crt_mod_switch_batch(uint32_t *dst, const uint32_t *src, int lvl, int batch, int dev, void *st) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl + 1 >= q.depth) return fail(CUHE_EINVAL, "modSwitch from level %d", lvl);
    if (batch < 1) return fail(CUHE_EINVAL, "batch %d", batch);
    const int np = q.numCrtPrimeAt(lvl);
    if (np < 2) return fail(CUHE_EINVAL, "modSwitch needs >= 2 primes");
    DevCtx &D = G_.dev[dev];
    hipLaunchKernelGGL(k_modswitch, dim3((q.modLen + 255) / 256, np - 1, batch), dim3(256), 0, S(st), dst, src, prime_tab(D),
                       D.invp, np, q.modLen, q.crtLen, q.modMsg, (long)np * q.crtLen, (long)(np - 1) * q.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// dst[t] = src[idx_a[t]] * src[idx_b[t]] (pointwise mod P) for t < npairs; ciphertexts of `np_rows` rows; the index
// arrays live in device memory
int cuhe_hip_ntt_mul_pairs(uint64_t *dst, const uint64_t *src, const int32_t *idx_a, const int32_t *idx_b, int npairs, int np_rows, int dev, void *st) {
    CHK(need_init(dev));
    if (npairs < 1 || np_rows < 1) return fail(CUHE_EINVAL, "npairs %d rows %d", npairs, np_rows);
    const long ct_pairs = (long)np_rows * ct_len() / 2;
    const int gx = (int)std::min<long>((ct_pairs + 255) / 256, 1024);
    hipLaunchKernelGGL(k_ntt_mul_pairs, dim3(gx, npairs), dim3(256), 0, S(st), (u64 *)dst, (const u64 *)src, idx_a, idx_b, ct_pairs);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// dst[o] = sum over list[off[o] .. off[o+1]) of CRT-domain ciphertexts (entries < nA from src_a, the rest from
// src_b) + add_const[o] on the constant coefficient, for o < nout, at level lvl; off / list / add_const in device memory
int cuhe_hip_crt_combine(uint32_t *dst, const uint32_t *src_a, int nA, const uint32_t *src_b, const int32_t *off, const int32_t *list,
                         const int32_t *add_const, int nout, int lvl, int dev, void *st) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (nout < 1) return fail(CUHE_EINVAL, "nout %d", nout);
    const int np = q.numCrtPrimeAt(lvl);
    DevCtx &D = G_.dev[dev];
    hipLaunchKernelGGL(k_crt_combine, dim3((q.modLen + 255) / 256, np, nout), dim3(256), 0, S(st), dst, src_a, nA, src_b, off, list, add_const,
                       prime_tab(D), np, q.modLen, q.crtLen);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

// `batch` independent full multiplications raw -> raw of one level in a single call (mulZZX without the host
// staging, CuHE.cu:259-268: CRT, NTT, pointwise product, INTT + reduction mod the polynomial modulus, ICRT), operands
// and results as u32[batch][rawLen][W].  Same arithmetic as `batch` single sequences; every stage runs once over
// batch (x np) rows.  A single multiplication at config 3 is seven launches of a few megabytes each and sits on
// launch and latency floors; a batch amortises them.
int cuhe_hip_mul_raw_batch(uint32_t *dst, const uint32_t *a, const uint32_t *b, int lvl, int batch, int dev, void *st_) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (batch < 1) return fail(CUHE_EINVAL, "batch %d", batch);
    hipStream_t st = S(st_);
    DevCtx &D = G_.dev[dev];
    const int np = q.numCrtPrimeAt(lvl), W = q.wordsCoeff(lvl), L = ct_len(), cl = q.crtLen;
    if (W > D.maxW) return fail(CUHE_EINVAL, "coefficient words %d exceed table %d", W, D.maxW);
    const int rows = batch * np;
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    Workspace &Ws = *Wp;
    // scratch: CRT rows of both operands (2*rows), their transforms (2*rows)
    if (Ws.n_mr < (size_t)batch) {
        size_t x = 0, y = 0;
        if (Ws.mr_ntt) { HIPCHK(hipFree(Ws.mr_ntt)); Ws.mr_ntt = nullptr; }
        if (Ws.mr_crt) { HIPCHK(hipFree(Ws.mr_crt)); Ws.mr_crt = nullptr; }
        CHK(ws_grow(&Ws.mr_ntt, &x, (size_t)2 * batch * q.numCrtPrime * L));
        CHK(ws_grow(&Ws.mr_crt, &y, (size_t)2 * batch * q.numCrtPrime * cl));
        Ws.n_mr = batch;
    }
    u32 *ca = Ws.mr_crt, *cb = Ws.mr_crt + (size_t)rows * cl;
    u64 *na = Ws.mr_ntt;
    if (q.modLen < cl) HIPCHK(hipMemsetAsync(Ws.mr_crt, 0, (size_t)2 * rows * cl * sizeof(u32), st));
    const size_t lds_crt = (size_t)((W + 7) & ~7) * kCrtCoef * 4;
    const dim3 gcrt((q.modLen + kCrtCoef - 1) / kCrtCoef, batch);
    hipLaunchKernelGGL(k_crt, gcrt, dim3(kCrtCoef * kCrtGroups), lds_crt, st, ca, a, prime_tab(D), np, W, q.modLen, cl, (long)q.rawLen * W, (long)np * cl);
    hipLaunchKernelGGL(k_crt, gcrt, dim3(kCrtCoef * kCrtGroups), lds_crt, st, cb, b, prime_tab(D), np, W, q.modLen, cl, (long)q.rawLen * W, (long)np * cl);
    HIPCHK(hipGetLastError());
    // transforms of the a operands, then those of the b operands with the pointwise product riding on their output
    // (kOutU64Mul with the a transforms as the table: row r of b times row r of a) -- no separate product pass
    u64 *nb = na + (size_t)rows * L;
    CHK(ct_forward(nb, cb, rows, dev, st));
    CHK(ct_forward(na, ca, rows, dev, st, nb));
    CHK(ct_inverse(ca, na, rows, 0, np, true, dev, st));
    if (q.modLen < q.rawLen) HIPCHK(hipMemsetAsync(dst, 0, (size_t)batch * q.rawLen * W * sizeof(u32), st));
    return launch_icrt(dst, ca, D, lvl, np, W, batch, (long)np * cl, (long)q.rawLen * W, st);
}

// ---------------------------------------------------------------- CRT-prime-sharded variants
int cuhe_hip_relin_range(uint64_t *dst, const uint32_t *raw, int lvl, int prime0, int count, int dev, void *st) {
    return relin_range(dst, raw, lvl, prime0, count, dev, st);
}
int cuhe_hip_ntt_rows(uint64_t *X, const uint32_t *x, int count, int dev, void *st) {
    CHK(need_init(dev));
    return ct_forward((u64 *)X, x, count, dev, S(st));
}
int cuhe_hip_ntt_mul_rows(uint64_t *z, const uint64_t *y, const uint64_t *x, int count, int dev, void *st) {
    CHK(need_init(dev));
    const long pairs = (long)count * ct_len() / 2;
    const int grid = (int)std::min<long>((pairs + 255) / 256, 8192);
    hipLaunchKernelGGL((k_ntt_binop<true>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)y, (const u64 *)x, pairs);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_intt_mod_range(uint32_t *x, const uint64_t *X, int lvl, int prime0, int count, int dev, void *st) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (prime0 < 0 || count < 1 || prime0 + count > q.numCrtPrimeAt(lvl)) return fail(CUHE_EINVAL, "prime range [%d,%d)", prime0, prime0 + count);
    return ct_inverse(x, (const u64 *)X, count, prime0, 0, true, dev, S(st));
}
int cuhe_hip_crt_range(uint32_t *dst, const uint32_t *src, int logq, int prime0, int count, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (prime0 < 0 || count < 1 || prime0 + count > np) return fail(CUHE_EINVAL, "prime range [%d,%d)", prime0, prime0 + count);
    DevCtx &D = G_.dev[dev];
    const Params &q = G_.prm;
    hipLaunchKernelGGL(k_crt, dim3((q.modLen + kCrtCoef - 1) / kCrtCoef), dim3(kCrtCoef * kCrtGroups), (size_t)((W + 7) & ~7) * kCrtCoef * 4, S(st), dst, src, prime_tab_at(D, prime0),
                       count, W, q.modLen, q.crtLen, 0L, 0L);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}


// ---------------------------------------------------------------- CRT-prime-sharded multiply + relinearise
// Rank / device r owns a contiguous block of the level's primes (comm::shard_bounds).  Pointwise product, inverse
// transform (+ reduction), the key-switch inner product over the OWNED primes' keys and the last inverse transform need
// no communication; the one exchange is the all-gather of the CRT rows before ICRT, which every participant repeats
// (26-100 us) together with the k window transforms.
static int ws_shard(Workspace &w) {
    if (w.sh_ready) return CUHE_OK;
    const Params &q = G_.prm;
    const size_t np = q.numCrtPrime, Lc = ct_len();
    CHK(ws_buffer(&w.sh_a, np * Lc)); CHK(ws_buffer(&w.sh_b, np * Lc));
    CHK(ws_buffer(&w.sh_rows, np * q.crtLen)); CHK(ws_buffer(&w.sh_raw, (size_t)q.rawLen * q.wordsCoeff(0))); CHK(ws_buffer(&w.sh_out, np * q.crtLen));
    w.sh_ready = true;
    return CUHE_OK;
}
// stage 1 on one participant: products of the owned rows, back to the CRT domain into rows[first ..) of the gather buffer
static int shard_stage1(u32 *rows, u64 *tmp, const u64 *a_own, const u64 *b_own, int first, int count, int dev, hipStream_t st) {
    const Params &q = G_.prm;
    const long pairs = (long)count * ct_len() / 2;
    hipLaunchKernelGGL((k_ntt_binop<true>), dim3((int)std::min<long>((pairs + 255) / 256, 8192)), dim3(256), 0, st, tmp, a_own, b_own, pairs);
    HIPCHK(hipGetLastError());
    return ct_inverse(rows + (size_t)first * q.crtLen, tmp, count, first, 0, true, dev, st);
}
// stage 2: ICRT of the gathered rows, key switch over the owned primes, back to the CRT domain
static int shard_stage2(u32 *out_own, u32 *raw, u64 *acc, const u32 *rows, int lvl, int first, int count, int dev, hipStream_t st) {
    const Params &q = G_.prm;
    if (q.modLen < q.rawLen) HIPCHK(hipMemsetAsync(raw, 0, (size_t)q.rawLen * q.wordsCoeff(lvl) * sizeof(u32), st));
    CHK(cuhe_hip_icrt(raw, rows, q.logCoeff(lvl), dev, (void *)st));
    CHK(relin_range((uint64_t *)acc, raw, lvl, first, count, dev, (void *)st));
    return ct_inverse(out_own, acc, count, first, 0, true, dev, st);
}

int cuhe_hip_shard_bounds(int lvl, int nranks, int rank, int *first, int *count) {
    if (!G_.params_set || lvl < 0 || lvl >= G_.prm.depth || nranks < 1 || rank < 0 || rank >= nranks || !first || !count)
        return fail(CUHE_EINVAL, "shard_bounds(lvl %d, nranks %d, rank %d)", lvl, nranks, rank);
    comm::shard_bounds(G_.prm.numCrtPrimeAt(lvl), nranks, rank, first, count);
    return CUHE_OK;
}
// ---- one process per GPU: RCCL
int cuhe_hip_comm_unique_id(void *id128) {
    comm::Api &A = comm::api();
    if (A.error) return fail(CUHE_EHIP, "RCCL: %s", A.error);
    ncclUniqueId id;
    const ncclResult_t r = A.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(CUHE_EHIP, "ncclGetUniqueId: %s", A.GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return CUHE_OK;
}
int cuhe_hip_comm_init(int nranks, int rank, const void *id128) {
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128) return fail(CUHE_EINVAL, "comm_init(%d, %d)", nranks, rank);
    comm::Api &A = comm::api();
    if (A.error) return fail(CUHE_EHIP, "RCCL: %s", A.error);
    comm::State &C = comm::state();
    if (C.comm) return fail(CUHE_EINVAL, "communicator already initialised");
    HIPCHK(hipSetDevice(phys_dev(0)));                   // the rank's GPU: cuhe_hip_set_device_base(LOCAL_RANK)
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    const ncclResult_t r = A.CommInitRank(&C.comm, nranks, id, rank);
    if (r != ncclSuccess) { C.comm = nullptr; return fail(CUHE_EHIP, "ncclCommInitRank(%d of %d): %s", rank, nranks, A.GetErrorString(r)); }
    C.nranks = nranks; C.rank = rank;
    return CUHE_OK;
}
int cuhe_hip_comm_destroy(void) {
    comm::State &C = comm::state();
    if (C.comm) { comm::api().CommDestroy(C.comm); C.comm = nullptr; }
    C.nranks = 1; C.rank = 0;
    return CUHE_OK;
}
int cuhe_hip_comm_size(void) { return comm::state().nranks; }
int cuhe_hip_comm_rank(void) { return comm::state().rank; }
// rows: u32[np][crtLen] of level lvl on this rank's device, the rank's own block already in place; on return (in stream
// order) every block is.  A group of broadcasts, root r sending its block in place, because the blocks differ in size
// when np is not a multiple of the number of ranks.
int cuhe_hip_allgather_rows(uint32_t *rows, int lvl, int dev, void *st) {
    CHK(need_init(dev));
    if (lvl < 0 || lvl >= G_.prm.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    comm::State &C = comm::state();
    if (C.nranks == 1) return CUHE_OK;
    if (!C.comm) return fail(CUHE_ENOTINIT, "cuhe_hip_comm_init has not been called");
    comm::Api &A = comm::api();
    const int np = G_.prm.numCrtPrimeAt(lvl), cl = G_.prm.crtLen;
    ncclResult_t r = A.GroupStart();
    for (int rk = 0; rk < C.nranks && r == ncclSuccess; ++rk) {
        int f, c; comm::shard_bounds(np, C.nranks, rk, &f, &c);
        if (c == 0) continue;
        u32 *blk = rows + (size_t)f * cl;
        r = A.Broadcast(blk, blk, (size_t)c * cl, ncclUint32, rk, C.comm, S(st));
    }
    const ncclResult_t e = A.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows: %s", A.GetErrorString(r));
    return CUHE_OK;
}
// cAnd + relin with the level's primes sharded over the ranks of the communicator: a_own, b_own = ct rows of the rank's
// own primes (u64[count][ct_len]), dst_own = the reduced CRT rows of the same primes (u32[count][crtLen]).  Everything,
// the all-gather included, is enqueued on `stream`.
int cuhe_hip_mul_relin_sharded(uint32_t *dst_own, const uint64_t *a_own, const uint64_t *b_own, int lvl, int dev, void *st_) {
    CHK(need_init(dev));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    if (lvl < 0 || lvl >= G_.prm.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    comm::State &C = comm::state();
    int f, c; comm::shard_bounds(G_.prm.numCrtPrimeAt(lvl), C.nranks, C.rank, &f, &c);
    if (c < 1) return fail(CUHE_EINVAL, "rank %d owns no prime at level %d (%d ranks)", C.rank, lvl, C.nranks);
    hipStream_t st = S(st_);
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    CHK(ws_shard(*Wp));
    CHK(shard_stage1(Wp->sh_rows, Wp->sh_a, (const u64 *)a_own, (const u64 *)b_own, f, c, dev, st));
    CHK(cuhe_hip_allgather_rows(Wp->sh_rows, lvl, dev, st_));
    return shard_stage2(dst_own, Wp->sh_raw, Wp->sh_a, Wp->sh_rows, lvl, f, c, dev, st);
}
// ---- one process, several devices (multiGPUs(n)): a, b = ct rows of ALL primes on device dev0, dst = reduced CRT rows of
// all primes on dev0.  Device d works on its own stream: it pulls its operand rows over the peer link, runs stage 1,
// pulls the other devices' CRT rows once they are ready (events), runs stage 2 and pushes its result rows to dev0; the
// caller's stream continues when every device is done.  Keys and constants are resident on every device (init).
int cuhe_hip_mul_relin_sharded_inproc(uint32_t *dst, const uint64_t *a, const uint64_t *b, int lvl, int dev0, void *st_) {
    CHK(need_init(dev0));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    const int nd = G_.ndev, np = q.numCrtPrimeAt(lvl), cl = q.crtLen;
    const size_t Lc = ct_len();
    if (np < nd) return fail(CUHE_EINVAL, "%d primes at level %d cannot be split over %d devices", np, lvl, nd);
    // the helper stream and the two stage events are per DEVICE, not per host thread: concurrent callers enqueue one after
    // the other (the enqueue is short; the work of successive calls still overlaps on the devices' streams)
    static std::mutex enqueue_mu;
    std::lock_guard<std::mutex> enqueue_lock(enqueue_mu);
    hipStream_t st0 = S(st_);
    std::vector<Workspace *> W(nd, nullptr);
    std::vector<hipStream_t> sd(nd, nullptr);
    for (int d = 0; d < nd; ++d) {
        CHK(need_init(d));
        DevCtx &D = G_.dev[d];
        if (!D.sh_stream) {
            HIPCHK(hipStreamCreateWithFlags(&D.sh_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&D.sh_e1, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&D.sh_e2, hipEventDisableTiming));
        }
        sd[d] = d == dev0 ? st0 : D.sh_stream;
        CHK(workspace(d, sd[d], &W[d]));
        CHK(ws_shard(*W[d]));
    }
    auto peer = [&](void *dp, int dd, const void *sp, int sdv, size_t bytes, hipStream_t s) -> int {
        if (G_.virtual_devices || dd == sdv) HIPCHK(hipMemcpyAsync(dp, sp, bytes, hipMemcpyDeviceToDevice, s));
        else HIPCHK(hipMemcpyPeerAsync(dp, phys_dev(dd), sp, phys_dev(sdv), bytes, s));
        return CUHE_OK;
    };
    // operands ready on dev0
    CHK(set_dev(dev0));
    Workspace &W0 = *W[dev0];
    if (!W0.ev_in) HIPCHK(hipEventCreateWithFlags(&W0.ev_in, hipEventDisableTiming));
    HIPCHK(hipEventRecord(W0.ev_in, st0));
    for (int d = 0; d < nd; ++d) {                      // stage 1 everywhere
        int f, c; comm::shard_bounds(np, nd, d, &f, &c);
        CHK(set_dev(d));
        const u64 *ao = (const u64 *)a + (size_t)f * Lc, *bo = (const u64 *)b + (size_t)f * Lc;
        if (d != dev0) {
            HIPCHK(hipStreamWaitEvent(sd[d], W0.ev_in, 0));
            CHK(peer(W[d]->sh_a, d, ao, dev0, (size_t)c * Lc * sizeof(u64), sd[d]));
            CHK(peer(W[d]->sh_b, d, bo, dev0, (size_t)c * Lc * sizeof(u64), sd[d]));
            ao = W[d]->sh_a; bo = W[d]->sh_b;
        }
        CHK(shard_stage1(W[d]->sh_rows, W[d]->sh_a, ao, bo, f, c, d, sd[d]));
        HIPCHK(hipEventRecord(G_.dev[d].sh_e1, sd[d]));
    }
    for (int e = 0; e < nd; ++e) {                      // the exchange, then stage 2
        int fe, ce; comm::shard_bounds(np, nd, e, &fe, &ce);
        CHK(set_dev(e));
        for (int d = 0; d < nd; ++d) {
            if (d == e) continue;
            int f, c; comm::shard_bounds(np, nd, d, &f, &c);
            HIPCHK(hipStreamWaitEvent(sd[e], G_.dev[d].sh_e1, 0));
            CHK(peer(W[e]->sh_rows + (size_t)f * cl, e, W[d]->sh_rows + (size_t)f * cl, d, (size_t)c * cl * sizeof(u32), sd[e]));
        }
        u32 *out = e == dev0 ? dst + (size_t)fe * cl : W[e]->sh_out;
        CHK(shard_stage2(out, W[e]->sh_raw, W[e]->sh_a, W[e]->sh_rows, lvl, fe, ce, e, sd[e]));
        if (e != dev0) {
            CHK(peer(dst + (size_t)fe * cl, dev0, out, e, (size_t)ce * cl * sizeof(u32), sd[e]));
            HIPCHK(hipEventRecord(G_.dev[e].sh_e2, sd[e]));
        }
    }
    CHK(set_dev(dev0));
    for (int e = 0; e < nd; ++e) if (e != dev0) HIPCHK(hipStreamWaitEvent(st0, G_.dev[e].sh_e2, 0));
    return CUHE_OK;
}

// ---------------------------------------------------------------- batched primitives
int cuhe_hip_ntt_prepare(int len, int dev) {
    CHK(set_dev(dev));
    return ensure_ntt(dev, len, 1 << 20);
}
int cuhe_hip_set_ntt_chunk(int chunk) {
    G_.ntt_chunk = chunk;
    return CUHE_OK;
}
int cuhe_hip_set_onewg(int mode, int rows64k) {
    if (mode < 0 || mode > 2 || rows64k < 0 || rows64k > 2) return fail(CUHE_EINVAL, "mode %d, rows64k %d", mode, rows64k);
    G_.onewg = mode; G_.onewg64 = rows64k;
    return CUHE_OK;
}
int cuhe_hip_set_ntt_overlap(int on) { G_.ntt_overlap = on != 0; return CUHE_OK; }
int cuhe_hip_set_ll_rows(int rows) { if (rows < 0) return fail(CUHE_EINVAL, "rows %d", rows); g_ll_rows = rows; return CUHE_OK; }
int cuhe_hip_ntt_fwd_batched(uint64_t *dst, const uint32_t *src, int len, int batch, long src_stride, int dev, void *st) {
    CHK(set_dev(dev));
    if (lg_index(len) < 0) return fail(CUHE_EINVAL, "length %d", len);
    if (src_stride < len / 2) return fail(CUHE_EINVAL, "src_stride %ld < len/2", src_stride);
    return run_ntt(len, kSrcU32Ext, dst, src, batch, src_stride, len, len, 0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_ntt_inv_batched(uint32_t *dst, const uint64_t *src, int len, int batch, long dst_stride, int nstore,
                             int prime0, int dev, void *st) {
    CHK(need_init(dev));
    if (lg_index(len) < 0) return fail(CUHE_EINVAL, "length %d", len);
    if (prime0 < 0 || prime0 + batch > G_.prm.numCrtPrime) return fail(CUHE_EINVAL, "prime range [%d,%d)", prime0, prime0 + batch);
    return run_ntt(len, kSrcU64Neg, dst, src, batch, len, dst_stride, nstore, prime0, WindowArgs{0, 0, 0}, dev, S(st));
}
int cuhe_hip_time_ntt_fwd(uint64_t *dst, const uint32_t *src, int len, int batch, int iters, int dev, void *st,
                          float *ms1, float *ms2, float *mst) {
    CHK(set_dev(dev));
    if (lg_index(len) < 0) return fail(CUHE_EINVAL, "length %d", len);
    EvTimer tm; tm.on = true;
    for (int it = 0; it < iters; ++it)
        CHK(run_ntt(len, kSrcU32Ext, dst, src, batch, len / 2, len, len, 0, WindowArgs{0, 0, 0}, dev, S(st), &tm));
    HIPCHK(hipStreamSynchronize(S(st)));
    float a = 0, b = 0, tot = 0;
    for (size_t i = 0; i + 2 < tm.ev.size(); i += 3) {
        float t1 = 0, t2 = 0;
        hipEventElapsedTime(&t1, tm.ev[i], tm.ev[i + 1]);
        hipEventElapsedTime(&t2, tm.ev[i + 1], tm.ev[i + 2]);
        a += t1; b += t2;
    }
    for (auto e : tm.ev) hipEventDestroy(e);
    // whole pipelined region (pass 1 / pass 2 overlapped as in production), bracketed on the launch stream
    hipEvent_t t0, t1;
    HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
    HIPCHK(hipEventRecord(t0, S(st)));
    for (int it = 0; it < iters; ++it)
        CHK(run_ntt(len, kSrcU32Ext, dst, src, batch, len / 2, len, len, 0, WindowArgs{0, 0, 0}, dev, S(st), nullptr));
    HIPCHK(hipEventRecord(t1, S(st)));
    HIPCHK(hipEventSynchronize(t1));
    hipEventElapsedTime(&tot, t0, t1);
    hipEventDestroy(t0); hipEventDestroy(t1);
    if (ms1) *ms1 = a;
    if (ms2) *ms2 = b;
    if (mst) *mst = tot;
    return CUHE_OK;
}

// ---------------------------------------------------------------- field test hooks
static int modp_op(int op, uint64_t *z, const uint64_t *x, const uint64_t *y, int l, size_t n, int dev, void *st) {
    CHK(set_dev(dev));
    const int grid = (int)((n + 255) / 256);
    if (op == 0) hipLaunchKernelGGL((k_modp_test<0>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, l, n);
    else if (op == 1) hipLaunchKernelGGL((k_modp_test<1>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, l, n);
    else if (op == 2) hipLaunchKernelGGL((k_modp_test<2>), dim3(grid), dim3(256), 0, S(st), (u64 *)z, (const u64 *)x, (const u64 *)y, l, n);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_modp_add(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *st) { return modp_op(0, z, x, y, 0, n, dev, st); }
int cuhe_hip_modp_sub(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *st) { return modp_op(1, z, x, y, 0, n, dev, st); }
int cuhe_hip_modp_mul(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *st) { return modp_op(2, z, x, y, 0, n, dev, st); }

int cuhe_hip_modp_shl(uint64_t *z, const uint64_t *x, int l, size_t n, int dev, void *st) {
    CHK(set_dev(dev));
    if (l < 0 || l >= 192 || l % 3) return fail(CUHE_EINVAL, "shift %d: multiples of 3 in [0,192) (cuhe/ModP.h:151)", l);
    bool done = false;
    shl_dispatch<0>(l, z, x, n, S(st), done);
    HIPCHK(hipGetLastError());
    return done ? CUHE_OK : fail(CUHE_EINVAL, "shift %d", l);
}

}  // extern "C"
