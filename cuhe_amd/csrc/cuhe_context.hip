// cuhe_context.hip -- the library's state behind the C ABI (include/cuhe_hip.h): parameters, device tables, per-thread
// workspaces, the pooled allocator, streams.  Replaces cuhe/DeviceManager.cu and the upload half of cuhe/Base.cu
// (cuhe/Base.cu:40-305, cuhe/DeviceManager.cu:40-138).  One of three translation units: cuhe_transforms.hip (NTT launch
// sequencing), cuhe_keyswitch.hip (CRT / ICRT, relinearisation, batched and sharded chains).
#include "cuhe_internal.hpp"
#include <sched.h>
#include <dirent.h>
#include "comm.hpp"

namespace cuhe_impl {

// ------------------------------------------------------------------ errors
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

Global G_;

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
thread_local int tls_lane = 0;
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
// Barrett / inttResult scratch for `rows` polynomial rows (at least one level-0 ciphertext)
int ws_barrett(Workspace &w, int rows) {
    const Params &q = G_.prm;
    size_t need = (size_t)std::max(rows, q.numCrtPrime);
    if (w.n_barrett >= need && w.hold) return CUHE_OK;
    if (w.hold) need = std::max(need, 2 * w.n_barrett);          // geometric growth: the re-allocation waits for the device
    size_t a = 0, b = 0, c = 0, d = 0;
    CHK(ws_grow(&w.b_mq, &a, need * q.nttLen)); CHK(ws_grow(&w.b_crt, &b, need * q.nttLen));
    CHK(ws_grow(&w.b_ntt, &c, need * q.nttLen)); CHK(ws_grow(&w.hold, &d, need * q.nttLen));
    w.n_barrett = need;
    return CUHE_OK;
}
// window rows and their transforms for `cts` ciphertexts
int ws_relin(Workspace &w, int cts) {
    const Params &q = G_.prm;
    if (w.n_relin >= (size_t)cts && w.relin) return CUHE_OK;
    size_t a = 0, b = 0;
    if (w.relin) { ws_retire(w.relin); w.relin = nullptr; }
    if (w.win) { ws_retire(w.win); w.win = nullptr; }
    CHK(ws_grow(&w.relin, &a, (size_t)cts * q.numEvalKey * q.nttLen)); CHK(ws_grow(&w.win, &b, (size_t)cts * q.numEvalKey * q.crtLen));
    w.n_relin = cts;
    return CUHE_OK;
}
static std::mutex g_retired_mu;
static std::vector<std::pair<int, void *>> g_retired;              // (physical device, pointer)
void ws_retire(void *p) {
    if (!p) return;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = -1; }
    std::lock_guard<std::mutex> lk(g_retired_mu);
    g_retired.push_back({d, p});
}
// the device `phys` is idle (or everything goes: phys < 0, at shutdown): the retired buffers can be freed
static void free_retired(int phys) {
    std::vector<std::pair<int, void *>> take;
    {
        std::lock_guard<std::mutex> lk(g_retired_mu);
        for (size_t i = 0; i < g_retired.size();)
            if (phys < 0 || g_retired[i].first == phys || g_retired[i].first < 0) { take.push_back(g_retired[i]); g_retired[i] = g_retired.back(); g_retired.pop_back(); } else ++i;
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto &e : take) {
        if (e.first >= 0 && (!have || e.first != cur)) { if (hipSetDevice(e.first) != hipSuccess) (void)hipGetLastError(); else cur = e.first; }
        if (hipFree(e.second) != hipSuccess) (void)hipGetLastError();
    }
}
int ws_slab(Workspace &w, int li, int which, size_t bytes, u64 **out) {        // grow-only
    if (w.slab_bytes[li][which] < bytes) {
        // grow geometrically, so that a caller whose row counts creep up (the gate scheduler's batches) pays for a handful of
        // re-allocations, not one per new maximum; the outgrown slab is retired (ws_retire), not freed
        if (w.slab[li][which]) { ws_retire(w.slab[li][which]); bytes = std::max(bytes, 2 * w.slab_bytes[li][which]); }
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
                    w->sh_a, w->sh_b, w->sh_rows, w->sh_raw, w->sh_out, w->pair_cnt, w->rc_acc, w->ls_crt, w->ls_ntt};
    for (void *p : ptrs) if (p) hipFree(p);
    if (w->ev) hipEventDestroy(w->ev);
    if (w->ev_lane) hipEventDestroy(w->ev_lane);
    if (w->ev_in) hipEventDestroy(w->ev_in);
    if (w->lane_stream) hipStreamDestroy(w->lane_stream);
    delete w;
}


// ------------------------------------------------------------------ helpers
int need_init(int dev) {
    if (!G_.inited) return fail(CUHE_ENOTINIT, "cuhe_hip_init has not been called");
    CHK(set_dev(dev));
    if (!G_.dev[dev].ready) return fail(CUHE_ENOTINIT, "device %d not initialised", dev);
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
        // the same constants for the matrix-core form (icrt_mfma.cuh): signed base-256 digits of 128^a (M / p_i), laid out as
        // the first operand of v_mfma_i32_32x32x32_i8 reads them; per-prime constants by K step; NM = 2^(32 NW) - M
        uint32_t pmax = 0;
        for (int i = 0; i < I.np; ++i) pmax = std::max(pmax, hp[i]);
        I.tiles = (I.W + 1 + 7) / 8; I.ksteps = (I.np + 7) / 8;
        if (pmax < (1u << 28) && I.np <= 128) {
            const int NW = 8 * I.tiles, WH = 4 * I.tiles, ND = 4 * NW, ks = I.ksteps;
            std::vector<signed char> dg((size_t)ks * 8 * 4 * ND, 0);                 // [prime][a][digit]
            std::vector<u32> cw(NW + 1);
            for (int i = 0; i < I.np; ++i)
                for (int a = 0; a < 4; ++a) {
                    std::fill(cw.begin(), cw.end(), 0u);
                    for (int k = 0; k < I.W; ++k) {
                        const u64 v = (u64)hmi[(size_t)i * W4 + k] << (7 * a);
                        cw[k] |= (u32)v; cw[k + 1] |= (u32)(v >> 32);
                    }
                    int carry = 0;
                    signed char *o = &dg[((size_t)i * 4 + a) * ND];
                    for (int d = 0; d < ND; ++d) {
                        const int v = (int)((cw[d >> 2] >> (8 * (d & 3))) & 0xff) + carry;
                        carry = v >= 128;
                        o[d] = (signed char)(carry ? v - 256 : v);
                    }
                    if (carry) return fail(CUHE_EINVAL, "ICRT digit table: value does not fit %d words", NW);
                }
            std::vector<unsigned char> tab((size_t)I.tiles * ks * 1024, 0);
            for (int m = 0; m < I.tiles; ++m)
                for (int s = 0; s < ks; ++s)
                    for (int l = 0; l < 64; ++l) {
                        const int rho = l & 31, hk = l >> 5, d = 4 * (((rho >> 2) & 1) * WH + 4 * m + (rho >> 3)) + (rho & 3);
                        for (int e = 0; e < 4; ++e)
                            for (int a = 0; a < 4; ++a)
                                tab[(((size_t)m * ks + s) * 64 + l) * 16 + e * 4 + a] = (unsigned char)dg[((size_t)(8 * s + 4 * hk + e) * 4 + a) * ND + d];
                    }
            std::vector<IcrtPrimeConst> hpc((size_t)ks * 8);
            for (int i = 0; i < ks * 8; ++i)
                hpc[i] = i < I.np ? IcrtPrimeConst{hp[i], hbi[i], (u32)(((u64)hbi[i] << 32) / hp[i]), 0u, hrp[i], 0}
                                  : IcrtPrimeConst{hp[0], 0u, 0u, 0u, 0.0, 0};
            std::vector<u32> hnm(NW);
            u64 c = 1;
            for (int k = 0; k < NW; ++k) { c += (u64)(u32)~(k < I.W ? hM[k] : 0u); hnm[k] = (u32)c; c >>= 32; }
            CHK(upload(&I.dig, tab)); CHK(upload(&I.pc, hpc)); CHK(upload(&I.nm, hnm));
        }
    }
    // ---- transforms + scratch (initNtt: cuhe/Operations.cu:173-184)
    CHK(ensure_ntt(dev, L, pnum));
    if (G_.nc) CHK(ensure_twist(dev, n));
    // the tables of the one-workgroup forms of every sub-transform size this parameter set can reach (rows of nttLen points, their halves,
    // the half-length transforms of the folded reductions and of the key-switch windows): here, like the reference's initNtt, not inside the
    // first gate that needs them -- a homomorphic PRINCE block lost 6 ms to three such first uses (profiles/r05_prince_gaps.txt)
    for (int lgh = 12; lgh <= 15 && (1 << lgh) <= L; ++lgh) {
        CHK(ensure_onewg(D.ow[lgh - 12], lgh));
        // ... and the kernels themselves (HIP loads a code object and builds a kernel's function at its first use)
        const hipError_t we = lgh == 12 ? cuhe::ow_prewarm_12() : lgh == 13 ? cuhe::ow_prewarm_13() : lgh == 14 ? cuhe::ow_prewarm_14() : cuhe::ow_prewarm_15();
        if (we != hipSuccess) return fail(CUHE_EHIP, "one-workgroup kernels of 2^%d points: %s", lgh, hipGetErrorString(we));
    }
    if (G_.nc && n == 65536) CHK(ensure_onewg_twist(D.ow[3], 15));
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

}  // namespace cuhe_impl

using namespace cuhe_impl;

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

// ---- the CPUs local to a device.  After the library is up the PCI function comes from HIP (hipDeviceGetPCIBusId); BEFORE any HIP call of the
// process it is read from sysfs alone -- AMD display / processing-accelerator functions in bus order, the n-th one the runtime will show (a plain
// integer list in HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES is followed; anything else: unknown) -- because the first HIP call already places the
// runtime's host-side state (kernel-argument pools, signals, its helper threads) on the NUMA node of the thread that makes it.
static bool read_line(const char *path, char *buf, size_t n) {
    buf[0] = 0;
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)n, f) != NULL;
    fclose(f);
    size_t k = strlen(buf);
    while (k && (buf[k - 1] == '\n' || buf[k - 1] == ' ')) buf[--k] = 0;
    return ok;
}
static std::string sysfs_gpu_bdf(int phys) {
    int pick = phys;
    for (const char *name : {"ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"}) {
        const char *e = getenv(name);
        if (!e || !*e) continue;
        std::vector<int> ids;
        for (const char *q = e; *q;) {
            char *end; long v = strtol(q, &end, 10);
            if (end == q || (*end && *end != ',')) return std::string();           // (UUIDs and the like: not guessed)
            ids.push_back((int)v); q = *end ? end + 1 : end;
        }
        if (pick < 0 || pick >= (int)ids.size()) return std::string();
        pick = ids[pick];
    }
    std::vector<std::string> bdfs;
    if (DIR *d = opendir("/sys/bus/pci/devices")) {
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            char path[320], v[64], c[64];
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/vendor", e->d_name);
            if (!read_line(path, v, sizeof v) || strcmp(v, "0x1002") != 0) continue;
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/class", e->d_name);
            if (!read_line(path, c, sizeof c)) continue;
            if (strncmp(c, "0x03", 4) != 0 && strncmp(c, "0x12", 4) != 0) continue;   // display controllers, processing accelerators (MI300 / MI355X: 0x1200xx)
            bdfs.push_back(e->d_name);
        }
        closedir(d);
    }
    std::sort(bdfs.begin(), bdfs.end());
    return pick >= 0 && pick < (int)bdfs.size() ? bdfs[pick] : std::string();
}
static std::atomic<bool> hip_in_use{false};             // some entry point of this library has called into HIP already
int cuhe_hip_device_local_cpus(int dev, char *buf, size_t buf_bytes) {
    if (!buf || buf_bytes == 0) return fail(CUHE_EINVAL, "device_local_cpus: no buffer");
    buf[0] = 0;
    if (dev < 0) return fail(CUHE_EINVAL, "device_local_cpus: bad device");
    std::string bdf;
    if (G_.inited || hip_in_use.load()) {
        char b[64] = {0};
        if (hipDeviceGetPCIBusId(b, (int)sizeof b, phys_dev(dev)) == hipSuccess) bdf = b; else (void)hipGetLastError();
        for (char &c : bdf) if (c >= 'A' && c <= 'F') c = (char)(c - 'A' + 'a');      // sysfs names are lower case
    } else bdf = sysfs_gpu_bdf(phys_dev(dev));
    if (bdf.empty()) return CUHE_OK;
    char path[320];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf.c_str());
    read_line(path, buf, buf_bytes);
    return CUHE_OK;
}
// narrow `want` to the listed CPUs the thread may use; false when nothing would change
static bool local_mask(int dev, cpu_set_t *want) {
    char list[1024];
    if (cuhe_hip_device_local_cpus(dev, list, sizeof list) != CUHE_OK || !list[0]) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed); CPU_ZERO(want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int chosen = 0; const int had = CPU_COUNT(&allowed);
    for (const char *p = list; *p;) {                       // "a-b,c,d-e"
        char *e; long a = strtol(p, &e, 10); if (e == p) break;
        long b = a; if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); if (e == p) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, want); ++chosen; }
        if (*e != ',') break;
        p = e + 1;
    }
    return chosen != 0 && chosen != had;                     // (nothing local is allowed / everything allowed is local already: leave it)
}
int cuhe_hip_pin_thread_to_device(int dev) {
    cpu_set_t want;
    if (!local_mask(dev, &want)) return 0;
    return sched_setaffinity(0, sizeof want, &want) == 0 ? 1 : 0;
}
// The thread that brings the library up (multiGPUs / initCuHE: the first HIP calls of a client process) runs on the CPUs local to its device
// WHILE it does so -- the HIP runtime's host-side state and helper threads then live next to the GPU -- and gets its own affinity back afterwards
// (CUHE_PIN_CLIENT=1: it stays narrowed; 0: untouched).  profiles/r06_numa_pinning.txt.
namespace {
struct ClientPin {
    cpu_set_t saved; bool narrowed = false;
    ClientPin() {
        static const int mode = getenv("CUHE_PIN_CLIENT") ? atoi(getenv("CUHE_PIN_CLIENT")) : 2;
        if (mode <= 0) return;
        cpu_set_t want;
        if (sched_getaffinity(0, sizeof saved, &saved) != 0 || !local_mask(0, &want)) return;
        narrowed = sched_setaffinity(0, sizeof want, &want) == 0 && mode == 2;
    }
    ~ClientPin() { if (narrowed) sched_setaffinity(0, sizeof saved, &saved); }
};
}
int cuhe_hip_multi_gpus(int num) {
    int cnt = 0;
    ClientPin local;
    hip_in_use.store(true);
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

static bool same_inputs(const Params &a, const Params &b) {
    return a.depth == b.depth && a.modMsg == b.modMsg && a.logRelin == b.logRelin && a.logCoeffMin == b.logCoeffMin &&
           a.logCoeffCut == b.logCoeffCut && a.mSize == b.mSize;
}
int cuhe_hip_same_ring(const int32_t *modulus, int ncoeffs) {
    if (!G_.inited || !G_.params_set || !same_inputs(G_.prm, G_.prm_init) || G_.nc_mode != G_.nc_mode_init) return 0;
    if (!modulus) return G_.modulus == host::cyclotomic(G_.prm.mSize) ? 1 : 0;
    return (size_t)ncoeffs == G_.modulus.size() && std::equal(modulus, modulus + ncoeffs, G_.modulus.begin()) ? 1 : 0;
}

int cuhe_hip_init(const int32_t *modulus, int ncoeffs) {
    if (!G_.params_set) return fail(CUHE_EINVAL, "setParameters must precede initCuHE");
    ClientPin local;
    hip_in_use.store(true);
    if (G_.inited) {
        // A second scheme object built in the same process (examples/DHS/simple_DHS.cu:176,186: CuDHS(string) runs setParameters
        // + initCuHE again while the first object is alive and is used afterwards): the reference re-creates its tables beside the
        // old ones and every live object stays valid.  On the SAME ring that is a no-op here -- tables, resident evaluation keys
        // and every block clients hold are kept; another ring needs cuhe_hip_shutdown first (it frees all of them).
        if (cuhe_hip_same_ring(modulus, ncoeffs) == 1) return CUHE_OK;
        return fail(CUHE_EINVAL, "already initialised on another ring (cuhe_hip_shutdown first)");
    }
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
    // k_modswitch lifts (src - dirty) by a multiple of p in [2^42, 2^43) instead of branching on its sign: |src - dirty| < 2^32 +
    // modMsg * p_t has to stay below 2^42 (the reference's int arithmetic wraps much earlier, Base.cu:1117-1123)
    {
        uint64_t pmax = 0;
        for (uint32_t p : G_.primes) pmax = std::max<uint64_t>(pmax, p);
        if ((uint64_t)q.modMsg * pmax + (1ull << 32) >= (1ull << 42))
            return fail(CUHE_EINVAL, "modMsg=%d is too large for the modulus switch with %d-bit CRT primes (modMsg * p must stay below 2^42)", q.modMsg, q.logCrtPrime);
    }
    G_.coeffModulus.assign(q.depth, BigU(1));                      // cuhe/Operations.cu:81-90
    for (int i = 0; i < q.depth; ++i)
        for (int j = 0; j < q.numCrtPrime - i; ++j) G_.coeffModulus[i].mul_small(G_.primes[j]);
    G_.dev.resize(G_.ndev);
    G_.inited = true; G_.prm_init = G_.prm; G_.nc_mode_init = G_.nc_mode;
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
// bumped by every cuhe_hip_shutdown: device blocks handed out before it are gone (callers that keep blocks across calls compare it)
unsigned long long cuhe_hip_generation(void) { return (unsigned long long)G_.generation; }
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
        for (auto &t : D.ow) { hipFree(t.TW1f); hipFree(t.TW1i); hipFree(t.TW1h); hipFree(t.TW2); hipFree(t.TW1g); hipFree(t.TW1hi); }
        for (Workspace *w : D.spaces) free_workspace(w);
        for (void *p : ptrs) if (p) hipFree(p);
        for (auto &I : D.icrt) { hipFree(I.M); hipFree(I.mi); hipFree(I.bi); hipFree(I.rp); hipFree(I.dig); hipFree(I.pc); hipFree(I.nm); }
        for (auto &kv : D.freeBlocks) hipFree(kv.second);
        for (auto &sb : D.streamBlocks) for (auto &kv : sb.second) hipFree(kv.second);     // parked in stream order
        for (auto &kv : D.allocated) hipFree(kv.first);
        std::set<hipStream_t> keep; keep.swap(D.ownStreams);       // streams outlive an initialisation (the gate scheduler keeps its workers' streams)
        D = DevCtx();
        D.ownStreams.swap(keep);
    }
    free_retired(-1);
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
static long long g_alloc_counters[4];      // hipMalloc calls, hits in the settled pool, hits in a stream's parked blocks, idle streams settled
// `count` blocks of `bytes` bytes into the settled pool of device `dev` (the pooled allocator's own reserve): the reference's allocator
// takes the device's whole memory when it starts (cuhe/DeviceManager.cu:56-64); here startAllocator takes a bounded reserve of the
// one block size the classes use while it is on, so that a client's first operations do not pay hipMalloc one block at a time
int cuhe_hip_reserve_blocks(int dev, size_t bytes, int count) {
    CHK(need_init(dev));
    if (bytes == 0 || count < 0) return fail(CUHE_EINVAL, "reserve_blocks(%zu bytes, %d)", bytes, count);
    CHK(set_dev(dev));
    DevCtx &D = G_.dev[dev];
    for (int i = 0; i < count; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }       // as much as there is: a reserve, not a requirement
        ++g_alloc_counters[0];
        std::lock_guard<std::mutex> lk(G_.mu);
        D.freeBlocks.insert({bytes, p}); D.cachedBytes += bytes;
    }
    return CUHE_OK;
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
int cuhe_hip_alloc_counters(long long *out4) {
    std::lock_guard<std::mutex> lk(G_.mu);
    for (int i = 0; i < 4; ++i) out4[i] = g_alloc_counters[i];
    return CUHE_OK;
}
// may the allocator touch this stream handle on its own?  (only handles known to be alive)
static inline bool probeable(const DevCtx &D, hipStream_t st) { return st == nullptr || D.ownStreams.count(st) != 0; }
static std::atomic<long> g_alloc_fail_after{-1};
int cuhe_hip_set_alloc_fail_after(long n) { g_alloc_fail_after.store(n < 0 ? -1 : n); return CUHE_OK; }
void *cuhe_hip_malloc(int dev, size_t bytes) {
    if (set_dev(dev) != CUHE_OK) return nullptr;
    if (g_alloc_fail_after.load(std::memory_order_relaxed) >= 0 && g_alloc_fail_after.fetch_sub(1) == 0) {   // test hook: an allocation failure on demand
        fail(CUHE_EHIP, "hipMalloc(%zu) failed (injected by cuhe_hip_set_alloc_fail_after)", bytes);
        return nullptr;
    }
    DevCtx &D = G_.dev[dev];
    std::unique_lock<std::mutex> lk(G_.mu);
    auto it = D.freeBlocks.find(bytes);
    if (it == D.freeBlocks.end()) {
        // a miss: blocks of this size parked in the order of a stream that has gone idle since are free (several streams
        // of one host thread under the gate scheduler, none of which is ever synchronised by the client)
        for (auto sb = D.streamBlocks.begin(); sb != D.streamBlocks.end();) {
            if (sb->second.find(bytes) != sb->second.end() && probeable(D, sb->first) && hipStreamQuery(sb->first) == hipSuccess) {
                for (auto &kv : sb->second) D.freeBlocks.insert(kv);
                sb = D.streamBlocks.erase(sb);
            } else { (void)hipGetLastError(); ++sb; }
        }
        it = D.freeBlocks.find(bytes);
    }
    if (it != D.freeBlocks.end()) {
        void *p = it->second;
        D.freeBlocks.erase(it); D.cachedBytes -= bytes; D.allocated[p] = bytes;
        ++g_alloc_counters[1];
        return p;
    }
    void *p = nullptr;
    ++g_alloc_counters[0];
    lk.unlock();                                          // hipMalloc takes tens of microseconds to milliseconds: not under the library's lock
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    lk.lock();
    if (e != hipSuccess) {
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
                ++g_alloc_counters[2];
                return p;
            }
        }
        // Nothing parked for this stream and nothing settled: take a block parked in the order of ANOTHER stream and
        // make this stream wait for everything enqueued there so far (the block's last use is part of it).  Several
        // streams of one client under the gate scheduler hand blocks to each other all the time -- a task frees on its
        // own stream what a task on another stream allocated -- and none of them is ever synchronised, so without this
        // every such hand-over ends in hipMalloc (hundreds of microseconds, under the library's lock).
        if (D.freeBlocks.find(bytes) == D.freeBlocks.end()) {
            for (auto &other : D.streamBlocks) {
                if (other.first == S(st) || !probeable(D, other.first)) continue;
                auto it = other.second.find(bytes);
                if (it == other.second.end()) continue;
                hipEvent_t ev = nullptr;
                if (!D.fenceEvents.empty()) { ev = D.fenceEvents.back(); D.fenceEvents.pop_back(); }
                else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); break; }
                const bool ok = hipEventRecord(ev, other.first) == hipSuccess && hipStreamWaitEvent(S(st), ev, 0) == hipSuccess;
                D.fenceEvents.push_back(ev);           // the wait captured the record: the event can serve again at once
                if (!ok) { (void)hipGetLastError(); break; }
                void *p = it->second;
                other.second.erase(it); D.cachedBytes -= bytes; D.allocated[p] = bytes;
                ++g_alloc_counters[3];
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
    { std::lock_guard<std::mutex> lk(G_.mu); G_.dev[dev].ownStreams.insert(s); }
    *out = (void *)s;
    return CUHE_OK;
}
int cuhe_hip_stream_destroy(int dev, void *st) {
    CHK(set_dev(dev));
    if (st) {
        HIPCHK(hipStreamSynchronize(S(st))); settle_stream_blocks(G_.dev[dev], S(st));
        { std::lock_guard<std::mutex> lk(G_.mu); G_.dev[dev].ownStreams.erase(S(st)); }
        HIPCHK(hipStreamDestroy(S(st)));
    }
    return CUHE_OK;
}
int cuhe_hip_stream_sync(int dev, void *st) {
    CHK(set_dev(dev));
    HIPCHK(hipStreamSynchronize(S(st)));
    settle_stream_blocks(G_.dev[dev], S(st));
    return CUHE_OK;
}

int cuhe_hip_event_create(int dev, void **out) {
    CHK(set_dev(dev));
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = (void *)e;
    return CUHE_OK;
}
int cuhe_hip_event_destroy(int dev, void *ev) { CHK(set_dev(dev)); if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev)); return CUHE_OK; }
int cuhe_hip_event_record(int dev, void *ev, void *st) { CHK(set_dev(dev)); HIPCHK(hipEventRecord((hipEvent_t)ev, S(st))); return CUHE_OK; }
int cuhe_hip_stream_wait_event(int dev, void *st, void *ev) { CHK(set_dev(dev)); HIPCHK(hipStreamWaitEvent(S(st), (hipEvent_t)ev, 0)); return CUHE_OK; }
int cuhe_hip_event_sync(int dev, void *ev) { CHK(set_dev(dev)); HIPCHK(hipEventSynchronize((hipEvent_t)ev)); return CUHE_OK; }
int cuhe_hip_event_query(int dev, void *ev) {
    CHK(set_dev(dev));
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return CUHE_OK;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    return fail(CUHE_EHIP, "hipEventQuery failed : %s", hipGetErrorString(e));
}

// ---- cuhe_hip_probe_valu: the dense 64-bit integer stream the transforms are measured against (include/cuhe_hip.h)
}  // extern "C"
namespace {
constexpr int kProbeIters = 2048;          // x 24 wave-instructions
__global__ __launch_bounds__(256) void k_probe_valu(unsigned *out, unsigned long long *ticks, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kProbeIters; ++i) {
        asm volatile("v_mad_u64_u32 %8, vcc, %0, %1, %8\n v_mad_u64_u32 %9, vcc, %1, %2, %9\n v_mad_u64_u32 %10, vcc, %2, %3, %10\n v_mad_u64_u32 %11, vcc, %3, %4, %11\n"
                     "v_mad_u64_u32 %12, vcc, %4, %5, %12\n v_mad_u64_u32 %13, vcc, %5, %6, %13\n v_mad_u64_u32 %14, vcc, %6, %7, %14\n v_mad_u64_u32 %15, vcc, %7, %0, %15\n"
                     "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n"
                     "v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8\n"
                     "v_cmp_lt_u64 vcc, %8, %9\n v_cmp_lt_u64 vcc, %9, %10\n v_cmp_lt_u64 vcc, %10, %11\n v_cmp_lt_u64 vcc, %11, %12\n"
                     "v_cmp_lt_u64 vcc, %12, %13\n v_cmp_lt_u64 vcc, %13, %14\n v_cmp_lt_u64 vcc, %14, %15\n v_cmp_lt_u64 vcc, %15, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : : "vcc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
}
}  // namespace
extern "C" {
int cuhe_hip_probe_valu(int dev, int waves_per_simd, int millis, double *lane_instr_per_s, double *shader_mhz, double *cycles_per_instr) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || millis < 1 || !lane_instr_per_s || !shader_mhz || !cycles_per_instr) return fail(CUHE_EINVAL, "probe_valu(%d waves, %d ms)", waves_per_simd, millis);
    if (hipSetDevice(G_.dev_base + (G_.virtual_devices ? 0 : dev)) != hipSuccess) return fail(CUHE_EHIP, "hipSetDevice");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, G_.dev_base + (G_.virtual_devices ? 0 : dev)));
    const int blocks = prop.multiProcessorCount * waves_per_simd;      // a 256-thread block = one wave per SIMD of a CU
    unsigned *out = nullptr; unsigned long long *ticks = nullptr;
    HIPCHK(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(unsigned)));
    HIPCHK(hipMalloc((void **)&ticks, (size_t)blocks * 4 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](int reps, float *ms) -> int {
        HIPCHK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_valu, dim3(blocks), dim3(256), 0, 0, out, ticks, 1u);
        HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(ms, e0, e1));
        return CUHE_OK;
    };
    float ms = 0;
    CHK(run(3, &ms));                                                   // warm-up and a duration estimate
    const int reps = std::max(4, (int)(millis / std::max(ms / 3, 1e-3f)));
    CHK(run(reps, &ms));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    HIPCHK(hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0; for (unsigned long long t : h) avg += (double)t; avg /= (double)h.size();
    const double instr = (double)kProbeIters * 24;
    *lane_instr_per_s = (double)reps * blocks * 256.0 * instr / (ms * 1e-3);
    *shader_mhz = avg / (ms * 1e3 / reps);                              // ticks of one wave per microsecond of one kernel
    *cycles_per_instr = avg / instr / waves_per_simd;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(out); hipFree(ticks);
    return CUHE_OK;
}

// ---- cuhe_hip_probe_copy: the streaming-copy ceiling of this box (include/cuhe_hip.h) -- the "measured peak" the HBM-bound kernels are
// priced against next to the 8 TB/s of the data sheet.  16-byte accesses; the launch shapes of kCopyShapes (grid-stride with 8 / 16 / 32
// workgroups per CU, one element per thread, 256 / 512 / 1024 threads, several loads in flight before the stores).
}  // extern "C"
namespace {
template <int UNROLL, bool NT> __global__ void k_probe_copy(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) { const float *p = (const float *)(src + i + u * stride);
                      v[u] = make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3)); }
            else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) { float *p = (float *)(dst + i + u * stride);
                      __builtin_nontemporal_store(v[u].x, p); __builtin_nontemporal_store(v[u].y, p + 1); __builtin_nontemporal_store(v[u].z, p + 2); __builtin_nontemporal_store(v[u].w, p + 3); }
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// a workgroup copies CONTIGUOUS chunks (UNROLL x blockDim float4 each), chunks handed out round-robin: every wave's loads of one iteration
// are UNROLL consecutive 1 KB segments instead of UNROLL segments a whole grid apart
template <int UNROLL> __global__ void k_probe_copy_chunks(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n) {
    const size_t chunk = (size_t)UNROLL * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t i = base + (size_t)u * blockDim.x + threadIdx.x; if (i < n) v[u] = src[i]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t i = base + (size_t)u * blockDim.x + threadIdx.x; if (i < n) dst[i] = v[u]; }
    }
}
struct CopyShape { int kind, unroll, threads, blocks_per_cu; const char *name; };       // blocks_per_cu 0: one element per thread (no loop)
const CopyShape kCopyShapes[] = {
    {0, 1, 256, 8, "float4, 256 thr, 8 wg/CU, grid-stride"},       {0, 4, 256, 8, "4 float4 in flight, 256 thr, 8 wg/CU"},
    {1, 4, 256, 8, "4 float4 in flight, non-temporal, 256 thr, 8 wg/CU"}, {0, 8, 256, 8, "8 float4 in flight, 256 thr, 8 wg/CU"},
    {0, 1, 256, 0, "float4, 256 thr, one element per thread"},     {0, 1, 1024, 0, "float4, 1024 thr, one element per thread"},
    {0, 1, 512, 16, "float4, 512 thr, 16 wg/CU, grid-stride"},     {0, 2, 1024, 2, "2 float4 in flight, 1024 thr, 2 wg/CU"},
    {0, 4, 1024, 2, "4 float4 in flight, 1024 thr, 2 wg/CU"},      {2, 4, 256, 8, "contiguous chunks of 4 float4 per thread, 256 thr, 8 wg/CU"},
    {2, 4, 512, 4, "contiguous chunks of 4 float4 per thread, 512 thr, 4 wg/CU"}, {2, 8, 256, 8, "contiguous chunks of 8 float4 per thread, 256 thr, 8 wg/CU"},
    {2, 2, 1024, 2, "contiguous chunks of 2 float4 per thread, 1024 thr, 2 wg/CU"}, {0, 1, 256, 32, "float4, 256 thr, 32 wg/CU, grid-stride"},
};
constexpr int kNumCopyShapes = (int)(sizeof(kCopyShapes) / sizeof(kCopyShapes[0]));
}  // namespace
extern "C" {
int cuhe_hip_probe_copy_shapes(void) { return kNumCopyShapes; }
const char *cuhe_hip_probe_copy_name(int variant) { return variant >= 0 && variant < kNumCopyShapes ? kCopyShapes[variant].name : ""; }
int cuhe_hip_probe_copy(int dev, size_t bytes, int variant, int reps, double *gb_per_s) {
    if (bytes < (1u << 20) || (bytes & 15) || reps < 1 || variant < 0 || variant >= kNumCopyShapes || !gb_per_s) return fail(CUHE_EINVAL, "probe_copy(%zu bytes, variant %d, %d reps)", bytes, variant, reps);
    const int phys = G_.dev_base + (G_.virtual_devices ? 0 : dev);
    if (hipSetDevice(phys) != hipSuccess) return fail(CUHE_EHIP, "hipSetDevice");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, phys));
    float4 *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc((void **)&a, bytes));
    if (hipMalloc((void **)&b, bytes) != hipSuccess) { hipFree(a); return fail(CUHE_EHIP, "probe_copy: hipMalloc"); }
    HIPCHK(hipMemset(a, 0x5a, bytes));
    const size_t n = bytes / 16;
    const CopyShape &S = kCopyShapes[variant];
    const unsigned blocks = S.blocks_per_cu ? (unsigned)(prop.multiProcessorCount * S.blocks_per_cu) : (unsigned)((n + S.threads - 1) / S.threads);
    auto launch = [&]() {
        const dim3 g(blocks), t(S.threads);
        if (S.kind == 2) {
            if (S.unroll == 2) hipLaunchKernelGGL((k_probe_copy_chunks<2>), g, t, 0, 0, b, a, n);
            else if (S.unroll == 4) hipLaunchKernelGGL((k_probe_copy_chunks<4>), g, t, 0, 0, b, a, n);
            else hipLaunchKernelGGL((k_probe_copy_chunks<8>), g, t, 0, 0, b, a, n);
        } else if (S.kind == 1) hipLaunchKernelGGL((k_probe_copy<4, true>), g, t, 0, 0, b, a, n);
        else if (S.unroll == 1) hipLaunchKernelGGL((k_probe_copy<1, false>), g, t, 0, 0, b, a, n);
        else if (S.unroll == 2) hipLaunchKernelGGL((k_probe_copy<2, false>), g, t, 0, 0, b, a, n);
        else if (S.unroll == 4) hipLaunchKernelGGL((k_probe_copy<4, false>), g, t, 0, 0, b, a, n);
        else hipLaunchKernelGGL((k_probe_copy<8, false>), g, t, 0, 0, b, a, n);
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    launch(); launch();
    HIPCHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *gb_per_s = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;          // bytes read + bytes written
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(a); hipFree(b);
    return hipGetLastError() == hipSuccess ? CUHE_OK : fail(CUHE_EHIP, "probe_copy: launch failed");
}

// waits for everything enqueued on the device; every block freed in stream order becomes an ordinary free block
int cuhe_hip_device_sync(int dev) {
    CHK(set_dev(dev));
    HIPCHK(hipDeviceSynchronize());
    { int phys = 0; if (hipGetDevice(&phys) == hipSuccess) free_retired(phys); else (void)hipGetLastError(); }
    DevCtx &D = G_.dev[dev];
    std::lock_guard<std::mutex> lk(G_.mu);
    for (auto &sb : D.streamBlocks) for (auto &kv : sb.second) D.freeBlocks.insert(kv);
    D.streamBlocks.clear();
    return CUHE_OK;
}

}  // extern "C"
