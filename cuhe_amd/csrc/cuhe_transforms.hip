// cuhe_transforms.hip -- launch sequencing of the transforms behind the C ABI: tables, the one-workgroup / two-pass /
// low-latency kernel choice, the folded Barrett reduction, the ciphertext-domain (cyclic or negacyclic) transforms and
// their drivers.  Replaces the NTT half of cuhe/Operations.cu (cuhe/Operations.cu:306-504) and cuhe/Base.cu:309-842.
#include "cuhe_internal.hpp"
#include "comm.hpp"

namespace cuhe_impl {

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
    const u64 l2inv = host::powP((u64)(2 * Lh), host::P - 2);
    std::vector<u64> f(Lh), fi(Lh), fh(2 * (size_t)Lh), fhi(2 * (size_t)Lh), t2(T);
    for (int ka = 0; ka < 32; ++ka)
        for (int m = 0; m < T; ++m) {
            const size_t o = (size_t)ka * T + m;
            f[o] = r[(2L * m * ka) % (2L * Lh)];                 // w_Lh^(m ka)
            fi[o] = host::mulP(f[o], linv);
            fh[o] = f[o];
            fh[Lh + o] = r[((long)m * (2 * ka + 1)) % (2L * Lh)]; // W^(m (2 ka + 1)): the odd outputs of the zero-padded transform
            fhi[o] = host::mulP(fh[o], l2inv); fhi[Lh + o] = host::mulP(fh[Lh + o], l2inv);
        }
    for (int kb = 0; kb < R; ++kb)
        for (int c = 0; c < 32; ++c) t2[(size_t)kb * 32 + c] = r[(64L * c * kb) % (2L * Lh)];      // w_T^(c kb) = w_Lh^(32 c kb)
    CHK(upload(&tab.TW1f, f)); CHK(upload(&tab.TW1i, fi)); CHK(upload(&tab.TW1h, fh)); CHK(upload(&tab.TW1hi, fhi)); CHK(upload(&tab.TW2, t2));
    tab.ready.store(1, std::memory_order_release);
    return CUHE_OK;
}
// tables of the two Lh-point halves of the NEGACYCLIC forward transform of L = 2 Lh points (ntt_onewg.cuh: StreamTwist), T = Lh / 32:
// TW1g[h][ka T + m] = psi^(m (1 + 2h + 4 ka)), c128 = psi^T, i4 = psi^Lh = +-2^48; psi = root_2len(L), the root of ensure_twist
int ensure_onewg_twist(OwTab &tab, int lgh) {
    CHK(ensure_onewg(tab, lgh));
    std::lock_guard<std::mutex> lk(G_.mu);
    if (tab.TW1g) return CUHE_OK;
    const int Lh = 1 << lgh, T = Lh / 32;
    const u64 psi = host::root_2len(2 * Lh);
    if (!psi) return fail(CUHE_EINVAL, "no primitive 2^%d-th root of unity found", lgh + 2);
    std::vector<u64> r((size_t)4 * Lh);
    r[0] = 1;
    for (size_t i = 1; i < r.size(); ++i) r[i] = host::mulP(r[i - 1], psi);
    const u64 i4 = r[Lh], p48 = (u64)1 << 48;
    if (i4 != p48 && i4 != host::P - p48) return fail(CUHE_EINVAL, "psi^%d is not +-2^48", Lh);
    std::vector<u64> g(2 * (size_t)Lh);
    for (int h = 0; h < 2; ++h)
        for (int ka = 0; ka < 32; ++ka)
            for (int m = 0; m < T; ++m) g[(size_t)h * Lh + (size_t)ka * T + m] = r[((long)m * (1 + 2 * h + 4 * ka)) & (4L * Lh - 1)];
    CHK(upload(&tab.TW1g, g));
    tab.c128 = r[T]; tab.i4neg = i4 == p48 ? 0 : 1;
    return CUHE_OK;
}
int onewg_launch(int lgh, int mode, int out, bool half, const OwArgs &a, hipStream_t st) {
    hipError_t e = lgh == 12 ? ow_launch_12(mode, out, half, a, st) : lgh == 13 ? ow_launch_13(mode, out, half, a, st)
                 : lgh == 14 ? ow_launch_14(mode, out, half, a, st) : ow_launch_15(mode, out, half, a, st);
    if (e != hipSuccess) return fail(CUHE_EHIP, "one-workgroup transform (2^%d points, source %d, store %d%s): %s", lgh, mode, out, half ? ", half" : "", hipGetErrorString(e));
    return CUHE_OK;
}

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

// what the calling thread's last transform call was dispatched to (cuhe_hip_last_dispatch_info: bench / tests)
struct DispatchInfo { const char *form = "none"; int len = 0, batch = 0, dev = 0; };
static thread_local DispatchInfo tls_dispatch;
static inline void note_dispatch(const char *form, int len, int batch) { tls_dispatch.form = form; tls_dispatch.len = len; tls_dispatch.batch = batch; }
// rendezvous counters of the persistent kernels + one slot that counts the workgroups that gave a rendezvous up
static int ws_pair_counters(Workspace &W) {
    if (W.pair_cnt) return CUHE_OK;
    CHK(ws_buffer(&W.pair_cnt, kOwPairCounters + 1));
    HIPCHK(hipMemset(W.pair_cnt, 0, (kOwPairCounters + 1) * sizeof(unsigned)));
    return CUHE_OK;
}

// one batched transform, chunked so that the pass-1 -> pass-2 slab stays cache resident
template <int LG>
int run_ntt_lg(int mode, void *dst, const void *src, int batch, long src_stride, long dst_stride, int nstore,
               int prime0, WindowArgs wa, DevCtx &D, Workspace &W, hipStream_t st, EvTimer *tm, const u64 *mul_tab, int np_mod,
               const Epilogue *ep, const RowRebase *rb) {
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
        // persistent form of the halves (every workgroup resident: 1 / 2 per CU at 32K / 16K points): from two halves per
        // workgroup on, rows aligned for the LDS-DMA of their samples
        // (32K-point rows: measured 3 % slower than one workgroup per half, 6.43 vs 6.64 M/s -- taken only when onewg64 = 3)
        const int gridp = lgh == 15 || (lgh == 14 && G_.onewg64 == 3) ? (D.cus << (15 - lgh)) & ~15 : 0;
        // (the persistent launch carries ~70 us of fill, first fetch and tail: 0.570 / 0.492 us per transform at 256 / 512 rows of
        // 64K points against 0.407 / 0.385 for the two-pass pair, 0.362 at 8192 -- the fit a + c / rows crosses the pair at ~2400 rows;
        // profiles/r03_perf_ntt_table.txt.  From 10 halves per workgroup on, so that time per transform never rises with the batch.)
        const long stream_min = G_.onewg == 2 ? 2L * gridp : 20L * gridp;
        const bool stream_ok = half && mode == kSrcU32Ext && gridp >= 16 && gridp / 2 <= kOwPairCounters && wgs >= stream_min &&
                               ((uintptr_t)src & 15) == 0 && (src_stride & 3) == 0 && !rb;      // (rows in separate blocks: ntt_onewg only)
        const bool rows64_onewg = G_.onewg64 == 1 || (G_.onewg64 >= 2 && stream_ok);
        // negacyclic forward transform of full 64K-point rows (the ciphertext domain of x^65536 + 1): the persistent form, two
        // 32K-point halves per row meeting before their interleaved stores, from two halves per workgroup on (or forced)
        if (LG == 16 && mode == kSrcU32Twist && !mul_tab && !rb && G_.onewg && G_.onewg64 >= 2 && grid64 >= 16 &&
            (G_.onewg == 2 || 2L * batch >= 2L * grid64) && ((uintptr_t)src & 15) == 0 && (src_stride & 3) == 0) {
            OwTab &ot = D.ow[3];
            CHK(ensure_onewg_twist(ot, 15));
            OwArgs a{dst, src, ot.TW1g, ot.TW2, src_stride, dst_stride, batch, nstore, wa, nullptr, D.p, D.pinv, prime0, np_mod, nullptr, 0, FoldGeom{0, 0, 0, 0, 0}, nullptr};
            if (tm && tm->on) for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
            CHK(ws_pair_counters(W));
            note_dispatch("persistent one-workgroup halves with rendezvous (negacyclic rows)", L, batch);
            hipError_t he = ow_launch_stream_15(kSrcU32Twist, kOutU64, a, grid64, W.pair_cnt, ot.c128, ot.i4neg, st);
            if (he != hipSuccess) return fail(CUHE_EHIP, "persistent one-workgroup transform (negacyclic rows): %s", hipGetErrorString(he));
            if (tm && tm->on) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
            return CUHE_OK;
        }
        // inverse negacyclic rows of 32K points (the ciphertext domain of x^32768 + 1), calls that fill the chip: SPLIT into the two
        // 16K-point transforms of their even and odd outputs -- two workgroups per CU overlap where the 32K-point one-workgroup
        // form runs alone: 366 / 440 us against 419 / 507 us per 1536 rows / product rows (profiles/r03_split_rows.txt).
        // onewg_split = 2 (tests, A/B): also the forward rows of 32K points (measured equal: 519 vs 512 us per 2304 rows -- the
        // pre-add and the sample twist cost what the overlap wins) and the inverse rows of 64K points (slower than the two passes)
        const bool split_inv = mode == kSrcU64Neg || mode == kSrcU64NegMul;
        // (a persistent split form of the inverse 64K-point rows with the rendezvous was built in round 5 and lost to the pair by 2 %: profiles/r05_split_inverse_ab.txt)
        if (G_.onewg && !half && !(ep && ep->kind) &&
            ((LG == 15 && ((split_inv && G_.onewg_split >= 1) || (mode == kSrcU32Twist && G_.onewg_split == 2))) || (LG == 16 && split_inv && G_.onewg_split == 2)) &&
            (G_.onewg == 2 || 2L * batch >= (long)D.cus * (1 << (16 - LG)))) {
            int out = -1; const u64 *xt = nullptr;
            if (mode == kSrcU32Twist) { out = mul_tab ? kOutU64Mul : kOutU64; xt = mul_tab; }
            else if (nstore == kNcInverse) { out = kOutModPNc; xt = tab.twinv; }
            if (out >= 0 && ow_split_supported(mode, out)) {
                OwTab &ot = D.ow[LG - 1 - 12];
                if (mode == kSrcU32Twist) CHK(ensure_onewg_twist(ot, LG - 1)); else CHK(ensure_onewg(ot, LG - 1));
                if (mode == kSrcU64NegMul && !mul_tab) return fail(CUHE_EINVAL, "second operand missing");
                if (out == kOutModPNc && !xt) return fail(CUHE_EINVAL, "negacyclic untwist table missing");
                OwArgs a{dst, src, mode == kSrcU32Twist ? ot.TW1g : ot.TW1hi, ot.TW2, src_stride, dst_stride, batch, L, wa, mode == kSrcU64NegMul ? mul_tab : nullptr,
                         D.p, D.pinv, prime0, np_mod, nullptr, 0, FoldGeom{0, 0, 0, 0, 0}, xt, ot.c128, ot.i4neg};
                a.rb = rb;
                if (tm && tm->on) for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                note_dispatch("split row: one workgroup per half-length sub-transform", L, batch);
                CHK(onewg_launch(LG - 1, mode, out, true, a, st));
                if (tm && tm->on) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                return CUHE_OK;
            }
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
                a.rb = rb;
                if (tm && tm->on) for (int i = 0; i < 2; ++i) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                // resident workgroups walk over their share of the halves, the next half's samples arriving by LDS-DMA beside stage 3
                // of the current one, the two halves of a row meeting before their interleaved stores
                const bool stream = G_.onewg64 >= 2 && stream_ok;
                if (stream) {
                    CHK(ws_pair_counters(W));
                    note_dispatch("persistent one-workgroup halves with rendezvous (zero-padded rows)", L, batch);
                    hipError_t he = lgh == 15 ? ow_launch_stream_15(kSrcU32Ext, out, a, gridp, W.pair_cnt, 0, 0, st)
                                              : ow_launch_stream_14(kSrcU32Ext, out, a, gridp, W.pair_cnt, 0, 0, st);
                    if (he != hipSuccess) return fail(CUHE_EHIP, "persistent one-workgroup transform (2^%d-point halves): %s", lgh, hipGetErrorString(he));
                } else { note_dispatch(half ? "one workgroup per half" : "one workgroup per row", L, batch); CHK(onewg_launch(lgh, mode, out, half, a, st)); }
                if (tm && tm->on) { hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, st); tm->ev.push_back(ev); }
                return CUHE_OK;
            }
        }
    }
    if (rb) return kNoListForm;          // rows in separate blocks exist in the one-workgroup kernels only: the caller gathers them
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
    note_dispatch(ll ? "two-pass pair, low-latency kernels" : "two-pass pair", L, batch);
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
            int prime0, WindowArgs wa, int dev, hipStream_t st, EvTimer *tm, const u64 *mul_tab, int np_mod, const Epilogue *ep, const RowRebase *rb) {
    if (batch <= 0) return CUHE_OK;
    CHK(ensure_ntt(dev, len, batch));
    DevCtx &D = G_.dev[dev];
    Workspace *W = nullptr;
    CHK(workspace(dev, st, &W));
    switch (len) {
        case 8192:  return run_ntt_lg<13>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep, rb);
        case 16384: return run_ntt_lg<14>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep, rb);
        case 32768: return run_ntt_lg<15>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep, rb);
        default:    return run_ntt_lg<16>(mode, dst, src, batch, src_stride, dst_stride, nstore, prime0, wa, D, *W, st, tm, mul_tab, np_mod, ep, rb);
    }
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
int need_cyclic() {
    if (G_.prm.ncOnly()) return fail(CUHE_EINVAL, "ring degree %d has only the negacyclic representation (cuhe_hip_ct_*): the cyclic transforms of the reference stop at 65536 points", G_.prm.modLen);
    return CUHE_OK;
}
// CRT rows u32[rows][crtLen] -> ct rows u64[rows][ct_len]; mul_tab: rows the outputs are multiplied by on the way out
int ct_forward(u64 *X, const u32 *x, int rows, int dev, hipStream_t st, const u64 *mul_tab, int np_mod, const RowRebase *rb) {
    const Params &q = G_.prm;
    if (G_.nc) return run_ntt(q.modLen, kSrcU32Twist, X, x, rows, q.crtLen, q.modLen, q.modLen, 0, WindowArgs{0, 0, 0}, dev, st, nullptr, mul_tab, np_mod, nullptr, rb);
    return run_ntt(q.nttLen, kSrcU32Ext, X, x, rows, q.crtLen, q.nttLen, q.nttLen, 0, WindowArgs{0, 0, 0}, dev, st, nullptr, mul_tab, np_mod, nullptr, rb);
}
// Phi_m = x^n + 1 with n = L/2 on the cyclic representation: INTT, mod p_i and the reduction in one pass-2 epilogue
bool fused_xn1() {
    return !G_.force_generic && G_.reduce_kind == 1 && G_.prm.modLen * 2 == G_.prm.nttLen && G_.prm.crtLen == G_.prm.modLen;
}
// ct rows -> CRT rows u32[rows][crtLen]; row r is reduced modulo prime prime0 + r (row r mod np_mod when np_mod > 0, prime0
// = 0 then); is_prod: the rows are products of two reduced polynomials (cyclic representation: reduce modulo Phi_m)
// Y != null: the rows are the pointwise products X * Y, multiplied as pass 1 loads them (kSrcU64NegMul)
// rb != null (Y == null): the rows X are read from separate blocks (RowRebase::src_adj; dst_adj all zero: dst is one array); kNoListForm when the
// call would not take a one-workgroup kernel -- nothing has been enqueued then
int ct_inverse(u32 *dst, const u64 *X, int rows, int prime0, int np_mod, bool is_prod, int dev, hipStream_t st, const u64 *Y, const RowRebase *rb) {
    const Params &q = G_.prm;
    const int n = q.modLen, L = q.nttLen, cl = q.crtLen;
    const WindowArgs wa{0, 0, 0};
    const int mode = Y ? kSrcU64NegMul : kSrcU64Neg;
    if (rb && Y) return fail(CUHE_EINVAL, "rows in separate blocks: no second operand");
    if (G_.nc) return run_ntt(n, mode, dst, X, rows, n, cl, kNcInverse, prime0, wa, dev, st, nullptr, Y, np_mod, nullptr, rb);
    if (!is_prod) return run_ntt(L, mode, dst, X, rows, L, cl, cl, prime0, wa, dev, st, nullptr, Y, np_mod, nullptr, rb);
    if (fused_xn1()) return run_ntt(L, mode, dst, X, rows, L, cl, kFoldXn1, prime0, wa, dev, st, nullptr, Y, np_mod, nullptr, rb);
    Workspace *Wp = nullptr;
    CHK(workspace(dev, st, &Wp));
    CHK(ws_barrett(*Wp, rows));
    { const int r = run_ntt(L, mode, Wp->hold, X, rows, L, L, L, prime0, wa, dev, st, nullptr, Y, np_mod, nullptr, rb); if (r != CUHE_OK) return r; }
    return barrett_impl(dst, Wp->hold, prime0, rows, dev, st, np_mod);
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

}  // namespace cuhe_impl

using namespace cuhe_impl;

extern "C" {

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

// which kernel form the calling thread's last transform call took, and how many workgroups of this thread's persistent launches
// have given a rendezvous up so far (their partner was not resident: the launch then runs on without the merged stores)
int cuhe_hip_last_dispatch_info(int dev, char *buf, size_t cap) {
    if (!buf || cap == 0) return fail(CUHE_EINVAL, "no buffer");
    unsigned gave_up = 0;
    Workspace *Wp = nullptr;
    if (G_.inited && dev >= 0 && dev < (int)G_.dev.size() && set_dev(dev) == CUHE_OK && workspace_of_thread(dev, &Wp) == CUHE_OK && Wp->pair_cnt)
        HIPCHK(hipMemcpy(&gave_up, Wp->pair_cnt + kOwPairCounters, sizeof(unsigned), hipMemcpyDeviceToHost));
    snprintf(buf, cap, "%s; %d rows of %d points; rendezvous given up by %u workgroups so far", tls_dispatch.form, tls_dispatch.batch, tls_dispatch.len, gave_up);
    return CUHE_OK;
}
// the u64[nttLen] transform scratch of the reference (d_swap[dev], cuhe/Operations.cu:171-190: "not called externally" but
// public): the calling thread's slab for transforms of nttLen points, at least one transform long
uint64_t *cuhe_hip_ntt_swap(int dev) {
    if (!G_.inited || dev < 0 || dev >= (int)G_.dev.size() || set_dev(dev) != CUHE_OK) return nullptr;
    const int li = lg_index(G_.prm.nttLen);
    Workspace *Wp = nullptr; u64 *slab = nullptr;
    if (li < 0 || workspace_of_thread(dev, &Wp) != CUHE_OK) return nullptr;
    if (ws_slab(*Wp, li, 0, std::max(Wp->slab_bytes[li][0], (size_t)G_.prm.nttLen * sizeof(u64)), &slab) != CUHE_OK) return nullptr;
    return (uint64_t *)slab;
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

// ---------------------------------------------------------------- batched primitives
int cuhe_hip_ntt_prepare(int len, int dev) {
    CHK(set_dev(dev));
    return ensure_ntt(dev, len, 1 << 20);
}
int cuhe_hip_set_ntt_chunk(int chunk) {
    G_.ntt_chunk = chunk;
    return CUHE_OK;
}
int cuhe_hip_set_onewg_split(int mode) {
    if (mode < 0 || mode > 2) return fail(CUHE_EINVAL, "split mode %d", mode);
    G_.onewg_split = mode;
    return CUHE_OK;
}
int cuhe_hip_set_onewg(int mode, int rows64k) {
    if (mode < 0 || mode > 2 || rows64k < 0 || rows64k > 3) return fail(CUHE_EINVAL, "mode %d, rows64k %d", mode, rows64k);
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
