// cuhe_keyswitch.hip -- CRT / ICRT drivers, relinearisation (evaluation keys resident in HBM, key-switch inner product on the
// matrix cores), the batched multiply + relinearise chain, gates on arrays of ciphertexts and the CRT-prime-sharded variants
// with their RCCL exchange.  Replaces cuhe/Relinearization.cu and the CRT half of cuhe/Operations.cu
// (cuhe/Relinearization.cu:37-88, cuhe/Operations.cu:211-304, cuhe/CuHE.cu:217-256).
#include "cuhe_internal.hpp"
#include "comm.hpp"

namespace cuhe_impl {

int g_icrt_acc64 = getenv("CUHE_ICRT_ACC64") ? atoi(getenv("CUHE_ICRT_ACC64")) : 1;      // (environment override: A/B runs)
// the column sums of the ICRT on the matrix cores (icrt_mfma.cuh) wherever that form applies; 0: the VALU kernel k_icrt
int g_icrt_mfma = getenv("CUHE_ICRT_MFMA") ? atoi(getenv("CUHE_ICRT_MFMA")) : 1;
int icrt_lds_attr(size_t lds) {                 // k_icrt needs the large-LDS attribute for many primes
    static AttrOnce once;
    if (lds <= 64 * 1024) return CUHE_OK;
    CHK(once.set(k_icrt<false>, 160 * 1024));
    static AttrOnce once64;
    return once64.set(k_icrt<true>, 160 * 1024);
}

// CRT of `batch` polynomials (raw words -> residues of the primes prime0 ...).  Where the sums  sum_k word_k (2^(32k) mod p)
// fit 64 bits (W 2^32 pmax <= 2^64: every parameter set of the reference's examples) the kernel keeps no carry word and
// reduces once; CUHE_CRT_ACC64=0 keeps the 96-bit form (A/B runs)
int g_crt_acc64 = getenv("CUHE_CRT_ACC64") ? atoi(getenv("CUHE_CRT_ACC64")) : 1;
int launch_crt(u32 *dst, const u32 *src, const DevCtx &D, int prime0, int np, int W, int batch, long src_ct_stride, long dst_ct_stride, hipStream_t st) {
    const Params &q = G_.prm;
    if (W > D.maxW) return fail(CUHE_EINVAL, "coefficient words %d exceed table %d", W, D.maxW);
    const dim3 grid((q.modLen + kCrtCoef - 1) / kCrtCoef, batch), block(kCrtCoef * kCrtGroups);
    const size_t lds = (size_t)((W + 7) & ~7) * kCrtRow * 4;
    unsigned long long pmax = 0;
    for (int i = prime0; i < prime0 + np && i < (int)G_.primes.size(); ++i) pmax = std::max<unsigned long long>(pmax, G_.primes[i]);
    const bool acc64 = g_crt_acc64 && pmax > 0 && (unsigned long long)W * pmax <= (1ull << 32);
    if (acc64) hipLaunchKernelGGL(k_crt<true>, grid, block, lds, st, dst, src, prime_tab_at(D, prime0), np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride);
    else hipLaunchKernelGGL(k_crt<false>, grid, block, lds, st, dst, src, prime_tab_at(D, prime0), np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

// ICRT of `batch` ciphertexts of level lvl (np primes, W words)
int launch_icrt(u32 *dst, const u32 *src, const DevCtx &D, int lvl, int np, int W, int batch, long src_ct_stride, long dst_ct_stride, hipStream_t st,
                IcrtWindows wo = IcrtWindows{nullptr, 0, 0, 0, 0}) {
    const Params &q = G_.prm;
    const IcrtLevel &I = D.icrt[lvl];
    IcrtTab it{I.M, I.mi, I.bi, I.rp};
    const dim3 grid((q.modLen + kIcrtCoef - 1) / kIcrtCoef, batch), block(kIcrtCoef * kIcrtGroups);
    if (g_icrt_mfma && icrt_mfma_supported(I) && np == I.np && W == I.W)
        return launch_icrt_mfma(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    const size_t lds = icrt_lds_bytes(np, W);
    CHK(icrt_lds_attr(lds));
    // 64-bit column sums where they cannot overflow (see icrt_mac4_64): sum_i t_i m_i[k] + q M[k] < np pmax 2^32 + np 2^32 with
    // t_i < p_i <= pmax, every word below 2^32, q < np
    unsigned long long pmax = 0;
    for (int i = 0; i < np && i < (int)G_.primes.size(); ++i) pmax = std::max<unsigned long long>(pmax, G_.primes[i]);
    const bool acc64 = g_icrt_acc64 && pmax > 0 && (unsigned long long)(np + 1) * (pmax + 1) < (1ull << 32);
    if (acc64) hipLaunchKernelGGL(k_icrt<true>, grid, block, lds, st, dst, src, prime_tab(D), it, np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride, wo);
    else hipLaunchKernelGGL(k_icrt<false>, grid, block, lds, st, dst, src, prime_tab(D), it, np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride, wo);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
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

}  // namespace cuhe_impl

using namespace cuhe_impl;

extern "C" {

// ---------------------------------------------------------------- drivers
int cuhe_hip_crt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    DevCtx &D = G_.dev[dev];
    return launch_crt(dst, src, D, 0, np, W, 1, 0L, 0L, S(st));
}
int cuhe_hip_icrt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (lvl < 0) return fail(CUHE_EINVAL, "icrt below level 0");
    DevCtx &D = G_.dev[dev];
    const Params &q = G_.prm;
    return launch_icrt(dst, src, D, lvl, np, W, 1, 0L, 0L, S(st));
}
// elementwise CRT-domain kernels: 16-byte accesses when the rows allow them (CUHE_ELEMENTWISE_VEC=0: never, A/B runs)
static int g_elem_vec = getenv("CUHE_ELEMENTWISE_VEC") ? atoi(getenv("CUHE_ELEMENTWISE_VEC")) : 1;
static bool rows_vec4(const void *a, const void *b, long s1, long s2) {
    const Params &q = G_.prm;
    return g_elem_vec && q.modLen % 4 == 0 && q.crtLen % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && s1 % 4 == 0 && s2 % 4 == 0;
}
int cuhe_hip_crt_add(uint32_t *sum, const uint32_t *x, const uint32_t *y, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    const Params &q = G_.prm;
    if (rows_vec4(sum, x, 0, 0) && ((uintptr_t)y % 16) == 0)
        hipLaunchKernelGGL(k_crt_add<4>, dim3((q.modLen / 4 + 255) / 256, np), dim3(256), 0, S(st), sum, x, y, prime_tab(G_.dev[dev]), q.modLen, q.crtLen);
    else
        hipLaunchKernelGGL(k_crt_add<1>, dim3((q.modLen + 255) / 256, np), dim3(256), 0, S(st), sum, x, y, prime_tab(G_.dev[dev]), q.modLen, q.crtLen);
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
// modulus switch of `batch` ciphertexts
static int launch_modswitch(u32 *dst, const u32 *src, const DevCtx &D, int np, int batch, long ss, long ds, hipStream_t st) {
    const Params &q = G_.prm;
    const int groups = (np - 1 + kModswPrimes - 1) / kModswPrimes;
    if (rows_vec4(dst, src, ss, ds))
        hipLaunchKernelGGL(k_modswitch<4>, dim3((q.modLen / 4 + 255) / 256, groups, batch), dim3(256), 0, st, dst, src, prime_tab(D), D.invp, np, q.modLen, q.crtLen, q.modMsg, ss, ds);
    else
        hipLaunchKernelGGL(k_modswitch<1>, dim3((q.modLen + 255) / 256, groups, batch), dim3(256), 0, st, dst, src, prime_tab(D), D.invp, np, q.modLen, q.crtLen, q.modMsg, ss, ds);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
int cuhe_hip_crt_mod_switch(uint32_t *dst, const uint32_t *src, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (np < 2) return fail(CUHE_EINVAL, "modSwitch needs >= 2 primes");
    const Params &q = G_.prm;
    DevCtx &D = G_.dev[dev];
    return launch_modswitch(dst, src, D, np, 1, 0L, 0L, S(st));
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
// key-switch inner product of ONE ciphertext over the primes [prime0, prime0 + count): dst rows u64[count][L], win = the k transformed windows u64[k][L].
// primes per workgroup (each window value fetched from cache serves PB key streams): as many as still leave ~6 workgroups per CU -- the
// kernel streams the keys from HBM and needs that many loads in flight (12 waves per CU reach 4.3 TB/s, 24 reach 6 TB/s: profiles/r02_experiments_log.txt)
static int launch_mac_single(u64 *dst, const u64 *win, const DevCtx &D, int prime0, int count, int k, hipStream_t st) {
    const Params &q = G_.prm;
    const int L = ct_len();
    const u64 *ekp = D.ek + (size_t)(prime0 - D.ek_first) * q.numEvalKey * L;
    const long target = 6L * 256;
    auto blocks = [&](int pb) { return (long)(L / 512) * ((count + pb - 1) / pb); };
    if (blocks(4) >= target || count <= 1)
        hipLaunchKernelGGL((k_relin_mac<4, 1>), dim3((unsigned)blocks(4)), dim3(256), 0, st, dst, win, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    else if (blocks(2) >= target || count <= 2)
        hipLaunchKernelGGL((k_relin_mac<2, 1>), dim3((unsigned)blocks(2)), dim3(256), 0, st, dst, win, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    else
        hipLaunchKernelGGL((k_relin_mac<1, 1>), dim3((unsigned)blocks(1)), dim3(256), 0, st, dst, win, ekp, k, (long)q.numEvalKey * L, L, count, 0L, 0L, 1);
    HIPCHK(hipGetLastError());
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
    if (prime0 < D.ek_first || prime0 + count > D.ek_first + D.ek_count)
        return fail(CUHE_EINVAL, "keys of primes [%d, %d) wanted, device %d holds [%d, %d)", prime0, prime0 + count, dev, D.ek_first, D.ek_first + D.ek_count);
    return launch_mac_single((u64 *)dst, Wp->relin, D, prime0, count, k, S(st));
}
int cuhe_hip_relinearization(uint64_t *dst, const uint32_t *src, int lvl, int dev, void *st) {
    return relin_range(dst, src, lvl, 0, G_.prm.numCrtPrimeAt(lvl < 0 ? 0 : lvl), dev, st);
}

// ---- CuCtxt::relin from the raw domain on -- relinearization ; n2c (cuhe/CuHE.cu:574-580) -- as ONE call: raw coefficients ->
// reduced CRT rows, the sums kept in the calling thread's scratch (no NTT-domain result buffer on the caller's side, one call instead
// of two: 0.475 -> 0.451 ms per multiply + relinearise on x^65536+1, 0.258 -> 0.256 on x^32768+1).  Round 5 also built the form with
// the HBM-bound key stream BESIDE the instruction-bound transforms (window groups on a helper stream, partial inner products as they
// arrive, the last group in blocks of primes whose sums are taken back while the next block is summed): bit-identical and SLOWER --
// 0.547 / 0.354 ms -- five inner-product launches instead of one (the blocks of primes re-read the windows), transform calls of 16-23
// rows, eight cross-stream events per chain; removed again (profiles/EXPERIMENTS.md section 0, profiles/r05_relin_overlap_ab.txt).
int cuhe_hip_relin_crt(uint32_t *dst, const uint32_t *src, int lvl, int dev, void *st_) {
    CHK(need_init(dev));
    if (!G_.relin_ready) return fail(CUHE_ENOTINIT, "initRelinearization has not been called");
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    const int np = q.numCrtPrimeAt(lvl);
    hipStream_t st = S(st_);
    Workspace *W0 = nullptr;
    CHK(workspace(dev, st, &W0));
    CHK(ws_buffer(&W0->rc_acc, (size_t)q.numCrtPrime * ct_len()));
    CHK(relin_range((uint64_t *)W0->rc_acc, src, lvl, 0, np, dev, st_));
    return ct_inverse(dst, W0->rc_acc, np, 0, 0, true, dev, st);
}

// ---- key digits for the inner product on the matrix cores (k_relin_mac_mfma): built on the first batched call that
// wants them, from the ct-domain keys; as large as the keys themselves.
// measured crossover at config 4: 2 and 4 ciphertexts are a little faster on the VALU kernel (0.190 / 0.127 vs 0.197 / 0.134 ms per
// ciphertext), 6 already on the matrix cores (0.110 vs 0.141: one half-filled tile instead of two VALU groups)
static int g_mac_mfma_min = getenv("CUHE_MAC_MFMA_MIN") ? atoi(getenv("CUHE_MAC_MFMA_MIN")) : 5;     // smallest batch that takes the MFMA kernel; 0 = never
int cuhe_hip_set_crt_acc64(int on) {
    if (on != 0 && on != 1) return fail(CUHE_EINVAL, "on %d", on);
    g_crt_acc64 = on;
    return CUHE_OK;
}
int cuhe_hip_set_icrt_mfma(int on) {
    if (on != 0 && on != 1) return fail(CUHE_EINVAL, "on %d", on);
    g_icrt_mfma = on;
    return CUHE_OK;
}
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
        const size_t cap = std::max<size_t>(batch, 2 * Ws.n_bt);      // geometric; the outgrown buffers are retired (ws_retire)
        if (Ws.bt_ntt) { ws_retire(Ws.bt_ntt); Ws.bt_ntt = nullptr; }
        if (Ws.bt_crt) { ws_retire(Ws.bt_crt); Ws.bt_crt = nullptr; }
        CHK(ws_grow(&Ws.bt_ntt, &x, cap * q.numCrtPrime * L));
        CHK(ws_grow(&Ws.bt_crt, &y, cap * q.numCrtPrime * cl));
        Ws.n_bt = cap;
    }
    CHK(ws_relin(Ws, (int)std::max<size_t>(batch, Ws.n_relin < (size_t)batch ? 2 * Ws.n_relin : 0)));
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
    return launch_modswitch(dst, src, D, np, batch, (long)np * q.crtLen, (long)(np - 1) * q.crtLen, S(st));
}
// `count` blocks of `bytes` bytes each (a multiple of 16, 16-byte aligned) between their own addresses and one contiguous
// array: gather (blocks -> array) / scatter (array -> blocks); the pointer list is HOST memory (it travels as a kernel argument)
static int move_blocks(bool gather, void *contig, void *const *blocks, int count, size_t bytes, int dev, void *st) {
    CHK(need_init(dev));
    if (count < 1 || (bytes & 15) || ((uintptr_t)contig & 15)) return fail(CUHE_EINVAL, "gather / scatter of %d blocks of %zu bytes", count, bytes);
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList L;
        for (int i = 0; i < n; ++i) { L.p[i] = blocks[c0 + i]; if ((uintptr_t)L.p[i] & 15) return fail(CUHE_EINVAL, "block %d is not 16-byte aligned", c0 + i); }
        for (int i = n; i < kPtrListMax; ++i) L.p[i] = nullptr;
        const int gx = (int)std::min<size_t>((bytes / 16 + 255) / 256, 256);
        char *base = (char *)contig + (size_t)c0 * bytes;
        if (gather) hipLaunchKernelGGL(k_move_blocks<true>, dim3(gx, n), dim3(256), 0, S(st), base, L, (long)bytes);
        else hipLaunchKernelGGL(k_move_blocks<false>, dim3(gx, n), dim3(256), 0, S(st), base, L, (long)bytes);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
static int fill_list(PtrList &L, const void *const *p, int c0, int n) {
    for (int i = 0; i < kPtrListMax; ++i) {
        L.p[i] = i < n ? (void *)p[c0 + i] : nullptr;
        if (i < n && (!L.p[i] || ((uintptr_t)L.p[i] & 15))) return fail(CUHE_EINVAL, "list entry %d is null or not 16-byte aligned", c0 + i);
    }
    return CUHE_OK;
}
// z[i] = x[i] * y[i] (mul != 0) or x[i] + y[i], pointwise modulo P, on `count` separately owned ct-domain ciphertexts of the
// level of logq (cAnd / cXor of CuCtxt in the NTT domain, CuHE.cu:101,545 -- one launch for the whole list); lists in HOST memory
int cuhe_hip_ct_binop_list(int mul, void *const *z, const void *const *x, const void *const *y, int count, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (count < 1) return fail(CUHE_EINVAL, "count %d", count);
    const long n2 = (long)np * ct_len() / 2;
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList Z, X, Y;
        CHK(fill_list(Z, (const void *const *)z, c0, n)); CHK(fill_list(X, x, c0, n)); CHK(fill_list(Y, y, c0, n));
        const int gx = (int)std::min<long>((n2 + 255) / 256, 512);
        if (mul) hipLaunchKernelGGL(k_ntt_binop_list<true>, dim3(gx, n), dim3(256), 0, S(st), Z, X, Y, n2);
        else hipLaunchKernelGGL(k_ntt_binop_list<false>, dim3(gx, n), dim3(256), 0, S(st), Z, X, Y, n2);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// z[i] = (a[i] + b[i]) mod p_row on `count` separately owned CRT-domain ciphertexts (cXor in the CRT domain, Operations.cu:264)
int cuhe_hip_crt_add_list(void *const *z, const void *const *a, const void *const *b, int count, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (count < 1) return fail(CUHE_EINVAL, "count %d", count);
    const Params &q = G_.prm;
    DevCtx &D = G_.dev[dev];
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList Z, A, B;
        CHK(fill_list(Z, (const void *const *)z, c0, n)); CHK(fill_list(A, a, c0, n)); CHK(fill_list(B, b, c0, n));
        bool vec = rows_vec4(nullptr, nullptr, 0, 0);
        for (int t = 0; t < n && vec; ++t) vec = (((uintptr_t)Z.p[t] | (uintptr_t)A.p[t] | (uintptr_t)B.p[t]) % 16) == 0;
        if (vec) hipLaunchKernelGGL(k_crt_add_list<4>, dim3((q.modLen / 4 + 255) / 256, np, n), dim3(256), 0, S(st), Z, A, B, prime_tab(D), q.modLen, q.crtLen);
        else hipLaunchKernelGGL(k_crt_add_list<1>, dim3((q.modLen + 255) / 256, np, n), dim3(256), 0, S(st), Z, A, B, prime_tab(D), q.modLen, q.crtLen);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// modSwitch of `count` separately owned CRT-domain ciphertexts of level lvl (np rows each): dst[i] <- src[i] one level down (np - 1 rows);
// dst[i] == src[i] switches a ciphertext inside its own block (CuCtxt::modSwitch, CuHE.cu:583-594, over a list: one launch per 64)
int cuhe_hip_crt_mod_switch_list(void *const *dst, const void *const *src, int lvl, int count, int dev, void *st) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl + 1 >= q.depth) return fail(CUHE_EINVAL, "modSwitch from level %d", lvl);
    if (count < 1) return fail(CUHE_EINVAL, "count %d", count);
    const int np = q.numCrtPrimeAt(lvl);
    if (np < 2) return fail(CUHE_EINVAL, "modSwitch needs >= 2 primes");
    DevCtx &D = G_.dev[dev];
    const int groups = (np - 1 + kModswPrimes - 1) / kModswPrimes;
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList Dl, Sl;
        CHK(fill_list(Dl, (const void *const *)dst, c0, n)); CHK(fill_list(Sl, src, c0, n));
        if (rows_vec4(nullptr, nullptr, 0, 0))           // (fill_list has checked every block for 16-byte alignment)
            hipLaunchKernelGGL(k_modswitch_list<4>, dim3((q.modLen / 4 + 255) / 256, groups, n), dim3(256), 0, S(st), Dl, Sl, prime_tab(D), D.invp, np, q.modLen, q.crtLen, q.modMsg);
        else
            hipLaunchKernelGGL(k_modswitch_list<1>, dim3((q.modLen + 255) / 256, groups, n), dim3(256), 0, S(st), Dl, Sl, prime_tab(D), D.invp, np, q.modLen, q.crtLen, q.modMsg);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// cNot over a list: z[i] = x[i] + a on the constant coefficient of every row (mod p), the other coefficients copied when z[i] != x[i]
// (cNot of CuCtxt, CuHE.cu:176-187 with crt_add_int, Base.cu:1096-1100: one launch per 64 ciphertexts)
int cuhe_hip_crt_add_int_list(void *const *z, const void *const *x, unsigned a, int count, int logq, int dev, void *st) {
    CHK(need_init(dev));
    int lvl, np, W; CHK(level_of(logq, &lvl, &np, &W));
    if (count < 1) return fail(CUHE_EINVAL, "count %d", count);
    const Params &q = G_.prm;
    DevCtx &D = G_.dev[dev];
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList Z, X;
        CHK(fill_list(Z, (const void *const *)z, c0, n)); CHK(fill_list(X, x, c0, n));
        bool copies = false;
        for (int t = 0; t < n; ++t) copies = copies || Z.p[t] != X.p[t];
        const int gx = copies ? (int)std::min<long>(((long)np * q.modLen + 255) / 256, 128) : (np + 255) / 256;
        hipLaunchKernelGGL(k_crt_add_int_list, dim3(gx, n), dim3(256), 0, S(st), Z, X, a, prime_tab(D), np, q.modLen, q.crtLen);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// dst[i] = src[i] for `count` separately owned blocks of `bytes` bytes (copy() of CuCtxt, CuHE.cu:81: one launch for the list)
int cuhe_hip_copy_list(void *const *dst, const void *const *src, int count, size_t bytes, int dev, void *st) {
    CHK(need_init(dev));
    if (count < 1 || (bytes & 15)) return fail(CUHE_EINVAL, "copy of %d blocks of %zu bytes", count, bytes);
    for (int c0 = 0; c0 < count; c0 += kPtrListMax) {
        const int n = std::min(kPtrListMax, count - c0);
        PtrList Dl, Sl;
        CHK(fill_list(Dl, (const void *const *)dst, c0, n)); CHK(fill_list(Sl, src, c0, n));
        const int gx = (int)std::min<size_t>((bytes / 16 + 255) / 256, 256);
        hipLaunchKernelGGL(k_copy_list, dim3(gx, n), dim3(256), 0, S(st), Dl, Sl, (long)bytes);
    }
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}
// n2c of `batch` NON-product ciphertexts in one array (cuhe_hip_intt_mod_batch is the form for products): inverse transform, % p
int cuhe_hip_intt_batch(uint32_t *dst, const uint64_t *src, int lvl, int batch, int dev, void *st_) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < -1 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (batch < 1) return fail(CUHE_EINVAL, "batch %d", batch);
    const int np = lvl < 0 ? 1 : q.numCrtPrimeAt(lvl);
    return ct_inverse(dst, (const u64 *)src, batch * np, 0, np, false, dev, S(st_));
}
// ---- transforms of ciphertexts that live in SEPARATE blocks (the members of a batch of scheduled gates, cuhe_amd/cxx/CuHE.cpp runBatch): the
// one-workgroup kernels address row r of block c = r / np through RowRebase (ntt_kernels.cuh), kRowBlocksMax blocks per launch -- no gather before,
// no scatter after.  A call that would not take such a kernel (few rows: the two-pass pair; rows of 64K points: the persistent forms) gathers
// into / scatters from a scratch array of the calling thread's workspace instead: same results, the traffic of round 4.
static int fill_rebase(RowRebase &R, int per, const void *const *src, long src_block_bytes, void *const *dst, long dst_block_bytes, int c0, int n) {
    R.per = per;
    for (int i = 0; i < kRowBlocksMax; ++i) { R.src_adj[i] = 0; R.dst_adj[i] = 0; }
    for (int i = 0; i < n; ++i) {
        if (src) {
            if (!src[c0 + i] || ((uintptr_t)src[c0 + i] & 15)) return fail(CUHE_EINVAL, "source block %d is null or not 16-byte aligned", c0 + i);
            R.src_adj[i] = (long)((const char *)src[c0 + i] - (const char *)src[c0]) - (long)i * src_block_bytes;
        }
        if (dst) {
            if (!dst[c0 + i] || ((uintptr_t)dst[c0 + i] & 15)) return fail(CUHE_EINVAL, "destination block %d is null or not 16-byte aligned", c0 + i);
            R.dst_adj[i] = (long)((char *)dst[c0 + i] - (char *)dst[c0]) - (long)i * dst_block_bytes;
        }
    }
    return CUHE_OK;
}
static int g_row_lists = getenv("CUHE_ROW_LISTS") ? atoi(getenv("CUHE_ROW_LISTS")) : 1;      // 0: always gather / scatter (A/B runs, tests)
int cuhe_hip_set_row_lists(int on) {
    if (on != 0 && on != 1) return fail(CUHE_EINVAL, "on %d", on);
    g_row_lists = on;
    return CUHE_OK;
}
// c2n of `count` ciphertexts of level lvl: dst[i] u64[np][ct_len] = transform of src[i] u32[np][crtLen]; returns in *direct (may be null) how many
// ciphertexts went through the kernels' own block addressing
int cuhe_hip_ct_ntt_list(uint64_t *const *dst, const uint32_t *const *src, int count, int lvl, int dev, void *st_, int *direct) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (count < 1 || !dst || !src) return fail(CUHE_EINVAL, "count %d", count);
    hipStream_t st = S(st_);
    const int np = q.numCrtPrimeAt(lvl), L = ct_len();
    const long cB = (long)np * q.crtLen * sizeof(u32), nB = (long)np * L * sizeof(u64);
    int done_direct = 0;
    for (int c0 = 0; c0 < count; c0 += kRowBlocksMax) {
        const int n = std::min(kRowBlocksMax, count - c0);
        RowRebase R;
        CHK(fill_rebase(R, np, (const void *const *)src, cB, (void *const *)dst, nB, c0, n));
        const int r = g_row_lists ? ct_forward((u64 *)dst[c0], src[c0], n * np, dev, st, nullptr, 0, &R) : kNoListForm;
        if (r == kNoListForm) {                 // everything that is left as ONE array call (more rows per launch than chunk by chunk)
            const int m = count - c0;
            Workspace *W = nullptr;
            CHK(workspace(dev, st, &W));
            CHK(ws_grow(&W->ls_crt, &W->n_ls_crt, (size_t)m * np * q.crtLen));
            CHK(ws_grow(&W->ls_ntt, &W->n_ls_ntt, (size_t)m * np * L));
            CHK(move_blocks(true, W->ls_crt, (void *const *)(src + c0), m, (size_t)cB, dev, st_));
            CHK(ct_forward(W->ls_ntt, W->ls_crt, m * np, dev, st));
            CHK(move_blocks(false, W->ls_ntt, (void *const *)(dst + c0), m, (size_t)nB, dev, st_));
            break;
        }
        CHK(r);
        done_direct += n;
    }
    if (direct) *direct = done_direct;
    return CUHE_OK;
}
// n2c of `count` ciphertexts of level lvl into ONE array: dst u32[count][np][crtLen] = inverse transform (products: + reduction modulo the
// polynomial modulus) of the blocks src[i] u64[np][ct_len]
int cuhe_hip_ct_intt_list(uint32_t *dst, const uint64_t *const *src, int count, int lvl, int is_prod, int dev, void *st_, int *direct) {
    CHK(need_init(dev));
    const Params &q = G_.prm;
    if (lvl < 0 || lvl >= q.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    if (count < 1 || !dst || !src) return fail(CUHE_EINVAL, "count %d", count);
    hipStream_t st = S(st_);
    const int np = q.numCrtPrimeAt(lvl), L = ct_len();
    const long nB = (long)np * L * sizeof(u64);
    int done_direct = 0;
    for (int c0 = 0; c0 < count; c0 += kRowBlocksMax) {
        const int n = std::min(kRowBlocksMax, count - c0);
        u32 *out = dst + (size_t)c0 * np * q.crtLen;
        RowRebase R;
        CHK(fill_rebase(R, np, (const void *const *)src, nB, nullptr, 0, c0, n));
        const int r = g_row_lists ? ct_inverse(out, (const u64 *)src[c0], n * np, 0, np, is_prod != 0, dev, st, nullptr, &R) : kNoListForm;
        if (r == kNoListForm) {
            const int m = count - c0;
            Workspace *W = nullptr;
            CHK(workspace(dev, st, &W));
            CHK(ws_grow(&W->ls_ntt, &W->n_ls_ntt, (size_t)m * np * L));
            CHK(move_blocks(true, W->ls_ntt, (void *const *)(src + c0), m, (size_t)nB, dev, st_));
            CHK(ct_inverse(out, W->ls_ntt, m * np, 0, np, is_prod != 0, dev, st));
            break;
        }
        CHK(r);
        done_direct += n;
    }
    if (direct) *direct = done_direct;
    return CUHE_OK;
}
int cuhe_hip_gather_blocks(void *dst, const void *const *srcs, int count, size_t bytes, int dev, void *st) {
    return move_blocks(true, dst, (void *const *)srcs, count, bytes, dev, st);
}
int cuhe_hip_scatter_blocks(void *const *dsts, const void *src, int count, size_t bytes, int dev, void *st) {
    return move_blocks(false, (void *)src, dsts, count, bytes, dev, st);
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
    if (rows_vec4(dst, src_a, 0, 0) && (!src_b || ((uintptr_t)src_b % 16) == 0))
        hipLaunchKernelGGL(k_crt_combine<4>, dim3((q.modLen / 4 + 255) / 256, np, nout), dim3(256), 0, S(st), dst, src_a, nA, src_b, off, list, add_const,
                           prime_tab(D), np, q.modLen, q.crtLen);
    else
        hipLaunchKernelGGL(k_crt_combine<1>, dim3((q.modLen + 255) / 256, np, nout), dim3(256), 0, S(st), dst, src_a, nA, src_b, off, list, add_const,
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
        if (Ws.mr_ntt) { ws_retire(Ws.mr_ntt); Ws.mr_ntt = nullptr; }
        if (Ws.mr_crt) { ws_retire(Ws.mr_crt); Ws.mr_crt = nullptr; }
        CHK(ws_grow(&Ws.mr_ntt, &x, (size_t)2 * batch * q.numCrtPrime * L));
        CHK(ws_grow(&Ws.mr_crt, &y, (size_t)2 * batch * q.numCrtPrime * cl));
        Ws.n_mr = batch;
    }
    u32 *ca = Ws.mr_crt, *cb = Ws.mr_crt + (size_t)rows * cl;
    u64 *na = Ws.mr_ntt;
    if (q.modLen < cl) HIPCHK(hipMemsetAsync(Ws.mr_crt, 0, (size_t)2 * rows * cl * sizeof(u32), st));
    CHK(launch_crt(ca, a, D, 0, np, W, batch, (long)q.rawLen * W, (long)np * cl, st));
    CHK(launch_crt(cb, b, D, 0, np, W, batch, (long)q.rawLen * W, (long)np * cl, st));
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
    return launch_crt(dst, src, D, prime0, count, W, 1, 0L, 0L, S(st));
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
    if (C.stage) { (void)hipFree(C.stage); C.stage = nullptr; C.stage_words = 0; C.stage_dev = -1; }
    if (C.stage_event) { (void)hipEventDestroy((hipEvent_t)C.stage_event); C.stage_event = nullptr; }
    C.stage_used = false; C.stage_stream = nullptr;
    C.nranks = 1; C.rank = 0; C.force_exchange = 0; C.exchanges = 0; C.last_path = "none yet";
    for (long &n : C.path_count) n = 0;
    return CUHE_OK;
}
int cuhe_hip_comm_size(void) { return comm::state().nranks; }
int cuhe_hip_comm_rank(void) { return comm::state().rank; }
int cuhe_hip_comm_force_exchange(int on) { comm::state().force_exchange = on < 0 ? 0 : on; return CUHE_OK; }
// what RCCL itself reports about the communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion) and which path the
// last exchange of CRT rows took: the first thing to read when a multi-GPU run misbehaves
int cuhe_hip_comm_info(char *buf, size_t cap) {
    if (!buf || cap == 0) return fail(CUHE_EINVAL, "no buffer");
    comm::Api &A = comm::api();
    comm::State &C = comm::state();
    if (A.error) { snprintf(buf, cap, "RCCL unavailable: %s", A.error); return CUHE_OK; }
    int ver = -1, cnt = -1, urank = -1;
    if (A.GetVersion) A.GetVersion(&ver);
    if (C.comm && A.CommCount) A.CommCount(C.comm, &cnt);
    if (C.comm && A.CommUserRank) A.CommUserRank(C.comm, &urank);
    snprintf(buf, cap, "rccl %d; communicator %s; ncclCommCount %d, ncclCommUserRank %d (library: %d ranks, rank %d); exchanges so far %ld "
             "(ncclAllGather in place %ld, padded %ld, broadcast group %ld), last: %s",
             ver, C.comm ? "initialised" : "not initialised", cnt, urank, C.nranks, C.rank, C.exchanges, C.path_count[1], C.path_count[2], C.path_count[3], C.last_path);
    return CUHE_OK;
}
// which form the exchange of a level takes for a communicator of `nranks` ranks (comm::exchange_path): 0 none, 1 one in-place
// ncclAllGather, 2 one ncclAllGather of padded blocks, 3 the group of broadcasts.  Host logic only: no GPU, no RCCL needed.
int cuhe_hip_exchange_path(int lvl, int nranks, int force) {
    if (!G_.params_set || lvl < 0 || lvl >= G_.prm.depth || nranks < 1) return fail(CUHE_EINVAL, "exchange_path(lvl %d, nranks %d)", lvl, nranks), -1;
    return comm::exchange_path(G_.prm.numCrtPrimeAt(lvl), nranks, force);
}
// rows: u32[np][crtLen] of level lvl on this rank's device, the rank's own block already in place; on return (in stream
// order) every block is.  ONE ncclAllGather: in place when the blocks are equal; of blocks padded to the largest, through a
// staging buffer and two strided copies, when they are not (np not a multiple of the number of ranks).
int cuhe_hip_allgather_rows(uint32_t *rows, int lvl, int dev, void *st) {
    CHK(need_init(dev));
    if (lvl < 0 || lvl >= G_.prm.depth) return fail(CUHE_EINVAL, "level %d", lvl);
    comm::State &C = comm::state();
    int force = C.comm ? C.force_exchange : 0;
    // (the environment override applies to an existing communicator only: without one there is nothing to exchange and the call stays a no-op)
    if (C.comm) if (const char *e = getenv("CUHE_EXCHANGE")) { if (!strcmp(e, "padded")) force = 2; else if (!strcmp(e, "bcast")) force = 3; }
    const int np = G_.prm.numCrtPrimeAt(lvl), cl = G_.prm.crtLen;
    const int path = comm::exchange_path(np, C.nranks, force);
    if (path == comm::kPathNone) { C.last_path = comm::path_name(path); return CUHE_OK; }
    if (!C.comm) return fail(CUHE_ENOTINIT, "cuhe_hip_comm_init has not been called");
    comm::Api &A = comm::api();
    // every failing call is named with its rank, root and RCCL's own message: the first multi-GPU contact must explain itself
    if (path == comm::kPathAllGather) {
        const size_t cnt = (size_t)(np / C.nranks) * cl;           // in place: this rank's block sits at rows + rank * cnt
        const ncclResult_t r = A.AllGather(rows + (size_t)C.rank * cnt, rows, cnt, ncclUint32, C.comm, S(st));
        if (r != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows, rank %d of %d: ncclAllGather(%zu words per rank): %s", C.rank, C.nranks, cnt, A.GetErrorString(r));
    } else if (path == comm::kPathAllGatherPadded) {
        const int base = np / C.nranks, extra = np % C.nranks, maxc = base + (extra ? 1 : 0);
        const size_t slot = (size_t)maxc * cl, need = slot * C.nranks;
        if (C.stage_words < need || C.stage_dev != dev) {          // grow-only; a re-allocation waits for the device (hipFree), as every workspace does
            if (C.stage) { HIPCHK(hipFree(C.stage)); C.stage = nullptr; C.stage_words = 0; }
            HIPCHK(hipMalloc((void **)&C.stage, need * sizeof(u32)));
            C.stage_words = need; C.stage_dev = dev;
        }
        // ONE staging buffer per communicator: an exchange enqueued on another stream than the previous one is ordered behind the previous
        // one's last read of the buffer (collectives of one communicator are issued in one order anyway; the copies around them are not)
        if (!C.stage_event) HIPCHK(hipEventCreateWithFlags((hipEvent_t *)&C.stage_event, hipEventDisableTiming));
        if (C.stage_used && C.stage_stream != st) HIPCHK(hipStreamWaitEvent(S(st), (hipEvent_t)C.stage_event, 0));
        int f, c; comm::shard_bounds(np, C.nranks, C.rank, &f, &c);
        u32 *mine = C.stage + (size_t)C.rank * slot;
        HIPCHK(hipMemcpyAsync(mine, rows + (size_t)f * cl, (size_t)c * cl * sizeof(u32), hipMemcpyDeviceToDevice, S(st)));
        const ncclResult_t r = A.AllGather(mine, C.stage, slot, ncclUint32, C.comm, S(st));
        if (r != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows, rank %d of %d: ncclAllGather(padded, %zu words per rank): %s", C.rank, C.nranks, slot, A.GetErrorString(r));
        // unpack: the first `extra` ranks hold base + 1 rows each, the others base rows: two strided copies (the own block is rewritten with itself)
        const size_t pitch = slot * sizeof(u32);
        if (extra) HIPCHK(hipMemcpy2DAsync(rows, (size_t)(base + 1) * cl * sizeof(u32), C.stage, pitch, (size_t)(base + 1) * cl * sizeof(u32), extra, hipMemcpyDeviceToDevice, S(st)));
        if (base) HIPCHK(hipMemcpy2DAsync(rows + (size_t)extra * (base + 1) * cl, (size_t)base * cl * sizeof(u32), C.stage + (size_t)extra * slot, pitch,
                                         (size_t)base * cl * sizeof(u32), C.nranks - extra, hipMemcpyDeviceToDevice, S(st)));
        HIPCHK(hipEventRecord((hipEvent_t)C.stage_event, S(st)));
        C.stage_used = true; C.stage_stream = st;
    } else {
        ncclResult_t r = A.GroupStart();
        if (r != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows, rank %d of %d: ncclGroupStart: %s", C.rank, C.nranks, A.GetErrorString(r));
        int bad_root = -1;
        for (int rk = 0; rk < C.nranks && r == ncclSuccess; ++rk) {
            int f, c; comm::shard_bounds(np, C.nranks, rk, &f, &c);
            if (c == 0) continue;
            u32 *blk = rows + (size_t)f * cl;
            r = A.Broadcast(blk, blk, (size_t)c * cl, ncclUint32, rk, C.comm, S(st));
            if (r != ncclSuccess) bad_root = rk;
        }
        const ncclResult_t e = A.GroupEnd();
        if (r != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows, rank %d of %d: ncclBroadcast(root %d): %s", C.rank, C.nranks, bad_root, A.GetErrorString(r));
        if (e != ncclSuccess) return fail(CUHE_EHIP, "all-gather of CRT rows, rank %d of %d: ncclGroupEnd: %s", C.rank, C.nranks, A.GetErrorString(e));
    }
    C.last_path = comm::path_name(path); ++C.exchanges; ++C.path_count[path];
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

}  // extern "C"
