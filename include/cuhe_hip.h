/*
 * cuhe_hip.h -- C ABI of the MI355X (gfx950) large-polynomial backend.
 *
 * This is the drop-in boundary underneath cuHE's C++ API.  The reference has no
 * C ABI (its Operations.h functions are C++-namespaced and CuHE.h carries NTL
 * types); the functions below are what a `libcuHE` built on this backend binds:
 * each one replaces the reference function cited next to it (file:line relative
 * to the vernamlab/cuHE tree) with the same argument meaning (raw device
 * pointers + logq / dev / stream).  No torch / NTL / C++ types cross this line.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CUHE_E* code otherwise
 *     and records a message retrievable with cuhe_hip_last_error(); the C++
 *     shim turns a non-zero status into the reference's "print file:line,
 *     exit(-1)" behaviour (cuhe/Debug.h:39-46).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     are asynchronous on that stream unless stated; the C++ shim adds the
 *     reference's per-op hipStreamSynchronize (cuhe/CuHE.cu:98,121,...).
 *   - layouts: raw  u32[rawLen][W] coefficient-major LE words,
 *     crt u32[np][crtLen] prime-major, ntt u64[np][nttLen]; level `lvl` uses
 *     the first numCrtPrime-lvl primes; W = wordsCoeff(lvl).
 *   - library-owned scratch is kept per HOST THREAD and device: calls made by one
 *     thread are ordered (also across the streams that thread uses), different
 *     threads may drive the same device concurrently on their own streams.  (The
 *     reference has one scratch set per device: one operation in flight per
 *     device, cuhe/Operations.cu:171-172,193-195.)  Parameter set-up, init,
 *     initRelinearization and shutdown are not concurrent with anything.
 */
#ifndef CUHE_HIP_H
#define CUHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUHE_OK 0
#define CUHE_EINVAL (-1)   /* bad argument / wrong state */
#define CUHE_EHIP (-2)     /* HIP runtime failure */
#define CUHE_ENOTINIT (-3) /* cuhe_hip_init not called */

/* cuhe/Parameters.h:34-62 (GlobalParameters), same field names */
typedef struct {
    int mSize, modLen, modLen2, rawLen, crtLen, nttLen;
    int logCoeffMax, logCoeffMin, logCoeffCut;
    int depth, modMsg, logMsg, wordsMsg;
    int logRelin, numEvalKey;
    int logCrtPrime, numCrtPrime;
} cuhe_params_t;

const char *cuhe_hip_last_error(void);
const char *cuhe_hip_version(void);

/* ---- parameters: setParameters / resetParameters (cuhe/CuHE.h:171,174; Parameters.cu:53-105) */
int cuhe_hip_set_parameters(int d, int p, int w, int min, int cut, int m);
int cuhe_hip_reset_parameters(void);
int cuhe_hip_get_parameters(cuhe_params_t *out);
/* per-level helpers (cuhe/Parameters.cu:107-145) */
int cuhe_hip_num_crt_prime(int lvl);
int cuhe_hip_log_coeff(int lvl);
int cuhe_hip_words_coeff(int lvl);
int cuhe_hip_num_eval_key(int lvl);
int cuhe_hip_get_level(int logq);

/* ---- devices: multiGPUs / numGPUs (cuhe/CuHE.h:161-163; DeviceManager.cu:36-45) */
int cuhe_hip_multi_gpus(int num);
int cuhe_hip_num_gpus(void);
/* Bind the single-device context to HIP device `dev` (default 0).  Used by the
 * one-process-per-GPU launcher so that rank r drives device r as "dev 0". */
int cuhe_hip_set_device_base(int dev);
/* test hook: back every logical device of multi_gpus(n) by the one physical device (own context each), so that the
   in-process multi-device paths can run on a single-GPU box; before init */
int cuhe_hip_set_virtual_devices(int on);

/* The CPUs local to logical device `dev` (the NUMA node its PCIe root hangs off), as the kernel's cpulist string ("0-63,128-191") in buf;
 * empty when the platform does not say.  On a two-socket MI355X host a thread that launches from the far socket pays every doorbell and every
 * completion signal across the socket link: the gate scheduler's workers launch 900 small kernels per PRINCE block, and blocks issued from the far
 * socket take 0.065-0.070 s against 0.058-0.059 s (profiles/r06_numa_pinning.txt).  No counterpart in the reference (one GPU per host thread,
 * placement left to OpenMP).  cuhe_hip_pin_thread_to_device restricts the CALLING thread to those CPUs (intersected with what it may use now;
 * nothing is changed when that is empty or the list is unknown): returns 1 if the affinity was narrowed, 0 if left alone. */
int cuhe_hip_device_local_cpus(int dev, char *buf, size_t buf_bytes);
int cuhe_hip_pin_thread_to_device(int dev);

/* ---- init: initCuHE (cuhe/CuHE.h:153; CuHE.cu:36-50 = initNtt + initCrt + initBarrett).
 * modulus: monic integer polynomial, modLen+1 coefficients low-to-high
 * (NULL => the cyclotomic polynomial Phi_m).  Synchronous. */
int cuhe_hip_init(const int32_t *modulus, int ncoeffs);
/* 1 if the library is initialised and the CURRENT parameters (cuhe_hip_set_parameters) + this modulus describe the ring it was
 * initialised on, else 0.  cuhe_hip_init on the same ring is then a no-op that keeps tables, resident evaluation keys and every
 * block handed out: a second scheme object built from a key string (examples/DHS/DHS.cu:57-118: setParameters + initCuHE again,
 * examples/DHS/simple_DHS.cu:176-190) leaves the first one usable, as in the reference. */
int cuhe_hip_same_ring(const int32_t *modulus, int ncoeffs);
int cuhe_hip_is_initialised(void);       /* 1 between a successful cuhe_hip_init and cuhe_hip_shutdown */
int cuhe_hip_shutdown(void);
/* coefficient modulus q_lvl as little-endian bytes (initCuHE's ZZ* output, Operations.cu:157-160) */
int cuhe_hip_get_coeff_modulus(int lvl, uint8_t *le_bytes, size_t cap, size_t *len);
int cuhe_hip_get_crt_primes(uint32_t *out, int cap);
/* which fused poly-reduction the context selected: 0 = generic NTT Barrett, 1 = x^n+1, 2 = prime m */
int cuhe_hip_reduce_kind(void);
/* tests: 1 = force the generic NTT-Barrett path on rings that have a special kernel; 2 = additionally take its
   five-transform form instead of the folded one; 0 = default */
int cuhe_hip_force_generic_reduce(int on);

/* ---- allocator: startAllocator / stopAllocator (cuhe/CuHE.h:156,159; DeviceManager.cu:50-138) */
int cuhe_hip_start_allocator(void);
int cuhe_hip_stop_allocator(void);
/* `count` blocks of `bytes` bytes into the free pool of device `dev` ahead of use (fewer if the device runs out: a reserve, not a
   requirement).  The C++ startAllocator() reserves the one block size its classes use while the pooled allocator is on -- the bounded
   counterpart of the reference's allocator, which takes the whole device memory at start (cuhe/DeviceManager.cu:56-64). */
int cuhe_hip_reserve_blocks(int dev, size_t bytes, int count);
void *cuhe_hip_malloc(int dev, size_t bytes);
int cuhe_hip_free(int dev, void *ptr);
/* freed blocks are parked for reuse (hipMalloc/hipFree cost more than a CRT or NTT stage): without limit between
   start/stopAllocator, up to this many bytes per device otherwise (default 4 GiB; 0 = plain hipMalloc/hipFree) */
int cuhe_hip_set_alloc_cache(size_t bytes);
/* diagnostics: out4 = { hipMalloc calls, allocations served from the settled pool, from the caller stream's parked blocks,
   blocks handed over from another stream's parked set behind an event } since the library was loaded */
int cuhe_hip_alloc_counters(long long *out4);
/* test hook (failure injection): the (n+1)-th allocation from now on -- cuhe_hip_malloc or cuhe_hip_malloc_stream reaching it -- fails the
   way an exhausted device does (returns NULL, cuhe_hip_last_error names it); n < 0 switches the hook off.  The reference's reaction to
   a failed cudaMalloc is CSC: message + exit(-1) (cuhe/Debug.h:35-66), which is what the C++ layer does with the NULL. */
int cuhe_hip_set_alloc_fail_after(long n);
/* counts the cuhe_hip_shutdown calls so far: every device block the library handed out before a shutdown is gone with it */
unsigned long long cuhe_hip_generation(void);
/* stream-ordered variants: a block freed with free_stream is reused only by malloc_stream calls for the same stream
   until cuhe_hip_stream_sync(stream) has returned; callers can then enqueue chains of operations without a host
   synchronisation between them */
void *cuhe_hip_malloc_stream(int dev, size_t bytes, void *stream);
int cuhe_hip_free_stream(int dev, void *ptr, void *stream);
/* pinned host staging buffers for z2r / r2z (cuhe/CuHE.cu:317-348 stages through pageable memory) */
void *cuhe_hip_host_alloc(size_t bytes);
int cuhe_hip_host_free(void *ptr);
int cuhe_hip_memset_async(int dev, void *ptr, int value, size_t bytes, void *stream);
int cuhe_hip_memcpy_h2d(int dev, void *dst, const void *src, size_t bytes, void *stream);
int cuhe_hip_memcpy_d2h(int dev, void *dst, const void *src, size_t bytes, void *stream);
int cuhe_hip_memcpy_d2d(int dev, void *dst, const void *src, size_t bytes, void *stream);
/* moveTo / copyTo transport (cuhe/CuHE.cu:217-256: cudaMemcpyPeerAsync) */
int cuhe_hip_memcpy_peer(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, void *stream);
/* Streams for callers without HIP headers (the C++ layer's cudaStream_t is this void*).  The library is re-entrant
   per host thread: each thread owns its scratch on every device, so T threads with T streams keep T independent
   ciphertext operations in flight on one GPU (the reference has one scratch set per device, Operations.cu:171-209). */
int cuhe_hip_stream_create(int dev, void **stream_out);
int cuhe_hip_stream_destroy(int dev, void *stream);
int cuhe_hip_stream_sync(int dev, void *stream);
/* waits for all work on the device (every stream); blocks freed in stream order become reusable by any stream */
int cuhe_hip_device_sync(int dev);
/* Events (hipEvent_t as void*, timing disabled): the ordering primitive of the C++ layer's gate scheduler
   (cuhe_amd/cxx/Scheduler.h), which runs independent gates of one client thread on several streams.  The reference has
   one stream per device and a cudaStreamSynchronize after every gate instead (cuhe/CuHE.cu:98,121,139,157).
   event_query returns CUHE_OK when the recorded work has finished and 1 while it is pending. */
int cuhe_hip_event_create(int dev, void **event_out);
int cuhe_hip_event_destroy(int dev, void *event);
int cuhe_hip_event_record(int dev, void *event, void *stream);
int cuhe_hip_stream_wait_event(int dev, void *stream, void *event);
int cuhe_hip_event_sync(int dev, void *event);
int cuhe_hip_event_query(int dev, void *event);

/* ---- operation drivers (cuhe/Operations.h:60-108), same argument order */
int cuhe_hip_crt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *stream);            /* Operations.cu:245 */
int cuhe_hip_icrt(uint32_t *dst, const uint32_t *src, int logq, int dev, void *stream);           /* Operations.cu:254 */
int cuhe_hip_crt_add(uint32_t *sum, const uint32_t *x, const uint32_t *y, int logq, int dev, void *stream);       /* :264 */
int cuhe_hip_crt_add_int(uint32_t *sum, const uint32_t *x, unsigned a, int logq, int dev, void *stream);          /* :272 */
int cuhe_hip_crt_add_nx1(uint32_t *sum, const uint32_t *x, const uint32_t *scalar, int logq, int dev, void *stream); /* :280 */
int cuhe_hip_crt_mul_int(uint32_t *prod, const uint32_t *x, int a, int logq, int dev, void *stream);              /* :288 */
int cuhe_hip_crt_mod_switch(uint32_t *dst, const uint32_t *src, int logq, int dev, void *stream);                 /* :296 */
int cuhe_hip_ntt(uint64_t *X, const uint32_t *x, int logq, int dev, void *stream);                /* Operations.cu:394 */
int cuhe_hip_nttw(uint64_t *X, const uint32_t *x, int logq, int dev, void *stream);               /* Operations.cu:399 */
int cuhe_hip_intt(uint32_t *x, const uint64_t *X, int logq, int dev, void *stream);               /* Operations.cu:419 */
int cuhe_hip_intt_hold(const uint64_t *X, int logq, int dev, void *stream);                        /* Operations.cu:405 */
int cuhe_hip_intt_double_deg(uint32_t *x, const uint64_t *X, int logq, int dev, void *stream);    /* Operations.cu:412 */
int cuhe_hip_intt_mod(uint32_t *x, const uint64_t *X, int logq, int dev, void *stream);           /* Operations.cu:429 */
uint32_t *cuhe_hip_intt_result(int dev);                                                           /* Operations.cu:185 */
/* diagnostics: the kernel form the calling thread's last transform call was dispatched to (two-pass pair / one workgroup per
   half / persistent with rendezvous / split rows), its rows and length, and how many workgroups of this thread's persistent
   launches have given their rendezvous up so far (performance only: such a launch runs on at the speed of the plain form) */
int cuhe_hip_last_dispatch_info(int dev, char *buf, size_t cap);
uint64_t *cuhe_hip_ntt_swap(int dev);     /* ptrNttSwap(dev), Operations.cu:190: the calling thread's u64[nttLen] transform scratch */
int cuhe_hip_ntt_mul(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *stream);         /* :435 */
int cuhe_hip_ntt_mul_nx1(uint64_t *z, const uint64_t *x, const uint64_t *scalar, int logq, int dev, void *stream); /* :441 */
int cuhe_hip_ntt_add(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *stream);         /* :447 */
int cuhe_hip_ntt_add_nx1(uint64_t *z, const uint64_t *x, const uint64_t *scalar, int logq, int dev, void *stream); /* :453 */
/* barrett(dst, src, lvl, ...) and barrett(dst, lvl, ...) (Operations.cu:460-504); src = u32[np][nttLen] */
int cuhe_hip_barrett(uint32_t *dst, const uint32_t *src, int lvl, int dev, void *stream);
int cuhe_hip_barrett_hold(uint32_t *dst, int lvl, int dev, void *stream);

/* ---- ciphertext-domain ("ct") transforms: the NTT representation CuCtxt / CuPtxt keep their polynomials in
 * (cuhe/CuHE.cu:383-408 c2n / n2c, :101-215 gates, :570-581 relin).  On general rings it IS the reference's: cyclic
 * transforms of nttLen points of the zero-padded residues, products reduced modulo Phi_m afterwards
 * (cuhe/Operations.cu:394-504) -- the functions above.  When the polynomial modulus is x^n + 1 with n a transform
 * length (16384 / 32768 / 65536) and every CRT prime obeys 2 n p^2 < P (and 2 k n 2^w p < P for the key switch), it
 * is the NEGACYCLIC transform of n points, X[k] = sum_j x[j] psi^(j (2k+1)) with psi a primitive 2n-th root of unity
 * (psi = g for n = 32768): half the points per polynomial, half the bytes per evaluation key, and a product of
 * transforms is already the product modulo x^n + 1 -- the reduction of cuhe/Operations.cu:460-501 disappears.  CRT- and
 * raw-domain results are bit-identical either way.  Ring degree 65536 (m = 131072, beyond the reference's 65536-point
 * limit, cuhe/Base.cu:59-62) exists in this representation only: there the cyclic functions above fail with CUHE_EINVAL
 * and setParameters picks primes of at most 23 bits.
 * Row length of a ct-domain polynomial: cuhe_hip_ct_len() (= modLen or nttLen).  ct arrays are u64[rows][ct_len]. */
int cuhe_hip_set_negacyclic(int mode);   /* before init: -1 = negacyclic wherever it applies (default), 0 = never */
/* How many PRODUCTS of reduced polynomials may be summed in the ct domain (cuhe_hip_ct_add of isProd operands) before
 * cuhe_hip_ct_intt(..., is_prod = 1): the integer coefficients of the sum must stay inside the range the inverse
 * transform recovers (P in the cyclic representation, +-P/2 in the negacyclic one, where it is ONE on the largest
 * rings: 2 n p^2 is just below P).  The C ABI does not track this; the C++ gates (cXor, CuHE.cpp) do and reduce first. */
int cuhe_hip_ct_prod_headroom(void);
int cuhe_hip_ct_negacyclic(void);        /* 1 if the initialised context uses the negacyclic representation */
int cuhe_hip_ct_len(void);
int cuhe_hip_ct_ntt(uint64_t *X, const uint32_t *x, int logq, int dev, void *stream);                /* c2n, CuHE.cu:383 */
/* n2c, CuHE.cu:394-408: is_prod != 0 -> the rows are products (inttMod: reduction modulo the polynomial modulus) */
int cuhe_hip_ct_intt(uint32_t *x, const uint64_t *X, int logq, int is_prod, int dev, void *stream);
int cuhe_hip_ct_mul(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *stream);       /* cAnd */
int cuhe_hip_ct_mul_nx1(uint64_t *z, const uint64_t *x, const uint64_t *scalar, int logq, int dev, void *stream);
int cuhe_hip_ct_add(uint64_t *z, const uint64_t *y, const uint64_t *x, int logq, int dev, void *stream);       /* cXor in the NTT domain */
int cuhe_hip_ct_add_nx1(uint64_t *z, const uint64_t *x, const uint64_t *scalar, int logq, int dev, void *stream);

/* single-polynomial forms _ntt / _nttw / _intt (cuhe/Operations.h:78-81) */
int cuhe_hip_ntt_one(uint64_t *X, const uint32_t *x, int dev, void *stream);
int cuhe_hip_nttw_one(uint64_t *X, const uint32_t *x, int coeffwords, int relinIdx, int dev, void *stream);
int cuhe_hip_intt_one(uint32_t *x, const uint64_t *X, int crtidx, int dev, void *stream);

/* ---- relinearisation (cuhe/Relinearization.h; CuHE.h:178) */
/* initRelinearization: evalkey = numEvalKey polynomials in raw layout at level 0,
 * HOST memory u32[numEvalKey][rawLen][W0]; keys are converted once and stay in HBM. */
int cuhe_hip_init_relin(const uint32_t *evalkey_raw_host);
/* The same for the keys of the CRT primes [prime0, prime0 + count) only: what ONE participant of the CRT-prime-sharded
 * multiply needs (the evaluation keys are partitioned with the primes; key memory / participants).
 * cuhe_hip_key_range gives the range that covers a participant's block at every level (its contiguous block moves down
 * as the levels drop primes).  Entry points that need every prime's keys (cuhe_hip_relinearization over a whole level,
 * the batched calls, the key cache) fail with a message on such a device; cuhe_hip_relin_range and
 * cuhe_hip_mul_relin_sharded work on the owned primes.  cuhe_hip_init_relin_sharded: device d of multiGPUs(n) keeps
 * the range of participant d (for cuhe_hip_mul_relin_sharded_inproc).  Replaces Relinearization.cu:43-75 like
 * cuhe_hip_init_relin. */
int cuhe_hip_init_relin_range(const uint32_t *evalkey_raw_host, int prime0, int count);
int cuhe_hip_key_range(int nranks, int rank, int *first, int *count);
int cuhe_hip_init_relin_sharded(const uint32_t *evalkey_raw_host);
/* relinearization(dst, src, lvl, dev, st) (Relinearization.cu:76-88): src raw, dst in the ct domain u64[np][ct_len]
 * (the keys are kept in the ct domain only; the reference's sole consumer of dst is n2c, CuHE.cu:570-581) */
int cuhe_hip_relinearization(uint64_t *dst, const uint32_t *src, int lvl, int dev, void *stream);
/* `batch` independent (cAnd ; CuCtxt::relin) chains of one level in one call (CuHE.cu:101,570-581 issue them one
   ciphertext at a time): a, b = ct-domain operands u64[batch][np][ct_len], dst = reduced CRT-domain results
   u32[batch][np][crtLen], np = primes of `lvl`.  Bit-identical to the single-ciphertext sequence
   ntt_mul, intt_mod, icrt, relinearization, intt_mod; every stage runs over batch*np rows and the key-switch inner
   product reads each key value once per four ciphertexts. */
int cuhe_hip_mul_relin_batch(uint32_t *dst, const uint64_t *a_ntt, const uint64_t *b_ntt, int lvl, int batch, int dev, void *stream);
/* CuCtxt::relin for `batch` reduced CRT-domain ciphertexts of one level: src, dst = u32[batch][np][crtLen]
   (dst may be src); bit-identical to icrt, relinearization, intt_mod per ciphertext */
int cuhe_hip_relin_batch(uint32_t *dst, const uint32_t *src_crt, int lvl, int batch, int dev, void *stream);
/* Optional: on large rings (>= 1 GiB of keys per level) the two calls above can hand groups of four ciphertexts round-robin
   to `n` streams: the caller's stream and n-1 helper streams of the calling thread, own scratch each; the caller's stream
   waits for the helpers before the call's work counts as done on it.  n = 1..4, default 1 = everything on the caller's
   stream in one launch sequence (fastest since the inner product shares key fetches across the whole batch);
   -n = n streams whatever the ring size (tests).  Results do not depend on n. */
int cuhe_hip_set_relin_lanes(int n);
/* The key-switch inner product of the two batched calls above runs on the matrix cores (signed base-256 digits,
   v_mfma_i32_16x16x64_i8, 16 ciphertexts x 16 primes x 64 windows per instruction; exact, so results do not change) for
   batches of at least `min_batch` ciphertexts and parameter sets with at most 128 evaluation keys; the key digits (as
   large as the keys) are laid out on the first such call; a device without room for them keeps the VALU kernel.
   Default 5 (the measured crossover); 0 = never (the VALU kernel). */
int cuhe_hip_set_relin_mfma(int min_batch);
/* The inverse CRT (cuhe_hip_icrt and every chain that contains it; cuhe/Base.cu:884-960) forms its column sums
   sum_i t_i (M / p_i) on the matrix cores (base-128 digits of t_i against signed base-256 digits of the constants,
   v_mfma_i32_32x32x32_i8, exact) for parameter sets with primes below 2^28 and at most 47 words per coefficient; others
   keep the VALU kernel.  on = 1 (default) / 0 = the VALU kernel everywhere.  Results do not depend on the setting. */
int cuhe_hip_set_icrt_mfma(int on);
/* The CRT (cuhe_hip_crt and every chain that contains it; cuhe/Base.cu:857-879) keeps its sums  sum_k word_k (2^(32k) mod p)
   in 64 bits without a carry word wherever they fit (W 2^32 pmax <= 2^64: one multiply-add per term, one reduction).
   on = 1 (default) / 0 = the 96-bit form everywhere.  Results do not depend on the setting. */
int cuhe_hip_set_crt_acc64(int on);
/* `batch` independent full multiplications raw -> raw of one level in one call (mulZZX without the host staging,
   CuHE.cu:259-268): a, b, dst = u32[batch][rawLen][W], W = words of the level's coefficients; bit-identical to the
   single sequence crt, crt, ntt, ntt, ntt_mul, intt_mod, icrt */
int cuhe_hip_mul_raw_batch(uint32_t *dst, const uint32_t *a_raw, const uint32_t *b_raw, int lvl, int batch, int dev, void *stream);
/* ---- gates on ARRAYS of ciphertexts of one level (no counterpart: the reference's gates, CuHE.cu:101-215,545-568,
   take one ciphertext per call).  Arrays are u32[count][np][crtLen] (CRT domain) or u64[count][np][ct_len] (ct domain);
   index arrays live in device memory.  Each call is bit-identical to the per-ciphertext gates it stands for. */
/* n2c of products: inverse transform + reduction modulo the polynomial modulus for `batch` ciphertexts */
int cuhe_hip_intt_mod_batch(uint32_t *dst_crt, const uint64_t *src_ntt, int lvl, int batch, int dev, void *stream);
/* modSwitch: src at level lvl (np rows each) -> dst at level lvl+1, packed u32[batch][np-1][crtLen] */
int cuhe_hip_crt_mod_switch_batch(uint32_t *dst, const uint32_t *src, int lvl, int batch, int dev, void *stream);
/* `count` equally sized blocks (bytes: a multiple of 16, blocks 16-byte aligned) between their own addresses and one contiguous
   array; the pointer list is HOST memory and travels as a kernel argument.  What lets the C++ layer's gate scheduler run the
   ready gates of one kind on separately owned ciphertexts as one call of the array entry points above and below. */
int cuhe_hip_gather_blocks(void *dst, const void *const *srcs, int count, size_t bytes, int dev, void *stream);
/* The two transforms that touch the ct-sized rows of such separately owned ciphertexts, WITHOUT the gather / scatter: the one-workgroup
   kernels find row r of ciphertext r / np in that ciphertext's own block (the block offsets travel as a kernel argument, 128 blocks per
   launch).  ct_ntt_list: c2n of every ciphertext, dst[i] u64[np][ct_len] <- src[i] u32[np][crtLen] (CuCtxt::x2n, cuhe/CuHE.cu:392-411, once
   per ciphertext there); ct_intt_list: n2c into ONE array dst u32[count][np][crtLen] <- src[i] u64[np][ct_len], is_prod: + the reduction
   modulo the polynomial modulus (CuHE.cu:412-431).  A call whose row count or length takes another kernel form gathers / scatters through
   scratch of its own instead (same results); *direct (may be NULL): how many ciphertexts went through the kernels' block addressing.
   cuhe_hip_set_row_lists(0): always the gather / scatter form (A/B runs; CUHE_ROW_LISTS=0). */
int cuhe_hip_ct_ntt_list(uint64_t *const *dst, const uint32_t *const *src, int count, int lvl, int dev, void *stream, int *direct);
int cuhe_hip_ct_intt_list(uint32_t *dst, const uint64_t *const *src, int count, int lvl, int is_prod, int dev, void *stream, int *direct);
int cuhe_hip_set_row_lists(int on);
int cuhe_hip_scatter_blocks(void *const *dsts, const void *src, int count, size_t bytes, int dev, void *stream);
/* elementwise gates over LISTS of separately owned ciphertexts of one level, one launch per 64: ct rows z = x * y (mul != 0) or
   x + y modulo P (cAnd / cXor in the NTT domain), CRT rows z = (a + b) mod p (cXor in the CRT domain); lists in host memory */
int cuhe_hip_ct_binop_list(int mul, void *const *z, const void *const *x, const void *const *y, int count, int logq, int dev, void *stream);
int cuhe_hip_crt_add_list(void *const *z, const void *const *a, const void *const *b, int count, int logq, int dev, void *stream);
/* modSwitch over a list: dst[i] <- src[i] one level down (np -> np - 1 rows), dst[i] == src[i] allowed (in place, in the ciphertext's own
   block: CuCtxt::modSwitch, CuHE.cu:583-594); cNot over a list: z[i] = x[i] + a on the constant coefficient of every row, other
   coefficients copied when z[i] != x[i] (cNot, CuHE.cu:176-187; crt_add_int, Base.cu:1096-1100).  One launch per 64 ciphertexts. */
int cuhe_hip_crt_mod_switch_list(void *const *dst, const void *const *src, int lvl, int count, int dev, void *stream);
int cuhe_hip_crt_add_int_list(void *const *z, const void *const *x, unsigned a, int count, int logq, int dev, void *stream);
/* dst[i] = src[i] for `count` separately owned blocks of `bytes` bytes (a multiple of 16): copy() over a list, one launch per 64 */
int cuhe_hip_copy_list(void *const *dst, const void *const *src, int count, size_t bytes, int dev, void *stream);
/* n2c of `batch` NON-product ciphertexts of one level in one array (intt_mod_batch is the form for products) */
int cuhe_hip_intt_batch(uint32_t *dst_crt, const uint64_t *src_ntt, int lvl, int batch, int dev, void *stream);
/* cAnd over index pairs: dst[t] = src[idx_a[t]] * src[idx_b[t]], ciphertexts of np_rows rows */
int cuhe_hip_ntt_mul_pairs(uint64_t *dst, const uint64_t *src, const int32_t *idx_a, const int32_t *idx_b, int npairs, int np_rows, int dev, void *stream);
/* cXor / cNot over index lists: dst[o] = sum of the listed ciphertexts (+ add_const[o] on the constant coefficient);
   list entries e < nA address src_a[e], the others src_b[e - nA]; list[off[o] .. off[o+1]) belongs to output o */
int cuhe_hip_crt_combine(uint32_t *dst, const uint32_t *src_a, int nA, const uint32_t *src_b, const int32_t *off, const int32_t *list,
                         const int32_t *add_const, int nout, int lvl, int dev, void *stream);
/* binary evaluation-key cache: the NTT-domain keys initRelinearization computes (u64[prime][key][nttLen],
   Relinearization.cu:45-55) behind a 96-byte header naming the parameter set and the CRT primes; import refuses
   an image made for other parameters / primes or with a damaged payload.  cache_size = 0 before init. */
size_t cuhe_hip_relin_cache_size(void);
int cuhe_hip_relin_export(void *dst_host, size_t capacity, int dev);
int cuhe_hip_relin_import(const void *src_host, size_t bytes);

/* ---- CRT-prime-sharded variants (new): one rank owns primes [prime0, prime0+count) of level
 * `lvl`; row pointers address the shard's own rows (row 0 = prime0), NTT-domain rows are ct rows (ct_len).  Per-prime stages need no communication;
 * the only exchange of a sharded multiply+relinearise is the all-gather of CRT rows before cuhe_hip_icrt. */
int cuhe_hip_ntt_rows(uint64_t *X, const uint32_t *x, int count, int dev, void *stream);
int cuhe_hip_ntt_mul_rows(uint64_t *z, const uint64_t *y, const uint64_t *x, int count, int dev, void *stream);
int cuhe_hip_intt_mod_range(uint32_t *x, const uint64_t *X, int lvl, int prime0, int count, int dev, void *stream);
int cuhe_hip_relin_range(uint64_t *dst, const uint32_t *raw, int lvl, int prime0, int count, int dev, void *stream);
int cuhe_hip_crt_range(uint32_t *dst, const uint32_t *src, int logq, int prime0, int count, int dev, void *stream);

/* CuCtxt::relin from the raw domain on -- relinearization ; n2c (cuhe/CuHE.cu:574-580) -- as ONE call: raw coefficients u32[rawLen][W] of
 * level lvl -> reduced CRT rows u32[np][crtLen]; the sums live in the calling thread's scratch.  Bit-identical to
 * cuhe_hip_relinearization followed by cuhe_hip_ct_intt(is_prod = 1). */
int cuhe_hip_relin_crt(uint32_t *dst_crt, const uint32_t *src_raw, int lvl, int dev, void *stream);

/* ---- multi-GPU: cAnd + relin of ONE ciphertext with the level's CRT primes sharded over GPUs (new; the reference's
 * multi-GPU mode is whole ciphertexts per GPU, cuhe/CuHE.cu:217-256, which the dev argument of every call above already
 * serves).  Participant r owns the contiguous block cuhe_hip_shard_bounds(lvl, n, r) of the level's primes; the only
 * exchange is the all-gather of CRT rows before ICRT.
 * (1) one process per GPU: RCCL, opened at run time (no link-time dependency).  Rank 0 calls comm_unique_id, the
 *     launcher distributes the 128 bytes (bench.py: torch.distributed broadcast), every rank calls comm_init after
 *     cuhe_hip_set_device_base(local rank).  The collective is enqueued on the caller's stream. */
int cuhe_hip_shard_bounds(int lvl, int nranks, int rank, int *first, int *count);
int cuhe_hip_comm_unique_id(void *id_128_bytes);
int cuhe_hip_comm_init(int nranks, int rank, const void *id_128_bytes);
int cuhe_hip_comm_destroy(void);
int cuhe_hip_comm_size(void);
int cuhe_hip_comm_rank(void);
/* diagnostics for the first contact with N > 1 GPUs: what RCCL itself reports about the communicator (version, ncclCommCount,
   ncclCommUserRank) beside the library's view, how many exchanges took which path and which the last one took.
   comm_force_exchange(on): tests -- on > 0 runs the exchange on a communicator of ONE rank too (by default one rank has nothing
   to exchange); 1 = the policy below, 2 = the padded all-gather, 3 = the group of broadcasts; 0 = off. */
int cuhe_hip_comm_info(char *buf, size_t cap);
int cuhe_hip_comm_force_exchange(int on);
/* the form the exchange of level lvl takes on nranks ranks: 0 none (one rank), 1 ONE in-place ncclAllGather (blocks equal: the
   level's primes are a multiple of nranks), 2 ONE ncclAllGather of blocks padded to the largest (staging buffer + two strided
   copies), 3 a group of ncclBroadcast (only when a rank owns no prime, or forced).  Host logic, no GPU: the policy is testable
   for every (level, nranks).  force as in comm_force_exchange.  -1 on bad arguments. */
int cuhe_hip_exchange_path(int lvl, int nranks, int force);
/* rows = u32[np][crtLen] of level lvl with this rank's block in place -> every block in place (stream ordered): the one
   collective of the sharded multiply ("RCCL all-gather only at the ICRT recombine step") */
int cuhe_hip_allgather_rows(uint32_t *rows, int lvl, int dev, void *stream);
/* a_own, b_own: ct rows of the rank's primes u64[count][ct_len]; dst_own: reduced CRT rows u32[count][crtLen] */
int cuhe_hip_mul_relin_sharded(uint32_t *dst_own, const uint64_t *a_own, const uint64_t *b_own, int lvl, int dev, void *stream);
/* (2) one process driving the multiGPUs(n) devices: a, b = ct rows of all primes on device dev0, dst = reduced CRT rows
 *     of all primes on dev0; rows travel by peer copies over xGMI ordered by events, each device on its own stream; the
 *     caller's stream continues when every device is done.  Host threads may call it concurrently: the enqueue is
 *     serialised inside (the helper streams and stage events are per device), the enqueued work still overlaps. */
int cuhe_hip_mul_relin_sharded_inproc(uint32_t *dst, const uint64_t *a, const uint64_t *b, int lvl, int dev0, void *stream);

/* ---- batched transform primitives (the shape tests/test_ntt.cu:67-100 times):
 * `batch` independent length-`len` transforms, len in {16384, 32768, 65536}. */
/* forward: src u32[batch][src_stride] (only the first len/2 of each row are read, zero padded),
 *          dst u64[batch][len], natural order. */
int cuhe_hip_ntt_fwd_batched(uint64_t *dst, const uint32_t *src, int len, int batch, long src_stride,
                             int dev, void *stream);
/* inverse: src u64[batch][len]; dst u32[batch][dst_stride], first `nstore` outputs of each row;
 *          row b is reduced modulo primes[prime0 + b] of the initialised context. */
int cuhe_hip_ntt_inv_batched(uint32_t *dst, const uint64_t *src, int len, int batch, long dst_stride,
                             int nstore, int prime0, int dev, void *stream);
/* standalone transform tables (no cuhe_hip_init needed): prepares twiddles + scratch for `len` on `dev` */
int cuhe_hip_ntt_prepare(int len, int dev);
/* batch chunk (transforms per launch pair) used to keep the pass-1 -> pass-2 slab cache resident; 0 = default */
int cuhe_hip_set_ntt_chunk(int chunk);
/* 1 (default): pass 2 of chunk c runs concurrently with pass 1 of chunk c+1 on two internal streams
 * (joined back into `stream` before the call returns control of it); 0: strictly serial launches */
int cuhe_hip_set_ntt_overlap(int on);
/* Transform calls of at most rows * 32768 points in total (`rows` rows of 32K points, half as many of 64K points) take
 * the low-latency kernel pair (4 values per thread, 4x the workgroups: the duration of a lone ciphertext operation is
 * the latency of its small kernels); larger calls the throughput pair (16 values per thread); pass 2 alone keeps the
 * low-latency form up to twice that size (the two forms share the slab layout).  Same results.  Default
 * 24 = the measured crossover (profiles/r02_small_batch_latency.txt); 0 = never; a large value = always (the parity
 * tests run both forms).  Environment CUHE_LL_ROWS overrides the default for A/B runs of whole programs. */
int cuhe_hip_set_ll_rows(int rows);
/* One-workgroup transforms (cuhe_amd/csrc/ntt_onewg.cuh): a sub-transform of 8K / 16K / 32K points in the registers of
 * one workgroup, ONE launch and no slab in HBM; the zero-padded forward transforms of 16K / 32K / 64K points run as their
 * two half-length halves.  mode 1 (default): wherever that form exists and the call has enough rows to fill the chip with
 * its workgroups (1 / 2 / 4 per CU at 32K / 16K / 8K points; below that the two-pass kernels finish sooner); 2: wherever
 * it exists, whatever the row count (the parity tests run every form); 0: the two-pass kernels only.
 * rows64k selects the form of zero-padded rows of 64K points, whose 32K-point halves leave room for ONE workgroup per
 * CU: 0 the two-pass kernels, 1 one workgroup per half, 2 (default) persistent workgroups (one per CU, the next half's
 * samples prefetched into LDS by DMA, the two halves of a row meeting before their interleaved stores) for calls that
 * give every workgroup at least two halves and 16-byte aligned rows, the two-pass kernels otherwise (2.71 vs 2.56 M
 * transforms/s, profiles/r03_onewg_ab.txt); 3: the persistent form for zero-padded rows of 32K points too (measured 3 %
 * slower than one workgroup per half there: two workgroups per CU already overlap).  Same results in every form.
 * cuhe_hip_set_onewg_split: full-length INVERSE negacyclic rows of 32K points (the ciphertext domain of x^32768 + 1) run,
 * when the call fills the chip, SPLIT into the two 16K-point transforms of their even and odd outputs, two workgroups per
 * CU (mode 1, default: 13 % faster than one 32K-point workgroup per row); 0: never; 2: also the forward rows of 32K points
 * and the inverse rows of 64K points (no gain measured: parity tests and A/B runs).  Environment CUHE_ONEWG_SPLIT.
 * Environment CUHE_ONEWG / CUHE_ONEWG64 override the defaults for A/B runs of whole programs.  Replaces the same
 * reference code as the two-pass kernels (cuhe/Base.cu:309-842, cuhe/Operations.cu:306-398). */
int cuhe_hip_set_onewg(int mode, int rows64k);
int cuhe_hip_set_onewg_split(int mode);
/* name / average duration bookkeeping for bench.py: time the dominant kernel with hipEvents on `stream`.
 * Runs `iters` forward batched transforms and returns total milliseconds in *ms_pass1 / *ms_pass2 / *ms_total. */
int cuhe_hip_time_ntt_fwd(uint64_t *dst, const uint32_t *src, int len, int batch, int iters, int dev, void *stream,
                          float *ms_pass1, float *ms_pass2, float *ms_total);

/* measurement of the limiter the transforms run against (bench.py roofline.valu_ceiling): a dense stream of the
 * 64-bit integer instructions the field arithmetic lowers to (v_mad_u64_u32, v_lshl_add_u64, v_cmp_lt_u64, equal parts), every
 * SIMD of the chip filled with `waves_per_simd` waves (1 ... 8), repeated for about `millis` ms.  Returns the sustained rate in
 * lane-instructions per second (*lane_instr_per_s), the shader clock the chip settled at under that load (*shader_mhz: s_memtime
 * ticks of a wave per microsecond of kernel time) and the issue cost (*cycles_per_instr: shader cycles per wave-instruction and
 * SIMD).  No counterpart in the reference; a diagnostic, never on the product path. */
int cuhe_hip_probe_valu(int dev, int waves_per_simd, int millis, double *lane_instr_per_s, double *shader_mhz, double *cycles_per_instr);
/* The streaming-copy ceiling of the box (bench.py roofline.measured_copy_GBs): a copy of `bytes` bytes with 16-byte accesses in launch
 * shape `variant` (0 ... cuhe_hip_probe_copy_shapes() - 1; cuhe_hip_probe_copy_name says which: grid-stride at several occupancies, one
 * element per thread, several loads in flight, contiguous chunks per workgroup, non-temporal accesses), `reps` timed launches between
 * hipEvents; *gb_per_s = (bytes read + bytes written) / time.  A diagnostic like cuhe_hip_probe_valu; the HBM-bound kernels of the path
 * (key stream, ICRT, pointwise) are priced against the best shape. */
int cuhe_hip_probe_copy_shapes(void);
const char *cuhe_hip_probe_copy_name(int variant);
int cuhe_hip_probe_copy(int dev, size_t bytes, int variant, int reps, double *gb_per_s);

/* ---- field arithmetic test hooks (tests/test_ModP.cu:50-135): elementwise over n u64 */
int cuhe_hip_modp_add(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *stream);
int cuhe_hip_modp_sub(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *stream);
int cuhe_hip_modp_mul(uint64_t *z, const uint64_t *x, const uint64_t *y, size_t n, int dev, void *stream);
int cuhe_hip_modp_shl(uint64_t *z, const uint64_t *x, int l, size_t n, int dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUHE_HIP_H */
