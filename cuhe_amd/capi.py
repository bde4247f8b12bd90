"""ctypes binding of libcuhe_hip.so (include/cuhe_hip.h).  No fallback path:
if the HIP library is missing or fails to load this module raises ImportError."""
import ctypes as C
import os

from . import build as _build

LIB_PATH = os.environ.get("CUHE_HIP_LIB", _build.LIB)   # override: A/B builds of the same ABI (tools/ab_bench.sh)
if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cuhe_amd: %s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the HIP path)" % LIB_PATH)
# PyTorch bundles its own HIP runtime under the soname the library links (libamdhip64.so.7).  Whoever loads first decides
# which copy the process uses; a process that loads this library first and torch afterwards ends up with /opt/rocm's runtime
# under torch and loses the device ("no ROCm-capable device is detected" at the first call here).  Python callers hand
# torch tensors to the library anyway, so torch's runtime is loaded first; C++ clients (libcuHE.so) never see torch.
try:
    import torch  # noqa: F401
except ImportError:
    pass
lib = C.CDLL(LIB_PATH)


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mSize", "modLen", "modLen2", "rawLen", "crtLen", "nttLen",
        "logCoeffMax", "logCoeffMin", "logCoeffCut",
        "depth", "modMsg", "logMsg", "wordsMsg",
        "logRelin", "numEvalKey", "logCrtPrime", "numCrtPrime")]


i32, u32, u64, vp, sz, lng = C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t, C.c_long

# name -> (restype, argtypes); mirrors include/cuhe_hip.h one to one
SIGNATURES = {
    "cuhe_hip_last_error": (C.c_char_p, []),
    "cuhe_hip_version": (C.c_char_p, []),
    "cuhe_hip_set_parameters": (i32, [i32] * 6),
    "cuhe_hip_reset_parameters": (i32, []),
    "cuhe_hip_get_parameters": (i32, [C.POINTER(Params)]),
    "cuhe_hip_num_crt_prime": (i32, [i32]),
    "cuhe_hip_log_coeff": (i32, [i32]),
    "cuhe_hip_words_coeff": (i32, [i32]),
    "cuhe_hip_num_eval_key": (i32, [i32]),
    "cuhe_hip_get_level": (i32, [i32]),
    "cuhe_hip_multi_gpus": (i32, [i32]),
    "cuhe_hip_num_gpus": (i32, []),
    "cuhe_hip_set_virtual_devices": (i32, [i32]),
    "cuhe_hip_set_device_base": (i32, [i32]),
    "cuhe_hip_init": (i32, [vp, i32]),
    "cuhe_hip_same_ring": (i32, [vp, i32]),
    "cuhe_hip_shutdown": (i32, []),
    "cuhe_hip_get_coeff_modulus": (i32, [i32, vp, sz, C.POINTER(sz)]),
    "cuhe_hip_get_crt_primes": (i32, [vp, i32]),
    "cuhe_hip_reduce_kind": (i32, []),
    "cuhe_hip_force_generic_reduce": (i32, [i32]),
    "cuhe_hip_device_local_cpus": (i32, [i32, vp, sz]),
    "cuhe_hip_pin_thread_to_device": (i32, [i32]),
    "cuhe_hip_start_allocator": (i32, []),
    "cuhe_hip_reserve_blocks": (i32, [i32, sz, i32]),
    "cuhe_hip_stop_allocator": (i32, []),
    "cuhe_hip_malloc": (vp, [i32, sz]),
    "cuhe_hip_free": (i32, [i32, vp]),
    "cuhe_hip_set_alloc_cache": (i32, [sz]),
    "cuhe_hip_alloc_counters": (i32, [vp]),
    "cuhe_hip_set_alloc_fail_after": (i32, [C.c_long]),
    "cuhe_hip_generation": (u64, []),
    "cuhe_hip_malloc_stream": (vp, [i32, sz, vp]),
    "cuhe_hip_free_stream": (i32, [i32, vp, vp]),
    "cuhe_hip_host_alloc": (vp, [sz]),
    "cuhe_hip_host_free": (i32, [vp]),
    "cuhe_hip_memset_async": (i32, [i32, vp, i32, sz, vp]),
    "cuhe_hip_memcpy_h2d": (i32, [i32, vp, vp, sz, vp]),
    "cuhe_hip_memcpy_d2h": (i32, [i32, vp, vp, sz, vp]),
    "cuhe_hip_memcpy_d2d": (i32, [i32, vp, vp, sz, vp]),
    "cuhe_hip_memcpy_peer": (i32, [vp, i32, vp, i32, sz, vp]),
    "cuhe_hip_stream_create": (i32, [i32, vp]),
    "cuhe_hip_stream_destroy": (i32, [i32, vp]),
    "cuhe_hip_device_sync": (i32, [i32]),
    "cuhe_hip_stream_sync": (i32, [i32, vp]),
    "cuhe_hip_event_create": (i32, [i32, vp]),
    "cuhe_hip_event_destroy": (i32, [i32, vp]),
    "cuhe_hip_event_record": (i32, [i32, vp, vp]),
    "cuhe_hip_stream_wait_event": (i32, [i32, vp, vp]),
    "cuhe_hip_event_sync": (i32, [i32, vp]),
    "cuhe_hip_event_query": (i32, [i32, vp]),
    "cuhe_hip_crt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_icrt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_crt_add": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_crt_add_int": (i32, [vp, vp, C.c_uint, i32, i32, vp]),
    "cuhe_hip_crt_add_nx1": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_crt_mul_int": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_mod_switch": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_nttw": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_intt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_intt_hold": (i32, [vp, i32, i32, vp]),
    "cuhe_hip_intt_double_deg": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_intt_mod": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_intt_result": (vp, [i32]),
    "cuhe_hip_ntt_swap": (vp, [i32]),
    "cuhe_hip_last_dispatch_info": (i32, [i32, vp, sz]),
    "cuhe_hip_ntt_mul": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt_mul_nx1": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt_add": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt_add_nx1": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_barrett": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_barrett_hold": (i32, [vp, i32, i32, vp]),
    "cuhe_hip_set_negacyclic": (i32, [i32]),
    "cuhe_hip_ct_negacyclic": (i32, []),
    "cuhe_hip_ct_len": (i32, []),
    "cuhe_hip_ct_ntt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_ct_intt": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_ct_mul": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ct_mul_nx1": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ct_add": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ct_add_nx1": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt_one": (i32, [vp, vp, i32, vp]),
    "cuhe_hip_nttw_one": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_intt_one": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_init_relin": (i32, [vp]),
    "cuhe_hip_relinearization": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_mul_raw_batch": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_intt_mod_batch": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_mod_switch_batch": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_ct_binop_list": (i32, [i32, vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_add_list": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_mod_switch_list": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_add_int_list": (i32, [vp, vp, u32, i32, i32, i32, vp]),
    "cuhe_hip_copy_list": (i32, [vp, vp, i32, sz, i32, vp]),
    "cuhe_hip_intt_batch": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_gather_blocks": (i32, [vp, vp, i32, sz, i32, vp]),
    "cuhe_hip_ct_ntt_list": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "cuhe_hip_ct_intt_list": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "cuhe_hip_set_row_lists": (i32, [i32]),
    "cuhe_hip_scatter_blocks": (i32, [vp, vp, i32, sz, i32, vp]),
    "cuhe_hip_ntt_mul_pairs": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_crt_combine": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_relin_batch": (i32, [vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_set_relin_lanes": (i32, [i32]),
    "cuhe_hip_set_relin_mfma": (i32, [i32]),
    "cuhe_hip_set_icrt_mfma": (i32, [i32]),
    "cuhe_hip_set_crt_acc64": (i32, [i32]),
    "cuhe_hip_mul_relin_batch": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cuhe_hip_relin_cache_size": (sz, []),
    "cuhe_hip_relin_export": (i32, [vp, sz, i32]),
    "cuhe_hip_relin_import": (i32, [vp, sz]),
    "cuhe_hip_ntt_rows": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_ntt_mul_rows": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_intt_mod_range": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "cuhe_hip_relin_range": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "cuhe_hip_crt_range": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "cuhe_hip_ntt_fwd_batched": (i32, [vp, vp, i32, i32, lng, i32, vp]),
    "cuhe_hip_ntt_inv_batched": (i32, [vp, vp, i32, i32, lng, i32, i32, i32, vp]),
    "cuhe_hip_init_relin_range": (i32, [vp, i32, i32]),
    "cuhe_hip_key_range": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "cuhe_hip_init_relin_sharded": (i32, [vp]),
    "cuhe_hip_is_initialised": (i32, []),
    "cuhe_hip_ct_prod_headroom": (i32, []),
    "cuhe_hip_shard_bounds": (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "cuhe_hip_comm_unique_id": (i32, [vp]),
    "cuhe_hip_comm_init": (i32, [i32, i32, vp]),
    "cuhe_hip_comm_destroy": (i32, []),
    "cuhe_hip_comm_size": (i32, []),
    "cuhe_hip_comm_rank": (i32, []),
    "cuhe_hip_comm_info": (i32, [vp, sz]),
    "cuhe_hip_comm_force_exchange": (i32, [i32]),
    "cuhe_hip_relin_crt": (i32, [vp, vp, i32, i32, vp]),
    "cuhe_hip_exchange_path": (i32, [i32, i32, i32]),
    "cuhe_hip_allgather_rows": (i32, [vp, i32, i32, vp]),
    "cuhe_hip_mul_relin_sharded": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_mul_relin_sharded_inproc": (i32, [vp, vp, vp, i32, i32, vp]),
    "cuhe_hip_set_ll_rows": (i32, [i32]),
    "cuhe_hip_set_onewg": (i32, [i32, i32]),
    "cuhe_hip_set_onewg_split": (i32, [i32]),
    "cuhe_hip_ntt_prepare": (i32, [i32, i32]),
    "cuhe_hip_set_ntt_chunk": (i32, [i32]),
    "cuhe_hip_set_ntt_overlap": (i32, [i32]),
    "cuhe_hip_probe_valu": (i32, [i32, i32, i32] + [C.POINTER(C.c_double)] * 3),
    "cuhe_hip_probe_copy": (i32, [i32, sz, i32, i32, C.POINTER(C.c_double)]),
    "cuhe_hip_probe_copy_shapes": (i32, []),
    "cuhe_hip_probe_copy_name": (C.c_char_p, [i32]),
    "cuhe_hip_time_ntt_fwd": (i32, [vp, vp, i32, i32, i32, i32, vp] + [C.POINTER(C.c_float)] * 3),
    "cuhe_hip_modp_add": (i32, [vp, vp, vp, sz, i32, vp]),
    "cuhe_hip_modp_sub": (i32, [vp, vp, vp, sz, i32, vp]),
    "cuhe_hip_modp_mul": (i32, [vp, vp, vp, sz, i32, vp]),
    "cuhe_hip_modp_shl": (i32, [vp, vp, i32, sz, i32, vp]),
}
for _n, (_r, _a) in SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype = _r
    _f.argtypes = _a


class CuheError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise CuheError("cuhe_hip status %d: %s" % (status, lib.cuhe_hip_last_error().decode()))
    return status


def get_params():
    q = Params()
    check(lib.cuhe_hip_get_parameters(C.byref(q)))
    return q
