"""ctypes binding of libtc_amd.so (the C ABI declared in include/tc_amd.h).

The shared library is the product: if it is missing or cannot create a context on a HIP
device the import / the call fails loudly -- there is no CPU or PyTorch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TC_AMD_LIB") or os.path.join(_HERE, "libtc_amd.so")  # TC_AMD_LIB: experiment builds

TC_OK = 0
TC_ERR_INVALID_ARG = -1
TC_ERR_HIP = -2
TC_ERR_NO_DEVICE = -3
TC_ERR_HOST = -4

JOB_OK = 0
JOB_NOT_ENOUGH_SHARES = 1
JOB_DUPLICATE_ENTRY = 2
JOB_INVALID_ENCODING = 3

_u8p = ctypes.c_void_p      # data pointers are passed as raw addresses (host or device)
_u64p = ctypes.c_void_p
_sz = ctypes.c_size_t
_ctx = ctypes.c_void_p

# name -> argument types (after the leading tc_ctx*); every function returns int
PROTOTYPES = {
    "tc_hash_g2_batch": [_u8p, _u64p, _sz, _u8p],
    "tc_hash_g1_g2_batch": [_u8p, _u8p, _u64p, _sz, _u8p, _u8p],
    "tc_g2_mul_batch": [_u8p, _u8p, _sz, _sz, _u8p, _u8p],
    "tc_g1_mul_batch": [_u8p, _u8p, _sz, _sz, _u8p, _u8p],
    "tc_sign_shares_g2_batch": [_u8p, _sz, _u64p, _u8p, _sz, _sz, _u8p, _u8p],
    "tc_sign_batch": [_u8p, _u8p, _u64p, _sz, _sz, _u8p, _u8p],
    "tc_combine_g2_batch": [_sz, _sz, _u64p, _u8p, _sz, _u8p, _u8p],
    "tc_combine_g1_batch": [_sz, _sz, _u64p, _u8p, _sz, _u8p, _u8p],
    "tc_g1_lincomb_batch": [_sz, _u8p, _u8p, _sz, _u8p, _u8p],
    "tc_g2_lincomb_batch": [_sz, _u8p, _u8p, _sz, _u8p, _u8p],
    "tc_decrypt_batch": [_sz, _sz, _u64p, _u8p, _u8p, _u64p, _sz, _u8p, _u8p],
    "tc_combine_g2_fr_batch": [_sz, _sz, _u8p, _u8p, _sz, _u8p, _u8p],
    "tc_combine_g1_fr_batch": [_sz, _sz, _u8p, _u8p, _sz, _u8p, _u8p],
    "tc_decrypt_fr_batch": [_sz, _sz, _u8p, _u8p, _u8p, _u64p, _sz, _u8p, _u8p],
    "tc_combine_signatures_wire_batch": [_sz, _sz, _u64p, _u8p, _sz, _u8p, _u8p],
    "tc_decrypt_wire_batch": [_sz, _sz, _u64p, _u8p, _u8p, _u64p, _sz, _u8p, _u8p],
    "tc_xor_with_hash_batch": [_u8p, _u8p, _u64p, _sz, _u8p, _u8p],
    "tc_pairing_check_batch": [_u8p, _sz, _u8p, _sz, _u8p, _sz, _u8p, _sz, _sz, _u8p],
    "tc_verify_g2_batch": [_u8p, _sz, _u8p, _u8p, _sz, _u8p],
    "tc_verify_sig_batch": [_u8p, _sz, _u8p, _u8p, _u64p, _sz, _u8p],
    "tc_verify_shares_rlc_batch": [_u8p, _sz, _u8p, _u8p, _u64p, _sz, ctypes.c_char_p, _u8p, ctypes.POINTER(ctypes.c_uint64)],
    "tc_verify_g2_rlc_batch": [_u8p, _u8p, _u8p, _sz, _sz, ctypes.c_char_p, _u8p, ctypes.POINTER(ctypes.c_uint64)],
    "tc_verify_sig_rlc_batch": [_u8p, _u8p, _u8p, _u64p, _sz, _sz, ctypes.c_char_p, _u8p, ctypes.POINTER(ctypes.c_uint64)],
    "tc_verify_decryption_shares_rlc_batch": [_u8p, _sz, _u8p, _u8p, _u8p, _u64p, _u8p, _sz, ctypes.c_char_p, _u8p,
                                              ctypes.POINTER(ctypes.c_uint64)],
    "tc_ciphertext_verify_batch": [_u8p, _u8p, _u64p, _u8p, _sz, _u8p],
    "tc_decrypt_share_batch": [_u8p, _u8p, _u8p, _u64p, _u8p, _sz, _u8p, _u8p],
    "tc_secret_key_decrypt_batch": [_u8p, _u8p, _u8p, _u64p, _u8p, _sz, _u8p, _u8p],
    "tc_verify_decryption_share_batch": [_u8p, _sz, _u8p, _u8p, _u8p, _u64p, _u8p, _sz, _u8p],
    "tc_encrypt_batch": [_u8p, _sz, _u8p, _u8p, _u64p, _sz, _u8p, _u8p, _u8p, _u8p],
    "tc_public_key_share_batch": [_u8p, _sz, _u64p, _sz, _u8p, _u8p],
    "tc_g1_commitment_batch": [_u8p, _sz, _u8p, _u8p],
    "tc_bivar_commitment_row_batch": [_u8p, _sz, _u64p, _sz, _u8p, _u8p],
    "tc_fr_interpolate_batch": [_sz, _u8p, _u8p, _sz, _u8p, _u8p],
    "tc_g1_subgroup_check_batch": [_u8p, _sz, _u8p],
    "tc_g2_subgroup_check_batch": [_u8p, _sz, _u8p],
    "tc_g1_compress_batch": [_u8p, _sz, _u8p, _u8p],
    "tc_g2_compress_batch": [_u8p, _sz, _u8p, _u8p],
    "tc_g1_decompress_batch": [_u8p, _sz, _u8p, _u8p],
    "tc_g2_decompress_batch": [_u8p, _sz, _u8p, _u8p],
}

CONTEXT_SYMBOLS = ["tc_ctx_create", "tc_ctx_destroy", "tc_ctx_set_device_io", "tc_ctx_set_stream", "tc_sync",
                   "tc_last_error", "tc_ctx_set_timing", "tc_last_kernel_ms", "tc_version", "tc_ctx_set_input_checks",
                   "tc_ctx_get_input_checks", "tc_ctx_transfer_bytes", "tc_ctx_trim", "tc_ctx_get_device_io", "tc_ctx_get_tuning"]

# the multi-GPU surface (tc_group_*): name -> (restype, argtypes); the group handle is an opaque pointer
_grp = ctypes.c_void_p
_szp = ctypes.POINTER(ctypes.c_size_t)
_u64out = ctypes.POINTER(ctypes.c_uint64)
GROUP_PROTOTYPES = {
    "tc_group_create": (ctypes.c_int, [ctypes.POINTER(_grp), ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "tc_group_destroy": (None, [_grp]),
    "tc_group_size": (ctypes.c_int, [_grp]),
    "tc_group_uses_rccl": (ctypes.c_int, [_grp]),
    "tc_group_ctx": (ctypes.c_void_p, [_grp, ctypes.c_int]),
    "tc_group_last_error": (ctypes.c_char_p, [_grp]),
    "tc_group_shard": (ctypes.c_int, [_grp, _sz, ctypes.c_int, _szp, _szp]),
    "tc_group_transfer_bytes": (ctypes.c_int, [_grp, _u64out, _u64out]),
    "tc_group_set_keyset": (ctypes.c_int, [_grp, _sz, _u8p]),
    "tc_group_get_keyset": (ctypes.c_int, [_grp, ctypes.c_int, _u8p]),
    "tc_group_combine_signatures": (ctypes.c_int, [_grp, _sz, _u64p, _u8p, _sz, _u8p, _u8p]),
    "tc_group_verify_g2": (ctypes.c_int, [_grp, _u8p, _u8p, _sz, _u8p, _u64out]),
    "tc_group_sign_combine_verify": (ctypes.c_int, [_grp, _u8p, _sz, _u64p, _sz, _u8p, _u64p, _sz, _u8p, _u8p, _u64out]),
}

ALL_SYMBOLS = CONTEXT_SYMBOLS + sorted(PROTOTYPES) + sorted(GROUP_PROTOTYPES)

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Loads libtc_amd.so; raises NativeLibraryMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "%s not found: build it with `python -m threshold_crypto_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # libtc_amd.so needs libamdhip64.so.7.  PyTorch-ROCm bundles its own copy (same SONAME); two
    # HIP/HSA runtimes in one process cannot both own the GPU, so when torch is installed let it
    # load its runtime first and bind to that one.  Torch is not otherwise used here.
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: the system ROCm runtime is used
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.tc_ctx_create.argtypes = [ctypes.POINTER(_ctx), ctypes.c_int]
    lib.tc_ctx_create.restype = ctypes.c_int
    lib.tc_ctx_destroy.argtypes = [_ctx]
    lib.tc_ctx_destroy.restype = None
    lib.tc_ctx_set_device_io.argtypes = [_ctx, ctypes.c_int]
    lib.tc_ctx_set_device_io.restype = ctypes.c_int
    lib.tc_ctx_set_stream.argtypes = [_ctx, ctypes.c_void_p]
    lib.tc_ctx_set_stream.restype = ctypes.c_int
    lib.tc_ctx_set_timing.argtypes = [_ctx, ctypes.c_int]
    lib.tc_ctx_set_timing.restype = ctypes.c_int
    lib.tc_ctx_set_input_checks.argtypes = [_ctx, ctypes.c_int]
    lib.tc_ctx_set_input_checks.restype = ctypes.c_int
    lib.tc_ctx_trim.argtypes = [_ctx]
    lib.tc_ctx_trim.restype = ctypes.c_int
    lib.tc_ctx_get_input_checks.argtypes = [_ctx]
    lib.tc_ctx_get_input_checks.restype = ctypes.c_int
    lib.tc_ctx_get_device_io.argtypes = [_ctx]
    lib.tc_ctx_get_device_io.restype = ctypes.c_int
    lib.tc_ctx_get_tuning.argtypes = [_ctx, ctypes.POINTER(ctypes.c_uint64)]
    lib.tc_ctx_get_tuning.restype = ctypes.c_int
    lib.tc_ctx_transfer_bytes.argtypes = [_ctx, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.tc_ctx_transfer_bytes.restype = ctypes.c_int
    lib.tc_last_kernel_ms.argtypes = [_ctx]
    lib.tc_last_kernel_ms.restype = ctypes.c_double
    lib.tc_sync.argtypes = [_ctx]
    lib.tc_sync.restype = ctypes.c_int
    lib.tc_last_error.argtypes = [_ctx]
    lib.tc_last_error.restype = ctypes.c_char_p
    lib.tc_version.argtypes = []
    lib.tc_version.restype = ctypes.c_char_p
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = [_ctx] + args
        fn.restype = ctypes.c_int
    for name, (res, args) in GROUP_PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib
