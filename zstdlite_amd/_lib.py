"""ctypes binding of libzl_hip.so (include/zl_hip.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback behind this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ZL_HIP_LIB names another build of the SAME library (an A/B variant made by `python -m zstdlite_amd.build --variant`,
# development only); whatever it names must export every symbol below or the import fails.
LIB_PATH = os.environ.get("ZL_HIP_LIB") or os.path.join(HERE, "libzl_hip.so")

_ERR_FLOOR = (1 << 64) - 120


class ZstdError(RuntimeError):
    """Raised where the reference glue calls Rf_error() (src/raw-file.c:81-83,190-192)."""


class FrameT(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("src_size", C.c_uint64), ("dst_off", C.c_uint64), ("dst_cap", C.c_uint64)]


_lib = None

# every symbol include/zl_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("zl_isError", C.c_uint, [C.c_size_t]),
    ("zl_getErrorName", C.c_char_p, [C.c_size_t]),
    ("zl_compressBound", C.c_size_t, [C.c_size_t]),
    ("zl_findFrameCompressedSize", C.c_size_t, [C.c_void_p, C.c_size_t]),
    ("zl_getFrameContentSize", C.c_ulonglong, [C.c_void_p, C.c_size_t]),
    ("zl_findDecompressedSize", C.c_ulonglong, [C.c_void_p, C.c_size_t]),
    ("zl_findAllFramesCompressedSize", C.c_size_t, [C.c_void_p, C.c_size_t]),
    ("zl_createCCtx", C.c_void_p, []),
    ("zl_freeCCtx", C.c_size_t, [C.c_void_p]),
    ("zl_CCtx_setParameter", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    ("zl_CCtx_getParameter", C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("zl_CCtx_loadDictionary", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("zl_compress2", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    ("zl_createDCtx", C.c_void_p, []),
    ("zl_freeDCtx", C.c_size_t, [C.c_void_p]),
    ("zl_DCtx_setParameter", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    ("zl_DCtx_getParameter", C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("zl_DCtx_loadDictionary", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("zl_decompressDCtx", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    ("zl_getDictID_fromDict", C.c_uint, [C.c_void_p, C.c_size_t]),
    ("zl_getDictID_fromFrame", C.c_uint, [C.c_void_p, C.c_size_t]),
    ("zl_compress_batch", C.c_size_t, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("zl_decompress_batch", C.c_size_t, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("zl_decompress_device_async", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameT), C.c_size_t, C.c_void_p]),
    ("zl_decompress_device_finish", C.c_size_t, [C.c_void_p, C.POINTER(C.c_size_t)]),
    ("zl_decompress_device_hint", C.c_size_t, [C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t]),
    ("zl_decompress_device_hint_host_copy", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("zl_compress_device_nframes", C.c_size_t, [C.c_size_t]),
    ("zl_compress_device_async", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("zl_compress_device_finish", C.c_size_t, [C.c_void_p, C.POINTER(C.c_size_t)]),
    ("zl_index_frames", C.c_size_t, [C.c_void_p, C.c_size_t, C.POINTER(FrameT), C.c_size_t]),
    ("zl_cctx_kernel_ms", C.c_size_t, [C.c_void_p, C.POINTER(C.c_double)]),
    ("zl_cctx_far_ms", C.c_size_t, [C.c_void_p, C.POINTER(C.c_double)]),
    ("zl_dctx_kernel_ms", C.c_size_t, [C.c_void_p, C.POINTER(C.c_double)]),
    ("zl_dctx_large_frames", C.c_size_t, [C.c_void_p]),
    ("zl_dctx_frame_runs", C.c_size_t, [C.c_void_p]),
    ("zl_cctx_last_slots", C.c_size_t, [C.c_void_p]),
    ("zl_dctx_last_slots", C.c_size_t, [C.c_void_p]),
    ("zl_compressStream2", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("zl_CCtx_setPledgedSrcSize", C.c_size_t, [C.c_void_p, C.c_ulonglong]),
    ("zl_CCtx_reset", C.c_size_t, [C.c_void_p, C.c_int]),
    ("zl_CStreamInSize", C.c_size_t, []),
    ("zl_CStreamOutSize", C.c_size_t, []),
    ("zl_decompressStream", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("zl_DCtx_reset", C.c_size_t, [C.c_void_p, C.c_int]),
    ("zl_DStreamInSize", C.c_size_t, []),
    ("zl_DStreamOutSize", C.c_size_t, []),
    ("zl_versionString", C.c_char_p, []),
    ("zl_deviceCount", C.c_int, []),
]


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in
    one process cannot both own the GPU (the second one sees no devices), so when torch is
    installed its runtime is loaded first and libzl_hip.so binds to it by soname -- whichever of
    the two modules the caller imports first.  Without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libzl_hip.so (once).  Raises ImportError with build instructions if absent."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -m zstdlite_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError == a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def is_error(code):
    return code > _ERR_FLOOR


def error_name(code):
    return lib().zl_getErrorName(code).decode()
