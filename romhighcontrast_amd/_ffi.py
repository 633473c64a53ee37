"""ctypes binding of libromhc.so (include/romhc.h).  Thin: prototypes, error mapping, handles.

There is no CPU fallback: if the shared library is missing or no MI355X is visible the import of
the product modules fails loudly (RomLibraryError).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np
from scipy.linalg import LinAlgError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libromhc.so")

ROM_OK, ROM_ERR_INVALID, ROM_ERR_HIP, ROM_ERR_NOT_SPD, ROM_ERR_COMM, ROM_ERR_NOMEM = range(6)


class RomLibraryError(RuntimeError):
    pass


_lib = None

_c_double_p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/romhc.h declares
PROTOTYPES = {
    "rom_last_error": (C.c_char_p, []),
    "rom_version": (C.c_int, []),
    "rom_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rom_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rom_shutdown": (C.c_int, [_vp]),
    "rom_synchronize": (C.c_int, [_vp]),
    "rom_set_workspace_limit": (C.c_int, [_vp, C.c_size_t]),
    "rom_device_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "rom_timer_start": (C.c_int, [_vp]),
    "rom_timer_stop": (C.c_int, [_vp, _c_double_p]),
    "rom_profile_enable": (C.c_int, [_vp, C.c_int]),
    "rom_profile_reset": (C.c_int, [_vp]),
    "rom_profile_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rom_profile_query": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_size_t, _c_double_p, C.POINTER(C.c_long),
                                    _c_double_p, _c_double_p]),
    "rom_buf_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rom_buf_free": (C.c_int, [_vp]),
    "rom_buf_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "rom_buf_upload": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t]),
    "rom_buf_download": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t]),
    "rom_buf_fill": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_double]),
    "rom_buf_copy": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t]),
    "rom_buf_equal": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]),
    "rom_buf_gather_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_size_t]),
    "rom_fem_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rom_fem_destroy": (C.c_int, [_vp]),
    "rom_fem_dims": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                               C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rom_fem_load_vector_host": (C.c_int, [_vp, _vp]),
    "rom_assemble_batch": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "rom_solve_batch": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64]),
    "rom_solve_batch_async": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64]),
    "rom_solve_status": (C.c_int, [_vp]),
    "rom_fem_reduced_stride": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rom_fem_expansion_is_linear": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rom_fem_reduced_layout": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rom_fem_compact_stride": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rom_fem_pack_reduced_async": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "rom_fem_unpack_reduced_async": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "rom_comm_allgather_packed_async": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_size_t, C.c_int]),
    "rom_solve_reduced_async": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64]),
    "rom_expand_batch_async": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64, _vp, C.c_int64]),
    "rom_solve_work": (C.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "rom_stencil_apply": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "rom_h10norm": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, _vp]),
    "rom_l2norm": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp]),
    "rom_gemm_nt": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_double, _vp, C.c_size_t, C.c_int64,
                              _vp, C.c_size_t, C.c_int64, C.c_double, _vp, C.c_size_t, C.c_int64]),
    "rom_gram": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, C.c_size_t, C.c_int64, _vp, C.c_size_t, C.c_int64]),
    "rom_gemm_nn": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_double, _vp, C.c_size_t, C.c_int64,
                              _vp, C.c_size_t, C.c_int64, C.c_double, _vp, C.c_size_t, C.c_int64]),
    "rom_reduced_solve_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "rom_buf_scale": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_double]),
    "rom_host_alloc": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_double))]),
    "rom_host_free": (C.c_int, [C.POINTER(C.c_double)]),
    "rom_center_rows": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp]),
    "rom_rows_scale": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp]),
    "rom_rows_sign_flip": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64]),
    "rom_evaluate_points": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "rom_project_h10": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "rom_galerkin_rom": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64, C.c_int, _vp, C.c_int64]),
    "rom_orthonormalize_rows": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp, C.c_int64]),
    "rom_greedy": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rom_pod": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "rom_fem_energy_map": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rom_h10norm_factored": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp]),
    "rom_greedy_factored": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rom_pod_factored": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "rom_pod_ex": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, _vp, C.c_int64, _vp, _vp]),
    "rom_symmetric_orthonormalize": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64]),
    "rom_complete_orthonormal": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int64]),
    "rom_small_eig_host": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp]),
    "rom_comm_unique_id": (C.c_int, [C.c_char_p, C.c_size_t]),
    "rom_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.c_int, C.c_int]),
    "rom_comm_destroy": (C.c_int, [_vp]),
    "rom_comm_allgather": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t]),
    "rom_comm_allgather_async": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_int]),
    "rom_comm_wait_slot": (C.c_int, [_vp, C.c_int]),
    "rom_comm_wait": (C.c_int, [_vp, C.c_int]),
    "rom_comm_allreduce_host": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
}


def load_library(path: str = LIB_PATH):
    """dlopen libromhc.so and attach prototypes.  No GPU is touched here."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RomLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C romhighcontrast_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().rom_last_error().decode("utf-8", "replace")


def check(status: int):
    """Map a C status to the exception types the reference raises."""
    if status == ROM_OK:
        return
    msg = last_error()
    if status == ROM_ERR_NOT_SPD:
        raise LinAlgError(msg)  # scipy.linalg.solve(assume_a='pos') raises this in the reference
    if status == ROM_ERR_NOMEM:
        raise MemoryError(msg)
    raise RomLibraryError(f"libromhc error {status}: {msg}")


# downloads of at least this many bytes land in page-locked host memory (0 disables: ROMHC_PINNED_DOWNLOAD_BYTES)
PINNED_DOWNLOAD_BYTES = int(os.environ.get("ROMHC_PINNED_DOWNLOAD_BYTES", str(8 << 20))) or (1 << 62)


_download_sizes: dict = {}  # size -> number of large downloads of that size so far


def _pinned_array(lib, n: int):
    """A writable float64 array of n entries in page-locked memory from libromhc's pool; the block returns to the pool
    when the array (and every view of it) is gone.  Pinning new memory costs ten copies' worth of time, so that only
    happens for a size that keeps coming back (third request on); before, a block is used if the pool has one.
    None: the caller falls back to np.empty."""
    seen = _download_sizes[n] = _download_sizes.get(n, 0) + 1
    p = C.POINTER(C.c_double)()
    if lib.rom_host_alloc(n, 0 if seen >= 3 else 1, C.byref(p)) != 0 or not p:
        return None
    addr = C.cast(p, C.c_void_p).value
    raw = (C.c_double * n).from_address(addr)
    weakref.finalize(raw, lib.rom_host_free, C.cast(addr, C.POINTER(C.c_double)))
    return np.frombuffer(raw, dtype=np.float64)


def _host(arr: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(arr, dtype=np.float64)


class Context:
    """One GPU + stream.  Process-wide singleton per device (``get_context``)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        n = C.c_int(0)
        try:
            check(self.lib.rom_device_count(C.byref(n)))
        except RomLibraryError as e:
            raise RomLibraryError(f"no usable HIP device: {e}. libromhc needs an MI355X (gfx950).") from None
        if n.value < 1:
            raise RomLibraryError("no HIP device visible; libromhc needs an MI355X (gfx950). No CPU fallback.")
        h = _vp()
        check(self.lib.rom_init(device, C.byref(h)))
        self.h = h
        self.has_comm = False
        self.device = device

    # -- buffers ---------------------------------------------------------------------------
    def alloc(self, n: int) -> "Buffer":
        return Buffer(self, int(n))

    def upload(self, arr) -> "Buffer":
        arr = _host(arr)
        b = Buffer(self, arr.size)
        b.upload(arr)
        return b

    def synchronize(self):
        check(self.lib.rom_synchronize(self.h))

    def solve_status(self):
        """Wait for the sweeps enqueued with wait=False; raises (scipy LinAlgError) if any pivot was not positive."""
        check(self.lib.rom_solve_status(self.h))

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(self.lib.rom_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_workspace_limit(self, nbytes: int):
        check(self.lib.rom_set_workspace_limit(self.h, int(nbytes)))

    # -- timing ----------------------------------------------------------------------------
    def timer_start(self):
        check(self.lib.rom_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_double(0)
        check(self.lib.rom_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, on: bool):
        check(self.lib.rom_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.rom_profile_reset(self.h))

    def profile_report(self):
        n = C.c_int(0)
        check(self.lib.rom_profile_count(self.h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            ms, fl, by = C.c_double(0), C.c_double(0), C.c_double(0)
            cnt = C.c_long(0)
            check(self.lib.rom_profile_query(self.h, i, name, 128, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = dict(total_ms=ms.value, launches=cnt.value, flops=fl.value, bytes=by.value)
        return out

    # -- dense ops ---------------------------------------------------------------------------
    def gemm_nt(self, m, n, k, A, a_off, lda, B, b_off, ldb, Cb, c_off, ldc, alpha=1.0, beta=0.0):
        check(self.lib.rom_gemm_nt(self.h, m, n, k, alpha, A.h, a_off, lda, B.h, b_off, ldb, beta, Cb.h, c_off, ldc))

    def gram(self, m, k, A, a_off, lda, Cb, c_off, ldc):
        check(self.lib.rom_gram(self.h, m, k, A.h, a_off, lda, Cb.h, c_off, ldc))

    def gemm_nn(self, m, n, k, A, a_off, lda, B, b_off, ldb, Cb, c_off, ldc, alpha=1.0, beta=0.0):
        check(self.lib.rom_gemm_nn(self.h, m, n, k, alpha, A.h, a_off, lda, B.h, b_off, ldb, beta, Cb.h, c_off, ldc))

    def reduced_solve_batch(self, n, kb, M, Ahat, w, rhs, rhs_per_system, c_out):
        check(self.lib.rom_reduced_solve_batch(self.h, n, kb, M, Ahat.h, w.h, rhs.h, 1 if rhs_per_system else 0,
                                               c_out.h))

    def center_rows(self, X: "Buffer", M, dim, mean: "Buffer", row0=0):
        check(self.lib.rom_center_rows(self.h, X.h, row0, M, dim, mean.h))

    def rows_scale(self, X: "Buffer", rows, dim, factors, row0=0):
        f = _host(np.asarray(factors, dtype=np.float64))
        check(self.lib.rom_rows_scale(self.h, X.h, row0, rows, dim, f.ctypes.data))

    def rows_sign_flip(self, X: "Buffer", rows, dim, row0=0):
        check(self.lib.rom_rows_sign_flip(self.h, X.h, row0, rows, dim))

    def l2norm(self, U: "Buffer", row0, K, dim) -> np.ndarray:
        out = np.empty(K)
        check(self.lib.rom_l2norm(self.h, U.h, row0, K, dim, out.ctypes.data))
        return out

    # -- basis stage (single C calls) ------------------------------------------------------------
    def orthonormalize_rows(self, X: "Buffer", n, dim, Q: "Buffer", x_row0=0, q_row0=0):
        check(self.lib.rom_orthonormalize_rows(self.h, X.h, x_row0, n, dim, Q.h, q_row0))

    def pod(self, X: "Buffer", M, dim, n, V: "Buffer", center=True, x_row0=0, v_row0=0, rel_floor=0.0):
        """rom_pod / rom_pod_ex: X is overwritten.  Returns (sigma (n,), info dict)."""
        sigma, info = np.zeros(max(n, 1)), np.zeros(8)
        check(self.lib.rom_pod_ex(self.h, X.h, x_row0, M, dim, n, 1 if center else 0, float(rel_floor), V.h, v_row0,
                                  sigma.ctypes.data, info.ctypes.data))
        keys = ("resolved_modes", "completed_modes", "gram_passes", "sketch_passes")
        d = {k: int(info[i]) for i, k in enumerate(keys)}
        d.update(executed_flops=float(info[4]), useful_flops=float(info[5]), subspace_iterations=int(info[6]),
                 stop_reason=("filled", "floor", "budget")[int(info[7])])
        return sigma[:n], d

    def symmetric_orthonormalize(self, V: "Buffer", n, dim, v_row0=0):
        check(self.lib.rom_symmetric_orthonormalize(self.h, V.h, v_row0, n, dim))

    def complete_orthonormal(self, V: "Buffer", found, rest, dim, v_row0=0):
        check(self.lib.rom_complete_orthonormal(self.h, V.h, v_row0, found, rest, dim))

    def small_eig(self, A, mode=0, rel_tol=0.0, gram_like=True):
        A = _host(A)
        n = A.shape[0]
        lam, T = np.empty(n), np.empty((n, n))
        check(self.lib.rom_small_eig_host(self.h, n, A.ctypes.data, mode, rel_tol, 1 if gram_like else 0, lam.ctypes.data,
                                          T.ctypes.data))
        return lam, T

    # -- RCCL ----------------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        check(self.lib.rom_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, nranks: int):
        check(self.lib.rom_comm_init(self.h, uid, len(uid), rank, nranks))
        self.has_comm = True

    def comm_destroy(self):
        self.has_comm = False
        check(self.lib.rom_comm_destroy(self.h))

    def allgather(self, send: "Buffer", send_off, recv: "Buffer", recv_off, count):
        check(self.lib.rom_comm_allgather(self.h, send.h, send_off, recv.h, recv_off, count))

    def allgather_async(self, send: "Buffer", send_off, recv: "Buffer", recv_off, count, slot=0):
        check(self.lib.rom_comm_allgather_async(self.h, send.h, send_off, recv.h, recv_off, count, slot))

    def comm_wait_slot(self, slot):
        check(self.lib.rom_comm_wait_slot(self.h, slot))

    def comm_wait(self, host_sync=True):
        check(self.lib.rom_comm_wait(self.h, 1 if host_sync else 0))

    def allreduce_host(self, vals, op="sum") -> np.ndarray:
        v = _host(np.atleast_1d(vals)).copy()
        check(self.lib.rom_comm_allreduce_host(self.h, v.ctypes.data, v.size, 1 if op == "max" else 0))
        return v


D2H_BYTES = [0]  # bytes copied from device buffers to the host through Buffer.download (tests assert on it)


class Buffer:
    """fp64 device buffer owned by the library; freed on garbage collection."""

    def __init__(self, ctx: Context, n: int):
        self.ctx = ctx
        self.n = int(n)
        h = _vp()
        check(ctx.lib.rom_buf_alloc(ctx.h, self.n, C.byref(h)))
        self.h = h

    def upload(self, arr, offset=0):
        arr = _host(arr)
        check(self.ctx.lib.rom_buf_upload(self.h, offset, arr.ctypes.data, arr.size))
        return self

    def download(self, n=None, offset=0, shape=None) -> np.ndarray:
        n = self.n - offset if n is None else int(n)
        D2H_BYTES[0] += 8 * n
        out = _pinned_array(self.ctx.lib, n) if n * 8 >= PINNED_DOWNLOAD_BYTES else None
        if out is None:
            out = np.empty(n, dtype=np.float64)
        check(self.ctx.lib.rom_buf_download(self.h, offset, out.ctypes.data, n))
        return out.reshape(shape) if shape is not None else out

    def fill(self, value=0.0, offset=0, n=None):
        check(self.ctx.lib.rom_buf_fill(self.h, offset, self.n - offset if n is None else n, float(value)))
        return self

    def copy_from(self, src: "Buffer", n, dst_off=0, src_off=0):
        check(self.ctx.lib.rom_buf_copy(self.h, dst_off, src.h, src_off, n))
        return self

    def same_bits_as(self, other: "Buffer", n, off=0, other_off=0) -> bool:
        eq = C.c_int(0)
        check(self.ctx.lib.rom_buf_equal(self.h, off, other.h, other_off, n, C.byref(eq)))
        return bool(eq.value)

    def gather_rows_from(self, src: "Buffer", rows, dim):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(self.ctx.lib.rom_buf_gather_rows(self.h, src.h, rows.ctypes.data, rows.size, dim))
        return self

    def scale(self, alpha, offset=0, n=None):
        check(self.ctx.lib.rom_buf_scale(self.h, offset, self.n - offset if n is None else n, float(alpha)))
        return self

    def free(self):
        if getattr(self, "h", None) is not None and self.h:
            self.ctx.lib.rom_buf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_contexts = {}


def get_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("ROMHC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


class Fem:
    """Handle of one FE space (blocks_geometry, N) on the device."""

    def __init__(self, ctx: Context, nrb: int, ncb: int, N: int):
        self.ctx = ctx
        h = _vp()
        check(ctx.lib.rom_fem_create(ctx.h, nrb, ncb, N, C.byref(h)))
        self.h = h
        nr, nc, ng, nt = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        dim = C.c_int64(0)
        check(ctx.lib.rom_fem_dims(h, C.byref(nr), C.byref(nc), C.byref(dim), C.byref(ng), C.byref(nt)))
        self.nr, self.nc, self.dim, self.n_interface, self.n_tiles = nr.value, nc.value, dim.value, ng.value, nt.value
        self.nrb, self.ncb, self.N, self.kblk = nrb, ncb, N, nrb * ncb

    def load_vector(self) -> np.ndarray:
        B = np.empty(self.dim)
        check(self.ctx.lib.rom_fem_load_vector_host(self.h, B.ctypes.data))
        return B

    def solve_batch(self, a: Buffer, M: int, U: Buffer, row0: int = 0, wait: bool = True):
        """wait=False only enqueues the sweep; call Context.solve_status() before trusting the rows."""
        fn = self.ctx.lib.rom_solve_batch if wait else self.ctx.lib.rom_solve_batch_async
        check(fn(self.h, a.h, M, U.h, row0))

    @property
    def reduced_stride(self) -> int:
        """doubles per system of the interface vector (what solve_reduced writes and expand reads)"""
        v = C.c_int64(0)
        check(self.ctx.lib.rom_fem_reduced_stride(self.h, C.byref(v)))
        return v.value

    @property
    def expansion_is_linear(self) -> bool:
        """True if expand() is a linear map of the interface vectors (U = Y B^T, `a` ignored)."""
        v = C.c_int(0)
        check(self.ctx.lib.rom_fem_expansion_is_linear(self.h, C.byref(v)))
        return bool(v.value)

    @property
    def reduced_inputs(self) -> np.ndarray:
        """positions of an interface vector that the expansion reads (the nodal edge blocks are its outputs)"""
        b, e = C.c_int64(0), C.c_int64(0)
        check(self.ctx.lib.rom_fem_reduced_layout(self.h, C.byref(b), C.byref(e)))
        return np.concatenate((np.arange(0, b.value), np.arange(e.value, self.reduced_stride)))

    @property
    def compact_stride(self) -> int:
        """doubles per system of the COMPACT interface vector (the entries the expansion reads: what travels between ranks)"""
        v = C.c_int64(0)
        check(self.ctx.lib.rom_fem_compact_stride(self.h, C.byref(v)))
        return v.value

    def pack_reduced(self, Y: Buffer, M: int, Yc: Buffer, y_row0: int = 0, c_row0: int = 0):
        """Yc[c_row0:c_row0+M] = the compact form of the interface vectors Y[y_row0:y_row0+M] (enqueued only)."""
        check(self.ctx.lib.rom_fem_pack_reduced_async(self.h, Y.h, y_row0, M, Yc.h, c_row0))

    def unpack_reduced(self, Yc: Buffer, M: int, Y: Buffer, c_row0: int = 0, y_row0: int = 0):
        """The inverse of pack_reduced: full-stride vectors with a zero nodal part (expand() fills it) (enqueued only)."""
        check(self.ctx.lib.rom_fem_unpack_reduced_async(self.h, Yc.h, c_row0, M, Y.h, y_row0))

    def allgather_packed_async(self, Y: Buffer, M: int, send: Buffer, recv: Buffer, recv_off: int = 0, slot: int = 0,
                               y_row0: int = 0):
        """Pack the M interface vectors Y[y_row0:] into `send` and all-gather them into `recv`, both on the
        communication stream (the compute stream is not blocked); slots as Context.allgather_async."""
        check(self.ctx.lib.rom_comm_allgather_packed_async(self.h, Y.h, y_row0, M, send.h, recv.h, recv_off, slot))

    def solve_reduced(self, a: Buffer, M: int, Y: Buffer, y_row0: int = 0):
        """Stage 1 of the sweep (enqueued only): interface vectors of the M systems into Y[y_row0:y_row0+M]."""
        check(self.ctx.lib.rom_solve_reduced_async(self.h, a.h, M, Y.h, y_row0))

    def expand(self, a: Buffer, M: int, Y: Buffer, U: Buffer, y_row0: int = 0, row0: int = 0):
        """Stage 2 (enqueued only): snapshot rows U[row0:row0+M] from the interface vectors Y[y_row0:y_row0+M]."""
        check(self.ctx.lib.rom_expand_batch_async(self.h, a.h, M, Y.h, y_row0, U.h, row0))

    def solve_work(self):
        v = [C.c_double(0) for _ in range(4)]
        check(self.ctx.lib.rom_solve_work(self.h, *[C.byref(x) for x in v]))
        return dict(flops_own=v[0].value, bytes_own=v[1].value, flops_banded=v[2].value, bytes_banded=v[3].value)

    def assemble_batch(self, a: Buffer, M: int, diag: Buffer, east: Buffer, north: Buffer):
        check(self.ctx.lib.rom_assemble_batch(self.h, a.h, M, diag.h, east.h, north.h))

    def stencil_apply(self, X: Buffer, K: int, Y: Buffer, a_one=None, x_row0=0, y_row0=0):
        if a_one is None:
            check(self.ctx.lib.rom_stencil_apply(self.h, None, 1, X.h, x_row0, K, Y.h, y_row0))
        else:
            a_one = _host(a_one)
            assert a_one.size == self.kblk
            check(self.ctx.lib.rom_stencil_apply(self.h, a_one.ctypes.data, 0, X.h, x_row0, K, Y.h, y_row0))

    def h10norm(self, U: Buffer, K: int, u_row0=0, V: Buffer | None = None, v_row0=0) -> np.ndarray:
        out = np.empty(K)
        check(self.ctx.lib.rom_h10norm(self.h, U.h, u_row0, V.h if V is not None else None, v_row0, K,
                                       out.ctypes.data))
        return out

    def project_h10(self, U: Buffer, M: int, Cb: Buffer | None, n: int, OUT: Buffer, u_row0=0, c_row0=0, out_row0=0):
        check(self.ctx.lib.rom_project_h10(self.h, U.h, u_row0, M, Cb.h if Cb is not None else None, c_row0, n, OUT.h,
                                           out_row0))

    def galerkin_rom(self, a: Buffer, M: int, Cb: Buffer | None, n: int, OUT: Buffer, c_row0=0, out_row0=0):
        check(self.ctx.lib.rom_galerkin_rom(self.h, a.h, M, Cb.h if Cb is not None else None, c_row0, n, OUT.h, out_row0))

    def greedy(self, U: Buffer, M: int, a: Buffer | None, h1norm, galerkin: bool, n: int, u_row0=0):
        """rom_greedy: (picks list, max relative errors list)."""
        h1 = _host(np.broadcast_to(np.asarray(h1norm, dtype=np.float64), (M,)))
        picks, errs = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1))
        check(self.ctx.lib.rom_greedy(self.h, U.h, u_row0, M, a.h if a is not None else None, h1.ctypes.data,
                                      1 if galerkin else 0, n, picks.ctypes.data, errs.ctypes.data))
        return [int(p) for p in picks[:n]], [float(e) for e in errs[:n]]

    # -- basis stage on compact interface vectors (factored snapshot blocks: rom_factored.hip) --------------
    def energy_map(self, parts=7):
        """rom_fem_energy_map: build / cache the geometry of the snapshots in interface-vector coordinates (1: H^1_0,
        2: Galerkin forms, 4: Euclidean).  Returns the ranks (k_h10, k_l2)."""
        k1, k2 = C.c_int(0), C.c_int(0)
        check(self.ctx.lib.rom_fem_energy_map(self.h, int(parts), C.byref(k1), C.byref(k2)))
        return k1.value, k2.value

    def h10norm_factored(self, Yc: Buffer, M: int, c_row0=0) -> np.ndarray:
        out = np.empty(max(M, 1))
        check(self.ctx.lib.rom_h10norm_factored(self.h, Yc.h, c_row0, M, out.ctypes.data))
        return out[:M]

    def greedy_factored(self, Yc: Buffer, M: int, a: Buffer | None, h1norm, galerkin: bool, n: int, c_row0=0):
        """rom_greedy_factored: (picks list, max relative errors list)."""
        h1 = _host(np.broadcast_to(np.asarray(h1norm, dtype=np.float64), (M,)))
        picks, errs = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1))
        check(self.ctx.lib.rom_greedy_factored(self.h, Yc.h, c_row0, M, a.h if a is not None else None, h1.ctypes.data,
                                               1 if galerkin else 0, n, picks.ctypes.data, errs.ctypes.data))
        return [int(p) for p in picks[:n]], [float(e) for e in errs[:n]]

    def pod_factored(self, Yc: Buffer, M: int, n: int, V: Buffer, center=True, c_row0=0, v_row0=0):
        """rom_pod_factored: modes as rows into V.  Returns (sigma (n,), info dict)."""
        sigma, info = np.zeros(max(n, 1)), np.zeros(8)
        check(self.ctx.lib.rom_pod_factored(self.h, Yc.h, c_row0, M, n, 1 if center else 0, V.h, v_row0, sigma.ctypes.data,
                                            info.ctypes.data))
        keys = ("resolved_modes", "completed_modes", "gram_passes", "sketch_passes")
        d = {k: int(info[i]) for i, k in enumerate(keys)}
        d.update(executed_flops=float(info[4]), subspace_iterations=int(info[6]), stop_reason=("filled", "floor", "budget")[int(info[7])])
        return sigma[:n], d

    def evaluate_points(self, U: Buffer, K: int, ix, iy, tx, ty, row0=0) -> np.ndarray:
        ix = np.ascontiguousarray(ix, dtype=np.int32)
        iy = np.ascontiguousarray(iy, dtype=np.int32)
        tx, ty = _host(tx), _host(ty)
        out = np.empty((K, ix.size))
        check(self.ctx.lib.rom_evaluate_points(self.h, U.h, row0, K, ix.size, ix.ctypes.data, iy.ctypes.data,
                                               tx.ctypes.data, ty.ctypes.data, out.ctypes.data))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.rom_fem_destroy(self.h)
                self.h = None
        except Exception:
            pass
