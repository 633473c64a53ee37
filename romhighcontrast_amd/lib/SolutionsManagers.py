"""Host-side mirror of the reference's ``src/lib/SolutionsManagers.py`` on top of libromhc (HIP).

Same names, argument meaning, shapes and error behaviour as the reference
(``SolutionsManagerFEM``, ``galerkin``; file:line citations below are relative to the reference
tree), so notebooks and ``src/experiments`` import it unchanged as ``src.lib.SolutionsManagers`` or
``lib.SolutionsManagers``.  Every numerical method runs on the GPU through the C-ABI; there is no
CPU path in this module (a missing library / GPU raises ``RomLibraryError``).

Differences that cannot be avoided (SURVEY.md sections 0, 7):
  * the dense tensor ``A_preassembled[nrb,ncb,dim,dim]`` (:217-218) would be 135 GB at 256x256
    cells; it is exposed as a lazily-built property for small spaces only and never used here;
  * ``num_cores`` is accepted and ignored (the sweep is batched on the device);
  * arrays may also be passed / returned as ``DeviceArray`` handles to skip the PCIe round trip.
"""
from __future__ import annotations

from typing import List, Tuple, Union

import numpy as np

from .. import _ffi

__all__ = ["galerkin", "SolutionsManager", "SolutionsManagerFEM", "DeviceArray"]

_DENSE_LIMIT_BYTES = 2 << 30  # largest A_preassembled we are willing to materialise on the host


class DeviceArray:
    """(rows, dim) fp64 matrix resident in HBM.  ``.numpy()`` downloads it.

    ``factored``: when the rows are snapshots that ``generate_solutions_device`` has just produced, the same block in
    factored form (``romhighcontrast_amd.factored.FactoredSnapshots``: the interface vectors the rows are a fixed linear
    image of).  The basis builders take it when they find it -- the H^1_0 greedy and the PCA run on M x ~300 numbers
    instead of M x dim -- so whoever rewrites the rows in place (``pod_modes`` centres its input) must drop it."""

    def __init__(self, buf: _ffi.Buffer, rows: int, dim: int, factored=None):
        self.buf, self.rows, self.dim = buf, int(rows), int(dim)
        self.factored = factored

    @property
    def shape(self):
        return (self.rows, self.dim)

    def __len__(self):
        return self.rows

    def numpy(self) -> np.ndarray:
        if self.rows * self.dim == 0:
            return np.zeros((self.rows, self.dim))
        return self.buf.download(self.rows * self.dim, shape=(self.rows, self.dim))

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out if dtype is None else out.astype(dtype)


def _as_device(ctx: _ffi.Context, x, dim: int) -> DeviceArray:
    if isinstance(x, DeviceArray):
        return x
    arr = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).reshape(-1, dim)
    return DeviceArray(ctx.upload(arr), arr.shape[0], dim)


def _check_method(method: str):
    # the reference raises at solve time (:39); 'lsq' (LAPACK posv) and 'lsqsparse' (SuperLU) are
    # both exact direct solves of the same SPD system and map to the one GPU direct solver.
    if method.lower() not in ("lsq", "lsqsparse"):
        raise Exception(f"Method {method} Not implemented.")


def galerkin(a, B_total, A_preassembled, method="lsq"):
    """``galerkin`` (:17-40) for *small dense* tensors: ``(sum_pq a_pq A_pq) x = B``.

    Runs on the GPU as one reduced SPD solve (any n the dense tensor allows; the matrix is LDS resident up to
    n = 140).  The full-space solve does not go through a dense tensor in this build: use
    ``SolutionsManagerFEM.generate_solutions``.
    """
    _check_method(method)
    A_preassembled = np.asarray(A_preassembled, dtype=np.float64)
    n = A_preassembled.shape[-1]
    kb = int(np.prod(A_preassembled.shape[:-2]))
    ctx = _ffi.get_context()
    Ahat = ctx.upload(A_preassembled.reshape(kb, n, n))
    w = ctx.upload(np.asarray(a, dtype=np.float64).reshape(1, kb))
    rhs = ctx.upload(np.asarray(B_total, dtype=np.float64).reshape(n))
    out = ctx.alloc(n)
    ctx.reduced_solve_batch(n, kb, 1, Ahat, w, rhs, False, out)
    return out.download(n)


class SolutionsManager:
    """Common solver / projector surface (:43-142), device backed."""

    def __init__(self, fem: _ffi.Fem, B_total, blocks_geometry, num_cores=1, method="lsq"):
        self.method = method
        self._fem = fem
        self._ctx = fem.ctx
        self.vspace_dim = len(B_total)
        self.blocks_geometry = tuple(blocks_geometry)
        self.B_total = B_total
        self.mapfunction = map  # (:51) num_cores ignored: the sweep is one batched launch sequence

    def __str__(self):
        return self.__class__.__name__

    # -- pickling: device handles are rebuilt on load (basis objects / sm may be joblib-dumped) ----
    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_fem", None)
        d.pop("_ctx", None)
        d.pop("_host_blocks", None)
        return d

    # -- host arrays that generate_solutions returned: their interface vectors stay on the device --------------------
    def _remember_host_block(self, arr, fs):
        """``arr`` (what generate_solutions is about to return) is the image of the interface vectors ``fs``: remembered by
        identity (a weak reference; the two most recent blocks), so that a builder handed the same array back can work on
        the interface vectors -- after checking on the device that the rows still are that image, bit for bit."""
        import weakref
        blocks = [b for b in getattr(self, "_host_blocks", []) if b[0]() is not None][-1:]
        blocks.append((weakref.ref(arr), fs))
        self._host_blocks = blocks

    def factored_of_host_rows(self, arr, U: "DeviceArray"):
        """The interface vectors remembered for the host array ``arr`` (uploaded as ``U``) -- None unless ``arr`` IS an array
        generate_solutions returned and its rows are unchanged (the expansion of the interface vectors and the uploaded rows
        are compared bit for bit on the device: a caller may have written into the array)."""
        for ref, fs in getattr(self, "_host_blocks", []):
            if ref() is arr and fs.M == U.rows and U.dim == self.vspace_dim:
                Ue = fs.rows()
                same = Ue.buf.same_bits_as(U.buf, U.rows * U.dim)
                del Ue
                return fs if same else None
        return None

    # -- norms ----------------------------------------------------------------------------------------
    def H10norm(self, solutions: Union[List[np.ndarray], DeviceArray]):
        """``sqrt(u_k^T A_1 u_k)`` (:56-58), A_1 = unit-coefficient stencil (:49)."""
        U = _as_device(self._ctx, solutions, self.vspace_dim)
        return self._fem.h10norm(U.buf, U.rows)

    def H10norm_diff(self, approx, solutions):
        """H10 norm of ``approx - solutions`` without forming the difference (greedy, RB :129)."""
        A = _as_device(self._ctx, approx, self.vspace_dim)
        U = _as_device(self._ctx, solutions, self.vspace_dim)
        assert A.rows == U.rows
        return self._fem.h10norm(A.buf, A.rows, V=U.buf)

    @staticmethod
    def l2norm(solutions: Union[List[np.ndarray], DeviceArray]):
        """(:60-62)."""
        ctx = _ffi.get_context()
        if isinstance(solutions, DeviceArray):
            return ctx.l2norm(solutions.buf, 0, solutions.rows, solutions.dim)
        arr = np.ascontiguousarray(np.asarray(solutions, dtype=np.float64))
        arr = arr.reshape(arr.shape[0], -1)
        return ctx.l2norm(ctx.upload(arr), 0, arr.shape[0], arr.shape[1])

    # -- snapshot sweep ---------------------------------------------------------------------------------
    def _a_batch(self, a2try) -> np.ndarray:
        a = np.asarray(list(a2try) if not isinstance(a2try, np.ndarray) else a2try, dtype=np.float64)
        k = int(np.prod(self.blocks_geometry))
        return np.ascontiguousarray(a.reshape(-1, k))

    def generate_solutions_device(self, a2try, keep_interface_vectors=True) -> DeviceArray:
        """The sweep with the (M, dim) result left in HBM.  Where the expansion of the FE space is a linear map of the
        interface vectors (every geometry that compresses its edges: N >~ 16), the sweep runs as its two stages --
        parameters -> interface vectors -> rows, bit-identical rows (tests: test_two_stage_sweep_is_bit_identical) -- and the
        block remembers its interface vectors (``DeviceArray.factored``, M x 784 doubles at 2x2 / N = 128 beside M x 65 025),
        which is what lets ``ReducedBasisGreedy.build`` / ``ReducedBasisPCA.build`` on this block run on them."""
        _check_method(self.method)
        a = self._a_batch(a2try)
        M = a.shape[0]
        U = self._ctx.alloc(max(M * self.vspace_dim, 1))
        fs = None
        if M:
            a_dev = self._ctx.upload(a)
            if keep_interface_vectors and self._fem.expansion_is_linear:
                from ..factored import FactoredSnapshots
                Y = self._ctx.alloc(M * self._fem.reduced_stride)
                self._fem.solve_reduced(a_dev, M, Y)
                self._fem.expand(a_dev, M, Y, U)
                self._ctx.solve_status()
                fs = FactoredSnapshots(self, Y, M)
            else:
                self._fem.solve_batch(a_dev, M, U)
        return DeviceArray(U, M, self.vspace_dim, factored=fs)

    def generate_solutions(self, a2try):
        """``generate_solutions`` (:64-68): (M, dim) ndarray, row m = A(a_m)^-1 B_total."""
        Ud = self.generate_solutions_device(a2try)
        arr = Ud.numpy()
        if Ud.factored is not None:
            self._remember_host_block(arr, Ud.factored)
        return arr

    def generate_riesz(self, x, norm="h10"):
        """(:70-86) -- l2 branch only, as in the reference (h10 raises there too)."""
        if norm == "l2":
            return self.evaluate_solutions(points=x, solutions=np.eye(self.vspace_dim)).T
        elif norm.lower() == "h10":
            raise Exception("Not implemented.")
        else:
            raise Exception("Not implemented.")

    # -- reduced operators: one C call each (rom_galerkin_rom / rom_project_h10) -------------------------
    def generate_fm_solutions_device(self, a, coefficients_rom) -> DeviceArray:
        _check_method(self.method)
        a = self._a_batch(a)
        M, dim, ctx = a.shape[0], self.vspace_dim, self._ctx
        out = ctx.alloc(max(M * dim, 1))
        if M == 0:
            return DeviceArray(out, 0, dim)
        n = len(coefficients_rom)
        C = _as_device(ctx, coefficients_rom, dim) if n else None          # empty basis: zeros (:89-91)
        self._fem.galerkin_rom(ctx.upload(a), M, C.buf if n else None, C.rows if n else 0, out)
        return DeviceArray(out, M, dim)

    def generate_fm_solutions(self, a: Union[np.ndarray, List[np.ndarray]], coefficients_rom: List[np.ndarray]):
        """Galerkin reduced-order solutions (:88-106)."""
        return self.generate_fm_solutions_device(a, coefficients_rom).numpy()

    def project_solutions_device(self, solutions, coefficients_rom) -> DeviceArray:
        _check_method(self.method)
        ctx, dim = self._ctx, self.vspace_dim
        U = _as_device(ctx, solutions, dim)
        M = U.rows
        out = ctx.alloc(max(M * dim, 1))
        if M == 0:
            return DeviceArray(out, 0, dim)
        n = len(coefficients_rom)
        C = _as_device(ctx, coefficients_rom, dim) if n else None          # empty basis: zeros (:109-111)
        self._fem.project_h10(U.buf, M, C.buf if n else None, C.rows if n else 0, out)
        return DeviceArray(out, M, dim)

    def project_solutions(self, solutions: List[np.ndarray], coefficients_rom: List[np.ndarray]):
        """H^1_0-orthogonal projection onto span(coefficients_rom) (:108-139)."""
        return self.project_solutions_device(solutions, coefficients_rom).numpy()

    def evaluate_solutions(self, points: np.ndarray, solutions: List[np.ndarray]) -> np.ndarray:
        raise Exception("Not implemented.")  # (:141-142)


class SolutionsManagerFEM(SolutionsManager):
    """P1 FEM on the structured SW-NE triangulation of an nrb x ncb grid of unit blocks (:145-244)."""

    def __init__(self, blocks_geometry: Tuple[int, int], N: int, num_cores=1, method="lsq", device=None):
        nrb, ncb = blocks_geometry
        self.N = N
        self.x_domain = (-ncb / 2.0, ncb / 2.0)  # (:149)
        self.y_domain = (-nrb / 2.0, nrb / 2.0)  # (:150)
        self.nc_inner_vertices = ncb * self.N - 1  # (:153)
        self.nr_inner_vertices = nrb * self.N - 1  # (:154)
        self.nc_cells = ncb * self.N + 1  # (:156)
        self.nr_cells = nrb * self.N + 1  # (:157)
        self.points_c = np.linspace(*self.x_domain, self.nc_cells)  # (:168)
        self.points_r = np.linspace(*self.y_domain, self.nr_cells)  # (:169)
        self._device = device
        self.num_cores = num_cores
        ctx = _ffi.get_context(device)
        fem = _ffi.Fem(ctx, int(nrb), int(ncb), int(N))
        super().__init__(fem, fem.load_vector(), (int(nrb), int(ncb)), num_cores=num_cores, method=method)

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._ctx = _ffi.get_context(self._device)
        self._fem = _ffi.Fem(self._ctx, self.blocks_geometry[0], self.blocks_geometry[1], self.N)

    # -- the dense tensors of the reference, small spaces only ------------------------------------------
    def stencil_arrays(self, a):
        """diag[M,nr,nc], east[M,nr,nc-1], north[M,nr-1,nc] of ``einsum('pqij,pq->ij')`` (:19-23)."""
        a = self._a_batch(a)
        M, nr, nc = a.shape[0], self.nr_inner_vertices, self.nc_inner_vertices
        ctx = self._ctx
        d, e, n = ctx.alloc(M * nr * nc), ctx.alloc(max(M * nr * (nc - 1), 1)), ctx.alloc(max(M * (nr - 1) * nc, 1))
        self._fem.assemble_batch(ctx.upload(a), M, d, e, n)
        return (d.download(M * nr * nc, shape=(M, nr, nc)), e.download(M * nr * (nc - 1), shape=(M, nr, nc - 1)),
                n.download(M * (nr - 1) * nc, shape=(M, nr - 1, nc)))

    @property
    def A_preassembled(self):
        """Dense (nrb, ncb, dim, dim) tensor (:217-218), built lazily from the device stencil."""
        if "_A_pre" not in self.__dict__:
            nrb, ncb = self.blocks_geometry
            dim, nr, nc = self.vspace_dim, self.nr_inner_vertices, self.nc_inner_vertices
            if nrb * ncb * dim * dim * 8 > _DENSE_LIMIT_BYTES:
                raise MemoryError(f"A_preassembled would need {nrb * ncb * dim * dim * 8 / 2 ** 30:.1f} GiB; "
                                  "the GPU build keeps the operator in stencil form")
            d, e, n = self.stencil_arrays(np.eye(nrb * ncb).reshape(nrb * ncb, nrb, ncb))
            A = np.zeros((nrb * ncb, dim, dim))
            idx = np.arange(dim).reshape(nr, nc)
            for b in range(nrb * ncb):
                A[b, idx, idx] = d[b]
                A[b, idx[:, :-1], idx[:, 1:]] = e[b]
                A[b, idx[:, 1:], idx[:, :-1]] = e[b]
                A[b, idx[:-1, :], idx[1:, :]] = n[b]
                A[b, idx[1:, :], idx[:-1, :]] = n[b]
            self.__dict__["_A_pre"] = A.reshape(nrb, ncb, dim, dim)
        return self.__dict__["_A_pre"]

    @property
    def A_preassembled4h1_norm(self):
        """(:49)."""
        return np.einsum("abij->ij", self.A_preassembled)

    def evaluate_solutions(self, points: np.ndarray, solutions: List[np.ndarray]) -> np.ndarray:
        """P1 point evaluation (:221-244): (n_solutions, m) values at the m points (x, y)."""
        P = np.asarray(points, dtype=np.float64).reshape(-1, 2)
        U = _as_device(self._ctx, solutions, self.vspace_dim)
        ix = np.searchsorted(self.points_c, P[:, 0]) - 1  # (:235)
        iy = np.searchsorted(self.points_r, P[:, 1]) - 1  # (:236)
        tx = (P[:, 0] - self.points_c[ix]) / (self.points_c[ix + 1] - self.points_c[ix])  # (:237)
        ty = (P[:, 1] - self.points_r[iy]) / (self.points_r[iy + 1] - self.points_r[iy])  # (:238)
        return self._fem.evaluate_points(U.buf, U.rows, ix, iy, tx, ty)
