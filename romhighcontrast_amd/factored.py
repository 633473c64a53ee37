"""Snapshot blocks in factored form (host side: argument marshalling; the algebra is libromhc's, csrc/rom_factored.hip).

A snapshot row of libromhc is a fixed linear image of its system's *interface vector* (reduced unknowns,
cross-point values, coefficient blocks and the scalars 1/(a_p+a_q), h^2/a_b; ``Fem.reduced_stride`` doubles:
784 against 65 025 at 256x256 / 2x2):

    U = Y B^T ,     B (dim x K) parameter independent  (``Fem.expansion_is_linear``)

so everything the basis stage needs from a snapshot block can be formed from ``Y`` -- in fact from its ``compact``
coordinates (``Fem.compact_stride``: the entries the expansion reads, what the ranks of a sharded sweep exchange):
H^1_0 norms, the strong greedy in both modes and the POD are single C calls on the (M, Kc) block
(``rom_h10norm_factored``, ``rom_greedy_factored``, ``rom_pod_factored``); the geometry of the FE space in those
coordinates is built once per space and cached by the library (``rom_fem_energy_map``).  The (M, dim) block itself is
never needed; rows are materialised on demand (``rows``).  The reference has no counterpart (its snapshots are plain
NumPy rows, src/lib/SolutionsManagers.py:64-68); the builders are its own (src/lib/ReducedBasis.py:112-139, :189-200).
"""
from __future__ import annotations

import numpy as np

from . import _ffi


class ExpansionMap:
    """The expansion of one FE space as a linear map of compact interface vectors."""

    def __init__(self, sm):
        fem, ctx = sm._fem, sm._ctx
        if not fem.expansion_is_linear:
            raise _ffi.RomLibraryError("this geometry recovers some edges node by node: its expansion is not a "
                                       "linear map of the interface vectors (use snapshot rows)")
        self.sm, self.fem, self.ctx = sm, fem, ctx
        self.K, self.dim = fem.reduced_stride, fem.dim
        self._ones = {}
        self.inputs = fem.reduced_inputs
        self.Kc = fem.compact_stride
        self._S = None

    def build(self, parts=7):
        """rom_fem_energy_map: the geometry of the snapshots in compact coordinates (1: H^1_0, 2: Galerkin forms,
        4: Euclidean), built once per FE space and cached by the library.  Returns the ranks (k_h10, k_l2)."""
        return self.fem.energy_map(parts)

    @property
    def S(self):
        """B^T B on the device (Kc x Kc): only ``FactoredSnapshots.gram`` uses it."""
        if self._S is None:
            Bt = self.ctx.alloc(self.Kc * self.dim)
            self.expand_compact(self.ctx.upload(np.eye(self.Kc)), self.Kc, Bt)
            self._S = self.ctx.alloc(self.Kc * self.Kc)
            self.ctx.gram(self.Kc, self.dim, Bt, 0, self.dim, self._S, 0, self.Kc)
            self.ctx.synchronize()
        return self._S

    def compact(self, Y, M):
        """(M, Kc) input coordinates of the interface vectors Y (M, K)."""
        Yc = self.ctx.alloc(max(M * self.Kc, 1))
        if M:
            self.fem.pack_reduced(Y, M, Yc)
        return Yc

    def expand_compact(self, Wc, n, U, row0=0):
        """U[row0:row0+n] = expansion of n vectors given in input coordinates (n, Kc)."""
        W = self.ctx.alloc(max(n * self.K, 1))
        if n:
            self.fem.unpack_reduced(Wc, n, W)
        self.expand_into(W, n, U, row0=row0)

    def _a_dummy(self, M):
        if M not in self._ones:
            self._ones = {M: self.ctx.upload(np.ones((M, self.fem.kblk)))}
        return self._ones[M]

    def expand_into(self, Y, M, U, y_row0=0, row0=0):
        """U[row0:row0+M] = Y[y_row0:y_row0+M] B^T (the parameters are not read by a linear expansion)."""
        self.fem.expand(self._a_dummy(M), M, Y, U, y_row0=y_row0, row0=row0)
        self.ctx.solve_status()


def expansion_map(sm) -> ExpansionMap:
    em = getattr(sm, "_expansion_map", None)
    if em is None:
        em = sm._expansion_map = ExpansionMap(sm)
    return em


class FactoredSnapshots:
    """M snapshots held as their interface vectors ``Y`` (device, (M, K) row-major)."""

    def __init__(self, sm, Y: "_ffi.Buffer", M: int):
        self.sm, self.Y, self.M = sm, Y, int(M)
        self.map = expansion_map(sm)

    @property
    def K(self):
        return self.map.K

    @property
    def Yc(self):
        """(M, Kc) input coordinates of the interface vectors (what the norms / POD / greedy calls take)."""
        if getattr(self, "_Yc", None) is None:
            self._Yc = self.map.compact(self.Y, self.M)
        return self._Yc

    def rows(self, lo=0, hi=None):
        """Materialise snapshot rows [lo, hi) as a DeviceArray."""
        from .lib.SolutionsManagers import DeviceArray
        hi = self.M if hi is None else hi
        n = max(hi - lo, 0)
        U = self.map.ctx.alloc(max(n * self.map.dim, 1))
        if n:
            self.map.expand_into(self.Y, n, U, y_row0=lo)
        return DeviceArray(U, n, self.map.dim)

    def take(self, idx) -> "FactoredSnapshots":
        """The sub-block of the snapshots ``idx`` (a gather of interface vectors)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        Y = self.map.ctx.alloc(max(idx.size * self.K, 1))
        if idx.size:
            Y.gather_rows_from(self.Y, idx, self.K)
        return FactoredSnapshots(self.sm, Y, idx.size)

    def gram(self):
        """U U^T as a device buffer (M x M), from Y alone."""
        ctx, K, M = self.map.ctx, self.map.Kc, self.M
        T = ctx.alloc(M * K)
        ctx.gemm_nt(M, K, K, self.Yc, 0, K, self.map.S, 0, K, T, 0, K)  # T = Y S (S symmetric)
        G = ctx.alloc(M * M)
        ctx.gemm_nt(M, M, K, T, 0, K, self.Yc, 0, K, G, 0, M)
        return G


def pod_modes_factored(fs: FactoredSnapshots, n: int, center=True):
    """Leading ``n`` POD modes / singular values of the snapshot block ``fs`` without ever forming it: ONE C call
    (``rom_pod_factored``).  With U = Y B^T and B^T B = E E^T the rows of Z = Yc E (M x k', k' <= a few hundred / thousand)
    have the Euclidean geometry of the snapshot rows, so the POD of the block IS the POD of Z -- ``rom_pod`` on a matrix
    dim / k' times narrower -- and a mode is expanded like any interface vector.  Returns (modes (n, dim) NumPy,
    singular values); rows follow scikit-learn's ``svd_flip(u_based_decision=False)`` sign convention (the PCA call at
    src/lib/ReducedBasis.py:196)."""
    em, M = fs.map, fs.M
    ctx, dim = em.ctx, em.dim
    n = min(n, M, dim)
    V = ctx.alloc(max(n * dim, 1))
    sig, info = em.fem.pod_factored(fs.Yc, M, n, V, center=center)
    pod_modes_factored.last_info = info
    if info["completed_modes"]:
        from .lib.ReducedBasis import warn_completed_modes
        warn_completed_modes(info, n, 0.0)
    if n == 0:
        return np.zeros((0, dim)), sig
    return V.download(n * dim, shape=(n, dim)), sig


def h10norm_factored(fs: FactoredSnapshots) -> np.ndarray:
    """H^1_0 norms of the snapshots from their interface vectors (= ``sm.H10norm(fs.rows())``): rom_h10norm_factored."""
    return fs.map.fem.h10norm_factored(fs.Yc, fs.M)


def greedy_factored(fs: FactoredSnapshots, a2train, n: int, galerkin: bool, h1norm):
    """The strong greedy of ReducedBasisGreedy.build (src/lib/ReducedBasis.py:105-139) on a training block held in
    factored form: ONE C call (``rom_greedy_factored``).  Returns (picks, max relative errors per iteration)."""
    em, M = fs.map, fs.M
    a_dev = None
    if galerkin:
        a_dev = em.ctx.upload(np.ascontiguousarray(np.asarray(a2train, dtype=np.float64).reshape(M, -1)))
    return em.fem.greedy_factored(fs.Yc, M, a_dev, h1norm, galerkin, n)
