"""The step before and after the hot path in the reference's experiment driver:
parameter sampling, error definitions and the statistics record of ``experiment()``
(reference: src/experiments/HighContrast.py:59-64, 67-82, 99-115, 176-214).  Host orchestration only --
every numerical call goes through the HIP-backed ``SolutionsManagerFEM`` / reduced-basis classes.
Plotting, the results-folder layout and the multiprocessing wrapper of the paper run are out of scope.
"""
from __future__ import annotations

from collections import namedtuple
from time import time
from typing import Callable

import numpy as np

from .lib.ReducedBasis import INFINIT_A
from .lib.SolutionsManagers import SolutionsManager, SolutionsManagerFEM

MachinePrecision = 1e-13  # (HighContrast.py:19)

TypeOfProblems = namedtuple(
    "TypeOfProblems",
    "forward_modeling projection state_estimation parameter_estimation_inverse parameter_estimation_linear")  # (:28-29)


def get_full_a(a_per_block, sm, high_contrast_blocks):
    """(:59-64) expand per-group coefficients to per-block ones; blocks in no group stay at 1.
    ``sm`` is a solutions manager or just its ``blocks_geometry`` tuple."""
    a = np.ones(((len(a_per_block),) + tuple(getattr(sm, "blocks_geometry", sm))))
    for a_vec, members in zip(np.asarray(a_per_block).T, high_contrast_blocks):
        for p, q in members:
            a[:, p, q] = a_vec
    return a


def calculate_time(func: Callable, verbose=True):
    """(:67-78) wall-clock wrapper returning (seconds, result)."""
    def timed(**kwargs):
        if verbose:
            print(f"calculating {func.__name__}")
        t0 = time()
        res = func(**kwargs)
        t = time() - t0
        if verbose:
            print(f"time spent: {t}")
        return t, res
    return timed


def calculate_relative_error(sm: SolutionsManager, solutions, approximate_solutions):
    """(:81-82)."""
    return sm.H10norm(approximate_solutions - solutions) / sm.H10norm(solutions)


def get_data(experiment_path):
    """(:93-96) load the experiment cache ``<experiment_path>/data.compressed`` (joblib) or start empty."""
    import os
    import joblib
    data_path = f"{experiment_path}/data.compressed"
    data = joblib.load(data_path) if os.path.exists(data_path) else dict()
    return data, data_path


def save_data(data, data_path):
    """The reference's only persistence format (:150,170,214): one joblib dump of the ``data`` dict.
    Basis objects pickle as plain NumPy arrays + parameters (device handles are rebuilt on load)."""
    import joblib
    joblib.dump(data, data_path)


def get_a2test_and_train(blocks_geometry, high_contrast_blocks, mesh_discretization_per_dim, diff_coef_refinement,
                         max_num_samples_offline, seed, num_cores=1, method="lsq"):
    """(:99-115) tensor grid in 1/a per coefficient group (uniform in 1/a between 1/INFINIT_A and 1),
    a seeded sub-sample of it, preceded by all INFINIT_A / 1 corner combinations."""
    sm = SolutionsManagerFEM(blocks_geometry, N=mesh_discretization_per_dim, num_cores=num_cores, method=method)
    a, a_high_contrast = sample_parameters(blocks_geometry, high_contrast_blocks, diff_coef_refinement,
                                           max_num_samples_offline, seed)
    return sm, a, a_high_contrast


def sample_parameters(blocks_geometry, high_contrast_blocks, diff_coef_refinement, max_num_samples_offline, seed):
    """The sampling half of ``get_a2test_and_train`` (no device needed): (a, a_high_contrast)."""
    ngroups = len(high_contrast_blocks)
    per_dim = min(diff_coef_refinement * int(np.log2(INFINIT_A)),
                  int(np.ceil(max_num_samples_offline ** (1 / ngroups))))
    axis = 1 / np.linspace(1 / INFINIT_A, 1, num=per_dim, endpoint=False)
    a_high_contrast = np.transpose([np.ravel(g) for g in np.meshgrid(*[axis] * ngroups)])
    np.random.seed(seed)
    a_inf = np.transpose([np.ravel(g) for g in np.meshgrid(*[[INFINIT_A, 1]] * ngroups)])
    if len(a_high_contrast) > max_num_samples_offline - len(a_inf):
        keep = np.random.choice(len(a_high_contrast), size=max(0, max_num_samples_offline - len(a_inf)), replace=False)
        a_high_contrast = a_high_contrast[keep]
    a_high_contrast = np.vstack((a_inf, a_high_contrast))
    return get_full_a(a_high_contrast, tuple(blocks_geometry), high_contrast_blocks), a_high_contrast


def experiment_statistics(sm, a, reduced_basis_builders, vn_max_dim=20, num_measurements=50, vn_max_dim2do_stats=None,
                          verbose=False, data=None, data_path=None):
    """Snapshots, bases and the per-dimension error / time records of ``experiment()`` (:144-214),
    returned as the same ``data`` dictionary (keys ``solutions``, ``solutions_H1norm``,
    ``time2calculate_*`` and, per builder name, ``basis`` / ``time2build`` / ``errors`` / ``times``).
    Like the reference it draws the measurement points from NumPy's global RNG right after the sweep,
    so calling it after ``get_a2test_and_train(..., seed)`` reproduces the reference's points.
    """
    vn_max_dim2do_stats = vn_max_dim if vn_max_dim2do_stats is None else vn_max_dim2do_stats
    data = {} if data is None else data
    if "solutions" not in data:
        data["time2calculate_solutions"], data["solutions"] = calculate_time(sm.generate_solutions, verbose)(a2try=a)
        data["time2calculate_h1norm"], data["solutions_H1norm"] = calculate_time(sm.H10norm, verbose)(
            solutions=data["solutions"])
    U, h1 = data["solutions"], data["solutions_H1norm"]
    measurement_points = np.random.uniform(size=(num_measurements, 2))  # (:155)
    measurements = sm.evaluate_solutions(measurement_points, U)
    data["measurement_points"] = measurement_points
    for builder in reduced_basis_builders:
        if builder.name not in data:
            data[builder.name] = {"errors": {}, "times": {}}
            data[builder.name]["time2build"], data[builder.name]["basis"] = calculate_time(builder.build, verbose)(
                n=vn_max_dim, sm=sm, solutions2train=U, a2train=a, optim_method="lsq", solutions2train_h1norm=h1)
    for n in range(1, vn_max_dim + 1):
        for builder in reduced_basis_builders:
            rec = data[builder.name]
            if n > vn_max_dim2do_stats or n in rec["errors"]:
                continue
            rb = rec["basis"][:n]
            se_time, (c, se_approx) = calculate_time(rb.state_estimation, verbose)(
                sm=sm, measurement_points=measurement_points, measurements=measurements, return_coefs=True)
            inv_time, _ = calculate_time(rb.parameter_estimation_inverse, verbose)(c=c)
            lin_time, _ = calculate_time(rb.parameter_estimation_linear, verbose)(c=c)
            rb.orthonormalize()  # (:189)
            fm_time, fm_approx = calculate_time(rb.forward_modeling, verbose)(sm=sm, a=a)
            pj_time, pj_approx = calculate_time(rb.projection, verbose)(sm=sm, true_solutions=U)
            rec["errors"][n] = TypeOfProblems(
                forward_modeling=sm.H10norm(fm_approx - U) / h1,
                projection=sm.H10norm(pj_approx - U) / h1,
                state_estimation=sm.H10norm(se_approx - U) / h1,
                parameter_estimation_inverse=np.abs(1 - np.array(rb.parameter_estimation_inverse(c)) / a),
                parameter_estimation_linear=np.abs(1 - np.array(rb.parameter_estimation_linear(c)) / a))
            rec["times"][n] = TypeOfProblems(fm_time, pj_time, se_time, inv_time, lin_time)
    if data_path is not None:
        save_data(data, data_path)
    return data
