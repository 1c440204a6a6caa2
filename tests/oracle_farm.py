"""Runs the CPU oracle (oracle/lbl_oracle.c, TEST INFRASTRUCTURE) for whole BASELINE-size
spectra on a pool of worker processes, so that the GPU parity tests can compare every point
of a 5 M / 10 M-point spectrum instead of a few windows.

The workers are `spawn`ed (they never touch the GPU: they only load oracle/liblbl_oracle.so)
and rebuild the deterministic synthetic line tables from a recipe instead of receiving them
through a pipe.  Jobs are queued heaviest first; a job is one (table, level, grid, pedestal
flag) call of the oracle, i.e. one call of the reference's absorption().
"""
import functools
import multiprocessing
import os

import numpy as np


@functools.lru_cache(maxsize=None)
def table_from_recipe(recipe):
    """recipe: ("uniform", formula, v_lo, v_hi) -> synthetic.line_table(formula, v_lo, v_hi);
    ("banded", formula, v_lo, v_hi, num_lines, bands, seed) -> synthetic.banded_line_table;
    ("hitran", v_lo, v_hi, num_lines, seed) -> tests.hitran_shapes.hitran_shaped_table."""
    from pylbl_amd import synthetic
    kind = recipe[0]
    if kind == "uniform":
        _, formula, v_lo, v_hi = recipe
        return synthetic.line_table(formula, v_lo, v_hi)
    if kind == "banded":
        _, formula, v_lo, v_hi, num_lines, bands, seed = recipe
        return synthetic.banded_line_table(formula, v_lo, v_hi, num_lines=num_lines,
                                           bands=bands, seed=seed)
    if kind == "hitran":
        from tests.hitran_shapes import hitran_shaped_table
        _, v_lo, v_hi, num_lines, seed = recipe
        return hitran_shaped_table(np.random.default_rng(seed), v_lo, v_hi, num_lines)
    raise ValueError(f"unknown table recipe {recipe!r}")


def run_job(spec):
    """Worker: (recipe, T, P, x, v0, vn, n_per_v, cut_off, remove_pedestal) -> (k, evals)."""
    from oracle import oracle
    recipe, t, p, x, v0, vn, npv, cut, ped = spec
    table = table_from_recipe(recipe)
    k, extras = oracle.absorption_port(table, t, p, x, v0, vn, npv, cut_off=cut,
                                       remove_pedestal=ped)
    return k, extras["evals"]


def estimated_cost(spec):
    recipe, _, _, _, v0, vn, npv, cut, ped = spec
    table = table_from_recipe(recipe)
    lines = np.count_nonzero((table.nu >= v0 - cut - 1) & (table.nu <= vn + cut + 1))
    return lines*((2*cut + 1)*npv + 1)*(1.3 if ped else 1.)


class OracleFarm(object):
    """submit() any number of jobs, start() them (heaviest first), fetch with result()."""
    def __init__(self, workers=None):
        if workers is None:
            workers = max(1, min(16, len(os.sched_getaffinity(0))))
        self.workers = workers
        self.specs = {}
        self.pending = {}
        self.pool = None

    def submit(self, key, recipe, t, p, x, v0, vn, npv, ped, cut=25):
        if key in self.specs:
            raise KeyError(f"duplicate oracle job {key!r}")
        self.specs[key] = (tuple(recipe), float(t), float(p), float(x), int(v0), int(vn),
                           int(npv), int(cut), bool(ped))

    def start(self):
        context = multiprocessing.get_context("spawn")
        self.pool = context.Pool(self.workers)
        order = sorted(self.specs, key=lambda key: -estimated_cost(self.specs[key]))
        for key in order:
            self.pending[key] = self.pool.apply_async(run_job, (self.specs[key],))

    def result(self, key, timeout=900.):
        """(k, evals) of job `key`; the result is dropped from the farm once fetched twice is
        not needed, so callers keep what they want to reuse."""
        return self.pending[key].get(timeout=timeout)

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None
