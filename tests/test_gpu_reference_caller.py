"""SURVEY 8b row b-1 on a GPU: the reference's call sequence (tests/reference_caller.py:
pyLBL/spectroscopy.py:53-69,163-205 replayed) drives ``pylbl_amd.Gas``, the MT-CKD classes and
``CrossSection`` through an object that has only what ``pyLBL.database.Database`` has, and every
mechanism slot is compared with its oracle -- the lines slot also with the reference's own
compiled C reading the file the reference's ORM wrote (tests/golden/refdb.db)."""
import numpy as np
import pytest

import pylbl_amd
from pylbl_amd import arts_crossfit, mt_ckd, synthetic
from pylbl_amd.database import Database
from tests import golden_io, reference_caller as ref

pytestmark = pytest.mark.gpu

GRID = np.arange(580., 720., 0.01)
# (p, t, vmr): two levels inside the fixture's 150-350 K partition sums
P = np.asarray([98388., 11419.])
T = np.asarray([288.99, 220.37])
GASES = {"H2O": np.asarray([6.637074e-3, 4.0e-6]), "CO2": np.asarray([3.6e-4, 3.6e-4]),
         "O2": np.asarray([0.208996, 0.208996]), "N2": np.asarray([0.78, 0.78]),
         "CFC11": np.asarray([2.3e-10, 2.0e-10])}


def lines_check(got, expect, n_per_v, remove_pedestal, label):
    if remove_pedestal:
        tol = golden_io.pedestal_tolerance(expect, n_per_v, 25, 1.e-6) + 1e-300
        assert np.max(np.abs(got - expect)/tol) <= 1., label
    else:
        np.testing.assert_allclose(got, expect, rtol=1.e-6, atol=0., err_msg=label)


@pytest.mark.parametrize("path", [None, "not-a-file"], ids=["by_path", "by_query_helpers"])
@pytest.mark.parametrize("remove_pedestal", [None, False], ids=["default", "no_pedestal"])
def test_reference_sequence_over_the_reference_database_object(
        tmp_path, oracle, continuum_oracle, path, remove_pedestal):
    from oracle import xsec_oracle
    bands = synthetic.cross_section_bands(seed=11)
    arts_crossfit.write_npz(tmp_path / "CFC11.npz", bands)
    db = ref.ReferenceDatabase(path=path, cross_sections={"CFC11": str(tmp_path / "CFC11.npz")})
    beta, cache = ref.replay_compute_absorption(
        (P, T, GASES), GRID, db, pylbl_amd.Gas, mt_ckd.CONTINUA, pylbl_amd.CrossSection,
        remove_pedestal=remove_pedestal)
    pedestal = True if remove_pedestal is None else remove_pedestal
    v0, vn, npv = synthetic.grid_arguments(GRID)
    file_db = Database(ref.ReferenceDatabase().path)
    assert set(beta) == {f"{x}_absorption" for x in GASES}
    for formula in GASES:
        values = beta[f"{formula}_absorption"]
        assert values.shape == (2, 3, GRID.size)
        for level in range(2):
            t, p, x = T[level], P[level], GASES[formula][level]
            n = ref.number_density(t, p, x)
            vmr = {g: GASES[g][level] for g in GASES}
            # slot 0: lines
            if formula in ("H2O", "CO2"):
                k, _ = oracle.absorption_port(file_db.line_table(formula), t, p, x, v0, vn, npv,
                                              remove_pedestal=pedestal)
                lines_check(values[level, 0], n*k[:GRID.size], npv, pedestal,
                            f"{formula} level {level} vs the C restatement")
                if oracle.have_reference():
                    rc, k = oracle.absorption_reference(file_db.path, formula, t, p, x, v0, vn,
                                                        npv, remove_pedestal=pedestal)
                    assert rc == 0
                    lines_check(values[level, 0], n*k[:GRID.size], npv, pedestal,
                                f"{formula} level {level} vs the reference's C on its own file")
                assert values[level, 0].any()
            else:
                assert not values[level, 0].any()       # no TIPS / no transitions: zeros
            # slot 1: continua (H2O has two, spectroscopy.py:58-61)
            owners = {"H2O": ["H2OForeign", "H2OSelf"], "CFC11": []}.get(formula, [formula])
            expect = np.zeros(GRID.size)
            for owner in owners:
                expect = expect + continuum_oracle.continuum(owner).spectra(t, p, vmr, GRID)
            scale = np.max(np.abs(expect))
            assert np.all(np.abs(values[level, 1] - expect) <= 1e-6*np.abs(expect) + 1e-12*scale)
            # slot 2: cross-section, CFC11 only
            if formula == "CFC11":
                expect = n*xsec_oracle.absorption_coefficient(bands, GRID, t, p)
                scale = np.max(np.abs(expect))
                assert scale > 0.
                assert np.all(np.abs(values[level, 2] - expect)
                              <= 1e-6*np.abs(expect) + 1e-12*scale)
            else:
                assert not values[level, 2].any()
    assert cache["CFC11"].gas is not None and cache["CFC11"].gas.molecule is None
    assert cache["CFC11"].gas_continua is None and cache["H2O"].cross_section is None
    if path is not None:
        assert ("gas", "CO2") in db.calls
    else:
        assert not [c for c in db.calls if c[0] in ("gas", "tips")]


def test_own_spectroscopy_over_the_reference_database_object(tmp_path, oracle):
    """pylbl_amd.Spectroscopy(atmosphere, grid, <reference's Database object>): same values as
    the replayed per-level loop, in the batched form."""
    db = ref.ReferenceDatabase(path="not-a-file")
    atmosphere = synthetic.Atmos(p=P, t=T, vmr={k: GASES[k] for k in ("H2O", "CO2", "O2", "N2")})
    spec = pylbl_amd.Spectroscopy(atmosphere, GRID, db, farfield=False)
    assert spec.list_molecules() == db.molecules()
    out = spec.compute_absorption("all")
    replayed, _ = ref.replay_compute_absorption(
        (P, T, atmosphere.vmr), GRID, db, pylbl_amd.Gas, mt_ckd.CONTINUA, pylbl_amd.CrossSection)
    for name, expect in replayed.items():
        got = np.asarray(out[name])
        scale = np.max(np.abs(expect), axis=-1, keepdims=True)
        assert np.all(np.abs(got - expect) <= 1e-9*np.abs(expect) + 1e-13*scale), name
