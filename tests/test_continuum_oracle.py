"""CPU checks of the MT-CKD continuum path: the numpy oracle against the reference's own
known answers, the coefficient fixture against the reference's data file (when it is there),
and the host-side table preparation of the product against the oracle's."""
import os

import numpy as np
import pytest

from pylbl_amd import mt_ckd, mt_ckd_data, synthetic

from tests.conftest import MT_CKD_TABLES

# tests/test_mt_ckd.py:15-26 of the reference: sum of each band's spectrum for the last level
# of the fixture atmosphere (tests/conftest.py:61-77), pressure handed over in Pa as that
# test does.
KNOWN_ANSWERS = {
    "CO2": [21.284607102488753],
    "H2OForeign": [131.87162317621952],
    "H2OSelf": [13.482864611247933],
    "N2": [0.7612890022253513, 0.5875825355004741, 0.00414557543788256],
    "O2": [0.24690308716508605, 0.11052072297118236, 0.03200556021322852, 0.04514938962400228,
           0.03897535512343981, 285.7607588975901, 4419601.794329887],
    "O3": [0.0006562127133778276, 1.7334221226752753, 0.05197265302394795],
}


def last_level():
    """Level -1 with the dictionary in the order of the reference's molecule_names fixture
    (tests/conftest.py:30-39), the order air_number_density adds the entries up in."""
    atmos = synthetic.fixture_atmosphere()
    vmr = {name: atmos.vmr[name][-1] for name in
           ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2")}
    return atmos.t[-1], atmos.p[-1], vmr


def test_oracle_reproduces_the_reference_known_answers(continuum_oracle):
    temperature, pressure, vmr = last_level()
    checked = 0
    for owner, answers in KNOWN_ANSWERS.items():
        spectra = continuum_oracle.continuum(owner).band_spectra(temperature, pressure, vmr)
        assert len(spectra) == len(answers)
        for spectrum, answer in zip(spectra, answers):
            # The reference asserts pytest.approx (1e-6); the restatement is ~1 ulp away.
            assert np.sum(spectrum) == pytest.approx(answer, rel=1e-13)
            checked += 1
    assert checked == 16


def test_fixture_atmosphere_is_the_reference_one():
    temperature, pressure, vmr = last_level()
    assert (temperature, pressure) == (288.99, 98388.)
    assert vmr["H2O"] == 6.637074e-03 and vmr["O2"] == 0.208996 and vmr["N2"] == 0.78
    assert vmr["CO"] == 1.482969e-07 and vmr["CH4"] == 1.700002e-06


def test_coefficient_fixture_matches_the_reference_file():
    source = "/root/reference/pyLBL/mt_ckd/mt-ckd.nc"
    if not os.path.isfile(source):
        pytest.skip("the reference tree is not mounted here")
    try:
        tables = mt_ckd_data.read_hdf5(source)
    except OSError as error:
        pytest.skip(str(error))
    fixture = mt_ckd_data.read_npz(str(MT_CKD_TABLES))
    assert set(fixture) == set(mt_ckd_data.VARIABLES) == set(tables)
    for name, table in tables.items():
        assert np.array_equal(fixture[name].data, table.data)
        assert (fixture[name].lower_bound, fixture[name].upper_bound, fixture[name].resolution) \
            == (table.lower_bound, table.upper_bound, table.resolution)


def test_host_table_preparation_matches_the_oracle(continuum_oracle):
    """Scale factors, spread sub-tables and analytic band shapes (the reference's
    constructors) are prepared by product code; same numbers as the restatement."""
    tables = mt_ckd_data.load(str(MT_CKD_TABLES))
    for owner, cls in mt_ckd.CONTINUA.items():
        expect = continuum_oracle.continuum(owner).bands
        assert len(cls.recipe) == len(expect)
        for prepare, (w, arrays, _) in zip(cls.recipe, expect):
            kind, lower, resolution, columns = prepare(tables)
            assert 0 <= kind < 16 and lower == w[0] and columns[0].size == w.size
            assert np.array_equal(lower + np.arange(w.size)*resolution, w)
            for column, reference in zip(columns, arrays.values()):
                np.testing.assert_allclose(column, reference, rtol=1e-14, atol=0.)


def test_oracle_interpolation_is_zero_outside_and_in_inverse_metres(continuum_oracle):
    temperature, pressure, vmr = last_level()
    co2 = continuum_oracle.continuum("CO2")
    w, _, _ = co2.bands[0]
    coarse = co2.band_spectra(temperature, pressure*0.01, vmr)[0]
    grid = np.asarray([w[0] - 1., w[0], w[7], 0.5*(w[7] + w[8]), w[-1], w[-1] + 1.])
    out = co2.spectra(temperature, pressure, vmr, grid)
    expect = [0., coarse[0], coarse[7], 0.5*(coarse[7] + coarse[8]), coarse[-1], 0.]
    np.testing.assert_allclose(out, 100.*np.asarray(expect), rtol=1e-14)


def test_table_search_order(tmp_path, monkeypatch):
    monkeypatch.setenv("PYLBL_MT_CKD", str(tmp_path / "absent.npz"))
    assert mt_ckd_data.default_path() == str(tmp_path / "absent.npz")
    monkeypatch.delenv("PYLBL_MT_CKD")
    with pytest.raises(FileNotFoundError):
        mt_ckd_data.default_path()
