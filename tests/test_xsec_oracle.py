"""CPU checks of the cross-section path: the oracle against vectors produced by the
reference's own calculate_xsec_fullmodel, the coefficient-file reader, the database lookup."""
import numpy as np
import pytest

from pylbl_amd import arts_crossfit, synthetic
from pylbl_amd.database import Database, write_database
from pylbl_amd.errors import AliasNotFoundError, CrossSectionNotFoundError
from tests.conftest import ROOT

GOLDEN = ROOT / "tests" / "golden" / "xsec_model.npz"


def golden_cases():
    with np.load(GOLDEN) as archive:
        for case in range(int(archive["cases"])):
            coeffs = archive[f"set{int(archive[f'case{case}_set'])}_coeffs"]
            temperature, pressure = archive[f"case{case}_state"]
            yield case, coeffs, float(temperature), float(pressure), archive[f"case{case}_xsec"]


def test_oracle_is_bit_identical_to_the_reference_model():
    from oracle import xsec_oracle
    kinds = set()
    for case, coeffs, temperature, pressure, expect in golden_cases():
        got = xsec_oracle.full_model(temperature, pressure, coeffs.copy())
        assert np.array_equal(got, expect), case
        raw = xsec_oracle.fit(temperature, pressure, coeffs)
        kinds.add((bool((raw < 0).any()), bool(raw.sum() >= 0)))
    # no negatives / negatives with rescaling / negatives without rescaling all occur
    assert kinds == {(False, True), (True, True), (True, False)}


def test_reader_orientation_sorting_and_round_trip(tmp_path):
    bands = synthetic.cross_section_bands(seed=1)
    path = tmp_path / "CFC11.npz"
    arts_crossfit.write_npz(path, bands)
    for (f, c), (f2, c2) in zip(bands, arts_crossfit.read_bands(path)):
        assert np.array_equal(f, f2) and np.array_equal(c, c2)
    # Stored [nfreq, 4] and descending, as a file may be: same bands after reading.
    flipped = [(f[::-1], c[:, ::-1].T) for f, c in bands]
    arts_crossfit.write_npz(path, flipped)
    for (f, c), (f2, c2) in zip(bands, arts_crossfit.read_bands(path)):
        assert np.array_equal(f, f2) and np.array_equal(c, c2)
    with pytest.raises(ValueError):
        arts_crossfit._as_matrix(np.zeros((3, 10)), 10)


def test_database_lists_cross_section_files(tmp_path):
    tables = [synthetic.line_table("CO2", 1., 50., num_lines=5, seed=1),
              synthetic.line_table("N2O", 1., 50., num_lines=5, seed=2)]
    path = tmp_path / "lines.db"
    write_database(path, tables, cross_sections={"N2O": "/somewhere/N2O.nc"})
    db = Database(str(path))
    assert db.arts_crossfit("N2O") == "/somewhere/N2O.nc"
    with pytest.raises(CrossSectionNotFoundError):
        db.arts_crossfit("CO2")
    with pytest.raises(AliasNotFoundError):
        db.arts_crossfit("XYZ")


def test_oracle_interpolation_conventions():
    """Zero outside a band, bands added up, frequency = wavenumber x c x 100
    (cross_section.py:31-47)."""
    from oracle import xsec_oracle
    bands = synthetic.cross_section_bands(seed=2, ranges=((100., 110.), (105., 120.)), spacing=0.5)
    grid = np.asarray([99., 100.5, 107., 119.9, 121.])
    out = xsec_oracle.absorption_coefficient(bands, grid, 250., 5e4)
    assert out[0] == 0. and out[-1] == 0. and out[2] != 0.
    single = [xsec_oracle.absorption_coefficient([b], grid, 250., 5e4) for b in bands]
    np.testing.assert_allclose(out, single[0] + single[1], rtol=1e-15)


def test_hdf5_coefficient_file(tmp_path):
    """A netCDF-4/HDF5 file laid out as cross_section.py:29-40 reads it: integer `bands`,
    `band<m>_fgrid` [nfreq] and `band<m>_coeffs` stored [nfreq, 4]."""
    from tests import hdf5_writer
    try:
        from pylbl_amd import hdf5_reader
        hdf5_reader.library()
    except OSError as error:
        pytest.skip(str(error))
    bands = synthetic.cross_section_bands(seed=7, ranges=((10., 12.), (30., 31.)), spacing=0.1)
    path = tmp_path / "CFC12.nc"
    arrays = {"bands": np.asarray([1, 2], dtype=np.int32)}
    for m, (frequency, coefficients) in zip((1, 2), bands):
        arrays[f"band{m}_fgrid"] = frequency
        arrays[f"band{m}_coeffs"] = np.ascontiguousarray(coefficients.T)
    hdf5_writer.write(path, arrays)
    with hdf5_reader.File(path) as source:
        assert source.has("band1_fgrid") and not source.has("band3_fgrid")
        assert source.array("band2_coeffs").shape == (bands[1][0].size, 4)
        with pytest.raises(KeyError):
            source.array("missing")
    for (f, c), (f2, c2) in zip(bands, arts_crossfit.read_bands(path)):
        assert np.array_equal(f, f2) and np.array_equal(c, c2)
    with pytest.raises(OSError):
        hdf5_reader.File(tmp_path / "absent.nc")
