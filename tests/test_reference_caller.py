"""The back ends under the *reference's* caller, on the CPU: what
``pyLBL.spectroscopy.MoleculeCache`` hands them is a ``pyLBL.database.Database``, whose surface
is ``.path / .gas() / .tips() / .molecules() / .arts_crossfit()`` and whose errors are its own
classes.  tests/reference_caller.py holds the stand-ins; the values behind them were produced
by the reference's own database.py (tests/golden/make_refdb.py).  Compute calls need the GPU:
tests/test_gpu_reference_caller.py."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from pylbl_amd import database, errors
from pylbl_amd.database import Database, line_table_of
from tests import reference_caller as ref

LINE_FIELDS = ("nu", "sw", "gamma_air", "gamma_self", "n_air", "delta_air", "elower")


def same_table(a, b):
    for x in LINE_FIELDS + ("local_iso_id", "tips_temperature", "tips_data"):
        assert np.array_equal(getattr(a, x), getattr(b, x)), x
    assert np.array_equal(a.mass_by_slot(), b.mass_by_slot())
    assert a.formula == b.formula and a.num_lines == b.num_lines


def test_own_reader_returns_what_the_reference_returned():
    """The stdlib-sqlite reader on the file the reference's ORM wrote against the values the
    reference's own gas()/tips()/molecules()/arts_crossfit() returned for it."""
    standin = ref.ReferenceDatabase()
    db = Database(standin.path)
    assert db.molecules() == standin.molecules()
    for formula in ("H2O", "CO2"):
        got = db.gas(formula)
        assert got.formula == str(standin.recorded(formula, "formula"))
        assert got.mass == standin.recorded(formula, "mass").tolist()
        for x in LINE_FIELDS + ("local_iso_id",):
            assert np.array_equal(got.transitions[x], standin.recorded(formula, x)), x
        t, q = db.tips(formula)
        assert np.array_equal(t, standin.recorded(formula, "tips_temperature"))
        assert np.array_equal(q, standin.recorded(formula, "tips_data"))
        # TotalPartitionFunction as pyLBL/tips.py:26-39 evaluates it
        expect = standin.recorded(formula, "q_288p99")
        mine = [got.partition_function.total_partition_function(288.99, i + 1)
                for i in got.partition_function.isotopologue]
        assert np.array_equal(mine, expect)
    assert db.arts_crossfit("CFC11") == str(standin.recorded("CFC11", "arts_crossfit"))
    # the same condition for every molecule that lacks something (class names; the reference
    # checks masses before transitions before TIPS in gas(), database.py:362-367)
    raised = dict(zip(standin.recorded("raised", "keys").tolist(),
                      standin.recorded("raised", "values").tolist()))
    for key, name in raised.items():
        method, formula = key.split(":")
        with pytest.raises(BaseException) as caught:
            getattr(db, method)(formula)
        assert errors.kind(caught.value) is not None, key
        if method != "gas":
            assert errors.kind(caught.value) == name, key


def test_line_table_from_the_reference_object_by_path_and_by_query_helpers():
    direct = {f: Database(ref.ReferenceDatabase().path).line_table(f) for f in ("H2O", "CO2")}
    # rows of CO2 carry local id 0 (the tenth isotopologue) and are not all ascending
    assert (direct["CO2"].local_iso_id == 0).any() and (np.diff(direct["CO2"].nu) < 0).any()
    assert direct["CO2"].mass_by_slot()[9] == 49.001675
    by_path = ref.ReferenceDatabase()
    for formula, table in direct.items():
        same_table(line_table_of(by_path, formula), table)
    assert by_path.calls == []                         # read through .path alone
    helpers_only = ref.ReferenceDatabase(path="sqlite-in-another-process")
    for formula, table in direct.items():
        same_table(line_table_of(helpers_only, formula), table)
        same_table(line_table_of(helpers_only, formula.lower()), table)     # alias
    assert ("gas", "CO2") in helpers_only.calls and ("tips", "CO2") in helpers_only.calls
    with pytest.raises(TypeError):
        line_table_of(object(), "CO2")
    # The reference's own classes come through the query-helper route untouched ...
    with pytest.raises(ref.AliasNotFoundError):
        line_table_of(helpers_only, "HCl")
    with pytest.raises(ref.TipsDataNotFoundError):
        line_table_of(helpers_only, "O2")
    # ... in the order the C looks things up: no partition sums wins (absorption.c:50-64)
    with pytest.raises(ref.TipsDataNotFoundError):
        line_table_of(helpers_only, "CFC11")
    with pytest.raises(ref.TransitionsNotFoundError):
        line_table_of(helpers_only, "N2")
    # ... and this package's from the file route.
    with pytest.raises(errors.AliasNotFoundError):
        line_table_of(by_path, "HCl")


class RecordingEngine(object):
    """Engine stand-in: no GPU here; records what the back ends upload."""
    handle = 1

    def __init__(self):
        self.tables, self.bands = [], []

    def load(self, table):
        self.tables.append(table)
        return len(self.tables)

    def free(self, molecule):
        pass

    def load_xsec(self, bands):
        self.bands.append(bands)
        return len(self.bands)

    def free_xsec(self, handle):
        pass


@pytest.mark.parametrize("path", [None, "not-a-file"], ids=["by_path", "by_query_helpers"])
def test_molecule_cache_sequence_of_the_reference(monkeypatch, tmp_path, path):
    """pyLBL/spectroscopy.py:53-69 replayed literally -- positional constructors, the
    reference's exception classes in the handlers -- over the object the reference would pass."""
    import pylbl_amd
    from pylbl_amd import arts_crossfit, gas_optics
    engine = RecordingEngine()
    monkeypatch.setattr(gas_optics, "default_engine", lambda device=0: engine)
    monkeypatch.setattr(arts_crossfit, "default_engine", lambda device=0: engine)
    bands = [(np.linspace(2e13, 3e13, 50), np.ones((4, 50)))]
    arts_crossfit.write_npz(tmp_path / "CFC11.npz", bands)
    db = ref.ReferenceDatabase(path=path, cross_sections={"CFC11": str(tmp_path / "CFC11.npz")})
    no_continua = {}
    caches = {name: ref.replay_molecule_cache(name, db, pylbl_amd.Gas, no_continua,
                                              pylbl_amd.CrossSection)
              for name in ("H2O", "CO2", "O2", "N2", "CFC11", "HCl")}
    assert [t.formula for t in engine.tables] == ["H2O", "CO2"]
    assert engine.tables[1].num_lines == 64 and engine.tables[1].mass_by_slot()[9] == 49.001675
    for name in ("H2O", "CO2"):
        gas = caches[name].gas
        assert gas.molecule is not None and gas.formula == name and gas.database == db.path
        assert caches[name].cross_section is None and caches[name].gas_continua is None
    # no TIPS rows / no transitions / neither: an object whose spectrum is zero
    # (absorption.c:41,53-59), built without an upload
    grid = np.arange(600., 640., 0.1)
    for name in ("O2", "N2", "CFC11"):
        gas = caches[name].gas
        assert gas is not None and gas.molecule is None
        k = gas.absorption_coefficient(250., 5.e4, 0.2, grid, remove_pedestal=True)
        assert k.shape == (410,) and not k.any()
    # unknown alias: the constructor succeeds, the call fails like the reference's return-code
    # hook (gas_optics.py:15-26; spectral_database.c:152-156)
    with pytest.raises(ValueError, match="Error inside c functions."):
        caches["HCl"].gas.absorption_coefficient(250., 5.e4, 0.2, grid)
    assert caches["HCl"].cross_section is None
    assert caches["CFC11"].cross_section is not None and len(engine.bands) == 1


def test_cross_section_constructor_never_raises_for_a_bad_file(monkeypatch, tmp_path):
    """The reference's constructor stores two strings (cross_section.py:10-19); its
    MoleculeCache would let an OSError from ours through (spectroscopy.py:66-69)."""
    from pylbl_amd import arts_crossfit
    monkeypatch.setattr(arts_crossfit, "default_engine", lambda device=0: RecordingEngine())
    cross = arts_crossfit.CrossSection("CFC11", str(tmp_path / "missing.nc"))
    assert cross.formula == "CFC11" and cross.path.endswith("missing.nc")
    with pytest.raises(OSError):
        cross.absorption_coefficient(np.arange(800., 900., 1.), 250., 5.e4)


def test_errors_derive_from_the_reference_classes_when_pylbl_is_importable(tmp_path):
    """With a ``pyLBL.database`` on the path, what this package raises is caught by handlers
    naming the reference's classes (pyLBL/spectroscopy.py:55-56,68); without, or with
    PYLBL_AMD_STANDALONE=1, the classes stand alone (BaseException, database.py:489-506)."""
    package = tmp_path / "pyLBL"
    package.mkdir()
    (package / "__init__.py").write_text("")
    (package / "database.py").write_text("".join(
        f"class {name}(BaseException):\n    pass\n\n\n" for name in errors.NAMES))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import sys
        sys.path.insert(0, {root!r})
        import pylbl_amd.errors as mine
        try:
            import pyLBL.database as theirs
        except ImportError:
            theirs = None
        for name in mine.NAMES:
            ours = getattr(mine, name)
            assert issubclass(ours, BaseException) and not issubclass(ours, Exception)
            assert mine.kind(ours("x")) == name
            if {linked}:
                assert issubclass(ours, getattr(theirs, name)), name
                try:
                    raise ours("x")
                except (theirs.AliasNotFoundError, theirs.CrossSectionNotFoundError,
                        theirs.IsotopologuesNotFoundError, theirs.TipsDataNotFoundError,
                        theirs.TransitionsNotFoundError):
                    pass
                assert mine.kind(getattr(theirs, name)("x")) == name
            else:
                assert ours.__mro__[1] is BaseException, name
        print("ok")
        """)
    for linked, extra in ((True, {}), (False, {"PYLBL_AMD_STANDALONE": "1"})):
        env = dict(os.environ, PYTHONPATH=str(tmp_path), **extra)
        done = subprocess.run([sys.executable, "-c", script.format(root=root, linked=linked)],
                              env=env, capture_output=True, text=True, cwd=str(tmp_path))
        assert done.returncode == 0 and "ok" in done.stdout, done.stderr
    # here (no pyLBL importable) they stand alone
    assert errors.AliasNotFoundError.__mro__[1] is BaseException
    assert database.AliasNotFoundError is errors.AliasNotFoundError
