"""Fixture: what the reference's own ``Database`` object hands a lines back end.

Drives /root/reference/pyLBL/database.py (loaded without running pyLBL/__init__.py -- that
needs the built C extension -- and with an empty stand-in for ``xarray``, which only
arts_crossfit/cross_section.py imports and nothing here calls): its SQLAlchemy table classes
write a small spectral database the way ``Database.create`` would (create_all + one
session.add per row, pyLBL/database.py:20-127), then ``molecules()``, ``gas()``, ``tips()``
and ``arts_crossfit()`` are called on it and their return values dumped.  Output, data only:

* tests/golden/refdb.db   the SQLite file as the reference wrote it;
* tests/golden/refdb.npz  the arrays its query helpers returned, per molecule, plus the names of
                          the exception classes the reference raised for the molecules that
                          lack something.

Run here (build container, /root/reference present):  python tests/golden/make_refdb.py
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATABASE = os.path.join(HERE, "refdb.db")
TARGET = os.path.join(HERE, "refdb.npz")

FIELDS = ("nu", "sw", "gamma_air", "gamma_self", "n_air", "delta_air", "elower")

# molecule id, formula, isotopologue ids in HITRAN's listing order, masses
MOLECULES = [
    (1, "H2O", [1, 2, 3, 4], [18.010565, 20.014811, 19.01478, 19.01674]),
    # twelve isotopologues: the tenth carries HITRAN's local id 0 (spectral_database.c:173-177)
    (2, "CO2", [1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 11, 12],
     [43.98983, 44.993185, 45.994076, 44.994045, 46.997431, 45.9974, 47.998322, 46.998291,
      45.998262, 49.001675, 48.001646, 47.0016182378]),
    (7, "O2", [1, 2], [31.98983, 33.994076]),         # transitions but no TIPS rows
    (22, "N2", [1], [28.006148]),                      # TIPS rows but no transitions
    (1001, "CFC11", [], []),                           # cross-section only
]
LINES = {"H2O": 48, "CO2": 64, "O2": 6}
TIPS = {"H2O": 4, "CO2": 12, "N2": 1}
TIPS_T = np.arange(150., 351., 1.)


def load_reference_database_module():
    """pyLBL.database without pyLBL/__init__.py."""
    package = types.ModuleType("pyLBL")
    package.__path__ = [os.path.join(REFERENCE, "pyLBL")]
    sys.modules["pyLBL"] = package
    if "xarray" not in sys.modules:
        try:
            import xarray  # noqa: F401
        except ImportError:
            stand_in = types.ModuleType("xarray")
            stand_in.open_dataset = None
            stand_in.DataArray = stand_in.Dataset = None
            sys.modules["xarray"] = stand_in
    return importlib.import_module("pyLBL.database")


def lines_of(formula, count, ids, rng):
    """Seeded line parameters; the CO2 rows include local id 0 and the rows are NOT all
    ascending (row order is what the reference's pedestal depends on)."""
    nu = np.sort(rng.uniform(600., 700., count))
    if formula == "CO2":
        nu[[5, 6]] = nu[[6, 5]]
    rows = dict(
        nu=nu, sw=10.**rng.uniform(-24., -19., count), gamma_air=rng.uniform(0.03, 0.12, count),
        gamma_self=rng.uniform(0.05, 0.5, count), n_air=rng.uniform(0.4, 0.85, count),
        delta_air=rng.uniform(-0.01, 0.002, count), elower=rng.uniform(0., 3000., count))
    rows["local_iso_id"] = rng.choice(ids, size=count)
    return rows


def main():
    db = load_reference_database_module()
    if os.path.exists(DATABASE):
        os.remove(DATABASE)
    database = db.Database(DATABASE)           # create_all (pyLBL/database.py:136-146)
    from sqlalchemy.orm import Session
    rng = np.random.default_rng(20261004)
    with Session(database.engine, future=True) as session:
        for id, formula, ids, masses in MOLECULES:
            session.add(db.MoleculeTable(id=id, stoichiometric_formula=formula,
                                         ordinary_formula=formula, common_name=formula.lower()))
            for alias in (formula, formula.lower()):
                session.add(db.MoleculeAliasTable(alias=alias, molecule=id))
            for isoid, mass in zip(ids, masses):
                session.add(db.IsotopologueTable(molecule_id=id, isoid=isoid,
                                                 iso_name=f"{formula}-{isoid}",
                                                 abundance=0.5, mass=mass))
            if formula in LINES:
                rows = lines_of(formula, LINES[formula], ids, rng)
                for i in range(LINES[formula]):
                    session.add(db.TransitionTable(
                        global_iso_id=100*id + int(rows["local_iso_id"][i]), molecule_id=id,
                        local_iso_id=int(rows["local_iso_id"][i]),
                        **{x: float(rows[x][i]) for x in FIELDS}))
            if formula in TIPS:
                # iso-major, T-minor, 0-based isotopologue column (pyLBL/database.py:109-127);
                # values rounded to float32 like the TIPS tables (webapi/tips_api.py:86-87)
                for iso in range(TIPS[formula]):
                    q = np.float32((20. + 7.*iso)*(TIPS_T/296.)**1.5).astype(np.float64)
                    for t, value in zip(TIPS_T, q):
                        session.add(db.TipsTable(molecule_id=id, isotopologue_id=iso,
                                                 temperature=float(t), data=float(value)))
        session.add(db.ArtsCrossFitTable(molecule_id=1001, path="/data/cross-sections/CFC11.nc"))
        session.commit()

    arrays = {"molecules": np.asarray(database.molecules())}
    raised = {}
    for _, formula, _, _ in MOLECULES:
        try:
            name, mass, transitions, partition = database.gas(formula)
        except BaseException as error:      # the reference's errors derive from BaseException
            raised[f"gas:{formula}"] = type(error).__name__
        else:
            arrays[f"{formula}_formula"] = np.asarray(name)
            arrays[f"{formula}_mass"] = np.asarray(mass, dtype=np.float64)
            for x in FIELDS + ("local_iso_id", "molecule_id", "global_iso_id"):
                arrays[f"{formula}_{x}"] = np.asarray([getattr(t, x) for t in transitions])
            arrays[f"{formula}_q_temperature"] = np.asarray(partition.temperature)
            arrays[f"{formula}_q_data"] = np.asarray(partition.data)
            arrays[f"{formula}_q_288p99"] = np.asarray(
                [partition.total_partition_function(288.99, i + 1)
                 for i in partition.isotopologue])
        try:
            temperature, data = database.tips(formula)
        except BaseException as error:
            raised[f"tips:{formula}"] = type(error).__name__
        else:
            arrays[f"{formula}_tips_temperature"] = temperature
            arrays[f"{formula}_tips_data"] = data
        try:
            arrays[f"{formula}_arts_crossfit"] = np.asarray(database.arts_crossfit(formula))
        except BaseException as error:
            raised[f"arts_crossfit:{formula}"] = type(error).__name__
    try:
        database.gas("HCl")
    except BaseException as error:
        raised["gas:HCl"] = type(error).__name__
    arrays["raised_keys"] = np.asarray(sorted(raised))
    arrays["raised_values"] = np.asarray([raised[k] for k in sorted(raised)])
    database.engine.dispose()
    np.savez_compressed(TARGET, **arrays)
    for key in sorted(raised):
        print(f"{key}: {raised[key]}")
    print(DATABASE, os.path.getsize(DATABASE), "bytes;", TARGET, os.path.getsize(TARGET), "bytes")


if __name__ == "__main__":
    main()
