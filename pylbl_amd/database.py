"""Read side of the spectral line database (SQLite file written by pyLBL).

Mirrors what the lines path needs from pyLBL/database.py: ``Database.path``
(:146), ``.molecules()`` (:340-348), ``.gas(name)`` (:350-367) and ``.tips(name)``
(:369-395).  Ingest from HITRAN/TIPS web services (``create``) is out of scope.

The file is read with the standard-library ``sqlite3`` using the same four SELECT
statements the reference's C engine issues, so rows come back in the same
(rowid) order the reference iterates them in:

* alias -> molecule id      pyLBL/c_lib/spectral_database.c:143
* TIPS rows                  pyLBL/c_lib/spectral_database.c:55
* isotopologue masses        pyLBL/c_lib/spectral_database.c:113
* transitions                pyLBL/c_lib/absorption.c:69-70

``LineTable`` is the struct-of-arrays form one molecule is turned into, once, before
it is uploaded to the GPU; it is also what the synthetic generators produce.
"""
from collections import namedtuple
from dataclasses import dataclass
import os
import sqlite3

import numpy as np

from .errors import AliasNotFoundError, CrossSectionNotFoundError, EngineError, \
                    IsotopologuesNotFoundError, TipsDataNotFoundError, TransitionsNotFoundError


TIPS_REFERENCE_TEMPERATURE = 296.  # pyLBL/tips.py:6

# The reference's C engine holds masses in a 32-slot buffer (absorption.c:62-64).
MASS_SLOTS = 32

LINE_COLUMNS = ("nu", "sw", "gamma_air", "gamma_self", "n_air", "elower", "delta_air")


@dataclass
class LineTable:
    """One molecule's transitions (reference row order), masses and TIPS table."""
    formula: str
    molecule_id: int
    nu: np.ndarray
    sw: np.ndarray
    gamma_air: np.ndarray
    gamma_self: np.ndarray
    n_air: np.ndarray
    elower: np.ndarray
    delta_air: np.ndarray
    local_iso_id: np.ndarray     # int32, raw column value (0 means isotopologue 10)
    isoid: np.ndarray            # int, one per isotopologue row
    mass: np.ndarray             # float64, one per isotopologue row
    tips_temperature: np.ndarray  # float64[num_t]
    tips_data: np.ndarray        # float64[num_iso, num_t]

    @property
    def num_lines(self):
        return int(self.nu.size)

    def mass_by_slot(self):
        """mass[isoid - 1], HITRAN isoid 0 stored as 10 (spectral_database.c:113-129)."""
        out = np.zeros(MASS_SLOTS, dtype=np.float64)
        for isoid, mass in zip(self.isoid, self.mass):
            slot = 10 if int(isoid) == 0 else int(isoid)
            if slot > MASS_SLOTS:
                raise ValueError(f"isotopologue id {isoid} does not fit {MASS_SLOTS} slots.")
            out[slot - 1] = mass
        return out

    def subset(self, mask):
        """Same molecule, a subset of rows (order kept)."""
        fields = {x: getattr(self, x)[mask] for x in LINE_COLUMNS}
        return LineTable(self.formula, self.molecule_id, local_iso_id=self.local_iso_id[mask],
                         isoid=self.isoid, mass=self.mass,
                         tips_temperature=self.tips_temperature, tips_data=self.tips_data,
                         **fields)


class TotalPartitionFunction(object):
    """Total internal partition sums Q(T) of one molecule, tabulated per isotopologue on a
    common temperature axis: the host-side twin of spectral_database.c:97-104 and of the
    object the reference's Database.gas() hands out (pyLBL/tips.py:9-39).

    Attributes:
        molecule: Chemical formula.
        temperature: float64[num_t], ascending [K].
        data: float64[num_iso, num_t].
    """
    def __init__(self, molecule, temperature, data):
        self.molecule = molecule
        self.temperature = np.asarray(temperature, dtype=np.float64)
        self.data = np.asarray(data, dtype=np.float64)
        if self.data.ndim != 2 or self.data.shape[1] != self.temperature.size:
            raise ValueError("data must be [num_iso, num_t] on the temperature axis.")

    @property
    def isotopologue(self):
        """Row indices of the table (0-based), one per isotopologue."""
        return list(range(len(self.data)))

    def total_partition_function(self, temperature, isotopologue):
        """Q(temperature) of the 1-based `isotopologue`, linear between the two tabulated
        temperatures that bracket it (the lower one strictly below `temperature`, as the
        reference's left-sided search chooses it)."""
        row = self.data[isotopologue - 1]
        axis = self.temperature
        lower = int(np.searchsorted(axis, temperature, side="left")) - 1
        t0, t1 = axis[lower], axis[lower + 1]
        weight = (temperature - t0)/(t1 - t0)
        return row[lower] + (row[lower + 1] - row[lower])*weight


GasData = namedtuple("GasData", ["formula", "mass", "transitions", "partition_function"])


class Database(object):
    """Spectral line parameter database (read side).

    Attributes:
        path: String path to the SQLite file (what the reference's Gas stores,
              pyLBL/c_lib/gas_optics.py:43).
    """
    def __init__(self, path):
        self.path = str(path)

    def _connect(self):
        return sqlite3.connect(f"file:{self.path}?mode=ro", uri=True)

    def _molecule_id(self, connection, name):
        try:
            row = connection.execute(
                "select molecule from molecule_alias where alias == ?", (name,)).fetchone()
        except sqlite3.OperationalError:
            row = None
        if row is None:
            raise AliasNotFoundError(f"{name} not found in database.")
        return int(row[0])

    def arts_crossfit(self, name):
        """Path of the molecule's cross-section coefficient file (pyLBL/database.py:397-415).

        Raises:
            CrossSectionNotFoundError if the database lists none.
        """
        with self._connect() as connection:
            id = self._molecule_id(connection, name)
            try:
                row = connection.execute(
                    "select path from artscrossfit where molcule_id == ?", (id,)).fetchone()
            except sqlite3.OperationalError:
                row = None
        if row is None:
            raise CrossSectionNotFoundError(f"No cross sections for {name}.")
        return row[0]

    def molecules(self):
        """Lists the chemical formulae of all molecules in the database."""
        with self._connect() as connection:
            rows = connection.execute("select ordinary_formula from molecule").fetchall()
        return [x[0] for x in rows]

    def tips(self, name):
        """Returns (temperature[num_t], data[num_iso, num_t]) for a molecule alias."""
        with self._connect() as connection:
            return self._tips(connection, self._molecule_id(connection, name), name)

    def _tips(self, connection, id, name):
        rows = connection.execute(
            "select isotopologue_id, temperature, data from tips where molecule_id == ?",
            (id,)).fetchall()
        if not rows:
            raise TipsDataNotFoundError(f"no tips data for {name}.")
        rows = np.asarray(rows, dtype=np.float64)
        num_iso = 1 + int(np.count_nonzero(np.diff(rows[:, 0])))
        if rows.shape[0] % num_iso:
            raise ValueError("tips data is not rectangular.")  # spectral_database.c:85-90
        num_t = rows.shape[0]//num_iso
        temperature = rows[:num_t, 1].copy()
        data = rows[:, 2].reshape(num_iso, num_t).copy()
        return temperature, data

    def line_table(self, name):
        """Reads everything the lines engine needs for one molecule, as arrays: through the
        engine library's C reader (lbl_table_read -- the row loop of absorption.c:76-86 without
        the per-row Python objects: 0.1 s instead of 0.4 s for 400 k transitions), or, where the
        library cannot be loaded (no hipcc, no HIP runtime: reading a file is not computing),
        through the standard-library sqlite3 with the same statements."""
        try:
            from . import engine
            status, message, fields = engine.read_line_table(self.path, name)
        except (EngineError, OSError):
            return self._line_table_sqlite3(name)
        if status == engine.LBL_OK:
            columns = {x: fields["columns"][i] for i, x in enumerate(LINE_COLUMNS)}
            return LineTable(
                formula=fields["formula"], molecule_id=fields["molecule_id"],
                local_iso_id=fields["local_iso_id"], isoid=fields["isoid"], mass=fields["mass"],
                tips_temperature=fields["tips_temperature"], tips_data=fields["tips_data"],
                **columns)
        if status == engine.TABLE_NO_ALIAS:
            raise AliasNotFoundError(f"{name} not found in database.")
        if status == engine.TABLE_NO_TIPS:
            raise TipsDataNotFoundError(f"no tips data for {name}.")
        if status == engine.TABLE_NOT_RECTANGULAR:
            raise ValueError("tips data is not rectangular.")  # spectral_database.c:85-90
        if status == engine.TABLE_NO_ISOTOPOLOGUES:
            raise IsotopologuesNotFoundError(message)
        if status == engine.TABLE_NO_TRANSITIONS:
            raise TransitionsNotFoundError(message)
        if status == engine.TABLE_OPEN_FAILED:
            raise sqlite3.OperationalError(f"unable to open database file: {self.path}")
        raise EngineError(message)

    def _line_table_sqlite3(self, name):
        """line_table() with the standard-library sqlite3 (same statements, same row order)."""
        with self._connect() as connection:
            id = self._molecule_id(connection, name)
            temperature, data = self._tips(connection, id, name)
            iso = connection.execute(
                "select isoid, mass from isotopologue where molecule_id == ?", (id,)).fetchall()
            if not iso:
                raise IsotopologuesNotFoundError(
                    f"isotopologues not found for molecule {id}.")
            rows = connection.execute(
                "select nu, sw, gamma_air, gamma_self, n_air, elower, delta_air, "
                "local_iso_id from transition where molecule_id == ?", (id,)).fetchall()
            if not rows:
                raise TransitionsNotFoundError(f"transitions not found for molecule {id}.")
            formula = connection.execute(
                "select ordinary_formula from molecule where id == ?", (id,)).fetchone()
        rows = np.asarray(rows, dtype=np.float64)
        columns = {x: np.ascontiguousarray(rows[:, i]) for i, x in enumerate(LINE_COLUMNS)}
        return LineTable(
            formula=formula[0] if formula else name, molecule_id=id,
            local_iso_id=np.ascontiguousarray(rows[:, 7]).astype(np.int32),
            isoid=np.asarray([x[0] for x in iso], dtype=np.int64),
            mass=np.asarray([x[1] for x in iso], dtype=np.float64),
            tips_temperature=temperature, tips_data=data, **columns)

    def gas(self, name):
        """Same return shape as pyLBL/database.py:350-367: (formula, masses in
        isotopologue-row order, transitions in row order, TotalPartitionFunction).
        Transitions come back as a numpy record array: ``t[i].nu``, ``t.nu``."""
        table = self.line_table(name)
        transitions = np.rec.fromarrays(
            [getattr(table, x) for x in LINE_COLUMNS] + [table.local_iso_id],
            names=list(LINE_COLUMNS) + ["local_iso_id"])
        tips = TotalPartitionFunction(name, table.tips_temperature, table.tips_data)
        return GasData(table.formula, list(table.mass), transitions, tips)


class MemoryDatabase(object):
    """The read surface of ``Database`` over LineTables already in memory (synthetic tables, or
    a line list the caller assembled): ``path`` is None, nothing is written to disk.  What the
    lines backend and ``Spectroscopy`` need -- ``molecules()``, ``line_table(name)``,
    ``gas(name)``, ``tips(name)`` -- behaves like the file-backed class, including
    AliasNotFoundError for an unknown name."""
    def __init__(self, tables, aliases=None, cross_sections=None):
        """cross_sections: optional {formula: path of its ARTS-crossfit coefficient file}, what
        the artscrossfit table holds in a file-backed database."""
        self.path = None
        self.tables = {table.formula: table for table in tables}
        self.aliases = {}
        for formula, names in (aliases or {}).items():
            for name in names:
                self.aliases[name] = formula
        self.cross_sections = dict(cross_sections or {})

    def molecules(self):
        return list(self.tables) + [x for x in self.cross_sections if x not in self.tables]

    def arts_crossfit(self, name):
        """Same outcomes as Database.arts_crossfit (pyLBL/database.py:397-415)."""
        formula = self.aliases.get(name, name)
        if formula in self.cross_sections:
            return self.cross_sections[formula]
        if formula not in self.tables:
            raise AliasNotFoundError(f"{name} not found in database.")
        raise CrossSectionNotFoundError(f"No cross sections for {name}.")

    def line_table(self, name):
        formula = self.aliases.get(name, name)
        if formula not in self.tables:
            if formula in self.cross_sections:
                # A molecule the database knows only through its cross-sections: the reference's
                # ingest gives it a molecule row and an alias but no TIPS rows or transitions
                # (pyLBL/database.py:213-262), i.e. a zero lines spectrum, not an error.
                raise TipsDataNotFoundError(f"tips data not found for molecule {name}.")
            raise AliasNotFoundError(f"{name} not found in database.")
        table = self.tables[formula]
        if table.tips_data is None or np.size(table.tips_data) == 0:
            raise TipsDataNotFoundError(f"tips data not found for molecule {name}.")
        if table.num_lines == 0:
            raise TransitionsNotFoundError(f"transitions not found for molecule {name}.")
        return table

    def tips(self, name):
        table = self.line_table(name)
        return table.tips_temperature, table.tips_data

    def gas(self, name):
        table = self.line_table(name)
        transitions = np.rec.fromarrays(
            [getattr(table, x) for x in LINE_COLUMNS] + [table.local_iso_id],
            names=list(LINE_COLUMNS) + ["local_iso_id"])
        tips = TotalPartitionFunction(name, table.tips_temperature, table.tips_data)
        return GasData(table.formula, list(table.mass), transitions, tips)


def _column(transitions, name, dtype):
    """One field of the reference's transition rows (a list of ORM objects,
    pyLBL/database.py:446-459) or of a record array, as a contiguous array."""
    if hasattr(transitions, "dtype") and transitions.dtype.names:
        return np.ascontiguousarray(transitions[name], dtype=dtype)
    return np.fromiter((getattr(row, name) for row in transitions), dtype=dtype,
                       count=len(transitions))


def table_from_gas(lines_database, name):
    """LineTable assembled from the two query helpers of the reference's Database object,
    ``.gas(name)`` (pyLBL/database.py:350-367) and ``.tips(name)`` (:369-395), for databases
    that are not a file this process can open.

    ``.gas()`` hands out the masses in isotopologue-row order without their ``isoid``
    column, whereas the reference's C reader files them under ``isoid`` (0 -> 10,
    spectral_database.c:113-129).  HITRAN numbers a molecule's isotopologues 1, 2, ... 9, 0,
    11, ... in the order it lists them, which is the order ``Database.create`` inserts them
    in (pyLBL/database.py:58-77), so row r is taken as local id r + 1.
    """
    # Partition sums before anything else, as absorption.c:50-64 looks things up: a molecule
    # without them yields zeros whatever else it lacks (TipsDataNotFoundError here).
    temperature, data = lines_database.tips(name)
    formula, mass, transitions, _ = lines_database.gas(name)
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 2:
        raise ValueError("tips data is not rectangular.")       # spectral_database.c:85-90
    n = len(transitions)
    if n == 0:
        raise TransitionsNotFoundError(f"transitions not found for molecule {name}.")
    columns = {x: _column(transitions, x, np.float64) for x in LINE_COLUMNS}
    try:
        molecule_id = int(transitions[0].molecule_id)
    except (AttributeError, TypeError, ValueError, IndexError):
        molecule_id = 0
    return LineTable(
        formula=formula, molecule_id=molecule_id,
        local_iso_id=_column(transitions, "local_iso_id", np.int64).astype(np.int32),
        isoid=np.arange(1, len(mass) + 1, dtype=np.int64),
        mass=np.asarray(mass, dtype=np.float64),
        tips_temperature=np.ascontiguousarray(temperature, dtype=np.float64),
        tips_data=np.ascontiguousarray(data), **columns)


def line_table_of(lines_database, name):
    """Everything the lines engine needs for one molecule, from whichever database object the
    caller holds -- the reference hands its backends ``pyLBL.database.Database``
    (pyLBL/spectroscopy.py:54), which has ``.path``, ``.gas()``, ``.tips()``,
    ``.molecules()`` and ``.arts_crossfit()`` but no ``line_table``:

    1. an object with ``line_table(name)`` (this package's Database / MemoryDatabase);
    2. an object whose ``.path`` names a readable SQLite file (pyLBL/database.py:146): the
       file is read with the reference C engine's own four SELECTs, so rows, masses by
       ``isoid`` and TIPS rows are exactly what ``absorption()`` would see;
    3. an object with ``.gas(name)`` / ``.tips(name)`` only: ``table_from_gas``.

    Raises whatever the source raises for a molecule it does not know (this package's
    errors, or the reference's own classes of the same names from routes 1 and 3).
    """
    reader = getattr(lines_database, "line_table", None)
    if callable(reader):
        return reader(name)
    path = getattr(lines_database, "path", None)
    if isinstance(path, (str, os.PathLike)) and os.path.isfile(path):
        return Database(path).line_table(name)
    if callable(getattr(lines_database, "gas", None)):
        return table_from_gas(lines_database, name)
    raise TypeError(
        f"{type(lines_database).__name__} is not a line database: it needs line_table(name), "
        "a .path to an SQLite file in pyLBL's schema, or gas(name)/tips(name).")


# Exact DDL of the reference's schema (pyLBL/database.py:418-486 as emitted by
# SQLAlchemy's create_all); note the misspelt "molcule_id" column of artscrossfit.
SCHEMA = """
CREATE TABLE molecule (id INTEGER NOT NULL, stoichiometric_formula VARCHAR,
    ordinary_formula VARCHAR, common_name VARCHAR, PRIMARY KEY (id));
CREATE TABLE isotopologue (id INTEGER NOT NULL, molecule_id INTEGER, isoid INTEGER,
    iso_name VARCHAR, abundance FLOAT, mass FLOAT, PRIMARY KEY (id),
    FOREIGN KEY(molecule_id) REFERENCES molecule (id));
CREATE TABLE molecule_alias (id INTEGER NOT NULL, alias VARCHAR, molecule INTEGER,
    PRIMARY KEY (id), FOREIGN KEY(molecule) REFERENCES molecule (id));
CREATE TABLE transition (id INTEGER NOT NULL, global_iso_id INTEGER, molecule_id INTEGER,
    local_iso_id INTEGER, nu FLOAT, sw FLOAT, gamma_air FLOAT, gamma_self FLOAT,
    n_air FLOAT, delta_air FLOAT, elower FLOAT, PRIMARY KEY (id),
    FOREIGN KEY(molecule_id) REFERENCES molecule (id));
CREATE TABLE tips (id INTEGER NOT NULL, molecule_id INTEGER, isotopologue_id INTEGER,
    temperature FLOAT, data FLOAT, PRIMARY KEY (id),
    FOREIGN KEY(molecule_id) REFERENCES molecule (id));
CREATE TABLE artscrossfit (id INTEGER NOT NULL, molcule_id INTEGER, path VARCHAR,
    PRIMARY KEY (id), FOREIGN KEY(molcule_id) REFERENCES molecule (id));
CREATE TABLE metadata (id INTEGER NOT NULL, molecule_id INTEGER, database VARCHAR,
    time VARCHAR, PRIMARY KEY (id), FOREIGN KEY(molecule_id) REFERENCES molecule (id));
"""


def write_database(path, tables, with_tips=None, aliases=None, cross_sections=None):
    """Writes LineTables into a fresh SQLite file with the reference's schema.

    Used for fixtures and benchmarks (no HITRAN data exists offline).  Row order of
    every table is the array order, which is what the reference's readers rely on
    (pyLBL/database.py:80-127).

    Args:
        path: Output file (overwritten).
        tables: Iterable of LineTable.
        with_tips: Optional set of formulae to write TIPS rows for (default: all).
        aliases: Optional dict formula -> extra alias strings.
        cross_sections: Optional dict formula -> path of its ARTS-crossfit coefficient file
                        (table artscrossfit, pyLBL/database.py:472-477).
    """
    import os
    if os.path.exists(path):
        os.remove(path)
    connection = sqlite3.connect(str(path))
    connection.executescript(SCHEMA)
    for table in tables:
        id = int(table.molecule_id)
        connection.execute("insert into molecule values (?, ?, ?, ?)",
                           (id, table.formula, table.formula, table.formula))
        names = [table.formula] + list((aliases or {}).get(table.formula, []))
        connection.executemany("insert into molecule_alias (alias, molecule) values (?, ?)",
                               [(x, id) for x in names])
        connection.executemany(
            "insert into isotopologue (molecule_id, isoid, iso_name, abundance, mass) "
            "values (?, ?, ?, ?, ?)",
            [(id, int(i), f"{table.formula}-{int(i)}", 1., float(m))
             for i, m in zip(table.isoid, table.mass)])
        connection.executemany(
            "insert into transition (global_iso_id, molecule_id, local_iso_id, nu, sw, "
            "gamma_air, gamma_self, n_air, delta_air, elower) values (?,?,?,?,?,?,?,?,?,?)",
            zip([0]*table.num_lines, [id]*table.num_lines, table.local_iso_id.tolist(),
                table.nu.tolist(), table.sw.tolist(), table.gamma_air.tolist(),
                table.gamma_self.tolist(), table.n_air.tolist(), table.delta_air.tolist(),
                table.elower.tolist()))
        if with_tips is None or table.formula in with_tips:
            num_iso, num_t = table.tips_data.shape
            t = table.tips_temperature.tolist()
            for iso in range(num_iso):
                connection.executemany(
                    "insert into tips (molecule_id, isotopologue_id, temperature, data) "
                    "values (?, ?, ?, ?)",
                    zip([id]*num_t, [iso]*num_t, t, table.tips_data[iso].tolist()))
    for table in tables:
        if cross_sections and table.formula in cross_sections:
            connection.execute("insert into artscrossfit (molcule_id, path) values (?, ?)",
                               (int(table.molecule_id), str(cross_sections[table.formula])))
    connection.commit()
    connection.close()
    return str(path)
