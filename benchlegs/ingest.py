"""What it costs to get a molecule's line table from the reference's SQLite file into HBM."""
import time

import numpy as np


def ingest_leg(engine, tables, db_path, atmos, grid_args):
    """What it costs to get a molecule's line table from the reference's SQLite file into HBM
    (line-table load time reported separately; the reference pays its read
    on every call, absorption.c:44-86): the three routes of pylbl_amd.database.line_table_of,
    the upload (lbl_molecule_load: sort by wavenumber, eleven arrays to HBM), and the
    same-signature C entry's first call on a file (SQLite read in C + upload + compute) against
    its second (line table found resident)."""
    from ctypes import c_char_p, c_double, c_int32
    from pylbl_amd import database
    out = {"database": f"SQLite file in the reference's schema, "
                       f"{'+'.join(f'{t.formula} {t.num_lines}' for t in tables)} transitions",
           "routes": {}, "per_molecule": {}}

    class PathOnly(object):             # what pyLBL.database.Database looks like from outside
        def __init__(self, path):
            self.path = path

    class QueriesOnly(object):          # a database object that cannot be opened as a file
        def __init__(self, inner):
            self.gas, self.tips = inner.gas, inner.tips
    file_backed = database.Database(db_path)
    routes = (("line_table", "an object with line_table(name) (this package's Database)",
               file_backed),
              ("path", "an object with .path only (pyLBL.database.Database as the reference hands "
                       "it over, spectroscopy.py:54): the C engine's own four SELECTs", PathOnly(db_path)),
              ("gas_tips", "an object with .gas(name) / .tips(name) only (record arrays; the "
                           "reference's ORM rows would add their own object construction)",
               QueriesOnly(file_backed)))
    loaded = {}
    for key, what, source in routes:
        seconds = {}
        for table in tables:
            start = time.perf_counter()
            loaded[table.formula] = database.line_table_of(source, table.formula)
            seconds[table.formula] = time.perf_counter() - start
        out["routes"][key] = {"what": what, "seconds": seconds, "total_s": sum(seconds.values())}
    upload = {}
    for table in tables:
        start = time.perf_counter()
        handle = engine.load(loaded[table.formula])
        upload[table.formula] = time.perf_counter() - start
        engine.free(handle)
    out["upload_s"] = upload
    # The drop-in C entry (absorption.c:19-30's signature): first call reads the file itself.
    lib = engine.lib
    v0, vn, n_per_v = grid_args
    k = np.zeros((vn - v0)*n_per_v)
    first, second = {}, {}
    for table in tables:
        args = (c_double(atmos.p[0]), c_double(atmos.t[0]), c_double(atmos.vmr[table.formula][0]),
                c_int32(v0), c_int32(vn), c_int32(n_per_v), k.ctypes.data,
                c_char_p(str(db_path).encode()), c_char_p(table.formula.encode()), c_int32(25),
                c_int32(0))
        for book in (first, second):
            start = time.perf_counter()
            status = lib.lbl_absorption(*args)
            book[table.formula] = time.perf_counter() - start
            if status != 0:
                raise RuntimeError("lbl_absorption failed")
    out["c_entry_first_call_s"] = first
    out["c_entry_second_call_s"] = second
    for table in tables:
        f = table.formula
        out["per_molecule"][f] = {
            "lines": int(table.num_lines),
            "read_s": out["routes"]["path"]["seconds"][f], "upload_s": upload[f],
            "c_entry_ingest_s": first[f] - second[f]}
    out["note"] = ("paid once per molecule and process (resident line tables; the C entry keys "
                   "them by path + mtime + inode); compare cpu_baseline.split."
                   "read_s_per_molecule, which the reference pays on every (level, molecule) call. "
                   "c_entry_*: host array in and out, so both calls include the 40 MB-class "
                   "copy back; their difference is the ingest")
    return out

