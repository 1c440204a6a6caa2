"""Back-end registry with the shape of the reference's (pyLBL/plugins.py:7-34):
``molecular_lines[name] -> class``, looked up by Spectroscopy; an unknown name is a KeyError
(pyLBL/spectroscopy.py:118).

The reference fills its dictionaries from the entry points of the distribution named
"pyLBL" only, so a third-party engine cannot register through entry points; it joins by
inserting itself, which is what ``register`` does (also into pyLBL's own dictionary when
that package is importable).
"""
from .arts_crossfit import CrossSection
from .gas_optics import Gas
from .mt_ckd import CONTINUA

molecular_lines = {"mi355x": Gas}
cross_sections = {"arts_crossfit": CrossSection}
# Continua: model name -> {"CO2": class, "H2OForeign": class, ...} (plugins.py:24-34).  The
# MT-CKD model keeps the reference's name, so Spectroscopy's default fills slot 1.
continua = {"mt_ckd": CONTINUA}
models = molecular_lines.keys()


def register(name="mi355x", into=None):
    """Adds the MI355X Gas class to a ``molecular_lines``-shaped dictionary.

    Args:
        name: Backend name to register under.
        into: Dictionary to insert into; default tries ``pyLBL.plugins.molecular_lines``.
    """
    if into is None:
        from pyLBL.plugins import molecular_lines as into  # noqa: raises if pyLBL is absent
    into[name] = Gas
    return into
