"""MI355X-native molecular-lines engine: a drop-in for pyLBL's lines backend, widened to the
other two mechanism slots of ``compute_absorption`` (MT-CKD continua, ARTS-crossfit
cross-sections).

Names mirror ``pyLBL/__init__.py:1-5``: ``Gas``, ``Database``, ``molecular_lines``,
``continua``, ``cross_sections``, ``Spectroscopy``.  Importing the package does not touch the
GPU; the first ``Gas``/``Engine``/continuum/``CrossSection`` does, and fails loudly if the HIP
library or the device is missing.
"""
from .database import Database, LineTable, MemoryDatabase, TotalPartitionFunction, \
                      write_database
from .errors import AliasNotFoundError, CrossSectionNotFoundError, EngineError, \
                    IsotopologuesNotFoundError, TipsDataNotFoundError, TransitionsNotFoundError
from .engine import DeviceSpectra, Engine, default_engine
from .arts_crossfit import CrossSection
from .gas_optics import Gas
from .plugins import continua, cross_sections, models, molecular_lines, register
from .spectroscopy import Atmosphere, Spectroscopy, number_density

__all__ = ["Gas", "CrossSection", "Database", "MemoryDatabase", "LineTable", "TotalPartitionFunction", "write_database",
           "Engine", "DeviceSpectra", "default_engine", "Spectroscopy", "Atmosphere",
           "number_density", "molecular_lines", "continua", "cross_sections", "models",
           "register", "AliasNotFoundError", "CrossSectionNotFoundError", "EngineError", "IsotopologuesNotFoundError",
           "TipsDataNotFoundError", "TransitionsNotFoundError"]
