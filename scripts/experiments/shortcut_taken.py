"""Is the whole-row shortcut of general_line taken?  Compares the sum of a spectrum computed by the
working library with one whose shortcut adds nothing (build/liblbl_midzero.so)."""
import os, shutil, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
code = '''
import sys, numpy as np
sys.path.insert(0, %r)
from pylbl_amd import synthetic
from pylbl_amd.engine import Engine
e = Engine()
t = synthetic.line_table("H2O", 1000., 1100.)
h = e.load(t)
print(repr(float(np.sum(e.compute(h, [296.], [101325.], [0.01], 1000, 1100, 1000)[0]))))
''' % ROOT
for name in ("mid", "midzero"):
    shutil.copy(os.path.join(ROOT, "build", f"liblbl_{name}.so"), os.path.join(ROOT, "pylbl_amd", "liblbl_amd.so"))
    print(name, subprocess.run([sys.executable, "-c", code], capture_output=True, text=True))
