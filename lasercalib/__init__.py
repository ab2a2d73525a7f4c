"""Import-path shim: ``from lasercalib.pySBA import PySBA`` resolves to the MI355X engine.

The reference package is called ``lasercalib``; its caller (scripts/calibrate_camera.py:7) and
the pickles it writes (calibrate_camera.py:86-88) name ``lasercalib.pySBA.PySBA``.  The product
code lives in ``lasercalib_amd``; this directory only forwards that one module.

To run the reference's other (out-of-scope) modules beside it -- sba_print, convert_params ... --
point LASERCALIB_UPSTREAM at the upstream ``lasercalib`` directory: it is appended to this
package's search path, so every module except pySBA comes from there.
"""
import os as _os

_up = _os.environ.get("LASERCALIB_UPSTREAM")
if _up and _os.path.isdir(_up):
    __path__.append(_up)
