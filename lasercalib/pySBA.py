"""``lasercalib.pySBA`` -> the MI355X engine's PySBA (same surface as the reference module)."""
from lasercalib_amd.pySBA import PySBA, SBAResult, assemble_jacobian  # noqa: F401
