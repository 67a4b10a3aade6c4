"""TEST INFRASTRUCTURE: the a-posteriori feasibility checkers live in the public obca_amd/validate.py (pure numpy, independent of the
HIP path and of the oracle); this module only re-exports them for the tests written against `checkers`."""
from obca_amd.validate import *            # noqa: F401,F403
from obca_amd.validate import DMIN, _dyn, _geom   # noqa: F401
