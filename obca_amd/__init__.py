"""obca_amd -- MI355X-native batched OBCA (optimization-based collision avoidance) parking NLP solver.

Host-side mirror of the reference's Julia entry points over the C ABI of libobca_hip.so (include/obca_hip.h).
The HIP library is the only compute path: importing works anywhere, but every solve raises if the library or a gfx950
device is missing -- there is no CPU fallback.
"""
from .api import (ObcaError, Context, Batch, ParkingSignedDist, ParkingDist, DualMultWS, parking_signed_dist_batch, dualmult_ws_batch,
                  default_opts, ipopt_opts, warm_restart_opts, selftest, library_path, build_library, QuadBatch, QuadcopterSignedDist, QuadcopterDist,
                  quadcopter_signed_dist_batch, quadcopter_default_opts, quadcopter_ipopt_opts)

__all__ = ["ObcaError", "Context", "Batch", "ParkingSignedDist", "ParkingDist", "DualMultWS", "parking_signed_dist_batch",
           "dualmult_ws_batch", "default_opts", "ipopt_opts", "warm_restart_opts", "selftest", "library_path", "build_library", "QuadBatch", "QuadcopterSignedDist", "QuadcopterDist",
           "quadcopter_signed_dist_batch", "quadcopter_default_opts", "quadcopter_ipopt_opts"]
