"""apr_amd -- MI355X (gfx950) native implementation of APR's per-point feature-extraction hot path.

Layout:
  csrc/ + lib/libapr_hip.so   hand-written HIP kernels behind the C ABI of include/apr_hip.h
  _lib.py, ops.py             ctypes binding + tensor-level operator layer (no CPU fallback)
  MinkowskiEngine/            ME-compatible operator subset used by FCGF_APR
  fcgf/                       FCGF_APR mirror: sparse ResUNet, feature matching, pose fit
  predator/                   Predator_APR mirror: grid subsampling, radius neighbours, KPConv encoder
  synth.py                    deterministic KITTI-shaped synthetic scans
"""
import sys

__version__ = "0.1.0"


def install_shims():
    """Make `import MinkowskiEngine as ME` (and `MinkowskiEngine.MinkowskiFunctional`) resolve to apr_amd."""
    from . import MinkowskiEngine as ME
    sys.modules.setdefault("MinkowskiEngine", ME)
    sys.modules.setdefault("MinkowskiEngine.MinkowskiFunctional", ME.MinkowskiFunctional)
    sys.modules.setdefault("MinkowskiEngine.utils", ME.utils)
    return ME
