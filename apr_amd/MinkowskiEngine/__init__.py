"""MinkowskiEngine-compatible operator subset backed by libapr_hip.so (gfx950).

Exactly the surface FCGF_APR touches; call shapes follow the
reference call sites:
  `ME.MinkowskiNetwork(D)`                         FCGF_APR/model/resunet.py:10,25
  `ME.MinkowskiConvolution(...)`                   resunet.py:31-78,122-140; residual_block.py:23-33
  `ME.MinkowskiConvolutionTranspose(...)`          resunet.py:83-117
  `ME.MinkowskiBatchNorm / MinkowskiInstanceNorm`  model/common.py:6-8
  `ME.MinkowskiFunctional.relu`                    resunet.py:146
  `ME.cat`                                         resunet.py:168
  `ME.SparseTensor(F, coordinates=C[, device=])`   scripts/test_apr.py:132-137, util/misc.py:91
  `ME.SparseTensor(F, coordinate_map_key=, coordinate_manager=)`   resunet.py:188-191
  `ME.utils.sparse_quantize / sparse_collate / batched_coordinates`
        lib/complement_data_loader.py:671-674,1310-1311; util/misc.py:80-83

`apr_amd.install_shims()` registers this package as `MinkowskiEngine` in
sys.modules so `import MinkowskiEngine as ME` in the reference's files resolves
here.  Inference (no_grad / eval) runs the fused HIP plan; when autograd is recording the same modules
run forward and backward through `ops.SparseConvFunction`.  GPU tensors only.
"""
from .core import (CoordinateManager, CoordinateMapKey, SparseTensor, MinkowskiNetwork,  # noqa: F401
                   MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm,
                   MinkowskiInstanceNorm, MinkowskiReLU, cat)
from . import MinkowskiFunctional  # noqa: F401
from . import utils  # noqa: F401

__version__ = "0.5.4+apr_amd"
