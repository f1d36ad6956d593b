"""`MinkowskiEngine.MinkowskiFunctional` subset (`MEF.relu`, FCGF_APR/model/resunet.py:146)."""
from .core import relu  # noqa: F401
