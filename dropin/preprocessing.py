"""Drop-in shim: `import preprocessing` resolves to the MI355X implementation.
Put this directory (and the repo root) on PYTHONPATH ahead of the reference's sources."""
from kg_instance_segmentation_amd.preprocessing import *  # noqa: F401,F403
from kg_instance_segmentation_amd import preprocessing as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
