"""Drop-in shim: `import eval_parts` resolves to the MI355X implementation.
Put this directory (and the repo root) on PYTHONPATH ahead of the reference's sources."""
from kg_instance_segmentation_amd.eval_parts import *  # noqa: F401,F403
from kg_instance_segmentation_amd import eval_parts as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
