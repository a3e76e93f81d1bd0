"""OPT-IN shim: `import preprocessing` resolves to the GPU ground-truth generator.
Not on the default shim path: the reference calls `preprocessing.get_ground_truth` inside `DataLoader` worker processes
(dataset_base.py:94-97, train.py --workers 4), which cannot touch the GPU.  `dropin/run.py` adds this directory only when
KG_GPU_GT=1 and then forces `num_workers=0`."""
from kg_instance_segmentation_amd.preprocessing import *  # noqa: F401,F403
from kg_instance_segmentation_amd import preprocessing as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
