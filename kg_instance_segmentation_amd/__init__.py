"""kg_instance_segmentation_amd -- MI355X-native (gfx950) implementation of KGnet's hot path.

Drop-in modules mirroring the reference's flat module surface:
    KGnet, loss, seg_loss, postprocessing, nms, config
backed by libkgnet_hip.so (hand-written HIP kernels behind the C ABI of include/kgnet_hip.h).
"""
__version__ = "0.1.0"
