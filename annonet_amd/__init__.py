"""annonet_amd — MI355X-native (gfx950) implementation of annonet's hot path: the segmentation net's training step and
tiled sliding-window inference, behind the reference's NetPimpl / tiling / annonet_infer interface.

The compute lives in annonet_amd/lib/libannonet_hip.so (hand-written HIP, C ABI in include/annonet_hip.h); this package
is the thin host-side mirror used by tests and bench.py.  There is no CPU fallback.
"""
from ._lib import ANH_BF16, ANH_FP32, LABEL_IGNORE, AnnonetHipError, build, lib  # noqa: F401
from .netpimpl import (Dataset, RuntimeNet, TrainingNet, argmax_device, annonet_infer, annonet_infer_device, count_steps_without_decrease, dnn_envelope_pack, dnn_envelope_unpack,  # noqa: F401
                       ignore_large_nonzero_regions,
                       net_config, net_layers,
                       op_conv_backward_data, op_conv_backward_data_bn, op_conv_backward_filter, op_conv_backward_filter_bn,
                       op_conv_forward, op_conv_forward_stats, outpaint,
                       random_rect_containing_point, set_devices, set_weights, shard_range, cross_replica_overlaps, tiling)
