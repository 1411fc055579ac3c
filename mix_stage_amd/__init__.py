"""mix_stage_amd -- MI355X-native (gfx950) implementation of the Mix-StAGE audio->pose conditional-mixture
GAN forward/backward path of chahuja/mix-stage, behind the reference's own nn.Module API.

The arithmetic lives in libmixstage_hip.so (hand-written HIP, C-ABI in include/mixstage.h); this package is
the host-side mirror of src/model/{layers,joint_late_cluster_soft_style,speech2gesture,gan}.py.
"""
from .layers import (AudioEncoder, ClusterClassify, ConvNormRelu, Curriculum, EmbLin, Group, PoseEncoder,  # noqa: F401
                     PoseStyleEncoder, TextEncoder1D, UNet1D, compute_dtype, set_compute_dtype, set_inference_folding)
from .speech2gesture import Speech2Gesture_D  # noqa: F401
from .joint_late_cluster_soft_style import JointLateClusterSoftStyle4_D, JointLateClusterSoftStyle4_G  # noqa: F401
from .gan import GAN  # noqa: F401

__all__ = ['ConvNormRelu', 'UNet1D', 'AudioEncoder', 'PoseEncoder', 'PoseStyleEncoder', 'TextEncoder1D',
           'ClusterClassify', 'Group', 'EmbLin', 'Curriculum', 'JointLateClusterSoftStyle4_G',
           'JointLateClusterSoftStyle4_D', 'Speech2Gesture_D', 'GAN', 'set_compute_dtype', 'set_inference_folding',
           'compute_dtype']
