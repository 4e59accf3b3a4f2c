from .deform_conv3d import (DeformConv3d, DeformConvFunction3D, DeformConvBf16Function3D, ModulatedDeformConv3d, deform_conv3d,
                            deform_conv3d_bf16)

__all__ = ['DeformConv3d', 'DeformConvFunction3D', 'DeformConvBf16Function3D', 'ModulatedDeformConv3d', 'deform_conv3d',
           'deform_conv3d_bf16']
