from .deform_conv3d import DeformConv3d, DeformConvFunction3D, ModulatedDeformConv3d, deform_conv3d

__all__ = ['DeformConv3d', 'DeformConvFunction3D', 'ModulatedDeformConv3d', 'deform_conv3d']
