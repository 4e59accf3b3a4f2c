"""Device-side input preparation (the data-dependent half of mmdet/datasets/coco_3d_2scales.py's prepare_train_img)."""
from .volume import prepare_volume, IMG_NORM_CFG
from .pipeline import RandomCrop3D, prepare_train_sample, load_npy_volume

__all__ = ['prepare_volume', 'IMG_NORM_CFG', 'RandomCrop3D', 'prepare_train_sample', 'load_npy_volume']
