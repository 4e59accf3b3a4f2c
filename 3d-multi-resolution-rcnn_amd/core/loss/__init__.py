from .losses import (weighted_binary_cross_entropy, weighted_smoothl1, weighted_cross_entropy, smooth_l1_loss, accuracy,
                     mask_cross_entropy)

__all__ = ['weighted_binary_cross_entropy', 'weighted_smoothl1', 'weighted_cross_entropy', 'smooth_l1_loss', 'accuracy',
           'mask_cross_entropy']
