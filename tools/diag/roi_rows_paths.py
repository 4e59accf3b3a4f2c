import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import mr3d_amd
from mr3d_amd import _hip
from oracle import binding as orc
dev = torch.device('cuda:0')
for P in ((3, 7, 7), (10, 14, 14)):
    rng = np.random.default_rng(100 + P[0])
    N, C, D, H, W = 2, 128, 12, 40, 300
    feat = torch.from_numpy(rng.normal(size=(N, C, D, H, W)).astype(np.float32)).to(dev).contiguous(memory_format=torch.channels_last_3d)
    rois = np.array([[0, 8, 8, 19, 40, 2, 10], [1, 8, 8, 60, 150, 1, 20], [0, 100, 20, 200, 60, 4, 9], [1, 300, 10, 520, 30, 0, 6],
                     [0, 100, 40, 700, 70, 3, 8], [1, 40, 2, 60, 158, 0, 23], [0, -30, -20, 50, 30, -6, 8], [1, 1100, 120, 1250, 170, 15, 40],
                     [0, 10, 10, 10, 10, 3, 3], [1, 30, 30, 20, 20, 5, 2]], np.float32)
    exp = orc.roi_align3d_forward(feat.cpu().numpy(), rois, *P, 0.25, 0.5, 2)
    for form in (0, 1, 4, 5, 6):
        out = torch.zeros(exp.shape, device=dev)
        _hip.lib().mr3d_tuning_set(7, form)
        _hip.roi_align3d_forward(feat, torch.from_numpy(rois).to(dev), *P, 0.25, 0.5, 2, out)
        torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - exp).reshape(len(rois), -1).max(1)
        chb = np.abs(out.cpu().numpy() - exp).reshape(len(rois), 2, 64, -1).max((2, 3))
        print(P, 'form', form, ' '.join(f'{e:.1e}' for e in err), '| by channel half roi4:', chb[4])
