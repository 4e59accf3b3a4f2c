"""RoIAlign3D forward of the bbox extractor shape against the oracle, with the first mismatching bins printed (development aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mr3d_amd
from mr3d_amd import _hip
from mr3d_amd.mmdet_ops import RoIAlign3D
from oracle import binding as orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import ref_impls as ri
dev = torch.device('cuda:0')
rng = np.random.default_rng(2)
N, C, D, H, W = 1, 64, 40, 128, 128
rois = ri.rand_rois(rng, 512, N, (512, 512, 80))
r = torch.from_numpy(rois).to(dev)
layer = RoIAlign3D(7, 3, 0.25, 0.5, sample_num=2)
f = torch.randn((N, C, D, H, W), device=dev).contiguous(memory_format=torch.channels_last_3d)
exp = orc.roi_align3d_forward(f.cpu().numpy(), rois[:64], 3, 7, 7, 0.25, 0.5, 2)
a = layer(f, r)[:64].cpu().numpy()
err = np.abs(a - exp)
bad = np.argwhere(err > 1e-5 + 1e-4 * np.abs(exp))
print('max abs err', err.max(), 'bad count', len(bad), 'first bad', bad[:3].tolist(), flush=True)
if len(bad):
    rr = bad[0][0]
    print('  roi', rois[rr], 'bad pd set', sorted(set(bad[bad[:, 0] == rr][:, 2].tolist())), 'ph', sorted(set(bad[bad[:, 0] == rr][:, 3].tolist())))
