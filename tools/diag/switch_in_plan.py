"""Do different-kernel transitions cost time inside a plan run?  Two records of the bf16 forward plan (a layer3 1x1x1 conv and a layer3
3x3x3 conv), N copies each, issued grouped (A..A B..B) and alternating (A B A B ..); same work either way."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import mr3d_amd
from mr3d_amd import _hip
from mr3d_amd.detector import ExtractFeat3D
dev = torch.device('cuda:0'); torch.manual_seed(0)
dt = torch.float32 if 'f32' in sys.argv else torch.bfloat16
x = torch.randn(1, 3, 80, 512, 512, device=dev)
eng = ExtractFeat3D(device=dev, dtype=dt)
eng.forward_features(x)
plan = eng._plan_for(x.shape)
ops = plan.fwd_ops
def positions(o): return o.i[0] * o.i[1] * o.i[2] * o.i[3]
a = next(o for o in ops if o.kind == _hip.OP_CONV_FWD and o.i[6] == 1 and o.i[4] == 256 and o.i[5] == 64)     # layer3 conv1 256 -> 64
b = next(o for o in ops if o.kind == _hip.OP_CONV_FWD and o.i[6] == 3 and o.i[4] == 64 and o.i[5] == 64 and positions(o) < 32768)
print('A: 1x1x1', a.i[4], '->', a.i[5], 'on', positions(a), 'positions;  B: 3x3x3', b.i[4], '->', b.i[5], 'on', positions(b))
N = 40
def timed(plan_, reps=20):
    for _ in range(3): plan_.run(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): plan_.run(dev)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
pa, pb = _hip.Plan([a] * (2 * N)), _hip.Plan([b] * (2 * N))
grouped = _hip.Plan([a] * N + [b] * N)
alt = _hip.Plan([a, b] * N)
ta, tb = timed(pa) / (2 * N), timed(pb) / (2 * N)
tg, tl = timed(grouped), timed(alt)
print(f'A alone {ta:.2f} us per launch, B alone {tb:.2f};  {N} A + {N} B grouped {tg:.1f} us, alternating {tl:.1f} us  ->  {(tl - tg) / (2 * N - 2):.2f} us per extra transition')
