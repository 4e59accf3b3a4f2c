#!/usr/bin/env python3
"""Headline benchmark: training iterations/sec of the ResNet3D-50 + FPN3D forward+backward hot path on
synthetic 1x3x80x512x512 fp32 volumes (BASELINE.json configs[1]), one volume per GPU (data parallel,
weak scaling).  For N > 1 every step ends with the reference's gradient exchange: one SUM all-reduce of the
flat fp32 gradient arena over RCCL followed by 1/world (mmdet/core/utils/dist_utils.py:9-41).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is measured live: HIP events (torch.cuda.Event on the stream the
kernels are launched on) bracket every launch of the dominant kernel inside the timed steps.
`cpu_baseline` times the oracle's plain-PyTorch CPU restatement of the same graph on the host cores
(rank 0, N == 1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic FLOPs of one forward+backward of backbone+FPN on a 1x3x80x512x512 volume
FLOP_FWD_BWD_512 = 1124.38e9
PEAK_FP32_MATRIX = 157.3e12          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 peak
PEAK_BF16_MATRIX = 2516.6e12         # dense bf16 MFMA (16x the fp32 matrix rate; no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--vol', type=int, nargs=3, default=[80, 512, 512], metavar=('D', 'H', 'W'),
                    help='volume size (default = the configuration the metric is quoted on)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-aux', action='store_true', help='skip the NMS3D / RoIAlign3D / two-stage-step side measurements '
                                                          '(profiling runs: only the headline step is traced)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the cpu_baseline sample')
    ap.add_argument('--per-kernel', action='store_true', help='also print a per-kernel time table to stderr')
    ap.add_argument('--per-op', action='store_true', help='also print every plan op with its time to stderr')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help='storage type of the activations: f32 = the headline metric (the reference arithmetic); bf16 = BASELINE '
                         'configs[4] (bf16 maps in HBM, bf16 MFMA with fp32 accumulation)')
    ap.add_argument('--config', choices=['backbone', 'two_stage'], default='backbone',
                    help="backbone = BASELINE configs[1] (the headline metric); two_stage = configs[2]/[3]: the whole "
                         "single-pathway training step with ONE all-reduce of the detector's 26 M-element gradient arena")
    ap.add_argument('--serial-allreduce', action='store_true',
                    help='N > 1: ONE all-reduce after the backward pass (the reference order) instead of the overlapped exchange')
    ap.add_argument('--backbone-only-allreduce', action='store_true',
                    help='N > 1, --config backbone: exchange only the 13.7 MB backbone+FPN gradients instead of the whole '
                         "detector's arena")
    ap.add_argument('--tune', default='', help='measurement knobs, key=value[,key=value...] for mr3d_tuning_set (include/mr3d.h); all 0 by default')
    return ap.parse_args()


def cpu_baseline(seconds):
    """oracle/torch_ref_model.py (pinned to the reference by tests/golden/backbone_fpn.npz) forward+backward on the
    host cores, on the reference's own CPU-runnable case (configs[0]: 1x3x32x128x128), scaled to 512x512x80
    volumes by the algorithmic FLOP ratio 432.34/10.81 = 40.0."""
    from oracle.torch_ref_model import RefExtractFeat
    # threads = the CPUs this process may really use: affinity mask, clipped by the cgroup CPU quota (the GPU box
    # shows 256 CPUs but grants 16: 256 oneDNN threads took 85 s per iteration there, 16 take 66 ms)
    avail = _host_cores()
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = RefExtractFeat()
    x = torch.randn(1, 3, 32, 128, 128)

    def it():
        m.zero_grad()
        outs = m(x)
        sum(o.sum() for o in outs).backward()

    it()                                   # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.perf_counter()
    while True:
        it()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    per_small = el / n
    scale = 432.34 / 10.81
    res = {'value': 1.0 / (per_small * scale), 'unit': 'iters/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'host_cpus': avail,
           'sample': f'{n} fwd+bwd iterations of the PyTorch-CPU (oneDNN) restatement on 1x3x32x128x128 '
                     f'({per_small * 1e3:.1f} ms each), scaled x{scale:.1f} by algorithmic FLOPs to 512x512x80'}
    # BASELINE.md section 3, C1 "optionally 1x3x80x512x512 (one iteration)": the metric's own size, measured, next to the
    # scaled figure -- skipped when the scaled estimate says it would not fit the budget (one warm-up + one timed iteration)
    if per_small * scale * 2 <= max(seconds, 20.0):
        try:
            xl = torch.randn(1, 3, 80, 512, 512)

            def big():
                m.zero_grad()
                sum(o.sum() for o in m(xl)).backward()
            big()
            t0 = time.perf_counter()
            big()
            t_big = time.perf_counter() - t0
            res['full_size'] = {'value': 1.0 / t_big, 'unit': 'iters/s', 'seconds_per_iter': t_big,
                                'sample': 'ONE timed fwd+bwd iteration (after one warm-up) of the same restatement on 1x3x80x512x512'}
        except Exception as exc:          # host memory: the activations of the full-size graph need several GB
            res['full_size'] = {'error': repr(exc)}
    return res


def cpu_baseline_allreduce(seconds):
    """the reference's gradient exchange (SUM all-reduce + 1/world of ONE flat 25.98 M-element fp32
    buffer, mmdet/core/utils/dist_utils.py:22-28) over gloo between 2 / 4 / 8 local CPU ranks: tools/gloo_mean_bench.py started
    once per rank as a child process (the parent only waits).  A sanity baseline for the RCCL curve, not a target."""
    import subprocess
    numel = 25_980_000
    out = {'unit': 'ms per exchange', 'mbytes': numel * 4 / 1e6, 'kind': 'port', 'host_cpus': _host_cores(),
           'sample': 'gloo all_reduce(SUM) + div_(world) of a 103.9 MB fp32 buffer between local CPU ranks (one thread each), '
                     'mean of 3 timed repetitions after a warm-up'}
    script = os.path.join(ROOT, 'tools', 'gloo_mean_bench.py')
    t_start = time.perf_counter()
    for world in (2, 4, 8):
        if time.perf_counter() - t_start > 2 * seconds:
            out[f'ranks_{world}'] = None
            continue
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, script, str(numel), '3'], env=rank_env(r, world, port), stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True) for r in range(world)]
        val = None
        for r, p in enumerate(procs):
            try:
                so, _ = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                so = ''
            if r == 0:
                for line in so.splitlines():
                    if line.startswith('ms_per_exchange'):
                        val = round(float(line.split()[1]), 2)
        out[f'ranks_{world}'] = val
    return out


def _host_cores():
    """CPUs this process may really use: affinity mask clipped by the cgroup quota"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return avail


def nms_bench_inputs(g):
    """SURVEY 8(d) NMS3D inputs: per pyramid level N in {2000,2000,2000,1280,192} boxes = anchors of that level's size
    jittered by N(0, 0.25*size), distinct U(0,1) scores; host float32 arrays [n,7]"""
    out = []
    for n, stride, zs in zip((2000, 2000, 2000, 1280, 192), (4, 8, 16, 32, 64), (2, 4, 8, 16, 32)):
        size = torch.tensor([stride * 2.0, stride * 2.0, zs * 2.0])
        ctr = torch.rand(n, 3, generator=g) * torch.tensor([512.0, 512.0, 80.0])
        ctr = ctr + torch.randn(n, 3, generator=g) * 0.25 * size
        half = size * (1 + 0.25 * torch.randn(n, 3, generator=g)).abs() / 2
        boxes = torch.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                             ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1)
        out.append(torch.cat([boxes, torch.rand(n, 1, generator=g)], 1))
    return out


def roi_bench_rois(rng, n):
    """SURVEY 8(d) RoIAlign3D inputs: centre uniform, xy size log-uniform [4,128], z size log-uniform [2,40], clipped"""
    s_ = np.exp(rng.uniform(np.log([4, 4, 2]), np.log([128, 128, 40]), (n, 3)))
    c = rng.uniform([0, 0, 0], [512, 512, 80], (n, 3))
    lo, hi = np.clip(c - s_ / 2, 0, [511, 511, 79]), np.clip(c + s_ / 2, 0, [511, 511, 79])
    return np.stack([np.zeros(n), lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1).astype(np.float32)


ROI_LEVELS = ((40, 128, 4, 2), (20, 64, 8, 4), (10, 32, 16, 8), (5, 16, 32, 16))      # (d, h=w, xy stride, z stride) of P2..P5


def roi_levels_of(rois, finest_scale=56, num_levels=4):
    """SingleRoIExtractor.map_roi_levels (single_level.py:73-81): sqrt of the VOLUME, floor(log2(scale/56 + 1e-6))"""
    sc = np.sqrt((rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1) * (rois[:, 6] - rois[:, 5] + 1))
    return np.clip(np.floor(np.log2(sc / finest_scale + 1e-6)), 0, num_levels - 1).astype(int)


def roialign_lb_bytes(rois, C, PD, PH, PW, sample_num=2):
    """LB = 4*R*C*PD*PH*PW (output written once) + 4*C*sum_r V_r (every distinct feature voxel under a
    roi's sample footprint read once) + 28*R (the rois), V_r = prod_axis(hi - lo + 1) with lo = floor(clamp(first
    sample)), hi = min(ceil(last sample), dim - 1) on the roi's pyramid level.  Returns (total, write, read) bytes."""
    lv = roi_levels_of(rois)
    R = rois.shape[0]
    vox = 0
    for r in range(R):
        d, hw, st, zst = ROI_LEVELS[lv[r]]
        v = 1
        for (a0, a1, scale, dim, P) in ((rois[r, 1], rois[r, 3], 1.0 / st, hw, PW), (rois[r, 2], rois[r, 4], 1.0 / st, hw, PH),
                                        (rois[r, 5], rois[r, 6], 1.0 / zst, d, PD)):
            start, end = a0 * scale, (a1 + 1) * scale
            binsz = max(end - start, 0.0) / P
            first = start + 0.5 * binsz / sample_num
            last = start + (P - 1) * binsz + (sample_num - 0.5) * binsz / sample_num
            lo = int(np.floor(min(max(first, 0.0), dim - 1)))
            hi = int(min(np.ceil(min(max(last, 0.0), dim - 1)), dim - 1))
            v *= max(hi - lo + 1, 1)
        vox += v
    write = 4 * R * C * PD * PH * PW
    read = 4 * C * vox
    return write + read + 28 * R, write, read


def cpu_baseline_ops(seconds):
    """BASELINE.md section 3, C2 and C3, timed on this box's host cores (bounded samples):
      C2  the numpy restatement of the reference's only CPU NMS3D (coco_utils.py:245-282; oracle/nms_ref.py, pinned by
          tests/golden/nms3d.npz) on the 5 per-level NMS bench inputs -> us per volume (x2 pathways);
      C3  the C restatement of RoIAlign3D forward (oracle/mr3d_oracle.c, roi_align_kernel.cu:214-291) on the bbox bench
          case, level by level like SingleRoIExtractor, single-thread and OpenMP."""
    from oracle import binding as orc
    from oracle.nms_ref import nms3d_numpy
    out = {}
    g = torch.Generator(device='cpu').manual_seed(0)
    dets = [d.numpy() for d in nms_bench_inputs(g)]
    t0, reps = time.perf_counter(), 0
    while reps < 1 or (time.perf_counter() - t0 < seconds / 3 and reps < 20):
        for d in dets:
            nms3d_numpy(d, 0.7)
        reps += 1
    per_pass = (time.perf_counter() - t0) / reps
    pairs = sum(d.shape[0] * (d.shape[0] - 1) // 2 for d in dets)
    out['nms3d'] = {'value': 2 * per_pass * 1e6, 'unit': 'us/volume', 'cores': 1, 'kind': 'port',
                    'pairs_per_s': pairs / per_pass,
                    'sample': f'{reps} passes of oracle/nms_ref.py (numpy, float64) over the 5 level inputs '
                              f'N=2000,2000,2000,1280,192 thr 0.7; x2 pathways'}
    rng = np.random.default_rng(0)
    rois = roi_bench_rois(rng, 512)
    lv = roi_levels_of(rois)
    rs = np.random.default_rng(1)
    feats = [rs.standard_normal((1, 64, d, hw, hw), dtype=np.float32) for d, hw, _, _ in ROI_LEVELS]
    cores = _host_cores()

    def run():
        for k, (d, hw, st, zst) in enumerate(ROI_LEVELS):
            r = rois[lv == k]
            if len(r):
                orc.roi_align3d_forward(feats[k], r, 3, 7, 7, 1.0 / st, 1.0 / zst, 2)
    for tag, nthreads in (('1thread', 1), ('openmp', min(cores, 32))):
        orc.set_num_threads(nthreads)
        run()
        t0, reps = time.perf_counter(), 0
        while reps < 1 or (time.perf_counter() - t0 < seconds / 3 and reps < 50):
            run()
            reps += 1
        per = (time.perf_counter() - t0) / reps
        out[f'roialign3d_{tag}'] = {'value': per * 1e6, 'unit': 'us/call', 'cores': nthreads, 'kind': 'port',
                                    'sample': f'{reps} forward passes of the C restatement, bbox case: 512 rois x 64 ch x (3,7,7), '
                                              'sample_num 2, P2..P5 of a 512x512x80 volume, one call per level'}
    return out


def ops_microbench(dev):
    """second half of BASELINE.json's metric: RoIAlign3D + NMS3D microseconds per volume (SURVEY.md section 8d inputs)"""
    import numpy as np
    from mr3d_amd.mmdet_ops import RoIAlign3D
    from mr3d_amd.mmdet_ops.nms import nms_cuda
    g = torch.Generator(device='cpu').manual_seed(0)
    out = {}

    def timed(fn, iters=20):
        # the median of three windows of `iters` calls (one window now and then comes out 1.5x long: 86 instead of 58 us for the
        # same kernel whose rocprofv3 average is 57.9 -- another process's burst, a clock step)
        fn(); torch.cuda.synchronize()
        wins = []
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record(); torch.cuda.synchronize()
            wins.append(s.elapsed_time(e) * 1e3 / iters)
        return sorted(wins)[1]

    # NMS3D: 5 pyramid levels x 2 pathways per volume; per level N in {2000,2000,2000,1280,192}, thr 0.7
    tot = 0.0
    level_dets = [d.to(dev) for d in nms_bench_inputs(g)]
    for dets in level_dets:
        tot += timed(lambda: nms_cuda.nms_3d(dets, 0.7))
    out['nms3d_unbatched_us_per_vol'] = 2 * tot                  # 10 separate calls, as the reference issues them
    from mr3d_amd import _hip as _h
    # the product path (detector/rpn_proposals.py): ONE batched call per pathway whose kept lists and counts stay on the
    # device for the merge kernel -- nothing is read back, so back-to-back calls measure the device time of the three kernels
    caps = [int(d.size(0)) for d in level_dets]
    cat = torch.cat(level_dets, 0).contiguous()
    out['nms3d_one_pathway_us'] = timed(lambda: _h.nms3d_batched_dev(cat, caps, None, 0.7), iters=50)
    # a volume = 2 pathways x 5 levels = 10 independent sets; the dual-pathway detector hands all ten to ONE launch sequence
    # (detector/heads.rpn_get_bboxes_pair -> rpn_proposals.get_bboxes_pair_device); the second pathway's sets are a second draw
    level_dets_2 = [d.to(dev) for d in nms_bench_inputs(g)]
    cat10 = torch.cat(level_dets + level_dets_2, 0).contiguous()
    caps10 = caps + [int(d.size(0)) for d in level_dets_2]
    out['nms3d_us_per_vol'] = timed(lambda: _h.nms3d_batched_dev(cat10, caps10, None, 0.7), iters=50)
    # the same call followed by a read-back of the kept counts (what mmdet's nms() wrapper needs to size its result)
    out['nms3d_with_readback_us_per_vol'] = 2 * timed(lambda: _h.nms3d_batched(level_dets, 0.7))
    # algorithmic work: N(N-1)/2 IoU evaluations per level; latency-bound, so pairs/s is the figure of merit
    pairs = sum(d.size(0) * (d.size(0) - 1) // 2 for d in level_dets + level_dets_2)
    out['nms3d_pairs_per_vol'] = pairs
    out['nms3d_gpairs_per_s'] = pairs / out['nms3d_us_per_vol'] * 1e-3
    # RoIAlign3D through SingleRoIExtractor over P2..P5: 512 rois x 64 ch x (3,7,7) and 128 rois x
    # (10,14,14); roi sizes log-uniform [4,128] px in xy and [2,40] in z, mapped to levels by the reference's rule
    from mr3d_amd.detector import SingleRoIExtractor
    feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
             for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
    rng = np.random.default_rng(0)

    for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
        ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64,
                                 [4, 8, 16, 32], [2, 4, 8, 16])
        r_host = roi_bench_rois(rng, n)
        r = torch.from_numpy(r_host).to(dev)
        out[f'roialign3d_{tag}_fwd_us'] = timed(lambda: ext([f.detach() for f in feats], r))
        # HBM roofline of the forward: LB bytes / time against 8 TB/s
        lb, wr, rd = roialign_lb_bytes(r_host, 64, od, osz, osz)
        out[f'roialign3d_{tag}_fwd_lb_mbytes'] = lb / 1e6
        out[f'roialign3d_{tag}_fwd_lb_read_mbytes'] = rd / 1e6
        out[f'roialign3d_{tag}_fwd_tbytes_per_s'] = lb / out[f'roialign3d_{tag}_fwd_us'] * 1e-6
        out[f'roialign3d_{tag}_fwd_hbm_frac'] = lb / out[f'roialign3d_{tag}_fwd_us'] * 1e-6 / 8.0
        top = torch.randn(n, 64, od, osz, osz, device=dev)

        def fb():
            for f in feats:
                f.grad = None
            ext(feats, r).backward(top)
        out[f'roialign3d_{tag}_fwd_bwd_us'] = timed(fb, iters=5)          # through autograd: + zero-fill of the 4 level gradients
        # the backward launch alone (mr3d_roi_align3d_backward_ml into the same pre-zeroed buffers: adds accumulate, timing only).
        # backward LB = read top_grad + read-modify-write of the touched voxels (2x the forward's read term)
        grads = [torch.zeros_like(f) for f in feats]
        scales, scales_d = [1.0 / s_ for s_ in (4, 8, 16, 32)], [1.0 / s_ for s_ in (2, 4, 8, 16)]
        out[f'roialign3d_{tag}_bwd_us'] = timed(lambda: _h.roi_align3d_backward_ml(top, r, grads, od, osz, osz, scales, scales_d, 2, 56.0))
        lb_b = wr + 2 * rd
        out[f'roialign3d_{tag}_bwd_lb_mbytes'] = lb_b / 1e6
        out[f'roialign3d_{tag}_bwd_hbm_frac'] = lb_b / out[f'roialign3d_{tag}_bwd_us'] * 1e-6 / 8.0
    return out



def rank_census(dev, world):
    """{'dist_world_size', 'rccl_ranks'}: what torch.distributed reports and the number of ranks a SUM all-reduce of ones
    over RCCL really counted (every rank calls this)"""
    import torch.distributed as dist
    if not dist.is_initialized():
        return {'dist_world_size': 1, 'rccl_ranks': 1}
    one = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(one)
    return {'dist_world_size': dist.get_world_size(), 'rccl_ranks': int(round(float(one.item())))}


def emit_json_line(obj):
    """the ONE JSON line, as the LAST line of stdout: RCCL writes its version banner through C stdio (block-buffered on a pipe,
    flushed at exit -- after Python's own line); flush C stdio first, then print"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(obj), flush=True)


def two_stage_step_ms(vol, steps=20, two_scales=False, static_mask_batch=False, bf16=False, dcn_neck=False):
    """BASELINE configs[2]: the whole single-pathway training step (backbone+FPN, RPN + loss + proposals/NMS3D,
    assign/sample 512, RoIAlign3D + bbox head, RoIAlign3D + mask head, all losses, backward) on one volume, timed by
    tools/two_stage_bench.py in a CHILD process (a fresh allocator / stream state, as a user would run it)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'two_stage_bench.py'), '--steps', str(steps), '--warmup', '3', '--json',
           '--vol', ','.join(str(v) for v in vol)] + (['--two-scales'] if two_scales else []) + \
          (['--static-mask-batch'] if static_mask_batch else []) + (['--bf16'] if bf16 else []) + (['--dcn-neck'] if dcn_neck else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)['two_stage_step_ms']
    raise RuntimeError('two_stage_bench.py failed: ' + out.stderr[-400:])


def two_stage_main(a, rank, world, dev):
    """--config two_stage: BASELINE configs[2] (N = 1) / configs[3] (N > 1, one volume per GPU): the whole single-pathway
    training step -- backbone+FPN, RPN + loss + proposals (NMS3D), assign/sample 512, RoIAlign3D + bbox head, RoIAlign3D +
    mask head, losses, backward -- then ONE SUM all-reduce of the detector's flat gradient arena (dist_utils.DetectorArena),
    L2 clip and SGD, i.e. dist_utils.train_step, the reference's DistOptimizerHook order."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from two_stage_bench import synthetic_gt
    from mr3d_amd.detector import TwoStageDetector3D
    from mr3d_amd.dist_utils import ArenaSGD, DetectorArena, broadcast_params, train_step
    D, H, W = a.vol
    torch.manual_seed(0)
    bf = getattr(a, 'dtype', 'f32') == 'bf16'           # BASELINE configs[4] storage type on the configs[2] / [3] step
    det = TwoStageDetector3D(device=dev, dtype=torch.bfloat16 if bf else torch.float32)
    det.extract.init_weights()
    arena = DetectorArena(det)
    if world > 1:
        broadcast_params(arena.flat_params)
    opt = ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=1e-4)     # configs/...py:235
    rng = np.random.default_rng(100 + rank)
    img = torch.randn(1, 3, D, H, W, device=dev)
    gt, masks = synthetic_gt(rng, 12, D, H, W, dev)
    gl = torch.ones(12, dtype=torch.long, device=dev)
    meta = [dict(img_shape=(H, W, 3, D), pad_shape=(H, W, 3, D), ori_shape=(H, W, 3))]
    batch = ((img, meta, [gt], [gl]), dict(gt_masks=[masks]))
    for _ in range(a.warmup):
        train_step(det, arena, opt, batch, world, freeze_gc=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        train_step(det, arena, opt, batch, world, freeze_gc=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    census = rank_census(dev, world)
    if rank == 0:
        ms = el * 1e3 / a.steps
        # algorithmic FLOPs of the step: backbone+FPN fwd+bwd 1124.38 + RPN head 3 x 165.8; roi heads depend on
        # the sampled batch and are left out of the fraction
        flops = (FLOP_FWD_BWD_512 + 3 * 165.8e9) * (D * H * W) / (80.0 * 512 * 512)
        emit_json_line({
            'metric': 'train iters/sec on 512x512x80 vols (full two-stage step: RPN3D + RoIAlign3D + NMS3D + bbox/mask heads, ' +
                      ('bf16 storage / fp32 accumulate)' if bf else 'fp32)'),
            'value': world * a.steps / el, 'unit': 'iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if bf else 'f32',
            'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[{2 if world == 1 else 3}]: 1x3x{D}x{H}x{W} fp32 volume per GPU, whole '
                                   'single-pathway training step + SGD' +
                                   ('' if world == 1 else f' + ONE all-reduce of the {arena.flat_grads.numel() * 4 / 1e6:.1f} MB detector arena over RCCL'),
                       'global_batch': world, 'parallelism': f'dp{world}'},
            'roofline': {'bound': 'mfma', 'kernel': 'whole step (backbone+FPN+RPN convolutions)', 'achieved': flops / (ms * 1e-3) / 1e12,
                         'peak': (PEAK_BF16_MATRIX if bf else PEAK_FP32_MATRIX) / 1e12, 'unit': 'TFLOP/s',
                         'frac': flops / (ms * 1e-3) / (PEAK_BF16_MATRIX if bf else PEAK_FP32_MATRIX), 'traffic': None},
            'arena_mbytes': arena.flat_grads.numel() * 4 / 1e6, **census})
    if dist.is_initialized():
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    """environment of one rank of a single-node job: what `python -m torch.distributed.launch --nproc_per_node=N`
    (tools/dist_train.sh:5 of the reference) hands each child"""
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0',
                'MR3D_BENCH_SPAWNED': '1'})
    return env


def launch_ranks(n, argv=None, device_count=None, script=None):
    """`bench.py --gpus N` outside a launcher: start N ranks of this script, one per GPU (the reference's
    tools/dist_train.sh:5 / mmdet/apis/train.py:85-100 shape).  The parent never touches a GPU: it only counts devices
    (torch.cuda.device_count() does not initialise HIP), starts the children, forwards their output and returns the worst
    exit code.  Fewer than N devices is an error, never a silent N=1 run."""
    import subprocess
    have = torch.cuda.device_count() if device_count is None else device_count
    if have < n:
        print(f'bench.py: --gpus {n} asked for {n} ranks but this node shows {have} GPU(s); refusing to run a smaller job '
              f'under the name of a {n}-GPU one', file=sys.stderr)
        return 2
    port = _free_port()
    argv = sys.argv[1:] if argv is None else argv
    procs = [subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=rank_env(r, n, port))
             for r in range(n)]
    worst = 0
    try:
        for p in procs:
            rc = p.wait()
            if rc != 0:
                worst = worst or (rc if rc > 0 else 1)
                for q in procs:              # a dead rank leaves the others in a collective: end exactly the PIDs started here
                    if q.poll() is None:
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        worst = 130
    return worst


def main():
    a = parse()
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} disagrees with WORLD_SIZE={world} of the launcher; the JSON line must '
                         'name the rank count that really ran')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP hot path has no CPU fallback')
    if torch.cuda.device_count() <= local:
        raise SystemExit(f'bench.py: rank {rank} has LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    force_dist = os.environ.get('MR3D_BENCH_FORCE_DIST') == '1'      # exercise the RCCL path on a 1-GPU box
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)          # backend "nccl" is RCCL on ROCm

    import mr3d_amd
    from mr3d_amd import _hip
    for kv in filter(None, a.tune.split(',')):
        _hip.lib().mr3d_tuning_set(*(int(v) for v in kv.split('=')))
    from mr3d_amd.detector import ExtractFeat3D
    from mr3d_amd.dist_utils import allreduce_flat_grads, allreduce_overlapped, broadcast_params

    torch.manual_seed(1234 + rank)
    D, H, W = a.vol
    if a.config == 'two_stage':
        return two_stage_main(a, rank, world, dev)
    eng = ExtractFeat3D(device=dev, dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
    reduce_buf = eng.flat_grads
    if (world > 1 or force_dist) and not a.backbone_only_allreduce:
        # The reference exchanges ONE flat buffer holding every gradient of the detector (dist_utils.py:22-28): 25.98 M fp32
        # = 103.9 MB for the shipped model, of which backbone + FPN are the first 3.4 M.  This workload only
        # produces those; the engine is re-homed to the front of an arena of the WHOLE model's size and the all-reduce
        # runs on all of it (the head slots stay zero), so the message on the wire is the real one.
        n_model = -(-25_980_000 // 64) * 64
        storage = torch.zeros(n_model + eng.num_stats, dtype=torch.float32, device=dev)
        grads = torch.zeros(n_model, dtype=torch.float32, device=dev)
        eng.rehome(storage, grads, 0, n_model)
        reduce_buf = grads
    if world > 1:
        broadcast_params(eng.flat_params)                       # identical initial weights on all ranks (mmcv DDP init)
    x = torch.randn(1, 3, D, H, W, device=dev)
    outs = eng.forward_features(x)
    cots = [torch.randn(o.shape, device=dev).to(eng.act_dtype).contiguous(memory_format=torch.channels_last_3d) for o in outs]
    plan = eng._plan_for(x.shape)
    scale = (D * H * W) / (80.0 * 512 * 512)

    def step(marks_f=None, marks_b=None, split=False, exchange=True):
        """split=True: every marked record is its own mr3d_plan_run call on the main stream (per-op table of the warm-up);
        otherwise each pass is ONE call -- the production launch sequence, side stream included -- and the marked records
        carry HIP timing events on the stream they really run on (mr3d_plan_run_marked)"""
        eng.zero_grad()
        eng.bind_input(plan, x)                  # dense NCDHW fp32: read in place, no staging copy
        ev = plan.fwd.run(dev, marks_f) if (split or not marks_f) else plan.fwd.run_marked(marks_f, dev)
        eng.bind_grads(plan, cots)               # channels-last P2..P4 gradients are read in place, P5/P6 copied
        evb = plan.bwd.run(dev, marks_b) if (split or not marks_b) else plan.bwd.run_marked(marks_b, dev)
        # the exchange: everything behind the stem's parameters starts on a communication stream as soon as its gradient
        # finishing is done, underneath the stem weight gradient; the stem's 66 KB follow (dist_utils.allreduce_overlapped)
        if (world > 1 or force_dist) and exchange:
            if a.serial_allreduce:
                allreduce_flat_grads(reduce_buf, world) if world > 1 else dist.all_reduce(reduce_buf)
            else:
                allreduce_overlapped(reduce_buf, eng.stem_elems, world, force_single_rank=force_dist, early_elems=eng.early_elems,
                                     early_slot=eng.MARK_GRADS_EARLY)
        return ev, evb

    # ---- warmup; the last warmup step is bracketed op by op to find the dominant kernel
    for _ in range(max(a.warmup - 1, 0)):
        step()
    ev, evb = step(list(range(plan.fwd.n)), list(range(plan.bwd.n)), split=True)     # op-by-op calls: each runs on the main stream
    torch.cuda.synchronize()
    groups = {}
    for which, events, p in (('f', ev, plan.fwd), ('b', evb, plan.bwd)):
        for k, (s, e) in events.items():
            key, flops = _hip.op_kernel_key(p.array[k])
            gk = groups.setdefault(key, {'ms': 0.0, 'flops': 0.0, 'ops': [], 'n': 0})
            gk['ms'] += s.elapsed_time(e); gk['flops'] += flops; gk['n'] += 1; gk['ops'].append((which, k))
    dom = max(groups, key=lambda k: groups[k]['ms'])
    if a.per_op and rank == 0:
        for which, events, p in (('f', ev, plan.fwd), ('b', evb, plan.bwd)):
            for k in sorted(events):
                key, flops = _hip.op_kernel_key(p.array[k])
                ms = events[k][0].elapsed_time(events[k][1])
                i = p.array[k].i
                print(f'{which}{k:4d} {key:38s} dims={list(i[1:6])} {ms:8.3f} ms {flops / ms / 1e9 if ms > 0 else 0:7.2f} TF', file=sys.stderr)
    if a.per_kernel and rank == 0:
        tot = sum(v['ms'] for v in groups.values())
        for k, v in sorted(groups.items(), key=lambda kv: -kv[1]['ms']):
            tf = v['flops'] / v['ms'] / 1e9 if v['ms'] > 0 else 0
            print(f'{k:40s} n={v["n"]:3d} {v["ms"]:8.3f} ms {100 * v["ms"] / tot:5.1f}%  {tf:7.2f} TFLOP/s', file=sys.stderr)
        print(f'sum of per-op times {tot:.3f} ms', file=sys.stderr)
    # the next two kernels by time carry event pairs too (`roofline.next_kernels`): the dominant kernel alone says little about a step
    # whose three largest kernels take 18 / 13 / 9 % of it
    # (kernels whose MFMA issue factor the library reports: the forward / data-gradient convolutions and the stem kernels)
    def priced(key):
        kinds = {(plan.fwd if w_ == 'f' else plan.bwd).array[k].kind for w_, k in groups[key]['ops']}
        return groups[key]['flops'] > 0 and kinds <= {_hip.OP_CONV_FWD, _hip.OP_STEM_FWD, _hip.OP_STEM_WGRAD}
    top = sorted((k for k in groups if priced(k)), key=lambda k: -groups[k]['ms'])[:3]
    if dom not in top:
        top = [dom] + top[:2]
    # Inside the TIMED steps only the dominant kernel's launches carry event pairs: an event record is a packet of its own in the
    # kernel's queue, and the kernel behind it starts ~6 us later than back to back (the step timeline: 0 us between unmarked
    # main-queue kernels, 6-7 us on both sides of every marked one -- with the three groups' 29 launches marked that was 0.3 ms of the
    # 8.9 ms step).  The two groups behind it are measured the same way in a few extra steps AFTER the timed region.
    marked = [op for op in groups[dom]['ops']]
    marks_f = sorted(k for w_, k in marked if w_ == 'f')
    marks_b = sorted(k for w_, k in marked if w_ == 'b')
    marked_next = [op for k in top if k != dom for op in groups[k]['ops']]
    marks_next_f = sorted(k for w_, k in marked_next if w_ == 'f')
    marks_next_b = sorted(k for w_, k in marked_next if w_ == 'b')

    if os.environ.get('MR3D_BENCH_GRAPH') == '1':          # A/B: hipGraph replay (no per-kernel event brackets possible)
        eng.capture(plan)
        marks_f = marks_b = None
        step(); torch.cuda.synchronize()
    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dom_events = []
    for _ in range(a.steps):
        e1, e2 = step(marks_f, marks_b)
        dom_events.append((e1, e2))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    next_events = []
    if marks_f is not None:                                # (not under MR3D_BENCH_GRAPH: a replayed graph carries no event brackets)
        for _ in range(min(a.steps, 5)):
            next_events.append(step(marks_next_f, marks_next_b))
        torch.cuda.synchronize()
    census = rank_census(dev, world)
    exch = None
    if world > 1 or force_dist:
        # ---- the exchange, measured on its own (SURVEY 8d "All-reduce"; reference: mmdet/core/utils/dist_utils.py:22-28,51-57) --
        # every rank runs this block: (1) the same K' steps WITHOUT the collectives -> what the exchange adds to a step that overlaps
        # it (`ms_exposed`); (2) the in-place mean alone, 10 repetitions on an idle device -> `ms_standalone` and the bus bandwidth
        # 2 (n - 1) / n * bytes / t against the xGMI links; (3) one more full step, then every rank's gradient arena must carry the
        # same bits (all-gather of a 64-bit checksum): a mismatch fails the run with exit code 3
        from mr3d_amd.dist_utils import allreduce_bus_bandwidth, ranks_hold_identical, time_allreduce_standalone
        k2 = max(min(a.steps, 10), 1)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step(exchange=False)
        torch.cuda.synchronize()
        noex = torch.tensor([(time.perf_counter() - t1) / k2], device=dev, dtype=torch.float64)
        dist.all_reduce(noex, op=dist.ReduceOp.MAX)
        ms_noex = float(noex.item()) * 1e3
        nbytes = reduce_buf.numel() * 4
        sec = time_allreduce_standalone(reduce_buf, world, reps=10, sync=torch.cuda.synchronize)
        step()
        torch.cuda.synchronize()
        same, sums = ranks_hold_identical(reduce_buf)
        exch = {'bytes': nbytes, 'ms_standalone': sec * 1e3, 'ms_step_without_exchange': ms_noex,
                'ms_exposed': el * 1e3 / a.steps - ms_noex, **allreduce_bus_bandwidth(nbytes, sec, world),
                'mode': ('serial after backward' if a.serial_allreduce else
                         'two collectives, the first under the backward pass (dist_utils.allreduce_overlapped)'),
                'grads_identical_across_ranks': same, 'grad_checksums': [f'{v:016x}' for v in sums],
                'note': ('ms_exposed = step with the exchange - the same step without it (max over ranks, ' + str(k2) + ' extra steps); '
                         'busbw = 2 (n - 1) / n * bytes / ms_standalone; frac_of_xgmi against 7 x 153 GB/s, frac_of_one_link against 153 GB/s' +
                         ('' if world > 1 else '; ONE rank (MR3D_BENCH_FORCE_DIST): the collective moves nothing, the fields show the line a '
                                               'multi-GPU run prints'))}

    def committed_traffic(kernel_key):
        """HBM bytes per launch of `kernel_key` from the committed rocprofv3 PMC passes of this same command
        (profiles/<round>/traffic.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE doubled per the
        gfx950 note of MI355X_MICROARCH.md); None when no pass covers the kernel."""
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'traffic.json')), reverse=True):
            try:
                k = json.load(open(path))['kernels'].get(kernel_key)
            except (OSError, ValueError, KeyError):
                continue
            if k:
                return k['hbm_read_bytes_per_launch'] + k['hbm_write_bytes_per_launch']
        return None

    if rank == 0:
        def group_ms(key, events=None):
            ops = set(groups[key]['ops'])
            return sum(ev[k][0].elapsed_time(ev[k][1]) for e1, e2 in (dom_events if events is None else events)
                       for w_, ev in (('f', e1), ('b', e2)) for k in ev if (w_, k) in ops)

        def issue_factor_of(key):
            """MFMA FLOPs the launches of a group really ISSUE / their algorithmic FLOPs (Winograd instances issue less; the gather form
            of the stem weight gradient issues none: vector-ALU work)"""
            issued, var = 0.0, {}
            for which, k in groups[key]['ops']:
                op = (plan.fwd if which == 'f' else plan.bwd).array[k]
                _, fl = _hip.op_kernel_key(op)
                factor, tag = 1.0, 'direct'
                if op.kind == _hip.OP_CONV_FWD and op.i[25] == 0:
                    v = _hip.op_conv_variant(op)
                    factor = float(v.mfma_issue_factor)
                    tag = ('streaming pointwise' if v.streaming_pointwise else
                           ('winograd F(4,3) along W x F(2,3) along H' if v.winograd == 8 else
                            (f'winograd F({v.winograd},3) along W' if v.winograd else 'direct')) +
                           f' tile {v.tile_d}x{v.tile_h}x{v.tile_w} cout {v.cout_tile}')
                elif op.kind == _hip.OP_STEM_FWD and op.i[25] == 0:
                    # stem.hip: F(2,3) along depth (10 MFMA groups per depth pair instead of 14), kw padded 7 -> 8: the counters agree
                    # (profiles/r04/insts_per_kernel.md: 6.881e7 MFMAs x 2048 FLOP = 140.9 of 172.6 GFLOP)
                    factor, tag = 10.0 / 14.0 * 8.0 / 7.0, 'winograd F(2,3) along D, kw 7->8'
                elif key == 'stem_wgrad_gather':
                    factor, tag = 0.0, 'gather over the pooled elements on the vector ALUs (no MFMA)'
                elif op.kind == _hip.OP_CONV_WGRAD:
                    factor = _hip.op_wgrad_issue_factor(op)
                    tag = {1.0: 'direct', 2.0 / 3.0: 'winograd F(3,2) along W', 1.0 / 3.0: 'winograd F(3,2) along H x F(3,4) along W'}.get(
                        min((1.0, 2.0 / 3.0, 1.0 / 3.0), key=lambda v: abs(v - factor)), 'direct')
                issued += fl * factor
                var[tag] = var.get(tag, 0) + 1
            return (issued / groups[key]['flops'] if groups[key]['flops'] else 1.0), var

        issued_step = priced_step = 0.0                      # MFMA FLOPs one step issues / the algorithmic FLOPs they stand for
        for gkey, gv in groups.items():
            if gv['flops'] > 0:
                issued_step += gv['flops'] * issue_factor_of(gkey)[0]
                priced_step += gv['flops']
        dom_ms = group_ms(dom)
        launches = a.steps * groups[dom]['n']
        alg = groups[dom]['flops'] * a.steps / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        issue_factor, variants = issue_factor_of(dom)
        ach = alg * issue_factor
        ms_per_step = el * 1e3 / a.steps
        # the engine's arithmetic decides the peak: every kernel of a bf16 run (stem included) multiplies on the bf16 matrix cores
        peak = PEAK_BF16_MATRIX if a.dtype == 'bf16' else PEAK_FP32_MATRIX
        msg = reduce_buf.numel() * 4 / 1e6
        res = {
            'metric': f'train iters/sec on 512x512x80 vols (ResNet3D-50+FPN3D forward+backward, {"fp32" if a.dtype == "f32" else "bf16 storage / fp32 accumulate"})',
            'value': world * a.steps / el, 'unit': 'iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype,
            'data': 'synthetic', **census,
            'config': {'workload': f'BASELINE configs[{1 if a.dtype == "f32" else 4}] backbone: 1x3x{D}x{H}x{W} fp32 volume per GPU, ResNet3D-50 + FPN3D '
                                   'forward+backward (eval-mode BN, random-init weights' + ('' if a.dtype == 'f32' else '; bf16 activations') + ')' +
                                   ('' if world == 1 else f' + one SUM all-reduce of a {msg:.1f} MB flat fp32 gradient arena over RCCL' +
                                    ('' if a.backbone_only_allreduce else " (the whole detector's message size, SURVEY 8e)")),
                       'global_batch': world, 'parallelism': f'dp{world}'},
            # frac = MFMA FLOP/s actually ISSUED by the dominant kernel / fp32 matrix peak (a utilisation, <= 1);
            # algorithmic_* = 2*MAC of the convolutions it computes / time (exceeds `achieved` for Winograd instances)
            'roofline': {'bound': 'mfma', 'kernel': dom, 'achieved': ach / 1e12, 'peak': peak / 1e12,
                         'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': committed_traffic(dom),
                         'algorithmic_achieved': alg / 1e12, 'algorithmic_frac': alg / peak,
                         'mfma_issue_factor': issue_factor, 'variants': variants,
                         'avg_launch_ms': dom_ms / launches if launches else None, 'launches_per_step': groups[dom]['n'],
                         'timing': 'HIP events recorded by mr3d_plan_run_marked on the stream the kernel is launched on, around every '
                                   'launch of this kernel inside the timed steps; the plan is not split (other kernels overlap '
                                   'it exactly as in production)',
                         'whole_step_frac': FLOP_FWD_BWD_512 * scale / (ms_per_step * 1e-3) / peak,
                         'whole_step_issued_frac': issued_step / (ms_per_step * 1e-3) / peak,
                         'whole_step_issued_gflop': issued_step / 1e9, 'whole_step_priced_gflop': priced_step / 1e9,
                         'whole_step_issued_note': 'sum over EVERY record of the forward and backward plans of (its algorithmic 2*MAC x the MFMA '
                                                   'issue factor the library reports for the instance it launches: 1 direct, 2/3, 1/2, 1/3 the '
                                                   'Winograd forms, 0 the vector-ALU gather of the stem weight gradient) / step time / peak: the '
                                                   'share of the step in which the matrix cores issue',
                         'whole_step_flops_note': 'algorithmic 2*MAC of the dense convolutions of one forward+backward pass (1124.4 GFLOP on the '
                                                  '512x512x80 volume); the Winograd kernels issue 1/3 - 2/3 of theirs on the matrix cores and the '
                                                  'gather form of the stem weight gradient multiplies only the <= 1/8 of dy that the max-pool '
                                                  'adjoint leaves non-zero (21.6 of 172.6 GFLOP, same result)',
                         'next_kernels': []},
        }
        for key in top:
            if key == dom:
                continue
            kms, n_ = group_ms(key, next_events), len(next_events) * groups[key]['n']
            kalg = groups[key]['flops'] * len(next_events) / (kms * 1e-3) if kms > 0 else 0.0
            kf, kvar = issue_factor_of(key)
            res['roofline']['next_kernels'].append({
                'kernel': key, 'avg_launch_ms': kms / n_ if n_ else None, 'launches_per_step': groups[key]['n'],
                'algorithmic_achieved': kalg / 1e12, 'algorithmic_frac': kalg / peak, 'mfma_issue_factor': kf,
                'achieved': kalg * kf / 1e12, 'frac': kalg * kf / peak, 'variants': kvar, 'traffic': committed_traffic(key),
                'timing': f'as the dominant kernel, in {len(next_events)} extra steps behind the timed region'})
        if world == 1 and not a.no_aux and a.dtype == 'f32':
            try:
                res['aux'] = {k: (round(v, 4) if ('frac' in k or 'per_s' in k) else round(v, 1)) for k, v in ops_microbench(dev).items()}
                res['aux']['roialign3d_plus_nms3d_us_per_vol'] = round(res['aux']['roialign3d_bbox_fwd_us'] + res['aux']['roialign3d_mask_fwd_us'] + res['aux']['nms3d_us_per_vol'], 1)
                ms = two_stage_step_ms((D, H, W))
                res['aux']['two_stage_step_ms'] = round(ms, 2)           # BASELINE configs[2]
                res['aux']['two_stage_iters_per_s'] = round(1e3 / ms, 2)
                # same step with the mask head always on its full 128-row capacity (no positive-count read-back)
                res['aux']['two_stage_step_static_mask_batch_ms'] = round(two_stage_step_ms((D, H, W), static_mask_batch=True), 2)
                ms2 = two_stage_step_ms((D, H, W), steps=8, two_scales=True)  # the shipped model: region + its 1.5x copy
                res['aux']['shipped_2scales_step_ms'] = round(ms2, 2)
                res['aux']['shipped_2scales_iters_per_s'] = round(1e3 / ms2, 2)
                # BASELINE configs[4] (build-defined: bf16 activation storage in backbone + FPN, DeformConv3d neck); NOT the
                # headline: the reference's arithmetic is fp32
                import subprocess
                o = subprocess.run([sys.executable, os.path.abspath(__file__), '--dtype', 'bf16', '--steps', '20', '--warmup', '5', '--no-aux',
                                    '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
                bf = json.loads([l for l in o.stdout.splitlines() if l.startswith('{')][-1])
                res['aux']['bf16_backbone_iters_per_s'] = round(bf['value'], 2)
                res['aux']['bf16_backbone_step_ms'] = round(bf['ms_per_step'], 3)
                # the bf16 numbers are NOT fp32-equivalent results: parity bounds of the bf16 storage path (configs[4] is build-defined)
                res['aux']['bf16_parity_bounds'] = ('vs the fp32 arithmetic of the same model: pyramid outputs 3e-2 of their scale, parameter '
                                                    'gradients 8e-2 in L2 (measured 1.4e-2 .. 2.9e-2), losses 2e-2 -- tests/test_gpu_bf16.py, '
                                                    'tests/test_gpu_configs.py::test_config4_dual_pathway_dcn_neck_bf16')
                res['aux']['bf16_two_stage_step_ms'] = round(two_stage_step_ms((D, H, W), bf16=True), 2)
                res['aux']['bf16_shipped_2scales_step_ms'] = round(two_stage_step_ms((D, H, W), steps=8, two_scales=True, bf16=True), 2)
                res['aux']['bf16_dcn_neck_2scales_step_ms'] = round(two_stage_step_ms((D, H, W), steps=6, two_scales=True, bf16=True, dcn_neck=True), 2)
            except Exception as exc:  # the headline number must still print
                res['aux'] = {'error': repr(exc)}
        if world == 1 and not a.no_cpu_baseline and a.dtype == 'f32':
            res['cpu_baseline'] = cpu_baseline(a.cpu_seconds)
            try:        # C2 (numpy NMS3D) and C3 (C RoIAlign3D, 1 thread / OpenMP)
                res['cpu_baseline']['ops'] = cpu_baseline_ops(min(a.cpu_seconds, 12.0))
            except Exception as exc:
                res['cpu_baseline']['ops'] = {'error': repr(exc)}
            try:        # C4: gloo mean of the 103.9 MB gradient buffer at 2 / 4 / 8 local ranks
                res['cpu_baseline'].setdefault('ops', {})['allreduce_gloo'] = cpu_baseline_allreduce(min(a.cpu_seconds, 12.0))
            except Exception as exc:
                res['cpu_baseline'].setdefault('ops', {})['allreduce_gloo'] = {'error': repr(exc)}
        if exch is not None:
            res['allreduce'] = exch
        emit_json_line(res)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if exch is not None and not exch['grads_identical_across_ranks']:
        print('bench.py: the ranks hold different gradients after the exchange: ' + ' '.join(exch['grad_checksums']), file=sys.stderr)
        raise SystemExit(3)


if __name__ == '__main__':
    main()
