#!/usr/bin/env python3
"""Headline benchmark: stitched frames/s of the VideMosaic hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic 720p panned video, ORB-1000 + BF-Hamming
cross-check + RANSAC findHomography + warpPerspective / blend into a 1440x1536 canvas.  A step
is VideMosaic.process_frame (reference main.py:710-759) on one frame; the frames are resident in
HBM and go through Context.process_video (rtvm_batch_detect / rtvm_batch_retry: detection,
matching and RANSAC in batches ahead of the warp + blend, which follows in pieces).  With
--gpus N every rank stitches its own, independently seeded stream on its own GPU (no data-path
collective; RCCL carries only the barrier and the max-over-ranks time).  --procs N starts N such
processes on ONE GPU (gloo): what N Python orchestrators cost each other on one host.

Other workloads (--config): 1080p and 4K / 16k-canvas ORB streams through the same pipeline, and
1080p_sift2000_l2: the reference's default detector at 2000 features, whose L2 knn-2 matcher is
the one dense contraction of the path (matrix-core roofline).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
synth = rtvm.synth

CONFIGS = {
    # name: frame h, w, canvas h, w, nfeatures
    "720p_orb1000": (720, 1280, 1440, 1536, 1000),
    "1080p_orb1000": (1080, 1920, 2160, 2304, 1000),
    "4k_orb4000_canvas16k": (2160, 3840, 4096, 16384, 4000),
    "360p_orb700": (360, 640, 720, 768, 700),
    "540p_orb1000": (540, 960, 1080, 1152, 1000),
    "1080p_sift2000_l2": (1080, 1920, 2160, 2304, 2000),
    # the reference's default detector (main.py:1603) through the same device-resident pipeline
    "720p_sift700": (720, 1280, 1440, 1536, 700),
    "1080p_sift2000": (1080, 1920, 2160, 2304, 2000),
}
SIFT_CONFIGS = ("720p_sift700", "1080p_sift2000")
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E
HBM_COPY_GBS = 6290.0        # measured float4 copy rate
MFMA_BF16_TFLOPS = 2500.0    # dense bf16 matrix-core peak
PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")


def pyramid_pixels(fh, fw):
    tot = 0
    for l in range(8):
        s = np.float32(np.float64(np.float32(1.2)) ** l)
        inv = np.float32(1.0) / s
        tot += int(np.rint(fw * inv)) * int(np.rint(fh * inv))
    return tot


def pyramid_level_pixels(fh, fw, l):
    s = np.float32(np.float64(np.float32(1.2)) ** l)
    inv = np.float32(1.0) / s
    return int(np.rint(fw * inv)) * int(np.rint(fh * inv))


def algorithmic_bytes(name, fh, fw, ch, cw, nkp, roi_px, batch):
    """Compulsory HBM bytes per frame of all launches of a kernel: every input read once, every
    output written once.  Every kernel is launched once per batch
    and serves all frames of the batch.

    The distance transform runs two jobs per frame: the new mask on the frame's region of the
    canvas (roi_px pixels) and the (predicted) old mask on the canvas content's bounding box, whose
    distances are only needed on the blend window (the region plus the 15-pixel blur margin)."""
    P, C, S = fh * fw, ch * cw, pyramid_pixels(fh, fw)
    win_h, win_w = min(ch, fh + 38), min(cw, fw + 38)
    win = win_h * win_w
    A = C + roi_px                      # pixels the diagonal scans cover (both jobs)
    rows_px = roi_px + win_h * cw       # rows of the output windows, full rectangle width
    cols_px = roi_px + win_w * ch       # columns of the output windows, full rectangle height
    win2 = roi_px + win                 # output windows of both jobs
    table = {
        "k_gray": 3 * P + P,
        "k_fast": S,                           # level pixels in (candidates out are a few KB)
        "k_resize_chain<3>": P + sum(pyramid_level_pixels(fh, fw, l) for l in (1, 2, 3)),   # level 0 in, levels 1-3 out
        "k_resize_chain<4>": sum(pyramid_level_pixels(fh, fw, l) for l in (3, 4, 5, 6, 7)),     # level 3 in, levels 4-7 out
        "k_warp<false>": 3 * P + 4 * roi_px + roi_px / 8,   # frame in, warped u8x3 + mask + zero bits out
        "k_warp<true>": 3 * P + 4 * roi_px + roi_px / 8,
        "k_dt_diag_a": A / 8 + A / 4,          # zero bit rows in, one bit per pixel and orientation out
        "k_dt_diag_b": A / 4 + 2 * A,          # diagonal bitmaps in, u16 step plane out
        # steps of the windows' rows and columns in (the window itself once); column and row summaries out
        "k_dt_sum": 2 * (rows_px + cols_px - win2) + cols_px / 4 + rows_px / 8,
        "k_dt_final": 2 * win2 + cols_px / 4 + rows_px / 8 + 4 * win2,   # steps and summaries in; f32 out
        "k_blur_rows": 4 * roi_px + 4 * win + 8 * roi_px,   # two distance planes in, two row-blurred weights out
        "k_blur_cols": 8 * win + roi_px + 8 * roi_px,       # row-blurred weights + mask in, final weights out
        # per frame: mask + warped + two weights in; per batch: canvas + mask in/out and their snapshot
        "k_commit<false>": 12 * roi_px + 12 * roi_px / batch,
        "k_commit4<false>": 12 * roi_px + 12 * roi_px / batch,
        # the column pass inside the commit: row-blurred weights and warped pixels in per frame; canvas + mask per batch
        "k_commit_cols<false>": 8 * win + 4 * roi_px + 12 * roi_px / batch,
        "k_hamming_argmin": 2 * nkp * 32 * 2,
        "k_rs_refine": nkp * 16,               # matched point pairs in (latency bound: fp64 solver chains)
    }
    # SIFT scale space: the doubled image and the octaves built layer by layer (the octaves of at most 4000
    # pixels are built in LDS by one workgroup); a blur reads a layer and writes the next one; the differences
    # of Gaussians are not stored: the extrema kernel reads the six Gaussian layers of an octave
    big, oh, ow = [], 2 * fh, 2 * fw
    while oh * ow > 4000 or not big:
        big.append(oh * ow)
        oh, ow = oh // 2, ow // 2
    table["k_sift_blur"] = 8 * big[0] + 8 * 5 * sum(big)
    table["k_sift_extrema"] = 4 * 6 * sum(big)          # every Gaussian plane once
    return table.get(name)


def warp_blend_stage_bytes(fh, fw, ch, cw, roi_px):
    """The warp + blend stage by the reference's own data flow (main.py:871-927): frame read once, canvas
    region read + written once, two full-canvas distance transforms (mask read + f32 distance written)
    and two 31x31 blurs of f32 weights (read + write per separable pass)."""
    return 3 * fh * fw + 6 * roi_px + 2 * 5 * ch * cw + 16 * roi_px


def warp_blend_compulsory_bytes(fh, fw, roi_px):
    """What the stage cannot avoid moving: the frame read once, the canvas region read and written once
    (B_warp = 3 P + 6 A_roi)."""
    return 3 * fh * fw + 6 * roi_px


def oracle_modules():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    from oracle_mosaic import OracleMosaic
    return oracle_ffi, OracleMosaic


def cpu_baseline(frames_np, cfg, n_frames, threads, detector="orb"):
    """The CPU restatement (oracle/) on the first n_frames of the same video: frames/s and the mosaic."""
    orc, OracleMosaic = oracle_modules()
    fh, fw, ch, cw, nf = cfg
    before = orc.set_threads(threads)
    try:
        om = OracleMosaic(frames_np[0], nfeatures=nf, canvas_size=(ch, cw), fixed=False,
                          glue="explicit", detector=detector)
        t0 = time.perf_counter()
        for f in frames_np[1:n_frames + 1]:
            om.process_frame(f)
        dt = time.perf_counter() - t0
    finally:
        orc.set_threads(before)
    return n_frames / dt, om


def max_over_ranks(seconds, world, device=None):
    """Slowest rank's time: what the whole job took (all_reduce MAX over the process group)."""
    if world <= 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(frames_per_rank, seconds, world, device=None):
    """Whole-job frames/s: every rank stitched frames_per_rank frames of its own stream."""
    return world * frames_per_rank / max_over_ranks(seconds, world, device)


def stream_seed(rank, world):
    """Independent streams: one seeded synthetic video per rank."""
    return 0 if world == 1 else 10 + rank


def per_rank_rates(frames, seconds, world, device=None):
    """Every rank's own frames/s, gathered on all ranks."""
    if world <= 1:
        return [frames / seconds]
    import torch
    import torch.distributed as dist
    mine = torch.tensor([frames / seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def merge_variants(table):
    """The per-kernel timing table with the variants of the SIFT blur kernels (one instantiation per kernel
    length) added up under the kernel's name."""
    out = {}
    for k, (cnt, ms, stream) in table.items():
        name = k.split("<")[0] if k.startswith("k_sift_blur") else k
        c0, m0, _ = out.get(name, (0, 0.0, stream))
        out[name] = (c0 + cnt, m0 + ms, stream)
    return out


def pmc_traffic(kernel, config):
    """HBM bytes per frame of `kernel` from the committed rocprofv3 --pmc summary (separate FETCH_SIZE and
    WRITE_SIZE passes over a known number of frames; the counters cannot be read from inside this process)."""
    try:
        path = PMC_FILE if config == "720p_orb1000" else PMC_FILE.replace(".json", "_%s.json" % config)
        with open(path) as fjs:
            pmc = json.load(fjs)
        if pmc.get("config") != config:
            return None, None
        # the instantiations of a templated kernel (k_sift_blur<11>, ...) add up under the kernel's name
        ents = [e for k, e in pmc["kernels"].items() if k == kernel or k.startswith(kernel + "<")]
        if not ents:
            return None, None
        return sum(e["hbm_bytes_per_frame"] for e in ents), pmc.get("_about")
    except (OSError, KeyError, ValueError):
        return None, None


def run_procs(args):
    """N independent single-GPU processes (own context, own seeded video) on ONE GPU, timed between common
    barriers (gloo): the cost of N Python orchestrators and N pipelines sharing a host and a device."""
    port = 29000 + os.getpid() % 3000
    procs = []
    for r in range(args.procs):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.procs), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RTVM_BENCH_SAME_GPU="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--config", args.config, "--batch", str(args.batch), "--cpu-frames", "0",
               "--dropin-frames", "0", "--host-frames", "0", "--sift-frames", "0"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = [p.wait() for p in procs]
    if any(rc):
        raise SystemExit("a child process failed: %s" % rc)
    sys.stdout.write(out.decode())


def bench_sift_l2(args, cfg, dev, local_rank):
    """cfg 3: 1080p, the reference's default detector with 2000 features.  value = stitched frames/s of
    VideMosaic(detector_type='sift') driven like main.py --hide drives the reference (one host frame per
    process_frame call, PCIe included; the class runs its batched device pipeline behind that); roofline =
    the L2 knn-2 matcher's matrix-core kernel on the descriptors of the video's last two frames."""
    fh, fw, ch, cw, nf = cfg
    K, W = args.steps, args.warmup
    T = 1 + W + K
    Gs, wh, ww = synth.bounded_path(T, fh, fw, seed=2)
    world_img = synth.make_world(wh, ww, seed=0)
    host = list(synth.render_video_torch(world_img, Gs, fh, fw, dev).cpu().numpy())
    vm = rtvm.VideMosaic(host[0], detector_type="sift", nfeatures=nf, canvas_size=(ch, cw), device=local_rank,
                         show_intermediate=False, visualize=False)
    for t, f in enumerate(host[1:1 + W], 1):
        vm.process_frame(f, t)
    vm.output_img_u8()
    t0 = time.perf_counter()
    for t, f in enumerate(host[1 + W:], 1 + W):
        vm.process_frame(f, t)
    vm.output_img_u8()      # brings the mosaic up to the last frame and copies it out
    elapsed = time.perf_counter() - t0
    q, t = vm.des_cur, vm.des_prev
    n_matches = len(vm.matches)
    vm.close()
    # the matcher on the last two frames' descriptors, kernels timed with HIP events on their stream
    ctx = rtvm.Context(device=local_rank)
    for _ in range(3):
        ctx.match_l2_knn2(q, t)
    reps = 50
    ctx.profile_start(None)
    for _ in range(reps):
        ctx.match_l2_knn2(q, t)
    tab = ctx.profile_stop()
    ctx.close()
    name = "k_l2_tile<8>"
    launches, total_ms, _ = tab[name]
    avg_s = total_ms / launches * 1e-3
    flop = 2.0 * len(q) * len(t) * 128
    all_us = sum(v[1] for v in tab.values()) / reps * 1e3
    roof = {"kernel": name, "bound": "mfma", "launches": launches, "avg_us": avg_s * 1e6,
            "achieved": flop / avg_s / 1e12, "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": flop / avg_s / 1e12 / MFMA_BF16_TFLOPS, "traffic": None,
            "algorithmic_flop": flop, "descriptors": [len(q), len(t), 128],
            "matcher_all_kernels_us": all_us,
            "note": "2*Nq*Nt*128 flop of one 128x128-tile launch over the whole distance matrix; at this size the "
                    "launch is one tile per workgroup and latency bound (16000 x 16000: 0.25 of the peak)"}
    cpu = None
    if args.cpu_frames > 0:
        n_cpu = min(args.cpu_frames, 4)
        fps1, _ = cpu_baseline(host, cfg, n_cpu, 1, "sift")
        cpu = {"value": fps1, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "first %d frames of the same video through the oracle/ C++ restatement (SIFT), one thread" % n_cpu}
    return {
        "metric": "stitched frames/sec @1080p SIFT-2000 + L2 knn-2 (matrix cores) on 1xMI355X",
        "value": K / elapsed, "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.config, "frame": [fh, fw], "canvas": [ch, cw], "nfeatures": nf,
                   "path": "VideMosaic.process_frame, headless (main.py --hide), one host frame per call, PCIe included",
                   "last_frame_matches": n_matches},
        "roofline": roof, "cpu_baseline": cpu,
    }


def leg_dropin(args, cfg, frames, local_rank, T):
    """The drop-in class as the reference's main loop drives it (main.py:1596-1634): one host frame per
    process_frame call, PCIe copies included, constructed with the reference's default display flags
    (visualize=True, show_intermediate=True, main.py:17).  process_frame only takes the frame and the work follows
    in batches; the clock stops after flush() + a device sync, i.e. with every frame stitched."""
    fh, fw, ch, cw, nf = cfg
    nd = min(args.dropin_frames, T - 1)
    n_warm = min(32, nd // 4)
    host_frames = frames[:nd + 1].cpu().numpy()
    kw = dict(detector_type="orb", nfeatures=nf, canvas_size=(ch, cw), device=local_rank, warp_mode=args.warp_mode)
    progress("drop-in leg: process_frame with the reference's default flags, %d frames" % nd)
    vm = rtvm.VideMosaic(host_frames[0], **kw)
    assert vm.visualize and vm.show_intermediate and vm._deferred
    for t in range(1, n_warm + 1):
        vm.process_frame(host_frames[t], t)
    vm.flush()
    vm._ctx.sync()
    td0 = time.perf_counter()
    for t in range(n_warm + 1, nd + 1):
        vm.process_frame(host_frames[t], t)
    vm.flush()
    vm._ctx.sync()
    td = time.perf_counter() - td0
    dropin = {"value": (nd - n_warm) / td, "unit": "frames/s", "frames": nd - n_warm,
              "pipeline": dict(vm._pipe.stats),
              "flags": "reference defaults: visualize=True, show_intermediate=True",
              "note": "Python VideMosaic.process_frame, one host frame per call: copy into the pinned "
                      "ring, PCIe copy, batched detection, host-side validate/smooth/chain, batched warp + blend"}
    # ... and as the reference's loop runs with a progress call-back (main.py:1630-1632): output_img.copy() after
    # every frame, i.e. the float64 mosaic read back and copied by the caller once per frame
    nc = min(48, nd - n_warm)
    if nc >= 8:
        t_read = 0.0
        td0 = time.perf_counter()
        for t in range(nd + 1 - nc, nd + 1):
            vm.process_frame(host_frames[t], t)
            tr0 = time.perf_counter()
            img = vm.output_img
            t_read += time.perf_counter() - tr0
            img = img.copy()
        td = time.perf_counter() - td0
        dropin["process_frame_then_output_img_copy"] = {
            "value": nc / td, "frames": nc, "ms_per_output_img": t_read / nc * 1e3, "mosaic_mbytes_float64": img.nbytes / 1e6,
            "note": "every call followed by output_img.copy(): the frame is stitched before the read returns, the u8 canvas "
                    "crosses PCIe and is widened to the reference's float64 on four host threads; the caller's copy of "
                    "the float64 array is in the time"}
    vm.close()
    # the same class with pipeline_batch=0: every call completes before it returns
    ns = min(136, nd)
    progress("drop-in leg: synchronous process_frame, %d frames" % ns)
    vm = rtvm.VideMosaic(host_frames[0], pipeline_batch=0, **kw)
    vm.process_frame(host_frames[1], 1)
    td0 = time.perf_counter()
    for t in range(2, ns + 1):
        vm.process_frame(host_frames[t], t)
    vm._ctx.sync()
    dropin["process_frame_synchronous"] = {"value": (ns - 1) / (time.perf_counter() - td0), "frames": ns - 1}
    vm.close()
    if ns >= 16:
        # ... and through process_frames: detection, matching and RANSAC on four worker threads
        progress("drop-in leg: process_frames with 4 workers")
        vm = rtvm.VideMosaic(host_frames[0], pipeline_batch=0, **kw)
        vm.process_frames(list(host_frames[1:9]), 1, workers=4)
        td0 = time.perf_counter()
        vm.process_frames(list(host_frames[9:ns + 1]), 9, workers=4)
        vm.output_img_u8()
        dropin["process_frames_4_workers"] = {"value": (ns - 8) / (time.perf_counter() - td0), "frames": ns - 8}
        vm.close()
    return dropin


def leg_sift(args, cfg, frames, local_rank, T):
    """The reference's default detector (main.py:1603 passes "sift") through the drop-in class with host frames:
    with the reference's default flags (the batched device pipeline behind process_frame), with pipeline_batch=0
    (every call completes before it returns), and process_frames with detection, matching and RANSAC on worker
    threads."""
    fh, fw, ch, cw, nf = cfg
    ns = min(args.sift_frames, T - 1)
    host_frames = list(frames[:ns + 1].cpu().numpy())
    kw = dict(detector_type="sift", nfeatures=700, canvas_size=(ch, cw), device=local_rank, warp_mode=args.warp_mode)
    # process_frame only takes the frame; 32 frames of warm-up (the pipeline's buffers are reserved on the first
    # call), then the rest, the clock stopping with every frame stitched
    progress("SIFT leg: process_frame with the reference's default flags, %d frames" % ns)
    vm = rtvm.VideMosaic(host_frames[0], **kw)
    assert vm.visualize and vm.show_intermediate and vm._deferred
    nw = min(32, ns // 2)
    for t in range(1, nw + 1):
        vm.process_frame(host_frames[t], t)
    vm.flush()
    vm._ctx.sync()
    ts0 = time.perf_counter()
    for t in range(nw + 1, ns + 1):
        vm.process_frame(host_frames[t], t)
    vm.flush()
    vm._ctx.sync()
    ts1 = time.perf_counter()
    leg = {"value": (ns - nw) / (ts1 - ts0), "unit": "frames/s", "frames": ns - nw, "nfeatures": 700,
           "flags": "reference defaults: visualize=True, show_intermediate=True",
           "note": "VideMosaic(detector_type='sift').process_frame, one host frame per call, PCIe included"}
    vm.close()
    # with pipeline_batch=0 every call completes before it returns
    progress("SIFT leg: synchronous process_frame and process_frames")
    vm = rtvm.VideMosaic(host_frames[0], pipeline_batch=0, **kw)
    n1 = min(16, ns - 1)
    vm.process_frame(host_frames[1], 1)
    ts0 = time.perf_counter()
    for t in range(2, 2 + n1):
        vm.process_frame(host_frames[t], t)
    vm._ctx.sync()
    ts1 = time.perf_counter()
    leg["process_frame_synchronous"] = {"value": n1 / (ts1 - ts0), "frames": n1}
    n2 = min(64, ns - 1 - n1)
    if n2 > 8:
        vm.process_frames(host_frames[2 + n1:2 + n1 + n2], 2 + n1, workers=4)
        vm._ctx.sync()
        leg["process_frames_4_workers"] = {"value": n2 / (time.perf_counter() - ts1), "frames": n2}
    vm.close()
    return leg


def leg_host_ring(args, cfg, frames, local_rank, T, B):
    """Frames in host memory through the ingest ring (pinned slots, copies on a stream of their own) into the same
    batched pipeline: the PCIe-inclusive rate of the throughput path."""
    fh, fw, ch, cw, nf = cfg
    nh = min(args.host_frames, T - 1)
    host_frames = frames[:nh + 1].cpu().numpy()
    progress("host ring leg, %d frames" % nh)
    ctx2 = rtvm.Context(fh, fw, ch, cw, nfeatures=nf,
                        flags=(rtvm._ffi.FLAG_WARP_FIXED if args.warp_mode == "fixed" else 0), device=local_rank)
    ctx2.process_host_video(host_frames[0], host_frames[1:1 + min(64, nh)], ch - fh, int(cw / 2 - fw / 2), batch=B)
    th0 = time.perf_counter()
    sth = ctx2.process_host_video(host_frames[0], host_frames[1:], ch - fh, int(cw / 2 - fw / 2), batch=B)
    th = time.perf_counter() - th0
    leg = {"value": nh / th, "unit": "frames/s", "frames": nh, "frames_not_ok": int((sth != 0).sum()),
           "note": "Context.process_host_video: host frames -> pinned ring -> device, batches of %d; "
                   "one memcpy per frame into the pinned slot (4 threads) and the PCIe copy included" % B}
    ctx2.close()
    return leg


def progress(msg):
    """A line on stderr per phase: a run that prints nothing for minutes is taken to be hung."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="720p_orb1000", choices=sorted(CONFIGS))
    ap.add_argument("--warp-mode", default="float", choices=["float", "fixed"])
    ap.add_argument("--batch", type=int, default=32, help="frames per detection batch (1..32)")
    ap.add_argument("--single-stream", action="store_true", help="queue every batch on one HIP stream")
    ap.add_argument("--cpu-frames", type=int, default=12, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--path", default="bounded", choices=["bounded", "drift"],
                    help="camera path: 'bounded' keeps every frame inside the canvas for any number of frames; "
                         "'drift' moves 2 px/frame right and up for ever (leaves the 1.2x-wide canvas after ~64 frames)")
    ap.add_argument("--canvas-batch", type=int, default=0,
                    help="frames per canvas-half piece (0 = half a batch)")
    ap.add_argument("--host-frames", type=int, default=512,
                    help="frames of the video also run from host memory through the ingest ring (0 = skip)")
    ap.add_argument("--sift-frames", type=int, default=288,
                    help="frames of the PCIe-inclusive sample through VideMosaic with the SIFT detector (0 = skip)")
    ap.add_argument("--dropin-frames", type=int, default=1056,
                    help="frames of the PCIe-inclusive sample through the Python VideMosaic class (0 = skip)")
    ap.add_argument("--procs", type=int, default=1,
                    help="start this many independent processes (own context and video) on ONE GPU and report the aggregate")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel timing table to this JSON file")
    args = ap.parse_args()

    if args.procs > 1:
        return run_procs(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    same_gpu = os.environ.get("RTVM_BENCH_SAME_GPU") == "1"   # children of --procs
    if world != args.gpus and not same_gpu:
        if rank == 0 and world > 1:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    import torch
    import torch.distributed as dist
    # RTVM_BENCH_STUB=1 (tests/test_bench_dist.py, no GPU): the rank / barrier / aggregation plumbing of this file end
    # to end with a stand-in for rtvm.Context that stitches nothing; the line it prints says so in "data"
    stub = os.environ.get("RTVM_BENCH_STUB") == "1"
    if stub:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_stub
        bench_stub.install(rtvm)
        dev, red_dev = torch.device("cpu"), None
        device_sync = lambda: None
        args.cpu_frames = args.dropin_frames = args.sift_frames = args.host_frames = 0
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        red_dev = dev
        device_sync = torch.cuda.synchronize
    if world > 1:
        if same_gpu or stub:
            dist.init_process_group(backend="gloo")
            red_dev = None
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    cfg = CONFIGS[args.config]
    fh, fw, ch, cw, nf = cfg
    sift = args.config in SIFT_CONFIGS
    if sift:   # the host-frame legs are ORB legs
        args.dropin_frames = args.sift_frames = args.host_frames = 0
    if args.config == "1080p_sift2000_l2":
        out = bench_sift_l2(args, cfg, dev, local_rank)
        if rank == 0:
            print(json.dumps(out))
        return
    K, W = args.steps, args.warmup
    # the host-frame legs may use more frames than the timed region
    T = 1 + max(W + K, args.dropin_frames if world == 1 else 0, args.host_frames if world == 1 else 0)
    # --- synthetic video, seeded per rank (independent streams), resident in HBM
    # the camera path stays inside the canvas for any number of frames, so that every frame of the
    # timed region is warped and blended in full
    seed = stream_seed(rank, world)
    if args.path == "bounded":
        Gs, side_h, side_w = synth.bounded_path(T, fh, fw, seed=seed + 2)
    else:
        side_h, side_w = fh + 2 * T + 256, fw + 2 * T + 256
        Gs = synth.camera_path(T, 64.0, side_h - fh - 64.0, seed=seed + 2)
    progress("synthetic world %d x %d, %d frames" % (side_h, side_w, T))
    world_img = synth.make_world(side_h, side_w, seed=seed)
    frames = synth.render_video_torch(world_img, Gs, fh, fw, dev)
    device_sync()
    progress("frames resident in HBM")

    flags = rtvm._ffi.FLAG_WARP_FIXED if args.warp_mode == "fixed" else 0
    if args.single_stream:
        flags |= rtvm._ffi.FLAG_SINGLE_STREAM
    ctx = rtvm.Context(fh, fw, ch, cw, detector=rtvm._ffi.DET_SIFT if sift else rtvm._ffi.DET_ORB, nfeatures=nf,
                       flags=flags, device=local_rank)
    row_off, col_off = ch - fh, int(cw / 2 - fw / 2)
    stride = fh * fw * 3
    base = frames.data_ptr()

    B = max(1, min(32, args.batch))
    ctx.reserve_batch(B)

    def run(first, count):
        # frames first .. first+count-1: detection in batches of up to B, the canvas half in pieces; frames that
        # come back pending (see Context.process_video) are queued again, so every frame is stitched
        st = ctx.process_video([base + (first + j) * stride for j in range(count)], batch=B,
                               canvas_batch=args.canvas_batch)
        return st, dict(ctx.video_stats)

    def restart():
        ctx.stream_begin(base, row_off, col_off)
        run(1, W)
        ctx.sync()

    # --- pass 1, not timed for the headline: the same K frames with every kernel carrying HIP events.
    # The canvas half (main stream) bounds the throughput; detection + matching and RANSAC of the following
    # batches run beside it on streams of their own.  The roofline kernel is the main-stream kernel with the
    # largest summed time over these K frames; the longest kernel of any stream is named next to it.
    restart()
    progress("instrumented pass")
    ctx.profile_start(None)
    _, vstats1 = run(1 + W, K)
    table = merge_variants(ctx.profile_stop())
    main_stream = {k: v for k, v in table.items() if v[2] == 0 or args.single_stream}
    # the stream whose kernels add up to the most time bounds the frame rate: the canvas half (main stream) with
    # ORB, detection + matching with SIFT; the roofline kernel is the longest one of that stream
    stream_names = ["main (canvas half)", "detect+match", "ransac search", "ransac polish"]
    stream_ms = [sum(v[1] for v in table.values() if v[2] == sidx) for sidx in range(4)]
    # (the two RANSAC streams hold one-wave-per-frame kernels whose time is latency, not work: no roofline
    # describes them; the longest of them is named under longest_kernel_any_stream)
    bound_stream = 0 if args.single_stream else int(np.argmax(stream_ms[:2]))
    bound = {k: v for k, v in table.items() if v[2] == bound_stream or args.single_stream}
    dominant = max(bound, key=lambda k: bound[k][1])
    longest = max(table, key=lambda k: table[k][1])

    # --- pass 2, the timed region: exactly frames 1+W .. W+K after W warm-up frames; only the dominant kernel
    # carries HIP events (on the stream it is launched on)
    progress("timed pass")
    restart()
    ctx.profile_start(dominant)
    if world > 1:
        dist.barrier()
    device_sync()
    ctx.sync()
    t0 = time.perf_counter()
    statuses, vstats = run(1 + W, K)
    ctx.sync()
    device_sync()
    t_own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    dom = merge_variants(ctx.profile_stop())[dominant]
    elapsed = max_over_ranks(t1 - t0, world, red_dev)
    rates = per_rank_rates(K, t_own, world, red_dev)
    st, nk, nm, H = ctx.frame_status()
    n_bad = int(np.count_nonzero(statuses & 11))
    if n_bad:
        print("warning: %d frames of the timed region were skipped or left pending" % n_bad, file=sys.stderr)

    if rank == 0:
        # region of interest of a frame on the canvas (near-translation): the frame plus the safety margin
        roi_px = min(ch * cw, (fh + 8) * (fw + 8))
        launches, total_ms = dom[0], dom[1]
        avg_s = total_ms / launches * 1e-3
        ab_frame = algorithmic_bytes(dominant, fh, fw, ch, cw, nf, roi_px, B)
        # frames that went through the canvas-half kernels: those of the pieces that were not void,
        # plus the ones whose canvas half was queued again
        K_back = vstats["canvas_frames"] - vstats["frames_void"] + vstats["frames_retried"]
        fpl = K_back / launches
        ab = ab_frame * fpl if ab_frame else None   # mean per launch
        traffic_frame, traffic_note = pmc_traffic(dominant, args.config)
        K1 = vstats1["canvas_frames"] - vstats1["frames_void"] + vstats1["frames_retried"]
        K_stream = K1 if bound_stream == 0 else K      # frames that went through the kernels of that stream
        by_time = sorted(bound, key=lambda k: -bound[k][1])
        if bound_stream != 0:
            fpl = K / launches
            ab = ab_frame * fpl if ab_frame else None
        roof = {"kernel": dominant, "bound": "hbm", "launches": launches, "avg_us": avg_s * 1e6,
                "achieved": (ab / avg_s / 1e9) if ab else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (ab / avg_s / 1e9 / HBM_PEAK_GBS) if ab else None,
                "frac_of_measured_copy_peak": (ab / avg_s / 1e9 / HBM_COPY_GBS) if ab else None, "algorithmic_bytes": ab,
                "algorithmic_bytes_per_frame": ab_frame, "frames_per_launch": fpl,
                "traffic": traffic_frame * fpl if traffic_frame else None,
                "traffic_bytes_per_frame": traffic_frame, "traffic_note": traffic_note,
                "chosen_as": "kernel with the largest summed HIP-event time, over the K frames of the instrumented pass, on "
                             "whichever of the two image streams (canvas half, detection + matching) adds up to the most "
                             "time; avg_us is from the timed pass",
                "stream": stream_names[bound_stream],
                "stream_kernel_ms_per_frame": {stream_names[i]: stream_ms[i] / K for i in range(4)},
                "stream_us_per_frame": {k: bound[k][1] * 1e3 / K_stream for k in by_time[:6]},
                "longest_kernel_any_stream": {"kernel": longest, "us_per_frame": table[longest][1] * 1e3 / K,
                                              "stream": stream_names[table[longest][2]]}}
        if table[longest][1] > bound[dominant][1]:
            # a kernel of another stream adds up to more time than the roofline kernel (the RANSAC polish: one wave per
            # frame, latency, no roofline of bytes or flops describes it)
            roof["dominant_by_time"] = {"kernel": longest, "us_per_frame": table[longest][1] * 1e3 / K,
                                        "stream": stream_names[table[longest][2]], "launches": table[longest][0],
                                        "avg_us": table[longest][1] / table[longest][0] * 1e3}
        # the warp + blend stage as a whole (everything on the main stream) over the K frames of the instrumented pass
        wb_bytes = warp_blend_stage_bytes(fh, fw, ch, cw, roi_px)
        wb_us = sum(v[1] for v in main_stream.values()) * 1e3 / K1
        warp_blend = {"algorithmic_bytes_per_frame": wb_bytes, "us_per_frame": wb_us,
                      "achieved": wb_bytes / wb_us / 1e3, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                      "frac": wb_bytes / wb_us / 1e3 / HBM_PEAK_GBS,
                      "compulsory_bytes_per_frame": warp_blend_compulsory_bytes(fh, fw, roi_px),
                      "frac_compulsory": warp_blend_compulsory_bytes(fh, fw, roi_px) / wb_us / 1e3 / HBM_PEAK_GBS,
                      "note": "warp GB/s vs the HBM roofline over the summed HIP-event time of all main-stream kernels, "
                              "K frames of the instrumented pass.  frac: bytes of the reference's data flow (frame, canvas "
                              "region, two canvas-sized distance transforms, two separable blurs; the implementation "
                              "narrows the transforms to the content box, so this credits bytes it never moves).  "
                              "frac_compulsory: frame read once + canvas region read and written once (3 P + 6 A_roi)"}
        extras = {}
        solo = world == 1
        progress("timed pass done: %.1f frames/s; host-frame legs" % (K / elapsed))
        if solo:
            # The legs below build contexts of their own; a caller has one context per video, so the timed
            # context goes before they start (the parity check builds its own).
            ctx.close()
            ctx = None
        if args.dropin_frames > 0 and solo:
            extras["dropin_pcie_inclusive"] = leg_dropin(args, cfg, frames, local_rank, T)
        if args.sift_frames > 0 and solo:
            extras["sift_dropin_pcie_inclusive"] = leg_sift(args, cfg, frames, local_rank, T)
        if args.host_frames > 0 and solo:
            extras["host_frames_pcie_inclusive"] = leg_host_ring(args, cfg, frames, local_rank, T, B)
        cpu, cpu_all, parity = None, None, None
        if args.cpu_frames > 0 and solo:   # reported at N = 1 only
            orc, _ = oracle_modules()
            n1 = args.cpu_frames
            progress("CPU baseline, %d frames on one thread" % n1)
            frames_np = frames[:2 * n1 + 1].cpu().numpy()
            fps_cpu, om = cpu_baseline(frames_np, cfg, n1, 1, "sift" if sift else "orb")
            cpu = {"value": fps_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "first %d frames of the same synthetic video through the oracle/ C++ restatement, "
                             "one thread" % n1}
            # the same frames through the GPU path: the mosaic must be the oracle's, pixel for pixel
            ctx = rtvm.Context(fh, fw, ch, cw, detector=rtvm._ffi.DET_SIFT if sift else rtvm._ffi.DET_ORB, nfeatures=nf,
                               flags=flags, device=local_rank)
            ctx.reserve_batch(B)
            ctx.stream_begin(base, row_off, col_off)
            st_p, _ = run(1, n1)
            _, _, _, H_p = ctx.frame_status()
            canvas_p = ctx.canvas_read()
            parity = {"frames": n1, "canvas_equal": bool(np.array_equal(canvas_p, om.canvas)),
                      "homography_equal": bool(np.array_equal(H_p, om.H)),
                      "statuses_equal": bool(st_p.tolist() == [0] * n1 and om.status == 0)}
            threads = orc.default_threads()
            n2 = min(2 * n1, len(frames_np) - 1)
            progress("parity checked; CPU baseline, %d frames on %d threads" % (n2, threads))
            fps_all, _ = cpu_baseline(frames_np, cfg, n2, threads, "sift" if sift else "orb")
            cpu_all = {"value": fps_all, "unit": "frames/s", "cores": threads, "kind": "port",
                       "host_cpus": os.cpu_count(),
                       "sample": "first %d frames, the restatement's per-pixel / per-row / per-descriptor loops on "
                                 "%d OpenMP threads (the box's CPU share; same results as one thread)" % (n2, threads)}
        out = {
            "metric": "stitched frames/sec @720p ORB-1000 on 1xMI355X; warp GB/s vs HBM roofline" if args.config == "720p_orb1000"
                      else "stitched frames/sec with the SIFT detector (knn-2 L2 + ratio) on 1xMI355X, %s" % args.config if sift
                      else "stitched frames/sec on 1xMI355X, %s; warp GB/s vs HBM roofline" % args.config,
            "value": world * K / elapsed, "unit": "frames/s", "n_gpus": 1 if same_gpu else world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if not stub else "none: stand-in context (RTVM_BENCH_STUB), nothing was stitched",
            "config": {"workload": args.config, "camera_path": args.path, "frame": [fh, fw], "canvas": [ch, cw], "nfeatures": nf,
                       "warp_mode": args.warp_mode, "batch": B, "hip_streams": 1 if args.single_stream else 4,
                       "streams": world, "parallelism": "one stream per GPU" if not same_gpu else "%d processes on one GPU" % world,
                       "last_frame": {"status": st, "keypoints": nk, "matches": nm},
                       "frames_queued": vstats["frames_queued"], "batches_queued": vstats["batches_queued"],
                       "batches_with_pending_frames": vstats["batches_with_pending_frames"],
                       "canvas_half_retries": vstats["retries"], "frames_retried": vstats["frames_retried"],
                       "frames_void": vstats["frames_void"], "canvas_pieces": vstats["canvas_pieces"],
                       "canvas_batch": args.canvas_batch or (B + 1) // 2},
            "roofline": roof, "warp_blend_stage": warp_blend, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "parity_checked": bool(all(v for k, v in parity.items() if k != "frames")) if parity else None,
            "parity": parity,
        }
        if world > 1:
            out["per_process_frames_per_s"] = rates
        if same_gpu:
            out["procs"] = world
        out.update(extras)
        if "dropin_pcie_inclusive" in extras or "sift_dropin_pcie_inclusive" in extras:
            # VideMosaic(frame) exactly as a caller of the reference writes it (main.py:17: both display flags on)
            out["dropin_default_flags"] = {
                "unit": "frames/s",
                "orb": extras.get("dropin_pcie_inclusive", {}).get("value"),
                "sift": extras.get("sift_dropin_pcie_inclusive", {}).get("value"),
                "note": "process_frame, one host frame per call, PCIe included, display flags left at the "
                        "reference's defaults (accepted, no window is opened)"}
        if args.kernel_table:
            rows = {k: {"launches": v[0], "total_ms": v[1], "avg_us": v[1] / v[0] * 1e3, "stream": v[2],
                        "us_per_frame": v[1] * 1e3 / (K1 if v[2] == 0 else K),
                        "algorithmic_bytes_per_frame": algorithmic_bytes(k, fh, fw, ch, cw, nf, roi_px, B)}
                    for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])}
            os.makedirs(os.path.dirname(os.path.abspath(args.kernel_table)), exist_ok=True)
            with open(args.kernel_table, "w") as f:
                json.dump({"frames": K, "canvas_half_frames": K1, "kernels": rows}, f, indent=1)
        print(json.dumps(out))
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
