#!/usr/bin/env python
"""bench.py — end-to-end FPS of the hot path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {0,1,2,3,4}] [--scaling {weak,strong}] [--extra 0,2,3,4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
(one process per GPU, RCCL) and prints the ranks' ONE JSON line; under an external launcher it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* as usual.

One "step" = one pass of the hot path over one batch.  The headline (`value`) is BASELINE.json configs[1]:
Lightweight-OpenPose (MobilenetDilated backbone) + PAF parser, batch 8 @ 368x432 per GPU:
    u8 HWC frames already resident in HBM -> (pre-processing fused into the first conv) -> conv stack on MFMA
    -> conf/paf fp32 maps in HBM -> PAF parser kernels -> hp_human lists written to pinned host memory.
Frames shard over GPUs with no steady-state collective; the only collective is the one-time RCCL broadcast of the weight
blob from rank 0 (outside the timed region).  `--scaling weak` (default): every rank processes its own full batch per
step; `--scaling strong`: the configuration's global batch is split contiguously over the ranks (32 -> 4,
64 -> 8 frames per GPU).  Timing: W untimed steps (followed by 0.3 s of the same loop, also untimed, so that the clocks
have settled whatever W is), then exactly K steps bracketed by barrier + torch.cuda.synchronize(), MAX over ranks, rank 0
prints ONE JSON line.

The other BASELINE configurations are measured the same way and reported under `workloads` in the same line
(`--extra`, default 0,2,3,4,5 at N = 1 - 5 = configs[1] behind data_type::kFLOAT, the fp32-faithful engine, as workloads["configs[1]/fp32"] -
and 3,4 strong-scaled at N > 1): configs[0] TinyVGG-V2 + PAF on a single image; configs[2] OpenPose-VGG19 + PAF, batch 16 @ 432x768;
configs[3] PoseProposal ResNet-50 + NMS decoder, batch 32 @ 384x384; configs[4] OpenPifPaf ResNet-50 + seed/grow
decoder, batch 64 @ 385x385 - each with its own `roofline` and (N = 1) `cpu_baseline`.

Parser input: the networks have synthetic (random) weights, so their own heat-maps contain no people.  Every timed step
runs the FULL conv stack AND parses seeded synthetic heat-maps with several people per frame that are resident in HBM
("injected": strictly more parser work, nothing skipped); the same loop parsing the network's own output is reported as
`fps_dnn_output`.

Extra objects: `roofline` for the dominant kernel (per-launch timestamps in schedule order; `bound` = the roof its arithmetic intensity
puts it under, with `frac_mfma` and `frac_hbm` both given; `committed_profile` = the same kernel in the newest committed rocprofv3 kernel
trace of this command, profiles/*_kernel_stats*.csv - read from the file, not measured in the run),
`parser_roofline` (HBM: frames/s of the parser alone x the compulsory bytes of SURVEY.md 8d / 8 TB/s), `cpu_baseline` (the
reference's CPU parser on this box's host cores, rank 0, N = 1), `single_pipe_fps` (one engine + parser pair, one batch in
flight), `h2d_inclusive` / top-level `value_h2d_inclusive` (the same step with network-sized u8 frames starting in pinned HOST memory - the
PCIe-inclusive rate, never `value`; measured on all ranks at once at N > 1), `from_host` (1280x720 camera frames through the GPU letterbox) and, at N > 1, `collective` (which backend
carried the start-up weight broadcast, its bytes and time).  Every secondary leg runs for a minimum wall time (0.3 s ramp +
>= 0.5 s timed) whatever --steps is, so the driver's short runs reproduce the long ones.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: the fp32 matrix pipe an HP_DTYPE_F32 (data_type::kFLOAT) engine computes on
# Independent engine+parser instances per GPU, one HIP stream each, batches round-robin over them.  Throughput depends
# on how those streams land on the runtime's hardware queues (measured on MI355X, tools/queue_probe.py, us/batch, config 1):
# 4 pipes on 2 queues (2+2) 606-617 | 3 pipes on 3 queues 657-667 | 6 pipes on 2 queues 668 | 4 pipes on 4 queues 790-820
# | 4 pipes on 1 queue 1030.  ROCm hands out hardware queues round-robin per created stream; every Pipe below creates an
# engine stream and then a parser stream, which puts the engine streams on alternating queues.
# compulsory parser bytes per frame (the network's output tensors read once; + <= 1.4 KB per human written)
PARSER_BYTES = {"paf": lambda h, w: 57 * (h // 8) * (w // 8) * 4, "ppn": lambda h, w: 855360 * (h // 32) * (w // 32) // 144,
                "pifpaf": lambda h, w: (17 * 5 + 19 * 9) * (((h - 1) // 8 + 1) * ((w - 1) // 8 + 1)) * 4}
PEAK_HBM_GBS = 8000.0
CONFIGS = {
    0: dict(label="configs[0]: TinyVGG-V2 + PAF parser, single 368x432 image (the reference's CPU-runnable plumbing case, src/fake)", arch="lw_openpose_vggtiny",
            w=432, h=368, batch=1, parser="paf", pipes=4, seed=20240, steps=400, people=(3,)),
    1: dict(label="configs[1]: Lightweight-OpenPose (MobilenetDilated) + PAF parser, batch 8 @ 368x432", arch="lw_openpose_mobilenet",
            w=432, h=368, batch=8, parser="paf", pipes=4, seed=20241, steps=400, people=(1, 2, 4, 8, 16, 3, 5, 6)),
    2: dict(label="configs[2]: OpenPose-COCO (VGG19) + PAF parser, batch 16 @ 432x768", arch="openpose_vgg19",
            w=768, h=432, batch=16, parser="paf", pipes=2, seed=20242, steps=24, people=(2, 4, 8, 16)),
    3: dict(label="configs[3]: PoseProposal ResNet-50 + NMS decoder, batch 32 @ 384x384", arch="pose_proposal_resnet50",
            w=384, h=384, batch=32, parser="ppn", pipes=8, seed=20243, steps=60, people=(1, 2, 3, 4)),
    4: dict(label="configs[4]: OpenPifPaf ResNet-50 + pif/paf seed-grow decoder, batch 64 @ 385x385", arch="pifpaf_resnet50",
            w=385, h=385, batch=64, parser="pifpaf", pipes=3, seed=20244, steps=16, people=(1, 2, 3, 4)),
    # configs[1] behind data_type::kFLOAT, the reference engine's default precision (include/hyperpose/operator/dnn/tensorrt.hpp:48): fp32
    # storage and fp32 matrix-pipe arithmetic (HP_DTYPE_F32, conv_fp32.hip), one launch per layer - the faithful mode, reported next to the
    # fp16 headline under workloads["configs[1]/fp32"] with its own roofline against the 157 TFLOP/s fp32 MFMA peak
    5: dict(label="configs[1] with data_type::kFLOAT (fp32 storage + fp32 MFMA): Lightweight-OpenPose + PAF parser, batch 8 @ 368x432", arch="lw_openpose_mobilenet",
            w=432, h=368, batch=8, parser="paf", pipes=4, seed=20241, steps=40, people=(1, 2, 4, 8, 16, 3, 5, 6), dtype="f32", key="configs[1]/fp32"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps of the headline workload (default: per config)")
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="headline workload (BASELINE.json configs index; 5 = configs[1] behind data_type::kFLOAT)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--extra", default=None, help="comma-separated configs also measured and reported under `workloads` ('' = none)")
    ap.add_argument("--pipes", type=int, default=0, help="engine+parser pairs per GPU (0 = per config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-from-host", action="store_true")
    ap.add_argument("--no-dnn-output", action="store_true", help="skip the second timed phase (parser fed by the network's own heat-maps)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ multi-GPU launch
def respawn(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks of this file under torch.distributed.run (RCCL over xGMI,
    rendezvous on 127.0.0.1) and hand their stdout (rank 0's JSON line) through."""
    from hyperpose_amd import _lib
    have = _lib.lib().hp_device_count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {max(have, 0)} HIP device(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rank_plan(cfg_batch: int, scaling: str, rank: int, world: int):
    """(frames this rank processes per step, global frames per step).  weak: every rank its own full batch; strong: the
    configuration's batch split contiguously (hyperpose_amd.dist.shard)."""
    from hyperpose_amd import dist as hd
    if scaling == "strong":
        _, cnt = hd.shard(cfg_batch, rank, world)
        return cnt, cfg_batch
    return cfg_batch, cfg_batch * world


# ------------------------------------------------------------------------------------------------ one workload
class Pipe:
    """engine + parser sharing one stream; at most one batch in flight per pipe."""

    def __init__(self, cfg, model, weights, injected, batch):
        from hyperpose_amd.engine import Engine
        from hyperpose_amd import parser as P
        from hyperpose_amd._lib import HpError
        self._HpError = HpError
        self.kind, self.batch = cfg["parser"], batch
        self.eng = Engine.from_model(model, weights, max_batch=batch, dtype=cfg.get("dtype", "f16"))
        self.stream = self.eng.stream
        outs = self.eng.outputs  # sorted by name = the parsers' argument order
        if self.kind == "paf":
            self.par = P.Paf(max_batch=batch)
            (_, self.s0, d0), (_, self.s1, d1) = outs
            self.dnn = (d0, d1)
        elif self.kind == "ppn":
            self.par = P.PoseProposal((cfg["w"], cfg["h"]), max_batch=batch)
            self.dnn = [d for _, _, d in outs]
            self.s0 = outs[0][1]
            e = outs[6][1]
            nn = int(round((e[0] // 17) ** 0.5))
            self.s1 = (17, nn, nn, e[1], e[2])
        else:
            self.par = P.PifPaf(cfg["h"], cfg["w"], max_batch=batch)
            self.dnn = (outs[0][2], outs[1][2])
            self.s0 = outs[1][1]  # pif [85, fh, fw]
        self.inj = injected
        self.busy = self.eng_only = False
        # frames the device decoders handed to the host statements (PoseProposal / PifPaf: hp_*_decode_flags > 0) and batches in which a
        # fixed-capacity list of the PAF parser overflowed (HP_ERR_CAPACITY; the reference's vectors are unbounded) - reported, never hidden
        self.declined_frames = self.capacity_truncations = self.frames_parsed = 0

    def submit(self, frames_dev, injected: bool, engine: bool = True, parser: bool = True):
        n = self.batch
        if engine:
            self.eng.enqueue_u8(frames_dev, n)
        if not parser:
            self.eng_only = True  # collect() waits for the stream
            return
        src = self.inj if injected else self.dnn
        if self.kind == "paf":
            self.par.enqueue(src[0], src[1], n, self.s0, self.s1, stream=self.stream)
        elif self.kind == "ppn":
            self.par.enqueue(src, n, self.s0, self.s1, stream=self.stream)
        else:
            self.par.enqueue(src[0], src[1], n, self.s0[1], self.s0[2], stream=self.stream)
        self.busy = True

    def collect(self):
        if self.eng_only:
            self.eng.synchronize()
            self.eng_only = False
        if not self.busy:
            return 0
        self.busy = False
        try:
            humans = self.par.collect()
        except self._HpError as e:
            if e.code != -3:  # HP_ERR_CAPACITY: counted; anything else is a failure of the run
                raise
            self.capacity_truncations += 1
            return 0
        self.frames_parsed += self.batch
        if self.kind != "paf":
            self.declined_frames += int(sum(1 for f in self.par.decode_flags(self.batch) if f > 0))
        return sum(len(h) for h in humans)


def run_loop(pipes, frames_dev, steps, injected):
    n_humans = 0
    for i in range(steps):
        p = pipes[i % len(pipes)]
        n_humans += p.collect()
        p.submit(frames_dev, injected)
    for p in pipes:
        n_humans += p.collect()
    return n_humans


def synth_inputs(cfg, batch, rank):
    """Per-rank seeded synthetic inputs of one workload: u8 frames + the parser's injected tensors (host numpy)."""
    from hyperpose_amd import synth
    idx = [k for k, v in CONFIGS.items() if v is cfg][0]
    rng = synth.rng_for(idx, salt=rank)
    frames = synth.images_u8(rng, batch, cfg["h"], cfg["w"])
    if cfg["parser"] == "paf":
        conf, paf, _ = synth.paf_maps(rng, batch, cfg["h"] // 8, cfg["w"] // 8, people=cfg["people"])
        maps = [conf, paf]
    elif cfg["parser"] == "ppn":
        maps = synth.ppn_maps(rng, batch, net=cfg["w"], grid=cfg["w"] // 32, people=cfg["people"])
    else:
        f = (cfg["w"] - 1) // 8 + 1
        maps = list(synth.pifpaf_maps(rng, batch, f, f, people=cfg["people"]))
    return frames, maps


def cpu_baseline(cfg, maps, budget_s=8.0):
    """The reference's CPU parser on this host's cores with the reference's shipping flags (-Ofast) and its own parallel
    model: one parser replica per pool thread, frames round-robin (include/hyperpose/utility/thread_pool.hpp:21,
    stream.hpp:139-144).  All three parsers are the reference's own sources compiled in oracle/_ref ("reference"); for PAF
    (src/paf.cpp + src/post_process.hpp behind the container shims of oracle/shim) the two OpenCV calls are the scalar
    restatements of oracle/paf_oracle.cpp, i.e. without OpenCV's SIMD kernels - the sample says so."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import loader
    ncpu = os.cpu_count() or 1
    threads = max(1, min(14, ncpu + 2, ncpu))
    kind = cfg["parser"]
    nb = maps[0].shape[0]
    if kind == "paf":
        if loader.have_ref_paf(fast=True):
            def one(i):
                return len(loader.ref_paf_process(maps[0][i], maps[1][i], fast=True, debug=False)[0])
            what, tag = ("reference src/paf.cpp + src/post_process.hpp compiled in oracle/_ref (its two OpenCV calls = scalar "
                         "restatements, no OpenCV SIMD), -Ofast -march=x86-64-v3"), "reference"
        else:
            def one(i):
                return len(loader.paf_process(maps[0][i], maps[1][i], fast=True)[0])
            what, tag = "restated reference PAF parser (no OpenCV SIMD), -Ofast -march=x86-64-v3", "port"
    elif kind == "ppn":
        if loader.ref_lib(fast=True) is None:
            return None
        def one(i):
            return len(loader.ref_ppn_process([m[i] for m in maps], net_w=cfg["w"], net_h=cfg["h"], fast=True))
        what, tag = "reference src/pose_proposal.cpp compiled in oracle/_ref, -Ofast", "reference"
    else:
        if loader.ref_lib(fast=True) is None:
            return None
        def one(i):
            return len(loader.ref_pifpaf_process(maps[0][i], maps[1][i], net_h=cfg["h"], net_w=cfg["w"], fast=True))
        what, tag = "reference src/pifpaf.cpp + src/pifpaf_decoder compiled in oracle/_ref, -Ofast", "reference"
    one(0)  # warm / build
    t0 = time.perf_counter()
    one(0)
    lat = time.perf_counter() - t0
    frames = int(max(threads * 2, min(400, budget_s / max(lat, 1e-4) * threads)))
    idx = [i % nb for i in range(frames)]
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(one, idx))
        dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 2), "unit": "frames/s (parser only; the reference's DNN stage is TensorRT and has no CPU path)",
            "cores": threads, "host_cores": ncpu, "kind": tag,
            "sample": f"{frames} injected heat-map frames of this workload (its {nb} frames cycled), {threads} threads on a host with "
                      f"{ncpu} logical cores, single-thread latency {lat * 1e3:.2f} ms/frame, {what}"}


def _host_pipeline_rate(model, weights, cfg, batch, pipes, steps, frame_wh, keep_ratio, min_s=0.6):
    import ctypes as C

    from hyperpose_amd import _lib
    from hyperpose_amd.pipeline import Pipeline
    w_, h_ = frame_wh
    nbytes = w_ * h_ * 3
    lib = _lib.lib()
    host = C.c_void_p()
    _lib.check(lib.hp_malloc_host(C.byref(host), C.c_size_t(nbytes * batch)))
    src = np.random.default_rng(7).integers(0, 256, nbytes * batch, dtype=np.uint8)
    C.memmove(host, src.ctypes.data, src.nbytes)
    ptrs = (C.POINTER(C.c_uint8) * batch)(*[C.cast(host.value + i * nbytes, C.POINTER(C.c_uint8)) for i in range(batch)])
    ws, hs = (C.c_int * batch)(*([w_] * batch)), (C.c_int * batch)(*([h_] * batch))
    pl = Pipeline(model, weights, max_batch=batch, n_pipes=pipes, keep_ratio=keep_ratio, max_frame_wh=frame_wh, parser=cfg["parser"],
                  dtype=cfg.get("dtype", "f16"))

    def run(n): # n more steps; the pipes stay full (a drain after every few steps would time the fill and the drain, not the pipeline)
        for _ in range(n):
            if pl.in_flight == pl.n_pipes:
                pl.collect()
            pl.submit_ptrs(ptrs, ws, hs, batch)

    def drain():
        while pl.in_flight:
            pl.collect()

    # clock ramp (0.3 s untimed, drained), then whole multiples of `chunk` steps until at least `min_s` have been timed - independent of
    # --steps; the timed region starts with empty pipes and ends when the last result is on the host
    chunk = max(2 * pipes, 4)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        run(chunk)
    drain()
    done = 0
    t0 = time.perf_counter()
    while done < steps or time.perf_counter() - t0 < min_s:
        run(chunk)
        done += chunk
    drain()
    dt = time.perf_counter() - t0
    pl.close()
    lib.hp_free_host(host)
    return batch * done / dt, nbytes * batch, done


def h2d_inclusive(model, weights, cfg, batch, pipes, steps):
    """SURVEY.md 8d / BASELINE.md 4.5: the same step with the u8 frames starting in pinned HOST memory at network size (one H2D copy
    per batch straight into the network's input buffer, then conv stack + parser on the network's own heat-maps, humans back on the
    host) - hp_pipeline_*.  PCIe-inclusive, therefore NOT `value`."""
    fps, nb, steps = _host_pipeline_rate(model, weights, cfg, batch, pipes, steps, (cfg["w"], cfg["h"]), False)
    return {"value": round(fps, 1), "unit": "frames/s", "steps": steps,
            "what": f"network-sized {cfg['w']}x{cfg['h']} u8 BGR frames in pinned host memory -> ONE H2D copy per batch ({nb / 1e6:.2f} MB) -> conv stack -> "
                    "parser (the network's own heat-maps) -> humans on the host; compare with fps_dnn_output (same work, frames resident)"}


def from_host(model, weights, cfg, batch, pipes, steps=8, frame_wh=(1280, 720)):
    """Camera-sized frames: per batch H2D copies, non_scaling_resize on the device, conv stack, parser, resume_ratio - the GPU form of
    hyperpose::stream (NOT `value`)."""
    fps, nb, steps = _host_pipeline_rate(model, weights, cfg, batch, pipes, steps, frame_wh, True)
    return {"value": round(fps, 1), "unit": "frames/s", "steps": steps,
            "what": f"{frame_wh[0]}x{frame_wh[1]} BGR frames in pinned host memory -> H2D ({nb / 1e6:.1f} MB per batch) -> "
                    "non_scaling_resize on the GPU -> conv stack -> parser (the network's own heat-maps) -> resume_ratio -> humans on the host"}


def kernel_label(tile: int):
    """(substring of the kernel symbol as rocprofv3 prints it, human-readable label) for a profile row's `tile` code
    (conv_kernels.hpp: conv_mfma_tile; engine.cpp: hp_engine_profile)."""
    sep = {7: ("sepconv_small_kernel<32,1,2>", "32-channel separable block, all channels of an 8x8 tile in LDS"),
           1: ("sepconv_small_kernel<", "separable block -> 128 channels, all channels of an 8x8 tile in LDS"),
           2: ("sepconv_small_kernel<", "separable block -> 128 channels, stride 2, all channels of a tile in LDS"),
           3: ("sepconv_slot_kernel<2,1,2,1,128,64>", "separable block 128 -> 256, stride 2, half-CU form"),
           4: ("sepconv_slot_kernel<1,2,1,1,256,64>", "separable block 256 -> 256, half-CU form"),
           5: ("sepconv_pipe3_kernel<1,false,true>", "separable block 256 / 512 -> 512: 12x8 pixels x all 512 output channels per block, eight wavefronts; per 64-channel "
               "chunk the depthwise taps of chunk k+1 are issued between the pointwise MFMAs of chunk k in the SAME wavefront (one stream of 24 slots)"),
           6: ("sepconv_pipe3_kernel<2,false,true>", "separable block 512 -> 512, dilation 2, same form"),
           20: ("sepconv_pair_kernel<32,64,128>", "the stem's separable blocks 32 -> 64 and 64 -> 128 (stride 2) in one launch, the 64-channel "
                "tensor between them in LDS only")}
    chain = {1: "false,0", 2: "false,1", 3: "false,2", 10: "true,0", 13: "true,3"}
    if tile >= 32000000:
        bm, bn = (tile - 32000000) // 1000, tile % 1000
        wm, wn = (1, 4) if (bm, bn) == (64, 128) else (2, 2)
        return (f"conv32_kernel<{bm},{bn},{wm},{wn}>", f"conv32_kernel<BM={bm},BN={bn}> (fp32 implicit GEMM on v_mfma_f32_32x32x2_f32: {bm} cout x {bn} pixels per block, "
                "A and B staged through LDS in fp32, K-steps of 16 channels)")
    if tile >= 9000000:
        v = tile - 9000000
        m, pj, mr, a3 = v // 1000 * 64, v // 100 % 10, v // 10 % 10 * 64, v % 10
        own, pj = pj >= 2, pj % 2
        what = (f"{'own 1x1 reduction -> ' if own else ''}{'3x3 -> ' if a3 else ''}1x1 {m} -> {4 * m} + {'projection' if pj else 'shortcut'}"
                f"{f' -> 1x1 {4 * m} -> {mr} of the next block' if mr else ''} in one launch, 8x8 pixels per block, intermediates in LDS")
        if m == 64:
            t = lambda x: "true" if x else "false"
            return (f"bottleneck64_kernel<{mr},{t(a3)},{t(pj)},{t(own)}>", "bottleneck64_kernel (" + what + ")")
        if a3:
            return (f"bottleneck_kernel<128,{mr},true>", "bottleneck_kernel (" + what + ")")
        return (f"bottleneck128_kernel<{mr}>", "bottleneck128_kernel (" + what + ")")
    if tile >= 7000000:
        v = tile - 7000000
        return (f"conv_chain_kernel<{chain.get(v, '')},", "conv_chain_kernel ([1x1 ->] 3x3 -> 3x3 [+ residual] on 128 channels in one launch: 8x8 output pixels x all "
                "128 channels per block, intermediates in LDS, weights in MFMA-fragment order straight from L2)")
    if tile >= 6000000 and tile % 1000 in (9, 25, 49):
        v = tile - 6000000
        cin, taps = v // 1000, v % 1000
        ks = int(round(taps ** 0.5))
        ck, nbuf = (128, 1) if cin == 128 else (64, 1 if cin == 64 else 2)
        return (f"conv_direct_kernel<{ks},{ck},{nbuf}>",
                f"conv_direct_kernel<KS={ks},CK={ck},NBUF={nbuf}> ({ks}x{ks} taps, {cin} input channels in {cin // ck} chunk(s); 8 wavefronts, 128 cout x 16x12 px "
                "per block, halo tile of a chunk in LDS for all taps, weights in MFMA-fragment order straight from L2)")
    if tile >= 6000000:
        return ("mlp_head", "mlp_head(_pair)_kernel (1x1 K1 -> 512 relu -> 1x1 512 -> 19 | 38, hidden tensor in registers)")
    if 5200000 <= tile < 5300000:
        tm, ntp = (tile - 5200000) // 1000, tile % 1000
        return (f"conv1x1_big_kernel<{tm},{ntp}>", f"conv1x1_big_kernel<TM={tm},NTP={ntp}> (pixel-block GEMM: {32 * ntp} pixels x {128 * tm} output channels per block, "
                "weights from L2 in fragment order, activations through producer wavefronts + LDS)")
    if 5100000 <= tile < 5200000:
        return ("conv1x1_small_kernel", "conv1x1_small_kernel (64 pixels x all input channels in LDS, fragment-ordered weights from L2)")
    if tile >= 5000000:
        return ("conv3x3_direct_kernel<128,", "conv3x3_direct_kernel<CIN=128> (64 cout x 16x12 px tile, input halo tile in LDS, weights in MFMA-fragment order straight from L2)")
    if tile >= 4000000:
        return sep.get(tile - 4000000, ("sepconv", "fused depthwise 3x3 + pointwise 1x1"))
    return (f"conv_mfma_kernel<{tile // 1000},{tile % 1000},", f"conv_mfma_kernel<BM={tile // 1000},BN={tile % 1000}> (implicit GEMM, A and B staged through LDS)")


def pmc_traffic(symbol_key: str, tag: str):
    """HBM bytes per launch of the kernel from the committed rocprofv3 PMC passes (profiles/<round>_pmc_traffic*.json):
    newest round first; kernel names compared with blanks removed."""
    import glob
    want = symbol_key.replace(" ", "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{tag}.json")), reverse=True):
        try:
            pmc = json.load(open(path))["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        for name, d in pmc.items():
            if want in name.replace(" ", "") and "hbm_bytes_per_launch" in d:
                return round(d["hbm_bytes_per_launch"]), os.path.basename(path)
    return None, None


def rocprof_avg_us(symbol_key: str, tag: str):
    """Average duration of the kernel in the NEWEST committed `rocprofv3 --kernel-trace --stats` summary of this bench command
    (profiles/<round>_kernel_stats<tag>.csv): (us, file name), or (None, file name) unless the key names EXACTLY ONE row of that file -
    an ambiguous or missing key (a renamed kernel, a template instance the key does not pin down) must not produce a number."""
    import csv
    import glob
    want = symbol_key.replace(" ", "")
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_stats{tag}.csv")), reverse=True)
    if not paths:
        return None, None
    try:
        with open(paths[0], newline="") as f:
            hits = [row for row in csv.DictReader(f) if want in row.get("Name", "").replace(" ", "")]
        if len(hits) == 1:
            return float(hits[0]["AverageNs"]) / 1e3, os.path.basename(paths[0])
    except (OSError, KeyError, ValueError):
        pass
    return None, os.path.basename(paths[0])


RIDGE_FLOP_PER_BYTE = 2500.0e12 / 8000.0e9  # 312.5: below it a kernel's binding roof is HBM, above it the matrix pipe


def roofline(pipe, batch, cfg_index, frames_dev=None, peak_tflops=PEAK_F16_TFLOPS):
    """Per-launch timestamps on the engine stream with the schedule run in order (hp_engine_profile_sequence: every kernel sees the
    cache state of a real inference, which is what rocprofv3's per-kernel averages over this bench see too).  achieved = algorithmic
    FLOPs of the dominant kernel's launches / their summed duration.  `back_to_back_us` is the same kernel re-launched 20 times in a
    row (weights warm in L2) for comparison."""
    iters = 20 if cfg_index == 1 else 4
    prof = pipe.eng.profile(batch, iters=iters, in_sequence=True)
    warm = pipe.eng.profile(batch, iters=iters) if cfg_index == 1 else None
    mfma = [p for p in prof if p["tile"] != 0]
    by_tile = {}
    for p in mfma:
        d = by_tile.setdefault(p["tile"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
        d["ms"] += p["ms"]
        d["flops"] += p["flops"]
        d["bytes"] += p["bytes"]
        d["n"] += 1
    dom_tile, dom = max(by_tile.items(), key=lambda kv: kv[1]["ms"])
    tot_ms = sum(p["ms"] for p in prof)
    mfma_ms = sum(p["ms"] for p in mfma)
    mfma_fl = sum(p["flops"] for p in mfma)
    tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9   # ALGORITHMIC bytes of the launches (input + output + weights once, engine.cpp st.bytes)
    intensity = dom["flops"] / dom["bytes"]
    ridge = peak_tflops * 1e12 / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if intensity >= ridge else "hbm"
    frac_mfma, frac_hbm = tflops / peak_tflops, gbs / PEAK_HBM_GBS
    key, label = kernel_label(dom_tile)
    tag = "" if cfg_index == 1 else "_config1_fp32" if cfg_index == 5 else f"_config{cfg_index}"
    traffic, src = pmc_traffic(key, tag)
    prof_us, prof_src = rocprof_avg_us(key, tag)
    out = {
        # the roof that binds THIS kernel: its arithmetic intensity against the ridge of 2.5 PFLOP/s / 8 TB/s = 312.5 FLOP/byte; `achieved`
        # / `peak` / `frac` are quoted on that roof, both fractions are given below
        "bound": bound,
        "achieved": round(tflops, 2) if bound == "mfma" else round(gbs, 1),
        "peak": peak_tflops if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round(frac_mfma if bound == "mfma" else frac_hbm, 4),
        "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 2), "mfma_peak_tflops": peak_tflops,
        "frac_mfma": round(frac_mfma, 4), "achieved_tflops": round(tflops, 2),
        "frac_hbm": round(frac_hbm, 4), "achieved_gbs": round(gbs, 1),
        # the largest fraction of the MFMA peak this kernel could reach at 8 TB/s given its intensity (1 when it is right of the ridge)
        "mfma_frac_ceiling_at_hbm_peak": round(min(1.0, intensity / ridge), 4),
        "traffic": traffic, "traffic_source": src,
        "kernel": label,
        "launches_per_step": dom["n"], "avg_launch_us": round(dom["ms"] / dom["n"] * 1e3, 2),
        "flops_per_launch": round(dom["flops"] / dom["n"]), "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
        # NOT measured in this run: the kernel's average duration in the newest COMMITTED rocprofv3 kernel trace of `bench.py --config N
        # --pipes 1` (the tracer adds ~1 us per launch) and this run's FLOPs over it - for the reader who recomputes the fraction from
        # profiles/; null unless the kernel's name matches exactly one row of that file
        "committed_profile": {"source": prof_src, "avg_launch_us": None if prof_us is None else round(prof_us, 2),
                              "frac_mfma": None if prof_us is None else round(dom["flops"] / dom["n"] / (prof_us * 1e-6) / 1e12 / peak_tflops, 4),
                              "frac_hbm": None if prof_us is None else round(dom["bytes"] / dom["n"] / (prof_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)},
        "all_mfma_convs": {"achieved": round(mfma_fl / (mfma_ms * 1e-3) / 1e12, 2), "frac": round(mfma_fl / (mfma_ms * 1e-3) / 1e12 / peak_tflops, 4),
                           "ms_per_step": round(mfma_ms, 4), "launches_per_step": len(mfma)},
        "serial_layer_ms_per_step": round(tot_ms, 4),
        "non_mfma_ms_per_step": round(tot_ms - mfma_ms, 4),
    }
    # the kernel with the second-largest share: on configs[1] the 512-output separable block and the 128-channel chain are within a
    # few percent of each other and swap places between runs / boxes - both are always on the line
    rest = sorted(((t, d) for t, d in by_tile.items() if t != dom_tile), key=lambda kv: -kv[1]["ms"])
    if rest:
        t2, d2 = rest[0]
        tf2, gb2, in2 = d2["flops"] / (d2["ms"] * 1e-3) / 1e12, d2["bytes"] / (d2["ms"] * 1e-3) / 1e9, d2["flops"] / d2["bytes"]
        out["runner_up"] = {"kernel": kernel_label(t2)[1], "launches_per_step": d2["n"], "avg_launch_us": round(d2["ms"] / d2["n"] * 1e3, 2),
                            "share_of_serial_step": round(d2["ms"] / tot_ms, 4), "intensity_flop_per_byte": round(in2, 1),
                            "bound": "mfma" if in2 >= ridge else "hbm", "frac_mfma": round(tf2 / peak_tflops, 4), "frac_hbm": round(gb2 / PEAK_HBM_GBS, 4)}
    out["share_of_serial_step"] = round(dom["ms"] / tot_ms, 4)
    if warm is not None:
        out["back_to_back_us"] = round(sum(p["ms"] for p in warm if p["tile"] == dom_tile) / dom["n"] * 1e3, 2)
    if frames_dev is not None:
        # the same timestamps with the parser in the loop, as in the timed region with one pipe (and as rocprofv3 sees the kernel in
        # profiles/r02_kernel_stats*.csv, collected from `bench.py --pipes 1`): the parser's kernels run between two engine passes and
        # evict the weights from L2
        ms = n = 0
        for _ in range(iters):
            pipe.submit(frames_dev, True, engine=False, parser=True)
            pipe.collect()
            for q in pipe.eng.profile(batch, iters=1, in_sequence=True):
                if q["tile"] == dom_tile:
                    ms += q["ms"]
                    n += 1
        out["avg_launch_us_parser_in_loop"] = round(ms / n * 1e3, 2)
    return out


def measure(cfg_index, args, rank, world, dev, scaling, steps, warmup, headline):
    """Time one BASELINE configuration; returns the dict of its numbers (identical on every rank where it matters)."""
    import torch
    import torch.distributed as dist

    from hyperpose_amd import _lib
    from hyperpose_amd import dist as hd
    from hyperpose_amd.engine import Model

    cfg = CONFIGS[cfg_index]
    batch, global_batch = rank_plan(cfg["batch"], scaling, rank, world)
    model = Model(cfg["arch"], cfg["w"], cfg["h"])
    # one-time weight broadcast from rank 0 over RCCL/xGMI (the only collective of the whole job)
    w_host = hd.broadcast_weights(model.init_weights(cfg["seed"]) if rank == 0 else None, model.n_weights, rank, world,
                                  device=hd.collective_device(dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    coll = None
    if world > 1:
        coll = {"backend": hd.LAST_BROADCAST.get("backend"), "bcast_bytes": hd.LAST_BROADCAST.get("bytes"), "bcast_ms": hd.LAST_BROADCAST.get("ms"),
                "what": "one broadcast of the fp32 weight blob from rank 0 at start-up (outside the timed region); the steady state has no collective"}
    n_pipes = args.pipes if args.pipes > 0 else cfg["pipes"]
    res = {"workload": cfg["label"], "frames_per_gpu_per_step": batch, "global_batch": global_batch, "scaling": scaling,
           "pipes_per_gpu": n_pipes, "gflop_per_frame": round(model.flops_per_frame / 1e9, 2)}
    if coll:
        res["collective"] = coll
    if batch == 0:  # strong scaling with more ranks than frames: this rank idles but still takes part in the barriers
        pipes, frames_dev, maps = [], None, None
    else:
        frames, maps = synth_inputs(cfg, batch, rank)
        frames_dev = _lib.DevBuf.from_numpy(frames)
        inj = [_lib.DevBuf.from_numpy(m) for m in maps]
        pipes = [Pipe(cfg, model, w_host, inj, batch) for _ in range(max(1, n_pipes))]

    def timed(injected):
        if pipes:
            run_loop(pipes, frames_dev, warmup, injected)
            # the GPU's clocks take a few hundred ms of load to settle (100 steps right after a short warm-up measure
            # ~12 % low): keep the same loop running, untimed, until 0.3 s have passed since the warm-up began
            t_ramp = time.perf_counter()
            while time.perf_counter() - t_ramp < 0.3:
                run_loop(pipes, frames_dev, len(pipes), injected)
        barrier()
        t0 = time.perf_counter()
        nh = run_loop(pipes, frames_dev, steps, injected) if pipes else 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dt = hd.max_over_ranks(dt, world, device=hd.collective_device(dev))
        barrier()
        return dt, nh

    dt, n_humans = timed(True)
    dt_dnn = None if (args.no_dnn_output or not headline) else timed(False)[0]
    total_frames = global_batch * steps
    fps = total_frames / dt
    # host fall-backs / truncations over every step this rank ran (warm-up and both timed phases): frames a device decoder declined and
    # handed to the host statements (PoseProposal / PifPaf), batches with an overflowed PAF list
    parsed = sum(p.frames_parsed for p in pipes)
    res.update({"device_declined_frames": sum(p.declined_frames for p in pipes), "capacity_truncations": sum(p.capacity_truncations for p in pipes),
                "frames_parsed_for_these_counts": parsed})
    res.update({"value": round(fps, 1), "unit": "frames/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
                "humans_per_step": n_humans / max(1, steps),
                "fps_dnn_output": round(total_frames / dt_dnn, 1) if dt_dnn else None,
                "conv_tflops_end_to_end": round(fps * model.flops_per_frame / 1e12, 2),
                "conv_frac_of_mfma_peak_end_to_end": round(fps * model.flops_per_frame / 1e12 / (PEAK_F32_TFLOPS if cfg.get("dtype") == "f32" else PEAK_F16_TFLOPS) / world, 4),
                "dtype": "f32 (fp32 storage, fp32 MFMA)" if cfg.get("dtype") == "f32" else "f16 (fp32 accumulate)"})
    if rank == 0 and pipes and not args.no_roofline:
        # where the step's time goes: the parser alone (injected maps: GPU kernels + the host tail in collect) and the conv stack
        # alone, each through ONE pipe, next to the end-to-end step above (in which several pipes overlap them)
        p0 = pipes[0]

        def leg(eng_on, par_on, min_s=0.5):
            """ms per step of ONE pipe running the given halves serially: 0.2 s ramp, then >= min_s timed (independent of --steps)."""
            t_r = time.perf_counter()
            while time.perf_counter() - t_r < 0.2:
                p0.submit(frames_dev, True, engine=eng_on, parser=par_on)
                p0.collect()
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while n < 4 or time.perf_counter() - t0 < min_s:
                p0.submit(frames_dev, True, engine=eng_on, parser=par_on)
                p0.collect()
                n += 1
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        res["parser_only_ms_per_step"] = round(leg(False, True), 4)
        res["engine_only_ms_per_step"] = round(leg(True, False), 4)
        res["parser_share_of_serial_step"] = round(res["parser_only_ms_per_step"] / (res["parser_only_ms_per_step"] + res["engine_only_ms_per_step"]), 4)
        # one engine + parser pair, one batch in flight at a time: what a caller that does not pipeline batches gets
        res["single_pipe_fps"] = round(batch / (leg(True, True) * 1e-3), 1)
        pb = PARSER_BYTES[cfg["parser"]](cfg["h"], cfg["w"])
        gbs = batch * pb / (res["parser_only_ms_per_step"] * 1e-3) / 1e9
        res["parser_roofline"] = {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 5),
                                  "bytes_per_frame": pb, "frames_per_s_parser_alone": round(batch / (res["parser_only_ms_per_step"] * 1e-3), 1),
                                  "what": "compulsory bytes (the network's output tensors read once, SURVEY.md 8d) x frames/s of the parser alone "
                                          "(one pipe, injected maps, GPU kernels + collect); latency-bound at these sizes, not bandwidth-bound"}
        del p0
    if rank == 0 and pipes:
        if not args.no_roofline:
            res["roofline"] = roofline(pipes[0], batch, cfg_index, frames_dev, PEAK_F32_TFLOPS if cfg.get("dtype") == "f32" else PEAK_F16_TFLOPS)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, maps)
    if not args.no_from_host:
        # the PCIe-inclusive step, on EVERY rank at once (each with its own pinned frames and pipes, all feeding from
        # the same host): the aggregate is the sum of the ranks' rates over a common window, which is what the host feed limits at N > 1
        del pipes[:]
        if world > 1:
            barrier()
        mine = h2d_inclusive(model, w_host, cfg, batch, n_pipes, 8) if batch else {"value": 0.0, "unit": "frames/s", "steps": 0, "what": "idle rank"}
        total = hd.sum_over_ranks(mine["value"], world, device=hd.collective_device(dev))
        if world > 1:
            barrier()
        if rank == 0:
            mine["value_per_rank0"] = mine["value"]
            mine["value"] = round(total, 1)
            mine["n_gpus"] = world
            res["h2d_inclusive"] = mine
            if headline and world == 1:
                res["from_host"] = from_host(model, w_host, cfg, batch, n_pipes)
    del pipes
    return res, model


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

    from hyperpose_amd import _lib
    from hyperpose_amd import dist as hd

    torch.cuda.set_device(local_rank)
    _lib.init(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        backend = hd.init_for_gpu(dev)  # RCCL, or - agreed by all ranks - gloo if it cannot be brought up (the hot path has no collective)

    cfg = CONFIGS[args.config]
    steps = args.steps if args.steps is not None else cfg["steps"]
    head, model = measure(args.config, args, rank, world, dev, args.scaling, steps, args.warmup, True)
    out = {
        "metric": "end-to-end FPS (preproc+DNN+PAF parse) @ 368x432" if args.config in (0, 1) else f"end-to-end FPS (preproc+DNN+parse), {cfg['label']}",
        "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f16 (fp32 accumulate; parsers fp32)", "data": "synthetic",
        "config": {"workload": cfg["label"] + " per GPU, frames u8 HWC resident in HBM, humans written to pinned host memory",
                   "global_batch": head["global_batch"], "frames_per_gpu_per_step": head["frames_per_gpu_per_step"],
                   "parallelism": f"frame-sharded x{world}, no steady-state collective",
                   "pipes_per_gpu": head["pipes_per_gpu"], "frames_in_flight_per_gpu": head["pipes_per_gpu"] * head["frames_per_gpu_per_step"],
                   "parser_input": "injected synthetic heat-maps (several people per frame); the full conv stack also runs",
                   "humans_per_step": head["humans_per_step"], "gflop_per_frame": head["gflop_per_frame"]},
        "fps_dnn_output": head["fps_dnn_output"],
        "conv_tflops_end_to_end": head["conv_tflops_end_to_end"],
        # SURVEY.md 8d's strictest reading: the same step with the u8 frames starting in pinned HOST memory (one H2D copy per batch) and the
        # parser fed by the network's own heat-maps; aggregate over the ranks.  `value` keeps the contract's definition (inputs resident in HBM).
        "value_h2d_inclusive": head.get("h2d_inclusive", {}).get("value"),
        "device_declined_frames": head["device_declined_frames"], "capacity_truncations": head["capacity_truncations"],
    }
    out["collective_backend"] = backend
    for k in ("roofline", "parser_roofline", "cpu_baseline", "single_pipe_fps", "h2d_inclusive", "from_host", "parser_only_ms_per_step", "engine_only_ms_per_step",
              "parser_share_of_serial_step", "collective"):
        if k in head:
            out[k] = head[k]
    extra = args.extra
    if extra is None:
        extra = "0,2,3,4,5" if world == 1 else "3,4"
        if args.config != 1:
            extra = ""
    workloads = {}
    for tok in [t for t in extra.split(",") if t.strip()]:
        k = int(tok)
        if k == args.config or k not in CONFIGS:
            continue
        c = CONFIGS[k]
        scal = "strong" if world > 1 else "weak"
        r, _ = measure(k, args, rank, world, dev, scal, c["steps"], max(2, min(args.warmup, c["steps"] // 4)), False)
        r["n_gpus"] = world
        workloads[c.get("key", f"configs[{k}]")] = r
    if workloads:
        out["workloads"] = workloads
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
