#!/usr/bin/env python
"""Throughput of the hot path on MI355X: utterances/s of the VoiceSplit mask-prediction path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--mode train|forward|longform]

Workload (BASELINE.json metric "utterances/sec (3 s clips, B=64) fwd+bwd"): B=64 synthetic
[64,301,601] spectrograms + 256-d d-vectors per GPU, random-init weights of the reference
architecture, batch-statistics BatchNorm (model.train(), train.py:84).  A "step" is what one
iteration of train.py:85-111 does: forward (with the tape), the reference's SI-SNR loss through a
GPU iSTFT (train.py:95-108, vs_sisnr_loss; --loss fixed replaces it by a fixed upstream gradient),
backward, the gradient exchange (one flat-bucket all-reduce, a no-op at N=1) and the Adam update.  Inputs
are resident in HBM before the timed region.  --mode forward times BASELINE configs[1] (forward only,
eval-mode BatchNorm); --mode longform times configs[4] (30 s clips cut into 301-frame windows, 256 windows
per forward batch, clips sharded over the ranks, no collective).

N>1: one rank per GPU over RCCL, data parallel, 64 utterances per rank ("weak" scaling), one all-reduce of the
75.5 MB gradient bucket per step over xGMI.  The driver's `python -m torch.distributed.run ... bench.py --gpus N`
runs the ranks directly; a plain `python bench.py --gpus N` (no RANK / WORLD_SIZE in the environment) launches the
same command itself on 127.0.0.1 and fails loudly when fewer than N devices are visible.

Rank 0 prints ONE JSON line (the last line on stdout).  Training mode (default): the whole-job utterances/s of BASELINE
configs[2] -- the bf16 configuration (channels-last bf16 activations / tape, csrc/conv_nhwc.hip; mask MSE <= 1e-4 vs the
reference, tests/test_gpu_bf16.py) -- plus
  roofline     -- the dominant kernel (5x5 64->64 conv: cnn3..cnn7 forward and their data gradients, 10 launches
                  per training step): algorithmic FLOPs per launch / mean launch time from HIP events recorded
                  inside the timed region; the weight-gradient kernel and the LSTM input GEMM next to it
  forward      -- BASELINE configs[1] timed in the same process: forward only in the fp32-class arithmetic (fp32 results
                  from split-f16 MFMAs, <= 1e-4 relative vs the reference); forward_bf16: the same in bf16
  fp32_class   -- the same training step in the fp32-class arithmetic (the headline of rounds 1-2), own roofline figures
  fp32_strict  -- the same step on the fp32 matrix cores (bitwise an fmaf chain), 3 steps
  longform     -- BASELINE configs[4] as a short sub-leg (51 clips of 30 s = 510 windows per step, 256 windows per forward batch),
                  in bf16 and in the fp32-class arithmetic (`--mode longform` is the full line of that configuration)
  cpu_baseline -- the oracle (torch CPU restatement of the reference) timed on the host cores, B=1 and B=4; in training mode
                  forward + autograd backward from a STAND-IN loss (a fixed d(loss)/d(mask)).
--conv-math f16x3 makes the fp32-class step the line and bf16 the sub-object.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic work per utterance at T=301, F=601; FLOP = 2*MAC
T_FRAMES, N_FREQ, EMB = 301, 601, 256
GFLOP_CONV5X5 = 2 * 64 * 64 * 25 * T_FRAMES * N_FREQ / 1e9       # 37.05 per layer per utterance
GFLOP_LSTM_GEMM = 2 * 4808 * 3200 * T_FRAMES / 1e9                # 9.26 per utterance (d-vector columns folded away)
GFLOP_FWD_TOTAL = 206.995
PEAK_FP32_MFMA_TFLOPS = 157.3                                      # MI355X_MICROARCH.md chip table
PEAK_F16_MFMA_TFLOPS = 2500.0                                      # dense f16/bf16 MFMA (same table)
PEAK_HBM_GBS = 8000.0
LONG_FRAMES = 3001                                                 # 30 s at hop 160 / 16 kHz

MATH_LABEL = {"fp32": "fp32 MFMA", "f16x3": "fp32 I/O, split-f16 MFMA (3 f16 products per fp32 product)",
              "bf16": "channels-last bf16 activations / tape, bf16 MFMA, fp32 accumulate / statistics / master weights"}


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def committed_pmc_traffic(tag, kernel_substr, instance=None, source=None):
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 PMC summary of THIS command
    (profiles/<tag>_rocprof/pmc_per_kernel.csv, written by tools/profile_gpu.sh + summarize_prof.py in
    separate --pmc passes): 2 x FETCH_SIZE + WRITE_SIZE, in GB.  The x2 is calibrated, not assumed: a 1 GiB
    read reports FETCH_SIZE = 0.5 GiB for dword and for 16-byte loads alike on this chip and tool version, and a
    1 GiB write reports WRITE_SIZE = 1 GiB (profiles/r02_calibration/).  bench.py itself cannot collect
    counters.  instance: a substring of the template arguments that picks ONE instantiation of the kernel (the forward
    conv and the dy-form data gradient are two instances of nhwc_conv_kernel with different traffic); without it the
    instance with the most dispatches.  source: the kernel's file under voicesplit_amd/csrc/ -- the figure is reported only
    while that file is byte-identical to the one the counters were collected on (sources.sha256 beside the CSV, written
    on the GPU box by tools/profile_gpu.sh); a profile without that record, or of an older source, gives (None, why).
    Returns (GB or None, provenance string)."""
    import csv
    import glob
    rounds = sorted({os.path.basename(p).split("_")[0] for p in glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{tag}_rocprof"))}, reverse=True)
    for rnd in rounds:          # the newest round's counters first
        d = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_rocprof")
        path = os.path.join(d, "pmc_per_kernel.csv")
        if not os.path.isfile(path):
            continue
        if source is not None:
            rec = os.path.join(d, "sources.sha256")
            if not os.path.isfile(rec):
                return None, f"{rnd}: no sources.sha256 beside the counters (profile predates the record)"
            want = {ln.split()[1].lstrip("*"): ln.split()[0] for ln in open(rec) if ln.strip()}
            cur = os.path.join(ROOT, "voicesplit_amd", "csrc", source)
            if want.get(source) != (_sha256(cur) if os.path.isfile(cur) else None):
                return None, f"{rnd}: {source} has changed since the counters were collected"
        best = None
        for r in csv.DictReader(open(path)):
            if kernel_substr in r["kernel"] and (instance is None or instance in r["kernel"]) and r.get("fetch_GB_x2") and r.get("write_GB"):
                n = int(r["dispatches"])
                if best is None or n > best[0]:
                    best = (n, float(r["fetch_GB_x2"]) + float(r["write_GB"]))
        if best is not None:
            return round(best[1], 2), rnd
    return None, "no committed profile of this command"


def _pick_threads():
    """Thread count for the CPU leg: the box may expose far more logical cores than its cgroup can
    run, where one thread per core is pathological (86 s/utterance at 256 threads on the first
    run).  Calibrate on one 5x5 conv layer and keep the fastest candidate."""
    import torch.nn.functional as F
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    x = torch.rand(1, 64, T_FRAMES, N_FREQ)
    w = torch.rand(64, 64, 5, 5)
    best = (None, 1e30)
    for n in (4, 8, 16, 32, 64, 128, 256):
        if n > avail:
            break
        torch.set_num_threads(n)
        with torch.no_grad():
            F.conv2d(x, w, padding=2)
            t0 = time.perf_counter()
            F.conv2d(x, w, padding=2)
            dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (n, dt)
        if dt > 4 * best[1]:
            break
    return best[0] or 1, avail


def cpu_baseline(mode, seconds_budget=14.0):
    """Oracle (reference restatement, torch CPU ops, nn.LSTM) on the host cores: forward, or forward + autograd
    backward from the same fixed upstream gradient, at B=1 (BASELINE configs[0]) and B=4."""
    from oracle import reference_backward as RB
    from oracle import reference_forward as R
    threads, avail = _pick_threads()
    torch.set_num_threads(threads)
    dims = R.default_dims()
    sd = R.build_state_dict(dims, 0)
    what = "forward + autograd backward from a stand-in loss (fixed d(loss)/d(mask), no iSTFT / SI-SNR head)" if mode == "train" else "forward"

    def run(B, budget):
        x, dvec = R.synthetic_inputs(B, T_FRAMES, dims, 0)
        w = RB.loss_weights(B, T_FRAMES, dims["fc2_dim"], 0)

        def once():
            if mode == "train":
                RB.gradients(sd, x, dvec, w, act="mish", training=True)
            else:
                with torch.no_grad():
                    R.forward(sd, x, dvec, act="mish")

        once()                                                     # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            once()
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= 30:
                break
        return n, el

    n1, el1 = run(1, seconds_budget)
    n4, el4 = run(4, seconds_budget * 0.6)
    fwd_b1 = None
    if mode == "train":
        # BASELINE configs[0]: the reference's own CPU-runnable case -- ONE 3 s utterance, forward only (what test.py / validation() run)
        mode_keep, mode = mode, "forward"
        try:
            nf, elf = run(1, seconds_budget * 0.4)
        finally:
            mode = mode_keep
        fwd_b1 = {"value": round(nf / elf, 4), "unit": "utterances/s",
                  "sample": f"BASELINE configs[0]: {nf} x forward of one [1,301,601] utterance in {elf:.1f} s, same threads"}
    out = {"value": round(n1 / el1, 4), "unit": "utterances/s", "cores": threads, "kind": "port",
           "metric": ("utterances/sec fwd+bwd with a stand-in loss (the model's share of the step: the SI-SNR head is not timed on the CPU)"
                      if mode == "train" else "utterances/sec forward"),
           "value_b4": round(4 * n4 / el4, 4),
           "sample": f"{n1} x {what} of one [1,301,601] utterance in {el1:.1f} s (value) and {n4} x the same at B=4 in "
                     f"{el4:.1f} s (value_b4), fp32, torch {torch.__version__} CPU ops, {threads} threads "
                     f"(fastest of a 4..256 sweep; {avail} logical cores visible)"}
    if fwd_b1 is not None:
        out["forward_b1"] = fwd_b1
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run the driver's own command line on this node."""
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node -- refusing to run a smaller job "
                         f"under the same name (one rank per GPU, no CPU fallback)")
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="utterances (windows) per GPU and forward batch: 64; 256 in longform mode")
    ap.add_argument("--mode", default="train", choices=["train", "forward", "longform"])
    ap.add_argument("--clips", type=int, default=128, help="longform mode: 30 s clips per GPU and step")
    ap.add_argument("--model", default="voicesplit", choices=["voicesplit", "voicefilter"])
    ap.add_argument("--conv-math", default=None, choices=["fp32", "f16x3", "bf16"],
                    help="arithmetic / layout of the conv stack and the LSTM GEMMs (default: the library default, f16x3)")
    ap.add_argument("--loss", default="sisnr", choices=["sisnr", "powerlaw", "fixed"],
                    help="training mode: the reference's SI-SNR loss through the GPU iSTFT (default), or a fixed upstream gradient")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward / bf16 / fp32_strict sub-objects of the training line")
    ap.add_argument("--force-collectives", action="store_true",
                    help="train mode at --gpus 1: run the N > 1 code path of the step (bucket all-reduces over a one-rank RCCL group, "
                         "loss through the host ring, skip flag in the bucket) -- what one GPU can measure of it")
    ap.add_argument("--split-allreduce", default="auto", choices=["auto", "on", "off"],
                    help="N > 1 (or --force-collectives): start the BiLSTM + head segment of the gradient all-reduce at the library's "
                         "leaves event, beside the conv backward.  auto: both forms are timed for a few untimed steps and the faster one is kept")
    ap.add_argument("--loss-lag", type=int, default=0, choices=[0, 1],
                    help="N > 1: 1 = train_step returns the loss of the step before (the host never waits for the device)")
    ap.add_argument("--data", default="resident", choices=["resident", "files"],
                    help="train mode: 'files' writes a synthetic on-disk training set to /tmp and times Trainer.fit over it through "
                         "BatchFeeder (worker processes, pinned memory, copy stream, GPU STFT) instead of one batch resident in HBM")
    ap.add_argument("--workers", type=int, default=None, help="--data files: DataLoader worker processes (default: min(14, cores - 2))")
    ap.add_argument("--serial-backward", action="store_true",
                    help="weight gradients in order on the one stream (default: on the library's side stream, "
                         "beside the BatchNorm backward passes; same results)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 256 if args.mode == "longform" else 64

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no device ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    forced = args.force_collectives and world == 1 and args.mode == "train"
    if forced:
        from voicesplit_amd.sharding import init_single_rank_group
        init_single_rank_group("nccl", device_id=dev)

    import voicesplit_amd as V
    from voicesplit_amd import _lib
    from voicesplit_amd import ops
    lib = _lib.load()
    # BASELINE.json: the fwd+bwd configuration is configs[2] (bf16, SI-SNR loss); forward-only is configs[1] (fp32).  So the
    # training line defaults to the bf16 configuration and carries the fp32-class training step as a sub-object; the
    # forward / longform lines default to the fp32-class arithmetic (f16x3: fp32 results, <= 1e-4 vs the reference).
    # The library / module default stays f16x3: bf16 is always asked for explicitly.
    if args.conv_math is None and args.mode == "train":
        args.conv_math = "bf16"
    if args.conv_math:
        ops.set_conv_math(args.conv_math)
    conv_math = ops.get_conv_math()
    if args.serial_backward:
        assert lib.vs_set_backward_overlap(0) == 0

    B = args.batch
    train = args.mode == "train"
    cls = V.VoiceSplit if args.model == "voicesplit" else V.VoiceFilter

    def new_model(training):
        torch.manual_seed(0)
        m = cls(V.default_config())
        m.train(training)
        with torch.no_grad():                                           # non-trivial BN statistics
            g_ = torch.Generator().manual_seed(1000)
            for mod in m.conv:
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.copy_(torch.randn(mod.num_features, generator=g_) * 0.1)
                    mod.running_var.copy_(torch.rand(mod.num_features, generator=g_) + 0.5)
                    mod.weight.copy_(torch.rand(mod.num_features, generator=g_) + 0.5)
                    mod.bias.copy_(torch.randn(mod.num_features, generator=g_) * 0.1)
        return m.to(dev)

    model = new_model(train)
    g = torch.Generator().manual_seed(77 + rank)
    nb = B if args.mode != "longform" else 1
    spec = torch.rand(nb, T_FRAMES, N_FREQ, generator=g).to(dev)      # resident in HBM before timing
    dvec = torch.randn(nb, EMB, generator=g)
    dvec = (dvec / dvec.norm(dim=1, keepdim=True)).to(dev)
    # the rest of a training batch (train.py:85-92): target spectrogram, mixture phase, lengths
    target = torch.rand(nb, T_FRAMES, N_FREQ, generator=g).to(dev)
    phase = ((torch.rand(nb, T_FRAMES, N_FREQ, generator=g) - 0.5) * 6.2831853).to(dev)
    seq_len = torch.full((nb,), 160 * (T_FRAMES - 1), dtype=torch.int32, device=dev)
    # --loss fixed: a fixed d(loss)/d(mask) instead of the loss head
    dmask = (torch.randn(nb, T_FRAMES, N_FREQ, generator=g) / nb).to(dev)

    def train_cfg():
        cfg = V.default_config(model_name=args.model)
        cfg.loss["loss_name"] = {"sisnr": "si_snr", "powerlaw": "power_law_compression", "fixed": "si_snr"}[args.loss]
        cfg.train_config["learning_rate"] = 1e-4                    # random data: keep the weights finite
        return cfg

    def make_trainer(m, split=None):
        # the product's own training step (voicesplit_amd/trainer.py = train.py:86-117 per rank):
        # forward, criterion, backward, the gradient all-reduce, Adam, the loss value on the host
        from voicesplit_amd.trainer import Trainer
        fixed = (lambda mask, mixed, tgt, sl, ph: (mask * dmask).sum()) if args.loss == "fixed" else None
        tr = Trainer(m, train_cfg(), rank, world, criterion=fixed, force_collectives=forced, loss_lag=args.loss_lag,
                     split_allreduce=(args.split_allreduce != "off") if split is None else split)
        if os.environ.get("VOICESPLIT_EARLY_LOSS") == "0":           # A/B: the blocking loss.item() behind optimizer.step()
            tr.early_loss_read = False
        return tr

    units_per_step = B                                              # utterances (windows) per rank and step
    split_choice = None
    if train:
        trainer = make_trainer(model)
        bucket = trainer.bucket
        batch = (dvec, target, spec, seq_len, None, phase)
        comm_path = world > 1 or forced

        def step():
            # N > 1: the steady state of Trainer.fit -- the skip decision came with an earlier step's bucket, this rank's flag for the
            # batch two steps ahead rides in this one (a resident batch is never missing)
            if comm_path:
                trainer.train_step(batch, have=True, next_missing=0.0)
            else:
                trainer.train_step(batch)
            return bucket.flat

        feed_info = None
        if args.data == "files":
            # the step fed from the on-disk training set instead of one resident batch.  A synthetic set in the
            # reference's layout (utils/dataset.py:8-41: *-emb.pt, *-target.pt, *-mixed.wav, *-target.wav), 4 batches per rank; read by
            # worker processes, pinned, copied on a side stream, STFT on the GPU (voicesplit_amd/trainer.py: BatchFeeder).
            import shutil
            import numpy as np
            from scipy.io import wavfile
            from voicesplit_amd.trainer import BatchFeeder, EpochShard, SpecWavDataset
            root = f"/tmp/vs_bench_data_r{rank}"
            shutil.rmtree(root, ignore_errors=True)
            os.makedirs(root)
            n_items = 4 * B
            rng = np.random.default_rng(5 + rank)
            gd = torch.Generator().manual_seed(5 + rank)
            t_w = time.perf_counter()
            for i in range(n_items):
                stem = os.path.join(root, "%06d" % i)
                e_ = torch.randn(EMB, generator=gd)
                torch.save(e_ / e_.norm(), stem + "-emb.pt")
                torch.save(torch.rand(T_FRAMES, N_FREQ, generator=gd), stem + "-target.pt")
                wavfile.write(stem + "-mixed.wav", 16000, (rng.standard_normal(160 * (T_FRAMES - 1)) * 0.1).astype(np.float32))
                wavfile.write(stem + "-target.wav", 16000, (rng.standard_normal(160 * (T_FRAMES - 1)) * 0.1).astype(np.float32))
            cfg_d = train_cfg()
            cfg_d.dataset = {"train_dir": root, "test_dir": root,
                             "format": {"emb": "*-emb.pt", "mixed": "*-mixed.pt", "target": "*-target.pt",
                                        "target_wav": "*-target.wav", "mixed_wav": "*-mixed.wav"}}
            ds = SpecWavDataset(cfg_d, train=True)
            nw = args.workers if args.workers is not None else max(1, min(14, (os.cpu_count() or 4) - 2))
            feeder = BatchFeeder(ds, EpochShard(len(ds), B, 0, 1, seed=1), dev, num_workers=nw)      # (each rank has its own directory)
            fed = feeder.epoch(0, chain=(args.steps + args.warmup) // 4 + 8)
            feed_info = {"items_on_disk": n_items, "workers": nw, "write_s": round(time.perf_counter() - t_w, 1),
                         "mb_per_batch_read": round(B * (2 * 4 * 160 * (T_FRAMES - 1) + 4 * T_FRAMES * N_FREQ + 4 * EMB) / 1e6, 1)}

            def step():                                                # noqa: F811
                b_ = next(fed)
                if comm_path:
                    trainer.train_step(b_, have=True, next_missing=0.0)
                else:
                    trainer.train_step(b_)
                return bucket.flat

        if comm_path and args.split_allreduce == "auto" and trainer.split_allreduce:
            # Which form of the exchange is faster on THIS node is a property of its fabric and of how RCCL's kernels share the CUs with
            # the conv backward: time both for a few steps (outside the timed region, all ranks agree on the maximum) and keep the winner.
            def probe(on, k=6):
                trainer.split_allreduce = on
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                if dist:
                    dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(k):
                    step()
                torch.cuda.synchronize()
                t_ = torch.tensor([(time.perf_counter() - t0_) / k * 1e3], dtype=torch.float64, device=dev)
                if dist:
                    dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                return float(t_.item())
            ms_on, ms_off = probe(True), probe(False)
            trainer.split_allreduce = ms_on <= ms_off
            split_choice = {"chosen": "split" if trainer.split_allreduce else "one collective behind the backward",
                            "ms_per_step_split": round(ms_on, 3), "ms_per_step_single": round(ms_off, 3),
                            "note": "6 untimed steps each before the warm-up, maximum over ranks"}
    elif args.mode == "forward":
        def step():
            with torch.no_grad():
                return model(spec, dvec)
    else:
        # BASELINE configs[4]: this rank's clips of the job (clips sharded over the ranks: voicesplit_amd/sharding.py)
        from voicesplit_amd import sharding, streaming
        lo, hi = sharding.shard_range(args.clips * world, rank, world)
        n_clips = hi - lo
        long_spec = torch.rand(n_clips, LONG_FRAMES, N_FREQ, generator=g).to(dev)
        long_dvec = torch.randn(n_clips, EMB, generator=g)
        long_dvec = (long_dvec / long_dvec.norm(dim=1, keepdim=True)).to(dev)
        units_per_step = n_clips * sharding.chunk_windows(LONG_FRAMES, T_FRAMES)

        def step():
            return streaming.separate_long_many(model, long_spec, long_dvec, window=T_FRAMES, max_batch=B)

    elapsed_local = [0.0]

    def timed(fn, steps, warmup, profile, calls_per_step=None):
        for _ in range(warmup):
            out_ = fn()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        if profile and rank == 0:
            _lib.check(lib.vs_profile_begin(steps * (calls_per_step or (6 if args.mode == "longform" else 1))), "vs_profile_begin")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out_ = fn()
        torch.cuda.synchronize()
        elapsed_local[0] = time.perf_counter() - t0               # this rank alone, before the closing barrier
        if dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(out_).all()
        ms = calls = None
        if profile and rank == 0:          # close the per-stage timers before anything else calls into the library
            ms = (ctypes.c_float * _lib.PROF_SLOTS)()
            calls = (ctypes.c_int * _lib.PROF_SLOTS)()
            _lib.check(lib.vs_profile_end(ms, calls), "vs_profile_end")
        return el, ms, calls

    if train and world > 1:
        # N > 1 self-diagnosis: HIP events around every gradient all-reduce; the communicator's
        # first collectives (ring setup, buffer registration) run in the warm-up -- at least one warm-up step even with --warmup 0
        if args.warmup < 1:
            step()
        trainer.set_comm_timing(True)
    elapsed, ms, calls = timed(step, args.steps, args.warmup, True)
    if train:
        assert torch.isfinite(bucket.flat).all()
    comm = None
    if train and world > 1:
        ar = bucket.collective_ms()[-args.steps:]
        ae = bucket.collective_ms(early=True)[-args.steps:]
        fl = trainer.flag_ms[-args.steps:]
        trainer.set_comm_timing(False)
        mean_ = lambda v: sum(v) / max(1, len(v))
        mine = torch.tensor([elapsed_local[0] / args.steps * 1e3, mean_(ar), max(ar) if ar else 0.0, mean_(ae), max(ae) if ae else 0.0,
                             float(len(fl))], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu()
        comm = {"step_ms_per_rank_min": round(float(rows[:, 0].min()), 3), "step_ms_per_rank_max": round(float(rows[:, 0].max()), 3),
                "allreduce_tail_ms": round(float(rows[:, 1].mean()), 3), "allreduce_tail_ms_max_over_ranks_and_steps": round(float(rows[:, 2].max()), 3),
                "allreduce_early_ms": round(float(rows[:, 3].mean()), 3), "allreduce_early_ms_max_over_ranks_and_steps": round(float(rows[:, 4].max()), 3),
                "split_allreduce": bool(trainer.split_allreduce), "loss_lag": trainer.loss_lag,
                "blocking_flag_reduces_in_the_timed_steps": int(rows[:, 5].max()),
                "note": "allreduce_tail_ms: HIP-event time of the collective behind the backward pass (bucket.all_reduce: the whole 75.5 MB bucket, or -- "
                        "split_allreduce -- the conv stack's 2.2 MB + the two spare slots); it includes waiting for the slowest rank's backward.  "
                        "allreduce_early_ms: the BiLSTM + head segment (73 MB) on the side stream, started at the library's leaves event beside the conv "
                        "backward.  Mean over ranks and timed steps.  No per-step flag collective and no .item() in the step (trainer.py); "
                        "step_ms_per_rank_*: each rank's own clock around its timed steps (before the closing barrier)"}
        if split_choice:
            comm["split_allreduce_calibration"] = split_choice

    if train and forced:
        comm = {"forced_collectives_on_one_rank": True, "split_allreduce": bool(trainer.split_allreduce), "loss_lag": trainer.loss_lag}
        if split_choice:
            comm["split_allreduce_calibration"] = split_choice
    # ---- RCCL leg (outside the timed region) ---------------------------------------------------------
    # N > 1: every rank must hold the same bucket after the step's all-reduce (a checksum per rank, gathered).
    # N = 1: the step's exchange is a no-op, so the device collective is exercised once on its own: a
    # one-rank process group, the 75.5 MB bucket through ncclAllReduce, timed with HIP events.
    rccl = None
    if train:
        try:
            import torch.distributed as tdist
            if world > 1:
                chk = torch.stack([bucket.flat.double().sum(), bucket.flat.double().abs().sum()])
                allc = [torch.empty_like(chk) for _ in range(world)]
                tdist.all_gather(allc, chk)
                same = all(torch.equal(allc[0], c) for c in allc)
                assert same, "gradient buckets differ between ranks after the all-reduce"
                rccl = {"rccl_ranks": world, "bucket_identical_on_all_ranks": True, "bucket_mb": round(bucket.flat.numel() * 4 / 1e6, 1)}
                if comm:
                    rccl.update(comm)
            else:
                # (under torch.distributed.run the group has to come from the launcher's store: see the helper)
                from voicesplit_amd.sharding import init_single_rank_group
                if not forced and not os.environ.get("VS_BENCH_GROUP_READY"):
                    init_single_rank_group("nccl", device_id=dev)
                keep = bucket.flat.clone()
                bucket.all_reduce(1, force=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    bucket.all_reduce(1, force=True)
                e1.record()
                torch.cuda.synchronize()
                assert torch.equal(bucket.flat, keep)
                rccl = {"rccl_ranks": 1, "bucket_mb": round(bucket.flat.numel() * 4 / 1e6, 1),
                        "one_rank_allreduce_ms": round(e0.elapsed_time(e1) / 5, 3),
                        "note": "one-rank ncclAllReduce of the gradient bucket on the device, outside the timed region"}
                if comm:
                    rccl.update(comm)
                tdist.destroy_process_group()
        except AssertionError:
            raise
        except Exception as exc:                      # RCCL unavailable on this box: report, do not fail the bench
            rccl = {"rccl_ranks": 0, "error": str(exc)[:200]}

    # The pool's boxes differ by several percent on every kernel: one fixed, self-contained launch timed
    # outside the region says which kind of box this line comes from -- the bf16 LSTM input contraction at the metric shape
    # (19264 x 3200 x 4808, the same seeded random operands everywhere: the clock of these kernels follows the operands' toggle rate).
    box = None
    if rank == 0:
        try:
            M_, N_, K_ = 19264, 3200, 4808
            Kp_ = (K_ + 63) // 64 * 64
            g_ = torch.Generator(device=dev).manual_seed(1234)
            a_ = torch.randn(M_, Kp_, generator=g_, device=dev).to(torch.bfloat16)
            b_ = (torch.randn(N_, Kp_, generator=g_, device=dev) * 0.02).to(torch.bfloat16)
            ops.gemm_bf16(a_, b_, M_, N_, K_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_bf16(a_, b_, M_, N_, K_)
            e1.record()
            torch.cuda.synchronize()
            cal_ms = e0.elapsed_time(e1) / 10
            box = {"kernel": "gemm_bf16 19264x3200x4808 on seeded random operands, mean of 10 launches outside the timed region",
                   "ms": round(cal_ms, 4), "tflops": round(2.0 * M_ * N_ * K_ / cal_ms / 1e9, 1),
                   "note": "the same launch on every box: its TFLOP/s is a proxy of the clock this box sustains under a matrix-pipe load "
                           "(the pool's boxes differ by +-3.5 %); compare two lines at equal calibration"}
            del a_, b_
        except Exception as exc:
            box = {"error": str(exc)[:200]}

    def stage_table(ms_, calls_, steps_):
        return {n: (ms_[i] / steps_ if calls_[i] else None) for i, n in enumerate(_lib.PROF_NAMES)}

    def conv_roofline(stage_ms, math, is_train, tag, nbatch=None, batch=None):
        """The dominant kernel: the 5x5 64->64 conv (cnn3..cnn7 forward, + their data gradients in training).
        nbatch: forward batches per timed step (long-form: the windows of a step go through in several batches), batch: their
        mean size -- the stage timers hold the sum over a step's batches."""
        if nbatch is None:
            nbatch = 5 if args.mode == "longform" else 1      # --mode longform: 1280 windows = 5 forward batches of 256 per step
        B = args.batch if batch is None else batch
        launches = [stage_ms[f"cnn{i}"] for i in range(3, 8)]
        if is_train:
            launches += [stage_ms[f"dgrad_cnn{i}"] for i in range(3, 8)]
        launches = [v / nbatch for v in launches]
        kinst = None
        mean_launch_ms = sum(launches) / len(launches)
        achieved = B * GFLOP_CONV5X5 / mean_launch_ms           # algorithmic GFLOP / ms == TFLOP/s
        act_bytes = 2 if math == "bf16" else 4
        if math == "bf16":
            kname, peak = "nhwc_conv_kernel<5,5> (channels-last bf16, LDS-DMA window buffers, register-resident weights)", PEAK_F16_MFMA_TFLOPS
            extra = {"mfma_pipe": "bf16 (v_mfma_f32_16x16x32_bf16), one MFMA product per product", "mfma_peak_tflops": PEAK_F16_MFMA_TFLOPS}
            ksub, ksrc = "nhwc_conv_kernel", "conv_nhwc.hip"
        elif math == "f16x3":
            # every fp32 product is three f16 MFMA products (csrc/conv_f16x3.hip): the pipe-level
            # ceiling for ALGORITHMIC flops is the dense f16 MFMA peak / 3
            kname, peak = "conv64_f16x3_pk_kernel<5,5> (persistent pipeline)", PEAK_F16_MFMA_TFLOPS / 3.0
            extra = {"mfma_pipe": "f16 (v_mfma_f32_32x32x16_f16), 3 MFMA products per fp32 product",
                     "mfma_rate_tflops": round(3 * achieved, 1), "mfma_peak_tflops": PEAK_F16_MFMA_TFLOPS,
                     "launch_ms_includes": "operand-scale kernel + weight pack (~15 us); |max| of the operands is tracked by their producers",
                     "empirical_ceiling": "a stream of nothing but this layer's MFMAs reaches 1659 TF on the f16 pipe with its real operand values "
                                          "(2345 TF on zero operands): the chip clocks down with operand toggling; 553 TF algorithmic = 0.66 of `peak` "
                                          "is the most any schedule of this arithmetic reaches on random data (profiles/r02_conv_ablation.md)"}
            ksub, ksrc = "conv64_f16x3_pk_kernel", "conv_f16x3_pk.hip"
            if not is_train:
                # eval-mode forward (configs[1], configs[4]): the channels-last split-f16 kernel (csrc/conv_nhwc_f16x3.hip, round 4)
                kname = "nhwc_conv_f16x3_kernel<5,5> (channels-last hi/lo f16 planes by LDS-DMA, weights in AGPRs, K halves across waves)"
                ksub, ksrc, kinst = "nhwc_conv_f16x3", "conv_nhwc_f16x3.hip", "<5, 5"      # (_kernel or _scalar_kernel: the build vs_set_option picks)
                extra["mfma_pipe"] = "f16 (v_mfma_f32_16x16x32_f16), 3 MFMA products per fp32 product"
                extra["launch_ms_includes"] = "the layer's plan kernel (output scale from the tracked |max| of its input); weights come prepared"
                extra["issue_model"] = ("one wave per SIMD: 16 cycles per MFMA + ~5 cycles for every other instruction, no overlap measured "
                                        "(s_memtime probes and ablations: profiles/r04_split_conv.md); 450 MFMAs + ~350 other instructions per group of 6 rows; cuts of the "
                                        "instruction count have not moved the time: the clock follows (power-bound)")
        else:
            kname, peak, extra, ksub, ksrc = "conv64_mfma_kernel<5,5>", PEAK_FP32_MFMA_TFLOPS, {}, "conv64_mfma_kernel", "conv_mfma.hip"
        no64 = (None, "counters were collected at B = 64 (forward, training step) and at 256 windows (long-form) only")
        if B == 256 and not is_train:                                       # the long-form batch has its own counters (tag longform[_math])
            tag = tag.replace("forward", "longform")
        traffic, rnd = committed_pmc_traffic(tag, ksub, kinst, source=ksrc) if B == 64 or tag.startswith("longform") else no64
        algo_gb = B * 2 * 64 * T_FRAMES * N_FREQ * act_bytes / 1e9
        by_instance = None
        if math == "bf16" and is_train:
            # two instances of the kernel share the 10 launches: the forward conv (reads a, writes z: 2 tensors) and the dy-form
            # data gradient (reads dz and the lower layer's z, writes dy: 3 tensors) -- each against ITS algorithmic bytes
            one = B * 64 * T_FRAMES * N_FREQ * act_bytes / 1e9
            t_f, r_f = committed_pmc_traffic(tag, "nhwc_conv_kernel<5, 5", "true, false>", source=ksrc) if B == 64 else no64
            t_d, r_d = committed_pmc_traffic(tag, "nhwc_conv_kernel<5, 5", "false, true>", source=ksrc) if B == 64 else no64
            by_instance = {"forward": {"launches_per_step": 5, "algorithmic_gb": round(2 * one, 2), "traffic_gb": t_f},
                           "dy_form_data_gradient": {"launches_per_step": 5, "algorithmic_gb": round(3 * one, 2), "traffic_gb": t_d}}
            algo_gb = 2.5 * one                                                 # mean over the 10 launches
            if t_f is not None and t_d is not None:
                traffic, rnd = round((t_f + t_d) / 2, 2), r_f
        roof = {"bound": "mfma",
                "kernel": kname + " (cnn3..cnn7 forward" + (" + data gradient" if is_train else "") + ")",
                "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "traffic_unit": (f"GB per launch: 2 x FETCH_SIZE (calibrated: profiles/r02_calibration) + WRITE_SIZE from the committed --pmc passes of this command "
                                 f"(profiles/{rnd}_{tag}_rocprof/pmc_per_kernel.csv, collected on a byte-identical {ksrc}: sources.sha256 there; not measured "
                                 f"in this run: bench.py cannot collect counters); " if traffic is not None else
                                 f"null: no committed counters that belong to this binary ({rnd}); ")
                                + f"algorithmic bytes {algo_gb:.2f} GB" + (" (mean over the two instances: traffic_by_instance)" if by_instance else ""),
                "launches_per_step": len(launches), "launch_ms": [round(v, 3) for v in launches],
                "hbm_frac_of_same_kernel": round(algo_gb / (mean_launch_ms / 1e3) / PEAK_HBM_GBS, 4)}
        roof.update(extra)
        if by_instance:
            roof["traffic_by_instance"] = by_instance
        if is_train:
            wg = [stage_ms[f"wgrad_cnn{i}"] for i in range(3, 8)]
            wg_mean = sum(wg) / 5.0
            wname = {"bf16": "nhwc_wgrad_kernel<5,5> (weight gradient of cnn3..cnn7: transposing LDS reads, incl. its reduce)",
                     "f16x3": "conv64_wgrad_ring4_kernel (weight gradient of cnn3..cnn7, 3 f16 MFMA products per fp32 product, incl. its reduce)",
                     "fp32": "conv64_wgrad_kernel<5> (weight gradient of cnn3..cnn7, fp32 MFMA, incl. its reduce)"}[math]
            roof["second_kernel"] = {"kernel": wname,
                                     "achieved": round(B * GFLOP_CONV5X5 / wg_mean, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                     "frac": round(B * GFLOP_CONV5X5 / wg_mean / peak, 4),
                                     "launch_ms": [round(v, 3) for v in wg],
                                     "note": ("timed while the BatchNorm backward pass of the layer below runs beside it on the library's side stream "
                                              "(bf16 configuration: the weight gradient on the caller's stream; vs_set_backward_overlap, DESIGN.md 6.7)"
                                              if not args.serial_backward else "serial schedule: the kernel alone")}
        if stage_ms.get("lstm_gemm"):
            nl = nbatch
            gemm_peak = PEAK_F16_MFMA_TFLOPS if math == "bf16" else PEAK_F16_MFMA_TFLOPS / 3.0 if math == "f16x3" else PEAK_FP32_MFMA_TFLOPS
            lg = B * GFLOP_LSTM_GEMM / (stage_ms["lstm_gemm"] / nl)
            roof["lstm_input_gemm"] = {"stage_ms": round(stage_ms["lstm_gemm"] / nl, 3), "achieved": round(lg, 1), "peak": round(gemm_peak, 1),
                                       "unit": "TFLOP/s", "frac": round(lg / gemm_peak, 4),
                                       "note": "x @ [W_ih; W_ih_reverse]^T (19264 x 3200 x 4808 at B=64) incl. the operand split passes and the d-vector fold"}
        return roof

    # ---- BASELINE configs[1] next to the training line: forward only, eval-mode BatchNorm ---------------
    def leg_roofline(ms_, calls_, steps_, math, tag, nbatch=1, batch=None):
        """`roofline` + per-stage times of a forward / long-form sub-leg from its own stage timers (rank 0)."""
        st_ms = stage_table(ms_, calls_, steps_)
        roof = conv_roofline(st_ms, math, False, tag, nbatch=nbatch, batch=batch)
        return roof, {k: round(v / nbatch, 3) for k, v in st_ms.items() if v is not None}

    def forward_leg(m, math):
        m.eval()
        prev = ops.get_conv_math()
        ops.set_conv_math(math)
        try:
            FK = 10

            def f():
                with torch.no_grad():
                    return m(spec, dvec)
            fel, fms, fcalls = timed(f, FK, 2, True)
        finally:
            ops.set_conv_math(prev)
            m.train()
        leg = {"metric": f"utterances/sec (3 s clips, B={B}/GPU) forward only, eval BatchNorm (BASELINE configs[1]), " + MATH_LABEL[math],
               "value": round(world * B * FK / fel, 2), "unit": "utterances/s", "steps": FK, "warmup": 2,
               "ms_per_step": round(1e3 * fel / FK, 3)}
        if fms is not None:
            leg["roofline"], leg["stage_ms"] = leg_roofline(fms, fcalls, FK, math, "forward" + ("" if math == "f16x3" else "_" + math))
        return leg

    # ---- the reference's own eval shapes: validation() runs B = 1 (utils/generic_utils.py:495), the fast test B = 5 (:545) ----
    def latency_leg(m):
        m.eval()
        prev = ops.get_conv_math()
        out = {"note": "eval-mode forward of ONE call at the reference's own evaluation batch sizes (validation(): B = 1, utils/generic_utils.py:495; "
                       "test_fast_all_checkpoints: B = 5, :545), 301 frames, weights prepared once; mean of 20 calls after 3 warm-up calls, "
                       "host-timed around a device synchronize (latency, not throughput); cpu_baseline.forward_b1 is the same call on the host cores"}
        try:
            for math in ("f16x3", "bf16"):
                ops.set_conv_math(math)
                for b_ in (1, 5):
                    xs, ds = spec[:b_].contiguous(), dvec[:b_].contiguous()
                    with torch.no_grad():
                        for _ in range(3):
                            m(xs, ds)
                        torch.cuda.synchronize()
                        t0_ = time.perf_counter()
                        for _ in range(20):
                            m(xs, ds)
                        torch.cuda.synchronize()
                    ms_ = (time.perf_counter() - t0_) / 20 * 1e3
                    out[f"b{b_}_{'fp32_class' if math == 'f16x3' else 'bf16'}_ms"] = round(ms_, 3)
                    out[f"b{b_}_{'fp32_class' if math == 'f16x3' else 'bf16'}_utt_per_s"] = round(b_ / ms_ * 1e3, 1)
        finally:
            ops.set_conv_math(prev)
            m.train()
        return out

    # ---- BASELINE configs[4] next to the training line: 30 s clips as independent 301-frame windows, 256 per forward batch ----
    def longform_leg(m, math, clips=51, steps=2):
        from voicesplit_amd import sharding, streaming
        m.eval()
        prev = ops.get_conv_math()
        ops.set_conv_math(math)
        try:
            gl = torch.Generator().manual_seed(99 + rank)
            ls = torch.rand(clips, LONG_FRAMES, N_FREQ, generator=gl).to(dev)
            ld = torch.randn(clips, EMB, generator=gl)
            ld = (ld / ld.norm(dim=1, keepdim=True)).to(dev)
            nwin = clips * sharding.chunk_windows(LONG_FRAMES, T_FRAMES)

            def f():
                return streaming.separate_long_many(m, ls, ld, window=T_FRAMES, max_batch=256)
            lel, lms, lcalls = timed(f, steps, 1, True, calls_per_step=-(-nwin // 256) + 1)
            del ls
        finally:
            ops.set_conv_math(prev)
            m.train()
        leg = {"metric": "windows/sec = utterances/sec (30 s clips cut into independent 301-frame windows, 256 windows per forward batch, eval "
                         "BatchNorm: BASELINE configs[4]), " + MATH_LABEL[math],
               "value": round(world * nwin * steps / lel, 2), "unit": "utterances/s", "clips_per_s": round(world * clips * steps / lel, 2),
               "clips_per_gpu_and_step": clips, "windows_per_step": nwin, "steps": steps, "warmup": 1, "ms_per_step": round(1e3 * lel / steps, 3)}
        if lms is not None:
            nb_ = -(-nwin // 256)                                   # forward batches per step; the roofline is per launch at their mean size
            leg["roofline"], leg["stage_ms"] = leg_roofline(lms, lcalls, steps, math, "forward" + ("" if math == "f16x3" else "_" + math),
                                                            nbatch=nb_, batch=nwin / nb_)
            leg["stage_ms_note"] = f"per forward batch ({nb_} batches of {nwin / nb_:.0f} windows per step)"
        return leg

    # ---- the same training step in another arithmetic (own model, own trainer, same inputs) ----------------
    def train_leg(math, steps, warmup):
        prev = ops.get_conv_math()
        ops.set_conv_math(math)
        try:
            m2 = new_model(True)
            tr2 = make_trainer(m2)

            def st():
                tr2.train_step(batch)
                return tr2.bucket.flat
            if rank == 0:
                _lib.check(lib.vs_profile_begin(steps + warmup), "vs_profile_begin")
            el, _, _ = timed(st, steps, warmup, False)
            leg = {"metric": f"utterances/sec (3 s clips, B={B}/GPU) fwd+bwd, " + MATH_LABEL[math],
                   "value": round(world * B * steps / el, 2), "unit": "utterances/s", "steps": steps, "warmup": warmup,
                   "ms_per_step": round(1e3 * el / steps, 3)}
            if rank == 0:
                ms2 = (ctypes.c_float * _lib.PROF_SLOTS)()
                calls2 = (ctypes.c_int * _lib.PROF_SLOTS)()
                _lib.check(lib.vs_profile_end(ms2, calls2), "vs_profile_end")
                st_ms = stage_table(ms2, calls2, steps + warmup)         # the timers also saw the warm-up steps
                leg["roofline"] = conv_roofline(st_ms, math, True, "train_" + math)
                leg["stage_ms"] = {k: (round(v, 3) if v is not None else None) for k, v in st_ms.items()}
                leg["tape_gb"] = round(ops.tape_pool_bytes(dev) / 1e9, 2)
            if math == "bf16":
                leg["forward"] = forward_leg(m2, math)
            leg["dtype"] = math
            del tr2, m2
            ops.release_workspaces()
            torch.cuda.empty_cache()
            return leg
        finally:
            ops.set_conv_math(prev)

    fwd = fwd16 = other = strict = lform = lat = None
    if not train:
        feed_info = None
    if train and not args.no_extras:
        fwd = forward_leg(model, "f16x3" if conv_math == "bf16" else conv_math)      # configs[1]: fp32-class forward
        if conv_math == "bf16":
            fwd16 = forward_leg(model, "bf16")
        if rank == 0 and B >= 5:
            lat = latency_leg(model)
        ops.release_workspaces()
        torch.cuda.empty_cache()
        lform = {m_: longform_leg(model, m_) for m_ in (("bf16", "f16x3") if conv_math == "bf16" else (conv_math,))}
        if conv_math in ("bf16", "f16x3"):
            ops.release_workspaces()
            torch.cuda.empty_cache()
            other = train_leg("f16x3" if conv_math == "bf16" else "bf16", 5, 2)
            strict = train_leg("fp32", 3, 1)

    line = None
    if rank == 0:
        # per step: total ms of each slot / steps (a slot may be entered once per layer)
        stage_ms = stage_table(ms, calls, args.steps)
        value = world * units_per_step * args.steps / elapsed
        roof = conv_roofline(stage_ms, conv_math, train, ("train" if train else "forward") + ("" if conv_math == "f16x3" else "_" + conv_math))
        workload = {
            "train": (f"BASELINE metric config: B={B}/GPU synthetic [B,301,601] spec + [B,256] dvec, {args.model} "
                      "training step = forward (batch-stat BN) + "
                      + {"sisnr": "SI-SNR loss through the GPU iSTFT (train.py:95-108)",
                         "powerlaw": "power-law compressed loss (train.py:74-75,108)",
                         "fixed": "fixed upstream gradient on the mask"}[args.loss]
                      + " + backward + gradient all-reduce + Adam, random-init weights"),
            "forward": (f"BASELINE configs[1]: B={B}/GPU synthetic [B,301,601] spec + [B,256] dvec, "
                        f"{args.model} forward-only, eval BN, random-init weights"),
            "longform": (f"BASELINE configs[4]: {args.clips} synthetic 30 s clips ([{LONG_FRAMES},601]) per GPU and step, each cut into "
                         f"{-(-LONG_FRAMES // T_FRAMES)} independent 301-frame windows, {B} windows per forward batch, {args.model} forward, "
                         "eval BN, clips sharded over the ranks, no collective"),
        }[args.mode]
        line = {
            "metric": (f"utterances/sec (3 s clips, B={B}/GPU) " + {"train": "fwd+bwd, ", "forward": "forward, ",
                                                                   "longform": "forward over 301-frame windows of 30 s clips, "}[args.mode]
                       + MATH_LABEL[conv_math]
                       + (f" [box calibration GEMM: {box['tflops']:.0f} TF]" if box and "tflops" in box else "")),
            "value": round(value, 2), "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "fp32",
                      "bf16": "bf16 (conv activations and tape stored as bf16; bf16 MFMA operands in the convs, the LSTM / head contractions and the BPTT, f16 operands in the forward recurrence; fp32 accumulate, BatchNorm statistics, gate arithmetic / cell state, loss head, master weights)",
                      "f16x3": "fp32 (64->64 convs + LSTM GEMMs: fp32 operands as 2xf16 halves, 3 f16 MFMA products, fp32 accumulate)"}[conv_math],
            "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": B, "global_batch": B * world, "frames": T_FRAMES, "num_freq": N_FREQ,
                       "conv_math": conv_math,
                       "parallelism": (f"dp{world}: one flat 75.5 MB fp32 gradient all-reduce per step" if train else
                                       f"batch-sharded x{world}, no data-path collective")},
            "roofline": roof,
            "stage_ms": {k: (round(v, 3) if v is not None else None) for k, v in stage_ms.items()},
            "stage_ms_note": ("HIP-event time of each stage on the stream it was launched on; with the side-stream backward schedule the "
                              "wgrad_* / bwd_bn / bwd_lstm_gemm / bwd_edge stages overlap in time and do not add up to the step"
                              if train and not args.serial_backward else "stages run in order on one stream"),
            "model_tflops": round(value / world * GFLOP_FWD_TOTAL * (3 if train else 1) / 1e3, 2),
        }
        if args.mode == "longform":
            line["clips_per_s"] = round(world * args.clips * args.steps / elapsed, 2)
        if train and feed_info is not None:
            line["data"] = "synthetic, read from files"
            line["feeding"] = dict(feed_info, note="every step's batch comes from disk through voicesplit_amd.trainer.BatchFeeder: DataLoader worker "
                                                   "processes over SpecWavDataset.__getitem__ (2 wav reads + 2 torch.load per item), pinned memory, "
                                                   "host -> device on a copy stream beside the step before, wav -> spectrogram + phase on the GPU")
        # the sub-legs' headline numbers inside `config` (the driver's record keeps metric / value / config / roofline / cpu_baseline in
        # full and only the NAMES of other keys)
        also = {}
        if fwd is not None:
            also["configs[1] forward fp32-class"] = {"utt_per_s": fwd["value"], "ms_per_step": fwd["ms_per_step"],
                                                     "conv5x5_frac_of_833TF": fwd.get("roofline", {}).get("frac"),
                                                     "lstm_gemm_frac": fwd.get("roofline", {}).get("lstm_input_gemm", {}).get("frac")}
        if fwd16 is not None:
            also["forward bf16"] = {"utt_per_s": fwd16["value"], "ms_per_step": fwd16["ms_per_step"]}
        if lform is not None:
            also["configs[4] long-form"] = {k_: {"windows_per_s": v_["value"], "conv5x5_frac": v_.get("roofline", {}).get("frac")} for k_, v_ in lform.items()}
        if other is not None:
            also["training step " + ("fp32-class" if conv_math == "bf16" else "bf16")] = {"utt_per_s": other["value"], "ms_per_step": other["ms_per_step"]}
        if lat is not None:
            also["eval latency (reference's B = 1 / B = 5)"] = {k_: v_ for k_, v_ in lat.items() if k_.endswith("_ms")}
        if roof.get("second_kernel"):
            also["weight gradient 5x5 frac"] = roof["second_kernel"]["frac"]
        if roof.get("lstm_input_gemm"):
            also["lstm input GEMM stage frac"] = roof["lstm_input_gemm"]["frac"]
        if also:
            line["config"]["also_measured"] = also
        if lat is not None:
            line["eval_latency"] = lat
        if fwd is not None:
            line["forward"] = fwd
        if fwd16 is not None:
            line["forward_bf16"] = fwd16
        if lform is not None:
            line["longform"] = lform
        if other is not None:
            line["fp32_class" if conv_math == "bf16" else "bf16"] = other
        if strict is not None:
            line["fp32_strict"] = strict
        if rccl is not None:
            line["rccl"] = rccl
        if box is not None:
            line["box_calibration"] = box
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline("train" if train else "forward")
    if dist:
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a file:
    # push it out first so that the JSON line is the LAST line on stdout
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
