#!/usr/bin/env python
"""Headline benchmark: denoising steps/s (network forward + reverse update) per complex on the
BASELINE.json configs[1] workload -- synthetic 256-residue + 64-atom-ligand complex (N = 320),
single_dim 512, pair_dim 64, 4 folding blocks, T = 1000 -- on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]          # N > 1: starts N ranks itself (see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU over RCCL (backend "nccl").  Samples of ``sample()`` never interact, so the path shards by
global sample index with no collective inside the loop (weak scaling: every rank runs ``--samples-per-gpu``
complexes); the ONE collective of the sampling job -- the all_gather that returns every rank's samples
(distributed.sample_sharded, ~2 MB for 64 samples) -- runs once at the end of the timed region.  A "step" is one
reverse-diffusion step of every complex on the rank, replayed from one captured hipGraph with all inputs
(weights, static embeddings, noise table) resident in HBM.  Rank 0 prints ONE JSON line.

``--gpus N`` without a launcher (no WORLD_SIZE in the environment): this process starts
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD and exits with its code.  That happens
before anything here touches the GPU (a process that has initialised HIP must never re-exec, and does not).
"""
import argparse
import json
import os
import shutil
import socket
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 at the fp32 vector rate
PEAK_16BIT_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
HBM_PEAK_GBS = 8000.0
S, P, NB, T = 512, 64, 4, 1000


def step_flops(N, S, P, H=4, c=16, nb=4):
    """Algorithmic flops of one denoising step of one complex (contractions only)."""
    Hc = H * c
    pre = 2 * N * N * 256 * P + 2 * N * 21 * S
    opm = 4 * N * S * (S // 4) + 2 * N * N * (S // 4) * P
    spa = 2 * N * N * P * H + 8 * N * S * H * S + 4 * H * N * N * S + 2 * N * H * S * S
    blk = (2 * N * N * P * H + 10 * N * S * Hc + 4 * H * N * N * c + 16 * N * S * S + 2 * N * N * S * P + 4 * N * S * P
           + 16 * N * N * P * P + 4 * P * N ** 3 + 8 * N * N * P * P + 20 * N * N * P * Hc + 8 * Hc * N ** 3 + 16 * N * N * P * P)
    heads = 2 * N * N * P * P + 2 * N * N * P + 2 * N * S * S + 42 * N * S
    return pre + opm + spa + nb * blk + heads


def tri_attn_core_flops(b, N, P, H=4, c=16):
    """q,k,v,g projections (8 N^2 P Hc) + QK^T and PV (4 Hc N^3) of ONE launch of tri_attn_core_kernel."""
    return b * (8 * N * N * P * H * c + 4 * H * c * N ** 3)


def tri_attn_core_bytes(b, N, P):
    """Algorithmic bytes of one launch: the pair tensor read once, the gated per-head output written once (2 U)."""
    return b * (N * N * P * 4 + N * N * 64 * 4)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--samples-per-gpu", type=int, default=1)
    ap.add_argument("--residues", type=int, default=256)
    ap.add_argument("--atoms", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the child rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-shard-check", action="store_true", help="N > 1: skip the sample_sharded == single-rank check")
    ap.add_argument("--kernel-only", action="store_true", help="(internal) run only the dominant kernel, for the PMC passes")
    ap.add_argument("--train", action="store_true",
                    help="NOT the headline: print the optimisation-step line of tools/train_bench.py (BASELINE configs[3] per-GPU share: "
                         "2 complexes of N = 320, training_step + backward + Adam + EMA) and exit")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a):
    """--gpus N without a launcher: run N ranks as a child torchrun.  Nothing in this process has touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if os.environ.get("PRD_BENCH_DRY_LAUNCH"):                  # tests: show the launch instead of performing it
        print(json.dumps({"launch": cmd}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def build_model(dev, graph=True):
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    from protein_redesign_amd.synthetic import deterministic_state_dict
    from protein_redesign_amd.weights import spec_tensors
    margs = make_args(single_dim=S, pair_dim=P, num_blocks=NB, num_steps=T, mask_prob=0.3)
    params = deterministic_state_dict(spec_tensors(margs), seed=1)
    model = ProteinReDiffModel(margs)
    model.load_state_dict(params)
    model = model.to(dev).eval()
    model.use_hip_graph = graph
    return model, margs, params


def dominant_kernel(model, mask, bpg, N, dev, reps):
    """tri_attn_core alone, starting / ending modes alternating as inside a step; returns its average launch time (us) measured
    with HIP events on the stream the kernel is launched on (torch's current stream)."""
    import torch
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(0)
    pair = torch.randn(bpg, N, N, P, generator=g).to(dev)
    ta = model.Denoiser.folding_blocks[0].pair_attn_starting.attn
    wts = ta.weights()[:5]
    og = torch.empty(bpg, N, N, 64, device=dev)
    for i in range(4):
        ops.tri_attn_core(pair, mask, wts, 4, 16, ending=bool(i & 1), og=og)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.tri_attn_core(pair, mask, wts, 4, 16, ending=bool(i & 1), og=og)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def measure_traffic(a):
    """Counters of the dominant kernel, measured NOW: three child ``rocprofv3 --kernel-trace --pmc`` passes of
    ``python3 bench.py --kernel-only`` (FETCH_SIZE and WRITE_SIZE need separate passes: 3 + 2 of the 4 TCC slots; the SQ / GRBM
    counters ride in a third).  Units / corrections per MI355X_MICROARCH.md §HBM: both TCC counters are KiB; on gfx950 FETCH_SIZE
    reports half the bytes of wide (16 B / lane) coalesced reads -> x2; WRITE_SIZE matched the byte count of this kernel's
    output exactly (profiles/README.md) and is used as is.  Returns (bytes or None, note, {counter fractions})."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found", {}
    vals = {}
    tmp = tempfile.mkdtemp(prefix="prd_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]):
            out = os.path.join(tmp, counters[0])
            cmd = [prof, "--kernel-trace", "--pmc"] + counters + ["-d", out, "-o", "t", "--",
                   "python3", os.path.abspath(__file__), "--kernel-only", "--residues", str(a.residues), "--atoms", str(a.atoms),
                   "--samples-per-gpu", str(a.samples_per_gpu)]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            env.pop("WORLD_SIZE", None)
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None, f"{counters[0]} pass failed (exit {r.returncode}): {r.stdout.decode(errors='replace')[-300:]}", {}
            db = sqlite3.connect(dbs[0])
            for counter in counters:
                rows = list(db.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? "
                                       "and kernel_name like '%tri_attn_core%' group by dispatch_id", (counter,)))
                if not rows:
                    return None, f"{counter}: no tri_attn_core dispatch in the counter database", {}
                vals[counter] = sum(v for _, v in rows) / len(rows)
    except Exception as e:      # profiling is evidence, never a reason to lose the bench line
        return None, f"PMC pass error: {e!r}", {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = int(2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
    pipes = {"mfma_busy_frac": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),       # 1024 SIMDs
             "valu_active_frac": round(4 * vals["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 4),      # the counter ticks in quad-cycles
             "kernel_cycles_profiled": int(cyc)}
    return traffic, (f"measured in this run: child rocprofv3 --pmc passes, FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB x2 (gfx950 wide-read "
                     f"correction) + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB per launch"), pipes


def cpu_baseline(a, N, margs, params):
    """The oracle (a port of the reference algorithm, oracle/prd_oracle.py) on this box's host cores: one network step of the
    same complex per measurement.  Thread counts {8, 16, 32, 64, physical cores} are swept (one step each after a warm-up),
    then the best one is re-run: min of 5 steps.  Reported baseline, not the target."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import prd_oracle as O
    from protein_redesign_amd.synthetic import NoiseSource, synthetic_batch
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    logical = os.cpu_count()
    pb = O.prepare_batch(synthetic_batch([(a.atoms, a.residues)], seed=0), 0.3, [NoiseSource(0, 0).randperm(a.residues)])
    g = torch.Generator().manual_seed(0)
    z, seq_t, t = torch.randn(1, N, 3, generator=g), torch.randn(1, N, 21, generator=g), torch.tensor([T // 2])

    def one_step():
        c0 = time.perf_counter()
        with torch.inference_mode():
            O.network_step(params, margs, pb, z, seq_t, pb["residue_and_atom_mask"], t)
        return time.perf_counter() - c0

    default_threads = torch.get_num_threads()
    one_step()                                                   # warm-up (allocator, oneDNN primitives)
    sweep = {}
    for n in sorted({n for n in (8, 16, 32, 64, phys) if n <= logical}):
        torch.set_num_threads(n)
        sweep[n] = one_step()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]] + [one_step() for _ in range(4)]
    torch.set_num_threads(default_threads)
    return {"value": round(1.0 / min(times), 4), "unit": "denoising-steps/s", "cores": best, "kind": "port",
            "physical_cores": phys, "logical_cpus": logical,
            "thread_sweep_s_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": f"oracle/prd_oracle.py network_step on the same N={N} complex (reverse update negligible): 1 warm-up, one step "
                      f"per thread count, then min of 5 steps at the best count ({best} threads)"}


def main():
    a = parse_args()
    if a.train:                                                  # a separate JSON line with its own metric, never the headline
        sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "train_bench.py")]
        sys.path.insert(0, os.path.dirname(sys.argv[0]))
        import train_bench
        return train_bench.main()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.kernel_only:
        sys.exit(self_launch(a))                                 # before any GPU call in this process

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); n_gpus reports the RCCL world size",
              file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run, also with one rank
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)           # "nccl" is RCCL on ROCm
        world = dist.get_world_size()

    from protein_redesign_amd import _lib, ops
    from protein_redesign_amd.diffusion_model import ReverseDiffusion
    from protein_redesign_amd.distributed import gather_samples, sample_sharded
    from protein_redesign_amd.synthetic import NoiseSource, batch_to, synthetic_batch

    model, margs, params = build_model(dev, graph=not a.no_graph)
    bpg = a.samples_per_gpu
    N = a.atoms + a.residues
    batch = synthetic_batch([(a.atoms, a.residues)] * bpg, seed=0)
    sources = [NoiseSource(0, rank * bpg + k) for k in range(bpg)]      # keyed by GLOBAL sample index
    loop = ReverseDiffusion(model, batch_to(batch, dev), sources)

    if a.kernel_only:                                            # child of measure_traffic(): only the dominant kernel
        dominant_kernel(model, loop.mask, bpg, N, dev, reps=6)
        return

    def advance(n):
        for _ in range(n):
            if loop.steps_done >= loop.T:
                loop.reset()
            loop.step()

    advance(2)                         # eager step + graph capture (setup, not warm-up)
    advance(a.warmup)
    gather_samples(*loop.result(), world * bpg)                  # warm the collective (communicator setup is not a step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    advance(a.steps)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                             # this rank's own loop, before it waits for anybody
    pos_all, logits_all = gather_samples(*loop.result(), world * bpg)   # the sampling job's one collective (RCCL all_gather)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rank_ms = [dt_own / a.steps * 1e3]
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        own = torch.tensor([dt_own / a.steps * 1e3], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(every, own)
        rank_ms = [float(v.item()) for v in every]
    # the guard of the split-16 default (diffusion_model.ReverseDiffusion.finite): the sticky flag the step-boundary kernel keeps
    # over ALL steps of the loop or'ed with a check of the final state -- one host read, outside the timed region
    finite = bool(loop.finite() and torch.isfinite(pos_all).all() and torch.isfinite(logits_all).all())

    # ---- N > 1: the product's sharded sampler end to end, against single-rank runs (bit-identical by construction) ----
    shard_check = None
    if dist is not None and not a.no_shard_check:
        model.num_steps, model.setup_schedule = 24, False        # a short loop: this is a correctness check, not the timed region
        one = batch_to(synthetic_batch([(a.atoms, a.residues)], seed=0), dev)
        c0 = time.perf_counter()
        pos_s, log_s = sample_sharded(lambda bt, src: model.sample(bt, sources=src), one, world * bpg, seed=0, batch_size=bpg)
        torch.cuda.synchronize()
        secs = time.perf_counter() - c0
        # the block of samples ANOTHER rank drew (this rank's own block when world = 1), recomputed here in a batch of the SAME
        # size: identical launch shapes give bit-identical results.  (A sample drawn alone agrees with the same sample drawn in a
        # batch only to fp32 round-off -- the row kernels order their tasks by batch size -- so that comparison is reported with
        # a bound, not required to be exact.)
        k = ((rank + 1) % world) * bpg
        from protein_redesign_amd.distributed import repeat_batch
        clone = lambda d: {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in d.items()}      # noqa: E731
        pos_b, log_b = model.sample(repeat_batch(clone(one), bpg), sources=[NoiseSource(0, k + j) for j in range(bpg)])
        same = torch.tensor([int(torch.equal(pos_s[k:k + bpg], pos_b) and torch.equal(log_s[k:k + bpg], log_b))], device=dev)
        pos_1, log_1 = model.sample(clone(one), sources=[NoiseSource(0, k)])
        alone = max(float((pos_s[k] - pos_1[0]).norm() / pos_1[0].norm()), float((log_s[k] - log_1[0]).norm() / log_1[0].norm()))
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        shard_check = {"num_samples": world * bpg, "num_steps": 24, "identical_to_same_batch_size_recompute": bool(same.item()),
                       "batch_size": bpg, "rel_l2_vs_sample_drawn_alone": float(f"{alone:.3g}"), "seconds": round(secs, 3)}
        if not same.item() or not alone < 2e-5:
            raise SystemExit("bench.py: sample_sharded over RCCL differs from the single-rank samples")
        model.num_steps, model.setup_schedule = T, False

    # ---- dominant kernel, measured live with HIP events on the launch stream (rank 0) ----
    roofline = None
    if rank == 0:
        kus = dominant_kernel(model, loop.mask, bpg, N, dev, reps=20)
        kfl = tri_attn_core_flops(bpg, N, P)
        ach = kfl / (kus * 1e-6) / 1e12
        split_mode_now = _lib.lib().prd_get_gemm_mode() == 1
        traffic, note, pipes = (None, "not measured (--no-traffic or N > 1)", {})
        if world == 1 and not a.no_traffic:
            traffic, note, pipes = measure_traffic(a)
        split_mode = _lib.lib().prd_get_gemm_mode() == 1
        variant = ops.tri_attn_variant(N, P)
        v2 = split_mode and variant in (0, 1, 2) and ops.tri_attn_v2_supported(N, P) and not os.environ.get("PRD_TA_VARIANT", "0").strip("0")
        split = split_mode and variant in (0, 2)
        kname = {1: "tri_attn_core_v2_kernel", 2: "tri_attn_core_v3_kernel", 3: "tri_attn_core_v2l_kernel"}[
            _lib.lib().prd_tri_attn_v2_form(N, P)] if v2 else {
            0: "tri_attn_core_split_kernel" if split else "tri_attn_core_kernel",
            1: "tri_attn_core_long_kernel", 2: "tri_attn_core_split_long_kernel", 3: "tri_attn_core_chunk_kernel"}[variant]
        # peak: the kernel issues on the 16-bit matrix pipe, where an fp32-accurate MAC costs three split products (hi*hi + hi*lo +
        # lo*hi): 2.5 PF/s / 3.  In fp32 mode (fp32 MFMA kernels) the peak is the fp32 MFMA rate.
        peak = PEAK_16BIT_TFLOPS / 3.0 if split_mode_now else FP32_PEAK_TFLOPS
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "frac_of_fp32_mfma_peak": round(ach / FP32_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_note": note,
                    "algorithmic_bytes_per_launch": tri_attn_core_bytes(bpg, N, P),
                    "kernel": kname, "launches_per_step": 2 * NB,
                    "flops_per_launch": kfl, "avg_launch_us": round(kus, 2),
                    "peak_note": "achieved = ALGORITHMIC fp32 flops of the launch / its duration; peak = the rate at which the pipe the kernel "
                                 "issues on delivers fp32-accurate MACs: 2.5 PF/s dense fp16 MFMA / 3 split products per MAC (fp32 mode: "
                                 "157.3 TF/s fp32 MFMA).  frac_of_fp32_mfma_peak is round 1-2's pricing (SURVEY 8d), kept for continuity: "
                                 "above 1 it only says the 16-bit pipe is in use.  frac_of_16bit_peak prices what the kernel EXECUTES "
                                 "(MFMA instructions x 32768 flops) against 2.5 PF/s; mfma_busy_frac / valu_active_frac are the measured "
                                 "pipe occupancies (eager launches under the profiler).  The kernel is VALU-bound, not MFMA-bound: one "
                                 "v_exp_f32 and one fp16 hi|lo split per logit (DESIGN.md 4.3; profiles/r06_roofline.txt covers every "
                                 "kernel of the step)"}
        if v2:         # every MFMA is a 32x32x16 (32768 flops): 36 per 32-position block of a row (3 row GEMMs x 4 k-steps x 3
            nqb = (N + 31) // 32                                     # products; long rows: [K|Q] + [V] + [Q|G] = 36 too),
            ex = bpg * N * 4 * (36 * nqb + 7 * nqb * nqb) * 32768   # 7 per 32 x 32 logit tile (3 QK^T + 4 PV)
        elif split:    # first generation: 3 fp16 products per projection / P*V MAC, 6 bf16 products per Q*K^T MAC (long rows: 4)
            ex = bpg * (3 * 8 * N * N * P * 64 + ((6 if variant == 0 else 4) + 3) * 2 * 64 * N ** 3)
        else:
            ex = None
        if ex:
            roofline.update({"executed_16bit_mfma_flops_per_launch": ex,
                             "executed_16bit_tflops": round(ex / (kus * 1e-6) / 1e12, 1), "peak_16bit_tflops": PEAK_16BIT_TFLOPS,
                             "frac_of_16bit_peak": round(ex / (kus * 1e-6) / 1e12 / PEAK_16BIT_TFLOPS, 4)})
        roofline.update(pipes)

    # ---- the same loop in the other arithmetic (rank 0, single process): one line shows both ----
    other_ms = None
    if rank == 0 and world == 1 and not a.no_traffic:
        cur = _lib.lib().prd_get_gemm_mode()
        _lib.lib().prd_set_gemm_mode(1 - cur)
        try:
            model2, _, _ = build_model(dev, graph=not a.no_graph)
            loop2 = ReverseDiffusion(model2, batch_to(batch, dev), [NoiseSource(0, k) for k in range(bpg)])
            for _ in range(2 + a.warmup):
                loop2.step()
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for _ in range(min(a.steps, 100)):
                loop2.step()
            torch.cuda.synchronize()
            other_ms = round((time.perf_counter() - c0) / min(a.steps, 100) * 1e3, 4)
        finally:
            _lib.lib().prd_set_gemm_mode(cur)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a, N, margs, params)

    if rank == 0:
        total_steps = world * bpg * a.steps
        value = total_steps / dt
        flops = step_flops(N, S, P, nb=NB)
        b3 = _lib.lib().prd_get_gemm_mode() == 1
        out = {
            "metric": "denoising-steps/sec (fwd+rev) per complex, N=256 res", "value": round(value, 3),
            "unit": "denoising-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic {a.residues}-residue + {a.atoms}-atom ligand (N={N}), "
                                   f"single_dim={S} pair_dim={P} num_blocks={NB} num_steps={T}, "
                                   f"{bpg} complex/GPU, sharded by sample index (no data-path collective; one all_gather "
                                   f"of the samples at the end of the timed region)",
                       "samples_per_gpu": bpg, "hip_graph": not a.no_graph, "outputs_finite": finite,
                       "row_gemm": _lib.row_gemm_description(b3),
                       ("fp32_mode_ms_per_step" if b3 else "split16_mode_ms_per_step"): other_ms,
                       "backend": "nccl (RCCL)" if dist is not None else "single process", "sharded_sample_check": shard_check,
                       # each rank's OWN loop time per step, before the closing all_gather / barrier: imbalance shows here, the
                       # headline value is priced by the slowest rank (max over ranks of the whole timed region)
                       "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                            "all": [round(v, 4) for v in rank_ms]}},
            "step_gflop": round(flops / 1e9, 1),
            "step_tflops": round(flops * bpg / (dt / a.steps) / 1e12, 2),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
