#!/usr/bin/env python3
"""Benchmark of the hot path: HigherHRNet-w48 forward + heatmap->keypoint decode
at 640x640 on N MI355X (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole path over one batch per GPU: forward (half
wrapper numerics, fp32 NCHW in/out) -> fused bilinear+NMS+top-k -> host
match_by_tag -> adjust/refine -> fixed-size keypoint records all-gathered over
the process group.  Steps are software-pipelined (the forward of step k+1 is
enqueued before the host part of step k's decode); the timed region contains
exactly K forwards and K decodes, the last decode included.  Inputs are synthetic and resident in HBM before the timed
region; weights are seeded random (no checkpoint / dataset is reachable).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFS = 2500.0       # dense fp16/bf16 MFMA


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--weights", default="W0", choices=["W0", "W1"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=24)
    ap.add_argument("--dump-ops", default="", help="write the per-op table to this file")
    return ap.parse_args()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # host thread pools: the box gives one GPU a CPU quota (cgroup); hundreds of idle-spinning OpenMP
    # threads (torch defaults to one per visible core) exhaust it and the kernel launches stall
    host_threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8, 8))
    torch.set_num_threads(host_threads)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from oracle import synth
    from rtpe import engine
    from rtpe.helpers import build_hrnet_w48_teacher

    # ---- weights: rank 0 builds them, everyone gets them over RCCL ------------
    with open(os.path.join(ROOT, "tests", "golden", "w48_shapes.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f)["shapes"].items()}
    sd = synth.make_state_dict(shapes, 0, args.weights) if rank == 0 else \
        {k: torch.zeros(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)
         for k, v in shapes.items()}
    sd = engine.broadcast_state_dict(sd, 0, dev)
    model = build_hrnet_w48_teacher({"1." + k: v for k, v in sd.items()})
    pipe = engine.TeacherPipeline(model, device=dev)
    eng = model[1]._engine(dev)

    B, S = args.batch, args.size
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.randn(B, 3, S, S, generator=g, device=dev)
    ids = [rank * B + i for i in range(B)]
    n_ops = len(eng.program.ops)
    op_ms = np.zeros(n_ops)
    people = [0]

    n_slots = 4
    for slot in range(n_slots):          # creates the per-op events of every record slot outside the timed region
        eng.forward_record(x, slot)

    def run_steps(k_steps, record):
        """k_steps pipelined steps: forward(k+1) is enqueued before decode(k)'s host work"""
        def fwd(k, xb):
            # per-op events for the first n_slots timed steps only (their markers cost launch gaps)
            return eng.forward_record(xb, k) if record and k < n_slots else eng.forward(xb)
        last = None
        for res in pipe.stream((x for _ in range(k_steps)), (S, S), on_forward=fwd):
            people[0] = sum(len(p) if p.ndim == 3 else 0 for p, _ in res)
            last = pipe.gather(ids, res, equal_counts=True)
        return last

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    rec = run_steps(args.steps, True)
    fence()
    dt = time.perf_counter() - t0
    for k in range(min(args.steps, n_slots)):
        op_ms[:] += np.asarray(eng.read_record(k))
    n_rec = min(args.steps, n_slots)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # forward-only and decode-only rates (outside the timed region, for the record)
    fence()
    t1 = time.perf_counter()
    for _ in range(max(2, args.steps // 2)):
        eng.forward(x)
    torch.cuda.synchronize(dev)
    fwd_s = (time.perf_counter() - t1) / max(2, args.steps // 2)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel --------------------------------------
    op_ms /= max(1, n_rec)
    names = eng.program.names
    costs = [eng.op_cost(i, B, S, S) for i in range(n_ops)]
    by_name = {}
    names = list(names)
    for i in range(n_ops):
        t = eng.op_tile(i, B, S, S)
        ds = eng.program.tensors[eng.program.ops[i].out_t].ds_log2 if eng.program.ops[i].out_t >= 0 else 1
        names[i] = "%s @/%d" % (names[i], 1 << ds) + (
            " [m%d n%d w%d %dx%d cc%d cb%d%s]" % (tuple(t[:7]) + ((" FUSED-BLOCK" if t[7] == -900001 else " (in fused block)" if t[7] == -900002 else " S%d/%d" % (-t[7] % 100000, -t[7] // 100000) if t[7] <= -100000 else " P%d" % -t[7]) if t[7] < 0 else "",)) if t[0] else "")
    for i, nm in enumerate(names):
        d = by_name.setdefault(nm, dict(n=0, ms=0.0, flops=0.0, bytes=0.0, res={}))
        d["n"] += 1
        d["ms"] += op_ms[i]
        d["flops"] += costs[i][0]
        d["bytes"] += costs[i][1]
    # dominant = the fused 3x3 C=48 BasicBlock conv at 160x160 (64 launches / forward)
    dom_idx = [i for i, nm in enumerate(names) if nm.startswith("conv 48->48 k3s1")
               and eng.program.tensors[eng.program.ops[i].out_t].ds_log2 == 2]
    dom_tiles = [eng.op_tile(i, B, S, S) for i in dom_idx]
    heads = [i for i, t in zip(dom_idx, dom_tiles) if t[7] == -900001]
    if heads and len(heads) * 2 == len(dom_idx):
        # the two convs of a BasicBlock run as ONE kernel: a launch = the block, its algorithmic bytes =
        # the layer-fused traffic of both convs (conv in + out, + residual), its time = the pair's
        dom_launches = len(heads)
        dom_ms = float(np.mean([op_ms[i] + op_ms[i + 1] for i in heads]))
        dom_bytes = float(np.mean([costs[i][1] + costs[i + 1][1] for i in heads]))
        dom_flops = float(np.mean([costs[i][0] + costs[i + 1][0] for i in heads]))
        dom_kernel = "conv_block_kernel (fused BasicBlock: conv+BN+ReLU+conv+BN+add+ReLU)"
    else:
        dom_launches = len(dom_idx)
        dom_ms = float(np.mean([op_ms[i] for i in dom_idx]))
        dom_bytes = float(np.mean([costs[i][1] for i in dom_idx]))
        dom_flops = float(np.mean([costs[i][0] for i in dom_idx]))
        dom_kernel = "conv_stream_kernel<3,*,4>" if all(t[7] <= -100000 for t in dom_tiles) else \
            "conv_mfma_kernel<3,*,4>" if all(t[7] > 0 for t in dom_tiles) else "conv_stream_kernel / conv_mfma_kernel"
    total_flops = sum(c[0] for c in costs)
    total_bytes = sum(c[1] for c in costs)
    fwd_ms_events = float(op_ms.sum())
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured with
    # rocprofv3 on this kernel and committed under profiles/ (bench.py cannot run the profiler on itself)
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("kernel", "").split("<")[0].split(" ")[0] == dom_kernel.split("<")[0].split(" ")[0] and tj.get("batch") == B:
                traffic, traffic_src = float(tj["hbm_bytes_per_launch"]["mean"]), "profiles/r01_hbm_traffic.json"
        except (ValueError, KeyError):
            pass
    roofline = {
        "kernel": "%s, 3x3 s1 48->48 @160x160, %d launches/forward" % (dom_kernel, dom_launches),
        "bound": "hbm", "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
        "traffic_source": traffic_src,
        "launch_us": round(dom_ms * 1e3, 2), "bytes_per_launch": dom_bytes,
        "mfma_tflops": round(dom_flops / (dom_ms * 1e-3) / 1e12, 1),
        "forward_tflops": round(total_flops / (fwd_ms_events * 1e-3) / 1e12, 1),
        "forward_mfma_frac": round(total_flops / (fwd_ms_events * 1e-3) / 1e12 / MFMA_PEAK_TFS, 4),
        "forward_hbm_gbs": round(total_bytes / (fwd_ms_events * 1e-3) / 1e9, 1),
    }
    if args.dump_ops:
        with open(args.dump_ops, "w") as f:
            f.write("# per-op-class HIP-event times, batch %d, %dx%d, avg over %d steps\n" % (B, S, S, args.steps))
            f.write("%-66s %4s %9s %9s %9s %8s\n" % ("op", "n", "ms_total", "TFLOP/s", "GB/s", "us/op"))
            for nm, d in sorted(by_name.items(), key=lambda kv: -kv[1]["ms"]):
                f.write("%-66s %4d %9.3f %9.1f %9.1f %8.1f\n" % (
                    nm, d["n"], d["ms"], d["flops"] / max(d["ms"], 1e-9) / 1e9,
                    d["bytes"] / max(d["ms"], 1e-9) / 1e6, d["ms"] / d["n"] * 1e3))
            f.write("forward total (events) %.3f ms; wall %.3f ms\n" % (fwd_ms_events, fwd_s * 1e3))

    # ---- CPU baseline: the oracle (a port of the reference path) on the host cores
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import decode_ref, hrnet_ref
        # the GPU box gives one GPU a share of the host: use the cores we are allowed
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, 16))
        torch.set_num_threads(cores)
        print("cpu baseline: oracle on %d threads ..." % cores, file=sys.stderr, flush=True)
        # fp32 weights: the fastest form of the reference's CPU path on any host (PyTorch-CPU fp16
        # convolutions, which the half wrapper uses, are 50x slower on CPUs without fp16 units)
        net = hrnet_ref.OracleNet(sd, half=False)
        xc = x[:max(1, args.cpu_images)].cpu()
        c0 = time.perf_counter()
        n_done = 0
        t_fwd = t_dec = 0.0
        for i in range(xc.shape[0]):
            a0 = time.perf_counter()
            p, r = net(xc[i:i + 1])
            a1 = time.perf_counter()
            hms = decode_ref.upsample_bilinear(r, S, S)
            aes = decode_ref.upsample_bilinear(p[:, 17:], S, S)
            decode_ref.HeatmapParserRef().parse(hms, aes.unsqueeze(-1))
            t_fwd += a1 - a0
            t_dec += time.perf_counter() - a1
            n_done += 1
            print("cpu baseline: image %d done at %.1f s" % (n_done, time.perf_counter() - c0), file=sys.stderr,
                  flush=True)
            if time.perf_counter() - c0 > 40:
                break
        cdt = time.perf_counter() - c0
        # the half wrapper itself (what get_hrnet_w48_teacher builds), on a 128x128 crop
        h0 = time.perf_counter()
        hrnet_ref.OracleNet(sd, half=True)(xc[:1, :, :128, :128])
        half_s = (time.perf_counter() - h0) * (S / 128.0) ** 2
        cpu = {"value": round(n_done / cdt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
               "sample": "%d image(s) of the same synthetic batch at %dx%d: oracle/ forward with fp32 weights "
                         "(%.2f s) + bilinear + parse (%.2f s) per image on %d torch threads; the half-wrapper "
                         "forward costs ~%.0f s per image on this host (128x128 crop, scaled by area)"
                         % (n_done, S, S, t_fwd / max(n_done, 1), t_dec / max(n_done, 1), cores, half_s)}

    value = world * B * args.steps / dt
    out = {
        "metric": "images/sec at 640x640 (HRNet-w48 fwd+decode)", "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 (fp32 accumulate, fp32 BatchNorm; fp32 NCHW in/out)",
        "data": "synthetic",
        "config": {"workload": "configs[2]: batch=%d per GPU, %dx%d, half-wrapper teacher forward + decode of "
                               "17 keypoint channels, weights %s (seeded random)" % (B, S, S, args.weights),
                   "batch_per_gpu": B, "people_per_batch_rank0": people[0],
                   "forward_only_images_per_sec_per_gpu": round(B / fwd_s, 1)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
