#!/usr/bin/env python3
"""Benchmark of the hot path: HigherHRNet-w48 forward + heatmap->keypoint decode
at 640x640 on N MI355X (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N ranks as a child)
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
    # 100 steps of ~14 ms: the fill / drain of the software pipeline and the two recorded steps that run alone are 3 % of 20
    # steps (2,274-2,329 img/s on one box) and 1 % of 100 (2,361-2,368)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--weights", default="W0", choices=["W0", "W1", "W2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-ops", default="", help="write the per-op table to this file")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default: nccl = RCCL on a GPU, gloo without)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="no GPU work: the control path of the benchmark only (rank start-up, process group, weight "
                         "broadcast, shards, record gather with faked decode results, barrier + max-over-ranks timing); "
                         "for CPU tests of the launcher, with --backend gloo.  The line says so and carries no rate")
    ap.add_argument("--list", default="", help="configs[3]: a file of image names (one per line, e.g. tests/golden/"
                                               "coco_minival2017_100.txt); a step = one pass over the whole list, "
                                               "sharded contiguously over the ranks, uneven shards, short last batches, "
                                               "all-gather of the keypoint records with counts")
    import glob
    summaries = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
    ap.add_argument("--pmc-json", default=summaries[-1] if summaries else os.path.join(ROOT, "profiles", "pmc_summary.json"),
                    help="per-kernel PMC summary written by tools/pmc_summary.py from rocprofv3 --pmc passes of THIS "
                         "command (default: the latest profiles/rNN_pmc_summary.json); roofline.traffic is read from it "
                         "when it was made from the same kernel sources")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 4],
                    help="BASELINE.json configs[]: 2 (default) = batch 32 fp16 teacher forward + decode, the headline; "
                         "1 = batch 1 fp32 teacher forward + decode (latency form of the same metric); 4 = the "
                         "AttentionStudent at 320x320, dual-head decode.  1 and 4 print their own JSON line")
    return ap.parse_args()


def csrc_sha16():
    """identity of the kernel sources a measurement belongs to (the GPU box has no .git): sha1 over csrc/ and the header"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "realtime-pose-estimation_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/rtpe_hip.h"]:
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(sd, x, S):
    """the CPU path on the host cores this process may use - N = 1 and N = 4, fp32 and the half
    wrapper, forward and decode separately, min and mean of 3 warm runs.  PyTorch-CPU runs the half wrapper's
    fp16 convolutions ~400x slower than fp32 on hosts without fp16 vector units (~115 s per 640x640 image on the
    GPU box), so that leg is timed on a 128x128 crop and scaled by area; everything else runs at full size."""
    from oracle import decode_ref, hrnet_ref
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    print("cpu baseline: oracle on %d threads ..." % cores, file=sys.stderr, flush=True)
    xc = x[:4].cpu()

    def timed(fn, runs=3):
        fn()                                                   # warm-up
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return out, min(ts), sum(ts) / len(ts)

    net32 = hrnet_ref.OracleNet(sd, half=False)
    rows = {}
    outs = {}
    for n in (1, 4):
        outs[n], mn, mean = timed(lambda: net32(xc[:n]))
        rows["forward_fp32_N%d" % n] = (mn, mean)
        print("cpu baseline: fp32 forward N=%d %.2f s" % (n, mean), file=sys.stderr, flush=True)

    def decode(n):
        p, r = outs[n]
        for i in range(n):
            hms = decode_ref.upsample_bilinear(r[i:i + 1], S, S)
            aes = decode_ref.upsample_bilinear(p[i:i + 1, 17:], S, S)
            decode_ref.HeatmapParserRef().parse(hms, aes.unsqueeze(-1))
    for n in (1, 4):
        _, mn, mean = timed(lambda: decode(n))
        rows["decode_N%d" % n] = (mn, mean)
        print("cpu baseline: decode N=%d %.2f s" % (n, mean), file=sys.stderr, flush=True)
    crop = 128
    net16 = hrnet_ref.OracleNet(sd, half=True)
    _, mn, mean = timed(lambda: net16(xc[:1, :, :crop, :crop]))
    scale = (S / float(crop)) ** 2
    rows["forward_half_N1_scaled"] = (mn * scale, mean * scale)
    per_img = rows["forward_fp32_N1"][1] + rows["decode_N1"][1]
    return {"value": round(1.0 / per_img, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "oracle/ on %d torch threads, min / mean of 3 warm runs at %dx%d: fp32 forward N=1 %.3f / %.3f s, "
                      "N=4 %.3f / %.3f s; bilinear + parse N=1 %.3f / %.3f s, N=4 %.3f / %.3f s; half-wrapper forward "
                      "N=1 ~%.0f / %.0f s (timed on a %dx%d crop, scaled by area; N=4 not run: 4x that).  value = "
                      "1 / (fp32 forward + decode, N=1 means): the fastest form of the reference's CPU path" % (
                          cores, S, S, *rows["forward_fp32_N1"], *rows["forward_fp32_N4"], *rows["decode_N1"],
                          *rows["decode_N4"], *rows["forward_half_N1_scaled"], crop, crop),
            "seconds": {k: [round(v[0], 4), round(v[1], 4)] for k, v in rows.items()}}


def run_list_mode(args, pipe, dev, rank, world):
    """configs[3]: the image list sharded over the ranks (rtpe.engine.run_sharded_list).  Images are synthetic
    (one seeded 640x640 input per image id, generated on the device); a step is one pass over the whole list."""
    import torch.distributed as dist
    from rtpe import engine
    names = [ln.strip() for ln in open(args.list) if ln.strip()]
    S, B = args.size, args.batch
    gen = torch.Generator(device=dev)
    people = [0]

    def infer(part):
        xb = torch.empty((len(part), 3, S, S), device=dev)
        for i, nm in enumerate(part):
            gen.manual_seed(engine.image_id_of(nm))
            xb[i] = torch.randn(3, S, S, generator=gen, device=dev)
        res = pipe(xb, out_hw=(S, S))
        people[0] += sum(len(p) if p.ndim == 3 else 0 for p, _ in res)
        return res

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)
    out = None
    for _ in range(max(1, args.warmup)):
        out = engine.run_sharded_list(names, infer, B, dev)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = engine.run_sharded_list(names, infer, B, dev)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if (args.backend or "nccl") == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        shards = [len(engine.shard_indices(len(names), r, world)) for r in range(world)]
        print(json.dumps({
            "metric": "images/sec at 640x640 (HRNet-w48 fwd+decode)", "value": round(len(names) * args.steps / dt, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16 (fp32 accumulate, fp32 BatchNorm; fp32 NCHW in/out)", "data": "synthetic",
            "config": {"workload": "configs[3]: %d image names of %s sharded over %d rank(s) as %s, batches of <= %d, "
                                   "one seeded synthetic %dx%d input per image id, keypoint records all-gathered with "
                                   "counts; every id came back exactly once on every rank" % (
                                       len(names), os.path.basename(args.list), world, shards, B, S, S),
                       "images_gathered": len(out), "backend": args.backend or ("nccl" if world > 1 else "none")},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_other_config(args, dev):
    """configs[1] and configs[4] of BASELINE.json as their own JSON lines (rank 0 of a 1-GPU run; the headline stays
    configs[2]).  Steps are synchronous forward + decode calls, as the reference's loops make them (batch 1)."""
    from oracle import synth
    from rtpe import engine
    from rtpe.third_party.group import HeatmapParser
    parser = HeatmapParser(num_joints=engine.NUM_HEATMAPS, **engine.HM_PARSER_PARAMS)
    if args.config == 1:
        from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
        with open(os.path.join(ROOT, "tests", "golden", "w48_shapes.json")) as f:
            shapes = {k: tuple(v) for k, v in json.load(f)["shapes"].items()}
        net = PoseHigherResolutionNet()
        net.load_state_dict(synth.make_state_dict(shapes, 0, args.weights), strict=True)
        net = net.to(dev).eval()
        B, S = 1, args.size
        x = torch.randn(B, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))

        def step():
            preds, refined = net(x)
            return parser.parse_lowres(refined, preds[:, engine.NUM_HEATMAPS:], (S, S))
        eng = net._engine(dev)
        name = "configs[1]: HigherHRNet-w48 %dx%d batch=1 fp32 (exact-fp32 MFMA, no half wrapper), forward + decode of 17 " \
               "keypoint channels, synchronous per image as validate_hhrnet.py:84-105 runs it" % (S, S)
        dtype, peak_tf = "f32", 157.3
    else:
        from rtpe.students import AttentionStudent
        with open(os.path.join(ROOT, "tests", "golden", "student_shapes.json")) as f:
            shapes = {k: tuple(v) for k, v in json.load(f)["shapes"].items()}
        stu = AttentionStudent(None, "cpu", 100, 17, 1, True, None, False).eval()
        stu.load_state_dict(synth.make_state_dict(shapes, 3, "W1"), strict=True)
        stu = stu.to(dev)
        B, S = args.batch, 320
        x = torch.randn(B, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))

        def step():
            att, det = stu(x)
            det = det.float()
            return parser.parse_lowres(det[:, :engine.NUM_HEATMAPS].contiguous(),
                                       det[:, engine.NUM_HEATMAPS:engine.NUM_HEATMAPS + 1].expand(-1, engine.NUM_HEATMAPS, -1, -1).contiguous(),
                                       (S, S))
        eng = stu._engine(dev) if hasattr(stu, "_engine") else None
        name = "configs[4]: AttentionStudent(inplanes=100) %dx%d batch=%d, fp16 stem (half wrapper) + fp32 heads, attention + " \
               "keypoint dual-head decode (rtpe.engine.eval_student's path)" % (S, S, B)
        dtype, peak_tf = "f16 stem / f32 heads", None
    with torch.no_grad():
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        # the same work through the pipelined loop (TeacherPipeline.stream: two forwards in flight, decode of image
        # k-1 beside forward k) - what a caller gets who feeds the images as a stream instead of one at a time
        piped = None
        if args.config == 1:
            pipe = engine.TeacherPipeline(net, device=dev)
            for _ in pipe.stream((x for _ in range(4)), (S, S)):
                pass
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n_p = 0
            for _ in pipe.stream((x for _ in range(2 * args.steps)), (S, S)):
                n_p += 1
            torch.cuda.synchronize(dev)
            piped = round(B * n_p / (time.perf_counter() - t1), 2)
        # forward alone, per-op events
        fwd = None
        roof = None
        if eng is not None:
            try:
                _, ms = eng.forward_timed(x if args.config == 1 else x)
                n_ops = len(eng.program.ops)
                costs = [eng.op_cost(i, B, S, S) for i in range(n_ops)]
                fl, by = sum(c[0] for c in costs), sum(c[1] for c in costs)
                fwd = float(sum(ms))
                i_dom = int(np.argmax(ms))
                # which roof binds: the fp32 ops (RTPE_F_F32: exact-fp32 MFMA at the fp32 vector rate, 157.3 TF) and the
                # fp16 ops (2,500 TF) each at their own matrix peak, against the layer-fused bytes at 8 TB/s
                from rtpe import _native as nat
                fl32 = sum(c[0] for i, c in enumerate(costs) if eng.program.ops[i].flags & nat.F_F32)
                t_mfma = fl32 / (157.3e12) + (fl - fl32) / (MFMA_PEAK_TFS * 1e12)
                t_hbm = by / (HBM_PEAK_GBS * 1e9)
                mfma_bound = t_mfma >= t_hbm
                if mfma_bound:
                    blended = fl / t_mfma / 1e12          # the peak this mix of precisions could reach
                    label = "mfma" if fl32 == 0 else ("mfma (fp32)" if fl32 >= 0.99 * fl else
                                                      "mfma (fp32 ops at 157.3 TF, %.0f %% of the FLOPs; fp16 ops at 2,500 TF)" % (100 * fl32 / fl))
                    roof = {"bound": label, "achieved": round(fl / (fwd * 1e-3) / 1e12, 2), "peak": round(blended, 1),
                            "unit": "TFLOP/s", "frac": round(t_mfma / (fwd * 1e-3), 4)}
                else:
                    roof = {"bound": "hbm", "achieved": round(by / (fwd * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(t_hbm / (fwd * 1e-3), 4)}
                roof.update({"kernel": "whole forward (%d ops); slowest op: %s" % (n_ops, eng.program.names[i_dom]),
                             "traffic": None, "forward_ms_events": round(fwd, 3), "slowest_op_us": round(float(ms[i_dom]) * 1e3, 1),
                             "hbm_frac": round(t_hbm / (fwd * 1e-3), 4), "mfma_frac": round(t_mfma / (fwd * 1e-3), 4)})
            except Exception as e:          # the student engine has no two-output forward_timed: report the wall time only
                roof = {"note": "per-op timing unavailable: %s" % e}
    people = sum(len(p) if getattr(p, "ndim", 0) == 3 else 0 for p, _ in res)
    print(json.dumps({
        "metric": "images/sec at %dx%d (%s)" % (S, S, "HRNet-w48 fwd+decode" if args.config == 1 else "AttentionStudent fwd+decode"),
        "value": round(B * args.steps / dt, 2), "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic", "config": dict({"workload": name, "batch_per_gpu": B, "people_last_step": people},
                                                             **({"pipelined_images_per_sec": piped} if piped is not None else {})),
        "roofline": roof, "cpu_baseline": None}))


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves, as a
    CHILD process (never exec: a process that has touched the GPU must not be replaced, and this one has not touched
    it yet), with the same arguments; rank 0's JSON line passes through on the inherited stdout, the child's return
    code is ours.  One rank per GPU over RCCL, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                 # what torchrun would set itself (and warn about)
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def run_rehearsal(args, rank, world):
    """--rehearsal: everything of a multi-rank benchmark run except the GPU work, on the host: process group (gloo),
    weight broadcast as packed buffers, contiguous shards, fixed-size records of faked decode results all-gathered
    per step, barrier-bracketed timing with the max over ranks.  The collective code is rtpe.engine's, the one that
    runs on RCCL; nothing here touches a device."""
    import torch.distributed as dist
    from rtpe import engine
    sd = {"w": torch.arange(12.).reshape(3, 4), "h": torch.ones(5).half(), "n": torch.tensor(7)} if rank == 0 else \
        {"w": torch.zeros(3, 4), "h": torch.zeros(5).half(), "n": torch.tensor(0)}
    sd = engine.broadcast_state_dict(sd, 0, "cpu")
    assert sd["w"][2, 3] == 11. and int(sd["n"]) == 7, "weight broadcast"
    B = args.batch
    ids = [rank * B + i for i in range(B)]
    fake = [(np.full((1 + i % 3, 17, 4), float(i), np.float32), [float(i % 7)] * (1 + i % 3)) for i in ids]

    def fence():
        if world > 1:
            dist.barrier()
    n_rec = 0
    for _ in range(args.warmup):
        engine.all_gather_records(engine.pack_records(ids, fake, "cpu"), equal_counts=True) if world > 1 else None
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = engine.pack_records(ids, fake, "cpu")
        allrec = engine.all_gather_records(rec, equal_counts=True) if world > 1 else rec
        n_rec = allrec.shape[0]
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    got = sorted(engine.unpack_records(allrec))
    assert got == list(range(world * B)), "gathered records of %d images, expected %d" % (len(got), world * B)
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec at 640x640 (HRNet-w48 fwd+decode)", "value": None, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tmax.item()) / max(1, args.steps) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none (rehearsal)", "data": "synthetic",
            "rehearsal": True,
            "config": {"workload": "REHEARSAL of the control path only: %d rank(s), %s backend, weight broadcast, %d faked "
                                   "records per rank and step all-gathered, no GPU work - not a measurement" % (
                                       world, args.backend or "gloo", B),
                       "records_gathered": int(n_rec)},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing has touched the GPU yet (importing torch does not): the ranks are a child of this process
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    import torch.distributed as dist
    if args.rehearsal:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend or "gloo", rank=rank, world_size=world)
        return run_rehearsal(args, rank, world)
    from rtpe import _native as nat
    # RTPE_BENCH_SHARE_GPU=1: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (ranks share
    # devices; use --backend gloo, RCCL refuses two ranks on one device).  Not a scaling measurement.
    if os.environ.get("RTPE_BENCH_SHARE_GPU", "0") == "1":
        local = local % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(args.backend or "nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # host thread pools: the box gives one GPU a CPU quota (cgroup); hundreds of idle-spinning OpenMP
    # threads (torch defaults to one per visible core) exhaust it and the kernel launches stall.  The budget
    # is the allowed cores divided among the ranks of the node (LOCAL_WORLD_SIZE), for torch's pool here and
    # for the matcher threads in rtpe.third_party.group
    host_threads = nat.host_threads(8)
    torch.set_num_threads(host_threads)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    if args.config != 2:
        if world > 1:
            raise SystemExit("--config 1 / 4 are single-GPU lines; the multi-GPU benchmark is the default configs[2]")
        return run_other_config(args, dev)
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
    if args.list:
        return run_list_mode(args, pipe, dev, rank, world)
    # four different synthetic batches, resident in HBM, fed in turn (a replayed single batch would hide
    # anything that depends on the data, e.g. buffers sized by the number of decoded people)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    xs = [torch.randn(B, 3, S, S, generator=g, device=dev) for _ in range(4)]
    x = xs[0]
    ids = [rank * B + i for i in range(B)]
    n_ops = len(eng.program.ops)
    op_ms = np.zeros(n_ops)
    people = [0]

    n_slots = 2          # per-op events in the first two timed steps only: ~330 marker packets cost ~1 ms of stream time per step
    for slot in range(n_slots):          # creates the per-op events of every record slot outside the timed region
        eng.forward_record(x, slot)

    def run_steps(k_steps, record):
        """k_steps pipelined steps: forward(k+1) is enqueued before decode(k)'s host work"""
        def fwd(k, xb):
            # per-op events for the first n_slots timed steps only (their markers cost launch gaps)
            return eng.forward_record(xb, k) if record and k < n_slots else eng.forward(xb)
        last = None
        # the recorded steps run alone (no other forward in flight): their per-op events time each kernel by itself
        for res in pipe.stream((xs[k % len(xs)] for k in range(k_steps)), (S, S), on_forward=fwd,
                               exclusive=(lambda k: record and k < n_slots)):
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
    t0, c0 = time.perf_counter(), time.process_time()
    rec = run_steps(args.steps, True)
    fence()
    dt = time.perf_counter() - t0
    host_cpu_ms = (time.process_time() - c0) / args.steps * 1e3      # CPU time of this rank's process (all threads) per step
    for k in range(min(args.steps, n_slots)):
        op_ms[:] += np.asarray(eng.read_record(k))
    n_rec = min(args.steps, n_slots)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if (args.backend or "nccl") == "nccl" else "cpu")
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
            " [m%s n%d w%d %dx%d cc%d cb%d%s]" % ((("%dx2" % -t[0]) if t[0] < 0 else str(t[0]),) + tuple(t[1:7]) + ((" FUSED-BLOCK" if t[7] == -900001 else " (in fused block)" if t[7] == -900002 else " PAIR-HEAD" if t[7] == -800001 else " PAIR" if t[7] == -800002 else " FUSED-STEM" if t[7] in (-600001, -600002) else " HEAD-DIRECT" if t[7] == -400001 else " DECONV48" if t[7] == -300001 else " CONV48S2" if t[7] == -200001 else " CONV48S2 (one launch with the next %d)" % (-t[7] - 200001) if t[7] in (-200002, -200003) else " (in the CONV48S2 launch above)" if t[7] == -200009 else " CONV64" if -600000 < t[7] <= -500000 else " S%d/%d" % (-t[7] % 100000, -t[7] // 100000) if t[7] <= -100000 else " P%d" % -t[7]) if t[7] < 0 else "",)) if t[0] else "")
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
        # 160 % 8 == 0 and 160 % 16 == 0: the producer / consumer kernel unless RTPE_BLOCK_PC=0
        hh = S // 4
        import ctypes
        opt_pc, opt_ring = ctypes.c_int32(), ctypes.c_int32()          # what the next launch will use (not the environment)
        nat.check(nat.lib().rtpe_get_option(b"block_pc", ctypes.byref(opt_pc)))
        nat.check(nat.lib().rtpe_get_option(b"block_ring", ctypes.byref(opt_ring)))
        pc = opt_pc.value != 0 and opt_ring.value == 0 and hh % 8 == 0 and hh % 16 == 0
        dom_kernel = ("conv_block_pc_kernel (fused BasicBlock: conv+BN+ReLU+conv+BN+add+ReLU; 4 producer + 4 consumer waves)"
                      if pc else
                      "conv_block_rw_kernel (fused BasicBlock: conv+BN+ReLU+conv+BN+add+ReLU, weights resident in LDS)")
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
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured with rocprofv3 on this
    # command (tools/pmc_passes.sh) and committed under profiles/: bench.py cannot run the profiler on itself, so these
    # are NOT measured in this run; they are attached only when the summary was made from the same kernel sources
    traffic = traffic_src = None
    pmc = None
    pj = {}
    same_sources = False
    if os.path.exists(args.pmc_json):
        try:
            with open(args.pmc_json) as f:
                pj = json.load(f)
            same_sources = pj.get("csrc_sha16") == csrc_sha16() and pj.get("batch") == B and pj.get("size") == S
            kname = dom_kernel.split("<")[0].split(" ")[0]
            ent = pj.get("kernels", {}).get(kname)
            if ent and same_sources:
                # all launches of this kernel in the bench run at 160x160 (the roofline class) and 320x320:
                # the summary keeps the classes apart by grid size
                cls = ent.get("classes", {}).get("160x160", ent)
                traffic = float(cls["hbm_bytes_per_launch"])
                traffic_src = os.path.relpath(args.pmc_json, ROOT)
                pmc = {k: cls[k] for k in ("mfma_util", "lds_bank_conflict_frac", "valu_mfma_coexec_frac", "wait_any_frac",
                                           "fetch_bytes", "write_bytes") if k in cls}
        except (ValueError, KeyError, TypeError):
            pass
    dom_s = dom_ms * 1e-3
    mfma_frac = dom_flops / dom_s / 1e12 / MFMA_PEAK_TFS
    hbm_meas_frac = (traffic / dom_s / 1e9 / HBM_PEAK_GBS) if traffic else None
    roofline = {
        "kernel": "%s, 3x3 s1 48->48 @160x160, %d launches/forward" % (dom_kernel, dom_launches),
        # the contract's figure: ALGORITHMIC bytes (the layer-fused traffic of the two convs of the block:
        # conv1 in + out, conv2 in + out, residual) over the launch time.  The fused kernel does not move them all:
        # what it physically does is in "physical" below
        "bound": "hbm", "achieved": round(dom_bytes / dom_s / 1e9, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(dom_bytes / dom_s / 1e9 / HBM_PEAK_GBS, 4),
        "achieved_basis": "algorithmic layer-fused bytes of both convs (5 tensor passes), not bytes moved",
        "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_this_run": False,
        "launch_us": round(dom_ms * 1e3, 2), "bytes_per_launch": dom_bytes,
        "mfma_tflops": round(dom_flops / dom_s / 1e12, 1),
        "physical": {
            "hbm_gbs_measured_traffic": round(traffic / dom_s / 1e9, 1) if traffic else None,
            "hbm_frac_measured_traffic": round(hbm_meas_frac, 4) if hbm_meas_frac is not None else None,
            "mfma_frac": round(mfma_frac, 4),
            "bound": ("mfma" if (hbm_meas_frac is None or mfma_frac >= hbm_meas_frac) else "hbm"),
            "note": "useful FLOPs / 2.5 PFLOP/s and counter bytes / 8 TB/s over the same launch time: the kernel reads x "
                    "once and writes y once and is bound on the compute side",
        },
        "forward_tflops": round(total_flops / (fwd_ms_events * 1e-3) / 1e12, 1),
        "forward_mfma_frac": round(total_flops / (fwd_ms_events * 1e-3) / 1e12 / MFMA_PEAK_TFS, 4),
        "forward_hbm_gbs": round(total_bytes / (fwd_ms_events * 1e-3) / 1e9, 1),
        "counters": pmc,
    }
    # the TIME-dominant kernel family: the 3x3 stride-1 convs with >= 96 channels on the streaming kernel (MFMA-bound)
    td_idx = [i for i, nm in enumerate(names) if nm.startswith("conv ") and " k3s1" in nm and
              eng.program.ops[i].cin >= 96 and eng.program.ops[i].cin == eng.program.ops[i].cout]
    td = None
    if td_idx:
        td_ms = float(sum(op_ms[i] for i in td_idx))
        td_flops = float(sum(costs[i][0] for i in td_idx))
        tiles = [eng.op_tile(i, B, S, S) for i in td_idx]
        kinds = sorted({"conv_stream_pc_kernel<%d>" % t[1] if t[7] <= -800000 else "conv_stream_kernel<3,%d,%d>" % (t[1], t[2]) if t[7] <= -100000
                        else "conv_mfma_kernel" for t in tiles})
        cnt = {}
        tr_sum = tr_n = 0.0
        for kn in kinds:
            ent = pj.get("kernels", {}).get(kn.replace(" ", "")) or pj.get("kernels", {}).get(kn.replace(",", ", "))
            if ent and same_sources:
                cls = next(iter(ent.get("classes", {}).values()), ent)
                cnt[kn] = {k: cls[k] for k in ("mfma_util", "wait_any_frac", "lds_bank_conflict_frac", "us_under_profiler",
                                               "hbm_bytes_per_launch") if k in cls}
                if "hbm_bytes_per_launch" in cls:
                    n_l = float(cls.get("launches_per_pass", 1))
                    tr_sum += float(cls["hbm_bytes_per_launch"]) * n_l
                    tr_n += n_l
        td_launch_s = td_ms * 1e-3 / len(td_idx)
        td = {"kernel": " + ".join(kinds) + ": 3x3 s1 convs with C in {96, 192, 384}, %d launches/forward "
                        "(the TIME-dominant family of the forward)" % len(td_idx),
              "bound": "mfma", "achieved": round(td_flops / (td_ms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s",
              "frac": round(td_flops / (td_ms * 1e-3) / 1e12 / MFMA_PEAK_TFS, 4),
              "achieved_basis": "algorithmic FLOPs of a launch (2 * 9 * Cin * Cout * pixels, 33.97 GFLOP for every class at batch "
                                "32) / mean HIP-event duration of a launch over the recorded steps",
              # HBM bytes per launch by the counters (mean over the family's launches of the PMC passes of this command,
              # committed under profiles/; attached only when made from the same kernel sources)
              "traffic": round(tr_sum / tr_n, 1) if tr_n else None,
              "traffic_source": os.path.relpath(args.pmc_json, ROOT) if tr_n else None, "traffic_measured_in_this_run": False,
              "launch_us": round(td_launch_s * 1e6, 2), "flops_per_launch": td_flops / len(td_idx),
              "ms_per_forward": round(td_ms, 3),
              "share_of_forward": round(td_ms / fwd_ms_events, 3), "counters": cnt or None}
    if args.dump_ops:
        with open(args.dump_ops, "w") as f:
            f.write("# per-op-class HIP-event times, batch %d, %dx%d, avg over %d steps\n" % (B, S, S, args.steps))
            f.write("%-66s %4s %9s %9s %9s %8s\n" % ("op", "n", "ms_total", "TFLOP/s", "GB/s", "us/op"))
            for nm, d in sorted(by_name.items(), key=lambda kv: -kv[1]["ms"]):
                f.write("%-66s %4d %9.3f %9.1f %9.1f %8.1f\n" % (
                    nm, d["n"], d["ms"], d["flops"] / max(d["ms"], 1e-9) / 1e9,
                    d["bytes"] / max(d["ms"], 1e-9) / 1e6, d["ms"] / d["n"] * 1e3))
            f.write("forward total (events) %.3f ms; wall %.3f ms\n" % (fwd_ms_events, fwd_s * 1e3))

    # ---- CPU baseline: the oracle (a port of the reference path) on the host cores -------------------
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(sd, x, S)

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
                   "pipeline": "software-pipelined steps (TeacherPipeline.stream): %s forward(s) in flight on internal streams, "
                               "decode of batch k-1 beside forward k; the %d recorded steps run alone" % (
                                   os.environ.get("RTPE_FORWARDS_IN_FLIGHT", "2"), n_slots),
                   "forward_only_images_per_sec_per_gpu": round(B / fwd_s, 1),
                   "host_cpu_ms_per_step": round(host_cpu_ms, 2), "host_threads": host_threads},
        # roofline = the time-dominant kernel family (MFMA-bound 3x3 convs with C >= 96); roofline_secondary = the fused
        # C = 48 BasicBlock (HBM-bound by SURVEY 8d's layer-fused bytes), the second-largest share of the forward
        "roofline": td if td is not None else roofline, "roofline_secondary": roofline if td is not None else None,
        "build_mode": entry.BUILD_INFO.get("mode"), "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
