"""Inference loops of the hot path.

``eval_student`` keeps the name and signature of the reference's
``rtpe/engine.py:21-75``.  ``TeacherPipeline`` is the accelerated body of the
per-image loops of ``validate_hhrnet.py:84-105`` / ``teacher_inference.py:67-90``:
batched forward on the HIP executor + fused decode, one process per GPU, with
the weights broadcast and the keypoints all-gathered over RCCL when a process
group is active.
"""
from collections import OrderedDict

import numpy as np
import torch

from .third_party.group import HeatmapParser

HM_PARSER_PARAMS = {"max_num_people": 30, "detection_threshold": 0.1, "tag_threshold": 1.0,
                    "use_detection_val": True, "ignore_too_much": False, "tag_per_joint": True,
                    "nms_ksize": 5, "nms_padding": 2}       # validate_hhrnet.py:40-47
NUM_HEATMAPS = 17
MAX_PEOPLE_RECORD = 30
RECORD_FLOATS = 2 + MAX_PEOPLE_RECORD + MAX_PEOPLE_RECORD * NUM_HEATMAPS * 4   # 8,288 B


def eval_student(model, hm_parser, val_dataloader, device,
                 plot_every=None, save_every=None, save_dir="/tmp"):
    """reference engine.py:21-75: run ``model`` over the loader, decode with
    ``hm_parser.parse`` and return the dataset's evaluation dict.

    Each batch is a tuple whose first two items are ``(img_id, img)``.  ``model``
    may return one tensor (heatmaps in channels [:17], tags in [17:], as the
    reference's students did), the teacher's ``[preds, refined]`` list, or the
    dual-head student's ``(att, det)`` tuple (config 5).
    Plotting / image saving (``plot_every`` / ``save_every``) are not part of
    the accelerated path and are ignored.
    """
    model.eval()
    all_preds, all_scores = [], []
    for batch_i, batch in enumerate(val_dataloader):
        img = batch[1]
        out_hw = tuple(img.shape[2:])
        img = img.to(device)
        with torch.no_grad():
            try:
                pred = model(img, out_hw)
            except TypeError:
                pred = model(img)
        if isinstance(pred, (list, tuple)) and pred[0].shape[1] == 1:
            # dual-head student (rtpe/students.py:724-771): (attention mask, det) with the heat maps in
            # det[:, :17] and the tags in det[:, 17:18], both at 1/4 resolution; decoded like the teacher's
            # maps (validate_hhrnet.py:93-101): bilinear to the image size, then parse
            det = pred[1].float()
            res = hm_parser.parse_lowres(det[:, :NUM_HEATMAPS].contiguous(), det[:, NUM_HEATMAPS:NUM_HEATMAPS + 1]
                                         .expand(-1, NUM_HEATMAPS, -1, -1).contiguous(), out_hw)
            grouped, scores = [res[0][0]], res[0][1]
        elif isinstance(pred, (list, tuple)):
            preds, refined = pred
            res = hm_parser.parse_lowres(refined.float(), preds[:, NUM_HEATMAPS:].float(), out_hw)
            grouped, scores = [res[0][0]], res[0][1]
        else:
            pred = pred.detach().float()
            grouped, scores = hm_parser.parse(pred[:, :NUM_HEATMAPS], pred[:, NUM_HEATMAPS:].unsqueeze(-1),
                                              adjust=True, refine=True)
        all_preds.append([x for x in grouped[0] if x.size > 0])
        all_scores.append(scores)
    dataset = getattr(val_dataloader, "dataset", None)
    if dataset is not None and hasattr(dataset, "evaluate"):
        eval_dict, _ = dataset.evaluate(all_preds, all_scores, ".", False, False)
        return eval_dict
    return OrderedDict(images=len(all_preds), people=sum(len(p) for p in all_preds))


_PIN_RING = {}


def pack_records(image_ids, results, device):
    """fixed-size keypoint records for the all-gather: per image
    ``[image_index, n_people, scores[30], kpts[30][17][4]]`` as float32 (image ids are exact up to 2^24;
    COCO's largest is 581,929).  On a GPU the records go
    through a small ring of pinned host buffers and an asynchronous copy, so packing never waits
    for the kernels queued on the stream."""
    device = torch.device(device)
    n_img = len(results)
    if device.type == "cuda":
        ring = _PIN_RING.setdefault(n_img, {"i": 0, "bufs": []})
        if len(ring["bufs"]) < 4:
            ring["bufs"].append(torch.zeros((n_img, RECORD_FLOATS), dtype=torch.float32, pin_memory=True))
        host = ring["bufs"][ring["i"] % len(ring["bufs"])]
        ring["i"] += 1
        host.zero_()
        rec = host.numpy()
    else:
        host = None
        rec = np.zeros((n_img, RECORD_FLOATS), np.float32)
    for i, (img_id, (people, scores)) in enumerate(zip(image_ids, results)):
        n = min(len(people) if people.ndim == 3 else 0, MAX_PEOPLE_RECORD)
        rec[i, 0], rec[i, 1] = img_id, n
        if n:
            rec[i, 2:2 + n] = np.asarray(scores[:n], np.float32)
            rec[i, 2 + MAX_PEOPLE_RECORD:2 + MAX_PEOPLE_RECORD + n * NUM_HEATMAPS * 4] = \
                people[:n, :, :4].reshape(-1)
    if host is not None:
        return host.to(device, non_blocking=True)
    return torch.from_numpy(rec).to(device)


def unpack_records(rec):
    out = {}
    for r in rec.cpu().numpy():
        n = int(r[1])
        kp = r[2 + MAX_PEOPLE_RECORD:2 + MAX_PEOPLE_RECORD + n * NUM_HEATMAPS * 4]
        out[int(r[0])] = (kp.reshape(n, NUM_HEATMAPS, 4).copy(), r[2:2 + n].copy())
    return out


class TeacherPipeline:
    """forward + decode for batches of pre-processed images on one GPU."""

    def __init__(self, model, parser=None, device=None):
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.model = model.to(self.device).eval()
        self.parser = parser or HeatmapParser(num_joints=NUM_HEATMAPS, **HM_PARSER_PARAMS)

    @torch.no_grad()
    def forward(self, images):
        return self.model(images)

    @torch.no_grad()
    def __call__(self, images, out_hw=None):
        """images (N,3,H,W) on the GPU -> list of (people, scores) per image;
        out_hw = decode resolution (original image size), default (H, W)."""
        preds, refined = self.model(images)
        hw = tuple(out_hw) if out_hw is not None else tuple(images.shape[2:])
        return self.parser.parse_lowres(refined, preds[:, NUM_HEATMAPS:], hw)

    def stream(self, batches, out_hw=None, on_forward=None, decode_stream=None, in_flight=None, exclusive=None):
        """Software-pipelined loop over an iterable of (N,3,H,W) GPU batches, in the order
        F(k) R(k-1) T(k)  (forward, adjust+refine of the previous batch, fused top-k).  The host part
        of the decode of batch k-1 (tag matching on the host cores) runs while the GPU executes F(k),
        whose launches are already queued; the GPU never waits for the host.  The decode tables travel
        through pinned host memory that the kernels read and write in place (no copy commands in the
        stream).

        ``decode_stream``: ``None`` (default) = environment ``RTPE_DECODE_STREAM`` (default "side");
        ``"side"`` puts R and T on a second HIP stream that waits for F(k) by event: the
        network's ~330 kernels per forward leave the chip partly idle at every kernel boundary (a
        persistent one-workgroup-per-CU kernel ends with its slowest workgroup), and the decode's many
        small workgroups fill those gaps instead of costing their own ~0.9 ms of stream time;
        ``"same"`` keeps everything on the current stream.

        ``in_flight`` (default: environment ``RTPE_FORWARDS_IN_FLIGHT``, default 2): forwards of consecutive batches
        alternate between that many internal HIP streams, each with an activation workspace of its own, so that the
        low-occupancy layers of one batch (stem, layer1, transitions, heads: HBM- and latency-bound) run beside the
        matrix-bound stages of the other - 14.5 -> 13.0-13.4 ms per forward at batch 32, 3.30 -> 2.13 ms at batch 1
        (`tools/two_stream_probe.py`).  The forwards of this loop run without the executor's parallel lanes (both at once
        are slower than either): a per-call flag, the process-wide option is not touched.  1 = every forward on the caller's stream.  ``exclusive(k)`` true: the forward of batch k
        runs alone - it starts when the forwards in flight are done and the next one starts behind it (bench.py times the
        kernels of such a step with per-op events: a kernel's duration beside another forward is not its own).

        Yields one ``[(people, scores)] * N`` list per batch, in order, two steps after the batch
        was submitted.  ``on_forward(k, x)`` may replace the plain forward (bench.py records op
        events).  Keep the host thread pools small (``torch.set_num_threads``): a burst of idle-
        spinning OpenMP threads can exhaust a container's CPU quota and stall the launches."""
        import os
        mode = decode_stream or os.environ.get("RTPE_DECODE_STREAM", "side")
        if mode not in ("side", "same"):
            raise ValueError("decode_stream must be 'side' or 'same', not %r" % (mode,))
        main = torch.cuda.current_stream(self.device)
        n_fwd = int(in_flight if in_flight is not None else os.environ.get("RTPE_FORWARDS_IN_FLIGHT", "2"))
        if n_fwd < 1 or n_fwd > 4:
            raise ValueError("in_flight must be 1..4, not %r" % (n_fwd,))
        fwd_streams = None
        if n_fwd > 1:
            from .third_party.pose_higher_hrnet import (FWD_NO_LANES, forward_streams, set_forward_flags,
                                                        set_workspace_slot)
            # per device, not per pipeline: slot 1 + i of a model is always on stream i, whoever loops over it
            fwd_streams = forward_streams(self.device, n_fwd)
        side = None
        if mode == "side":
            side = self.__dict__.get("_decode_stream")
            if side is None:
                # normal priority (RTPE_DECODE_PRIORITY=-1: high).  A HIGH-priority decode stream beside the
                # forward's internal lane streams (option "lanes") more than halves the throughput - batch 1:
                # 123 img/s against 327, batch 32 with lanes on: 1,481 against 2,117 (profiles/r03_lanes_decode_stream.txt)
                # - and buys nothing without them (2,154 against 2,149)
                side = self._decode_stream = torch.cuda.Stream(
                    self.device, priority=int(os.environ.get("RTPE_DECODE_PRIORITY", "0")))
        after_alone = None    # the stream of an exclusive forward that the next forward has to wait for
        topk_done = None      # batch k-1: top-k enqueued
        refine_done = None    # batch k-2: refine enqueued
        P = self.parser

        def on_decode_stream(fn, *args, after=None, uses=()):
            if side is None:
                return fn(*args)
            with torch.cuda.stream(side):
                if after is not None:
                    side.wait_event(after)
                for t in uses:                  # allocated on the main stream, read on the side stream
                    t.record_stream(side)
                return fn(*args)

        # ``on_forward`` must return FRESH output tensors for every batch (the plain forward does): the decode of
        # batch k runs on the side stream while F(k+1) runs on the main one, and nothing makes F(k+1) wait for it.
        try:
            with torch.no_grad():
                for k, x in enumerate(batches):
                    fs = main
                    if fwd_streams is not None:
                        # forward k on stream k % n with workspace slot 1 + k % n; the input was produced on `main`
                        fs = fwd_streams[k % n_fwd]
                        fs.wait_stream(main)
                        alone = exclusive is not None and bool(exclusive(k))
                        if alone or after_alone is not None:
                            for other in fwd_streams[:n_fwd]:
                                if other is not fs:
                                    fs.wait_stream(other)
                        after_alone = fs if alone else None
                        x.record_stream(fs)
                        # lanes off for THIS call only (a per-call flag of the ABI: nothing process-wide is touched,
                        # and nothing stays changed while the generator is suspended or if it is abandoned)
                        prev_slot = set_workspace_slot(1 + k % n_fwd)
                        prev_flags = set_forward_flags(0 if os.environ.get("RTPE_STREAM_LANES", "0") == "1" else FWD_NO_LANES)
                        try:
                            with torch.cuda.stream(fs):
                                preds, refined = on_forward(k, x) if on_forward is not None else self.model(x)
                        finally:
                            set_forward_flags(prev_flags)
                            set_workspace_slot(prev_slot)
                    else:
                        preds, refined = on_forward(k, x) if on_forward is not None else self.model(x)
                    hw = tuple(out_hw) if out_hw is not None else tuple(x.shape[2:])
                    f_done = None
                    if side is not None or fs is not main:
                        f_done = torch.cuda.Event()
                        f_done.record(fs)
                        if side is None:
                            main.wait_event(f_done)          # the decode runs on the caller's stream
                            preds.record_stream(main)
                            refined.record_stream(main)
                    if topk_done is not None:
                        on_decode_stream(P.lowres_match, topk_done)     # host matching overlaps F(k) on the GPU
                    st = on_decode_stream(P.lowres_topk, refined, preds[:, NUM_HEATMAPS:], hw, after=f_done,
                                          uses=(preds, refined))
                    if refine_done is not None:
                        yield P.lowres_finish(refine_done)
                    refine_done, topk_done = topk_done, st
                if topk_done is not None:
                    on_decode_stream(P.lowres_match, topk_done)
                if refine_done is not None:
                    yield P.lowres_finish(refine_done)
                if topk_done is not None:
                    yield P.lowres_finish(topk_done)
        finally:
            # also when the consumer stops early or an exception propagates: whoever continues on the main stream
            # sees the decode as done
            if side is not None:
                main.wait_stream(side)
            if fwd_streams is not None:
                for fs in fwd_streams[:n_fwd]:
                    main.wait_stream(fs)

    def gather(self, image_ids, results, equal_counts=False, force_collective=False):
        """all-gather of the decoded keypoints over the process group (RCCL).  ``equal_counts``:
        every rank contributes the same number of images (no count exchange, no host sync).
        ``force_collective``: issue the collective also in a process group of ONE rank (a world of one is
        otherwise answered locally; the switch lets a single GPU exercise the RCCL calls)."""
        rec = pack_records(image_ids, results, self.device)
        if not _collectives_on(force_collective):
            return rec
        return all_gather_records(rec, equal_counts)


def _collectives_on(force_collective=False):
    """a process group is up and has more than one rank - or one rank and the caller insists"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(force_collective)


def _collective_tensor(t):
    """RCCL moves device memory; gloo (CPU rehearsals, or several ranks sharing one GPU) wants host memory"""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        return t.cpu(), t.device
    return t, None


def all_gather_records(rec, equal_counts=False):
    """records of all ranks, in rank order.  Variable count per rank: pad to the max, gather,
    strip the padding (one count exchange that the host has to read); with ``equal_counts`` a
    single collective and nothing for the host to wait for."""
    import torch.distributed as dist
    world = dist.get_world_size()
    rec, home = _collective_tensor(rec)
    if home is not None:
        return all_gather_records(rec, equal_counts).to(home)
    if equal_counts:
        out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec.contiguous())
        return out
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    m = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((m, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    out = torch.empty((world * m, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    keep = [out[r * m:r * m + int(counts[r].item())] for r in range(world)]
    return torch.cat(keep)


# --------------------------------------------------------------------------- #
# teacher-prediction files (teacher_inference.py:67-90 writes them, dataloaders.py:140-165 reads them)
# --------------------------------------------------------------------------- #
HEATMAPS_ORDER = ["nose", "leye", "reye", "lear", "rear", "lshould", "rshould", "lelbow", "relbow", "lwrist",
                  "rwrist", "lhip", "rhip", "lknee", "rknee", "lankle", "rankle"]   # teacher_inference.py:38-40


def teacher_prediction_path(out_dir, img_path):
    """teacher_inference.py:68-69; numpy appends ``.npz``, the reader asks for
    ``<img_id>.jpg_w48_predictions.npz`` (dataloaders.py:149-150)"""
    import os
    return os.path.join(out_dir, os.path.basename(img_path)) + "_w48_predictions"


def save_teacher_predictions(out_path, preds, refined):
    """teacher_inference.py:83-90: ``preds`` (34,h,w) and ``refined`` (17,2h,2w) of ONE image (tensors or
    arrays, a leading batch axis of 1 is squeezed) -> compressed npz with the reference's four keys"""
    preds = np.asarray(preds.detach().cpu() if torch.is_tensor(preds) else preds, np.float32).squeeze()
    refined = np.asarray(refined.detach().cpu() if torch.is_tensor(refined) else refined, np.float32).squeeze()
    if preds.ndim != 3 or preds.shape[0] != 2 * NUM_HEATMAPS or refined.ndim != 3 or refined.shape[0] != NUM_HEATMAPS:
        raise ValueError("save_teacher_predictions: expected (34,h,w) and (17,H,W), got %s and %s"
                         % (preds.shape, refined.shape))
    np.savez_compressed(out_path, pred_heatmaps=preds[:NUM_HEATMAPS], embeddings=preds[NUM_HEATMAPS:],
                        heatmaps_refined=refined, heatmaps_order=HEATMAPS_ORDER)


def load_teacher_predictions(path, out_hw=None, device=None):
    """dataloaders.py:140-165 ``_get_teacher_data``: -> ``(t_hms, t_ae)`` float32 tensors
    ``heatmaps_refined`` (17,H,W) and ``embeddings`` (17,h,w); with ``out_hw`` both are upsampled with
    ``F.interpolate(mode="bilinear", align_corners=True)`` semantics on the GPU (``device`` required)."""
    npz = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
    t_hms = torch.from_numpy(np.ascontiguousarray(npz["heatmaps_refined"], np.float32))
    t_ae = torch.from_numpy(np.ascontiguousarray(npz["embeddings"], np.float32))
    if device is not None:
        t_hms, t_ae = t_hms.to(device), t_ae.to(device)
    if out_hw is not None:
        from .third_party.group import upsample_bilinear
        t_hms = upsample_bilinear(t_hms.unsqueeze(0), out_hw)[0]
        t_ae = upsample_bilinear(t_ae.unsqueeze(0), out_hw)[0]
    return t_hms, t_ae


def export_teacher_predictions(model, items, out_dir, workers=4):
    """teacher_inference.py:67-90 as a loop: ``items`` yields ``(img_path, t)`` with ``t`` (1,3,H,W) on
    the GPU (see ``rtpe.third_party.transforms.warp_normalize``).  The forward of the next image is
    enqueued while a small thread pool compresses and writes the previous ones (the ~3 MB of
    deflate per image is the cost of this path).  Returns the list of files written."""
    from concurrent.futures import ThreadPoolExecutor
    written, futures = [], []
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool, torch.no_grad():
        for img_path, t in items:
            preds, refined = model(t)
            out_path = teacher_prediction_path(out_dir, img_path)
            p_host = torch.empty(preds.shape, dtype=torch.float32, pin_memory=True)
            r_host = torch.empty(refined.shape, dtype=torch.float32, pin_memory=True)
            p_host.copy_(preds, non_blocking=True)
            r_host.copy_(refined, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()

            def job(ev=ev, p=p_host, r=r_host, path=out_path):
                ev.synchronize()
                save_teacher_predictions(path, p.numpy(), r.numpy())
                return path + ".npz"
            futures.append(pool.submit(job))
        for f in futures:
            written.append(f.result())
    return written


def shard_indices(n_items, rank, world):
    """contiguous blocks, sizes differing by at most one (100 images over 8 ranks
    -> 13,13,13,13,12,12,12,12)"""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def image_id_of(name):
    """COCO image id of a file name such as ``000000576052.jpg`` (the entries of
    assets/coco_minival2017_100.txt of the reference); the id is what travels in the keypoint records"""
    import os
    stem = os.path.splitext(os.path.basename(str(name).strip()))[0]
    return int(stem)


def run_sharded_list(names, infer, batch_size, device, force_collective=False):
    """configs[3]: one image list, one process per GPU.  ``names`` is the WHOLE list, identical on every rank;
    this rank takes its contiguous block (``shard_indices``: 100 names over 8 ranks -> 13,13,13,13,12,12,12,12),
    runs it in batches of ``batch_size`` (the last batch of a shard is short, the shards are uneven) through
    ``infer(list_of_names) -> [(people, scores)] per name``, and the decoded keypoints of ALL ranks are
    all-gathered as fixed-size records, with a count exchange because the shards differ in length.  No other
    collective touches the data path.  Returns ``{image_id: (kpts (P,17,4), scores (P))}`` for the whole list
    on every rank, after checking that every image of the list came back exactly once.  ``force_collective``: run
    the count exchange and the gather also when the process group has a single rank."""
    import torch.distributed as dist
    on = _collectives_on(force_collective)
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    mine = shard_indices(len(names), rank, world)
    ids = [image_id_of(names[i]) for i in range(len(names))]
    if len(set(ids)) != len(ids):
        raise ValueError("run_sharded_list: duplicate image ids in the list")
    recs = []
    for o in range(0, len(mine), batch_size):
        part = mine[o:o + batch_size]
        results = infer([names[i] for i in part])
        if len(results) != len(part):
            raise RuntimeError("run_sharded_list: infer returned %d results for %d images" % (len(results), len(part)))
        recs.append(pack_records([ids[i] for i in part], results, device))
    rec = torch.cat(recs) if recs else torch.zeros((0, RECORD_FLOATS), dtype=torch.float32, device=device)
    allrec = all_gather_records(rec) if on else rec
    out = unpack_records(allrec)
    got = [int(r) for r in allrec[:, 0].cpu().tolist()]
    if sorted(got) != sorted(ids):
        missing = sorted(set(ids) - set(got))
        dup = sorted({g for g in got if got.count(g) > 1})
        raise RuntimeError("run_sharded_list: gathered %d records for %d images (missing %s, duplicated %s)"
                           % (len(got), len(ids), missing[:5], dup[:5]))
    return out


def broadcast_state_dict(sd, src=0, device=None, force_collective=False):
    """rank ``src`` holds the checkpoint; every rank gets the tensors over
    RCCL as ONE packed buffer per dtype.  ``force_collective``: broadcast also in a
    process group of one rank."""
    import torch.distributed as dist
    if not _collectives_on(force_collective):
        return sd
    keys = sorted(sd.keys())
    out = {}
    for dt in (torch.float32, torch.float16, torch.int64):
        ks = [k for k in keys if sd[k].dtype == dt]
        if not ks:
            continue
        flat = torch.cat([sd[k].reshape(-1) for k in ks]).to(device)
        flat, home = _collective_tensor(flat)
        dist.broadcast(flat, src)
        o = 0
        for k in ks:
            n = sd[k].numel()
            out[k] = flat[o:o + n].reshape(sd[k].shape).cpu()
            o += n
    return out
