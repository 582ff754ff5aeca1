"""GPU-box probe: taps intermediate tensors of the student program and compares with the oracle."""
import json, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "realtime-pose-estimation_amd"))
from oracle import synth, student_ref
from oracle.student_ref import _cam, _pool, _cbr, _bn
from oracle.hrnet_ref import _Net
from rtpe import _native as nat
from rtpe.students import AttentionStudent
from rtpe.third_party.pose_higher_hrnet import ProgramBuilder, Engine

def program(self, tap):
    b = ProgramBuilder(f32=True)
    t = self.stem[1].emit(b)
    m = self.mid_stem
    t = b.conv(t, m[0], m[1], relu=True)
    stem_out = b.conv(t, m[3], m[4], relu=True)
    T = {"stem_out": stem_out}
    cam = self.att_hi[0]
    if tap.startswith("cam_"):
        T["cam_res"] = b.conv(stem_out, cam.residual[0], cam.residual[1], relu=True)
        hc = cam.hdcs[0][0].out_channels; nd = len(cam.hdcs); hp = (hc + 3) // 4 * 4
        cat = b.new_tensor(hp * nd, 2)
        for i, hdc in enumerate(cam.hdcs):
            b.conv(stem_out, hdc[0], hdc[1], relu=True, out=(cat, hp * i), cout_store=hp)
        T["cam_cat"] = cat
        w = cam.hdc_top[0].weight.detach().cpu()
        wp = torch.zeros((w.shape[0], hp * nd, 1, 1), dtype=w.dtype)
        for i in range(nd):
            wp[:, hp * i:hp * i + hc] = w[:, hc * i:hc * (i + 1)]
        T["cam_top"] = b.conv(cat, cam.hdc_top[0], cam.hdc_top[1], relu=True, weight=wp, cin=hp * nd)
        gate = cam.se.emit(b, stem_out)
        T["cam_out"] = b.cam_combine(T["cam_top"], T["cam_res"], gate)
    hi = self.att_hi[0].emit(b, stem_out)
    mid = self.att_mid[1].emit(b, b.avgpool(stem_out))
    lo = self.att_lo[1].emit(b, b.avgpool(mid))
    att = b.fuse([(hi, 0), (lo, 2), (lo, 2)], relu=False)
    T["att_sum"] = att
    logits = b.conv(att, self.att_top[0], None)
    stem2 = b.sigmoid_add(logits, stem_out, out_flag=nat.F_OUT_PREDS)
    T["stem2"] = stem2
    if tap != "stem2":
        hi = self.det_hi[0].emit(b, stem2)
        T["det_hi"] = hi
        if tap != "det_hi":
            pl = b.avgpool(hi)
            T["det_pool"] = pl
            if tap != "det_pool":
                lo = self.det_lo[1].emit(b, pl)
                T["det_lo"] = lo
                if tap != "det_lo":
                    T["det_sum"] = b.fuse([(hi, 0), (lo, 1), (lo, 1)], relu=False)
    t = T[tap]
    c = b.tensors[t][0]
    ident = torch.nn.Conv2d(c, c, 1, bias=False)
    ident.weight.data = torch.eye(c).reshape(c, c, 1, 1)
    b.conv(t, ident, None, out_flag=nat.F_OUT_REFINED, nhwc=False)
    return b.finish()

def oracle_taps(sd, x):
    stem = _Net({k[len("stem.1."):]: v for k, v in sd.items() if k.startswith("stem.1.")}, False)
    s = stem.stem(x).float()
    f = {k: v.float() for k, v in sd.items() if not k.startswith("stem.")}
    s = _cbr(s, f, "mid_stem.")
    s = F.relu(_bn(F.conv2d(s, f["mid_stem.3.weight"], None, 1, 1), f, "mid_stem.4."))
    T = {"stem_out": s}
    hw = s.shape[-2:]
    p = "att_hi.0."
    T["cam_res"] = _cbr(s, f, p + "residual.")
    T["cam_cat"] = torch.cat([_cbr(s, f, p + "hdcs.%d." % i, i + 1) for i in range(5)], 1)
    T["cam_top"] = _cbr(T["cam_cat"], f, p + "hdc_top.")
    T["cam_out"] = _cam(s, f, p)
    hi = _cam(s, f, "att_hi.0."); mid = _cam(_pool(s), f, "att_mid.1."); lo = _cam(_pool(mid), f, "att_lo.1.")
    up = F.interpolate(lo, hw, mode="nearest")
    T["att_sum"] = hi + up + up
    att = torch.sigmoid(F.conv2d(T["att_sum"], f["att_top.0.weight"], f["att_top.0.bias"], 1, 1) / 20)
    s = s + att.expand(s.shape)
    T["stem2"] = s
    hi = _cam(s, f, "det_hi.0."); T["det_hi"] = hi
    T["det_pool"] = _pool(hi)
    lo = _cam(T["det_pool"], f, "det_lo.1."); T["det_lo"] = lo
    up = F.interpolate(lo, hw, mode="nearest")
    T["det_sum"] = hi + up + up
    return T

with torch.no_grad():
    stu = AttentionStudent(None, "cpu", 100, 17, 1, False, None, False).eval()
    shapes = {k: tuple(v.shape) for k, v in stu.state_dict().items()}
    sd = synth.make_state_dict(shapes, 3, "W0")
    stu.load_state_dict(sd, strict=True)
    x = synth.make_images(1, 128, 160, seed=5)
    OT = oracle_taps(sd, x)
    for tap in ("cam_res", "cam_cat", "cam_top", "cam_out", "stem_out", "att_sum", "stem2", "det_hi", "det_pool", "det_lo", "det_sum"):
        eng = Engine(program(stu, tap), 0)
        a, d = eng.forward(x.to("cuda:0"), torch.float32)
        want = OT[tap]
        if tap == "cam_cat":
            got = torch.cat([d.cpu()[:, 28 * i:28 * i + 25] for i in range(5)], 1)
            e = (got - want).abs()
            print("   per-branch err:", [float(e[:, 25 * i:25 * i + 25].max()) for i in range(5)])
        got = d.cpu()[:, :want.shape[1]]
        e = (got - want).abs()
        print("%-9s shape %s  max err %.3e  (range %.3f)  pad-ch max %.3e" % (
            tap, tuple(want.shape), e.max().item(), want.abs().max().item(),
            d.cpu()[:, want.shape[1]:].abs().max().item()), flush=True)
