"""The dual-head student of config 5 on MI355X.

Drop-in for the parts of the reference's ``rtpe/students.py`` that
``BASELINE.json`` configs[4] exercises: ``SELayer`` (:118-142),
``ContextAwareModule`` (:145-201), ``StemHRNet`` (:206-283),
``get_pretrained_stem`` (:286-298), ``init_weights`` (:20-32) and
``AttentionStudent`` (:595-771) - same constructor signatures, same attribute
names (hence the same state-dict keys: the bundled
``assets/pretrained_segm_4MB/*.statedict`` load through ``load_state_dicts``),
same ``forward(x, out_hw=None, return_intermediate=False) -> (att, det)``.
The other student classes of that file are abandoned experiments (SURVEY.md
section 2) and are not built.

As for the teacher, the ``torch.nn`` leaves only hold parameters; ``forward``
compiles the tree into one program (fp16 stem under the half wrapper, then a
cast, then fp32 ops: dilated 3x3 convs on the exact-fp32 MFMA, SE gates, CAM
combine, average pooling, sigmoid) and runs it on the HIP executor.  The quirks
of the reference forward are reproduced as they are: ``att = hi + 2*up(lo)``
(:739-742), ``det_hi`` used for both ``hi`` and ``mid`` and ``det_mid`` never
called (:759-760).
"""
import torch

from . import _native as nat
from .third_party.fp16_utils.fp16util import network_to_half, tofp16
from .third_party.pose_higher_hrnet import BN_MOMENTUM, Bottleneck, CompiledModule, ProgramBuilder


def init_weights(module, init_fn=torch.nn.init.kaiming_normal_, bias_val=0.0):
    """reference :20-32"""
    if isinstance(module, (torch.nn.Linear, torch.nn.Conv2d)):
        init_fn(module.weight)
        if module.bias is not None:
            module.bias.data.fill_(bias_val)


class SELayer(torch.nn.Module):
    """squeeze-excitation gate (reference :118-142): returns the (N,C,1,1) gate only"""

    def __init__(self, in_chans, hidden_chans=None, bn_momentum=0.1):
        super().__init__()
        hidden_chans = in_chans // 4 if hidden_chans is None else hidden_chans
        self.avg_pool = torch.nn.AdaptiveAvgPool2d(1)
        self.fc = torch.nn.Sequential(torch.nn.Linear(in_chans, hidden_chans, bias=True),
                                      torch.nn.ReLU(inplace=True),
                                      torch.nn.Linear(hidden_chans, in_chans, bias=True),
                                      torch.nn.Sigmoid())

    def emit(self, b, x):
        return b.se(x, self.fc[0], self.fc[2])

    def forward(self, x):
        raise RuntimeError("rtpe: SELayer runs only inside a compiled student program")


def _cbr(cin, cout, k, dilation=1, bn_momentum=0.1):
    return torch.nn.Sequential(
        torch.nn.Conv2d(cin, cout, kernel_size=k, stride=1, dilation=dilation,
                        padding=dilation * (k // 2), bias=False),
        torch.nn.BatchNorm2d(cout, momentum=bn_momentum), torch.nn.ReLU(inplace=True))


class ContextAwareModule(torch.nn.Module):
    """CAM (reference :145-201): relu(residual(x) + hdc_top(cat(hdc_d(x))) * se(x))"""

    def __init__(self, in_chans, se_chans=None, hdc_dilations=[1, 2, 3, 4], hdc_chans=None, bn_momentum=0.1):
        super().__init__()
        self.residual = _cbr(in_chans, in_chans, 1, bn_momentum=bn_momentum)
        self.se = SELayer(in_chans, se_chans, bn_momentum)
        hdc_chans = in_chans // 4 if hdc_chans is None else hdc_chans
        self.hdcs = torch.nn.ModuleList([_cbr(in_chans, hdc_chans, 3, d, bn_momentum) for d in hdc_dilations])
        self.hdc_top = _cbr(hdc_chans * len(hdc_dilations), in_chans, 1, bn_momentum=bn_momentum)
        self.final_relu = torch.nn.ReLU(inplace=True)

    def emit(self, b, x, col_map=None):
        """``col_map``: physical channel of x for every logical input channel (x may carry pad channels in the
        middle of a concat): the input-channel axis of every weight that reads x is re-indexed"""
        if col_map is not None:
            return self._emit_mapped(b, x, col_map)
        res = b.conv(x, self.residual[0], self.residual[1], relu=True)
        gate = self.se.emit(b, x)
        # torch.cat of the dilated branches (:191) without a copy: every branch writes its
        # channels side by side; each slice is padded to a multiple of 4 channels (16 bytes)
        hc = self.hdcs[0][0].out_channels
        hp = (hc + 3) // 4 * 4
        nd = len(self.hdcs)
        cat = b.new_tensor(hp * nd, b.tensors[x][1])
        for i, hdc in enumerate(self.hdcs):
            b.conv(x, hdc[0], hdc[1], relu=True, out=(cat, hp * i), cout_store=hp)
        w = self.hdc_top[0].weight.detach().cpu()                      # (C, hc*nd, 1, 1)
        wp = torch.zeros((w.shape[0], hp * nd, 1, 1), dtype=w.dtype)
        for i in range(nd):
            wp[:, hp * i:hp * i + hc] = w[:, hc * i:hc * (i + 1)]
        top = b.conv(cat, self.hdc_top[0], self.hdc_top[1], relu=True, weight=wp, cin=hp * nd)
        return b.cam_combine(top, res, gate)

    def _emit_mapped(self, b, x, col_map):
        phys = b.tensors[x][0]

        def remap(w):                                   # (Co, C_logical, k, k) -> (Co, C_physical, k, k)
            w = w.detach().float().cpu()
            out = torch.zeros((w.shape[0], phys) + tuple(w.shape[2:]))
            out[:, col_map] = w
            return out
        res = b.conv(x, self.residual[0], self.residual[1], relu=True, weight=remap(self.residual[0].weight), cin=phys)
        w1 = torch.zeros((self.se.fc[0].out_features, phys))
        w1[:, col_map] = self.se.fc[0].weight.detach().float().cpu()
        gate = b.se(x, self.se.fc[0], self.se.fc[2], w1=w1)
        hc = self.hdcs[0][0].out_channels
        hp = (hc + 3) // 4 * 4
        nd = len(self.hdcs)
        cat = b.new_tensor(hp * nd, b.tensors[x][1])
        for i, hdc in enumerate(self.hdcs):
            b.conv(x, hdc[0], hdc[1], relu=True, out=(cat, hp * i), cout_store=hp, weight=remap(hdc[0].weight), cin=phys)
        w = self.hdc_top[0].weight.detach().cpu()
        wp = torch.zeros((w.shape[0], hp * nd, 1, 1), dtype=w.dtype)
        for i in range(nd):
            wp[:, hp * i:hp * i + hc] = w[:, hc * i:hc * (i + 1)]
        top = b.conv(cat, self.hdc_top[0], self.hdc_top[1], relu=True, weight=wp, cin=hp * nd)
        return b.cam_combine(top, res, gate)

    def forward(self, x):
        raise RuntimeError("rtpe: ContextAwareModule runs only inside a compiled student program")


class StemHRNet(torch.nn.Module):
    """stem + layer1 of HigherHRNet (reference :206-283); same keys as the teacher's stem"""
    INPLANES = 64

    def __init__(self):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(3, self.INPLANES, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(self.INPLANES, momentum=BN_MOMENTUM)
        self.conv2 = torch.nn.Conv2d(self.INPLANES, self.INPLANES, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = torch.nn.BatchNorm2d(self.INPLANES, momentum=BN_MOMENTUM)
        self.relu = torch.nn.ReLU(inplace=True)
        proj = torch.nn.Sequential(torch.nn.Conv2d(64, 256, kernel_size=1, stride=1, bias=False),
                                   torch.nn.BatchNorm2d(256, momentum=BN_MOMENTUM))
        self.layer1 = torch.nn.Sequential(Bottleneck(64, 64, 1, proj), *[Bottleneck(256, 64) for _ in range(3)])

    def emit(self, b):
        t = b.stem(self.conv1, self.bn1)
        t = b.conv(t, self.conv2, self.bn2, relu=True)
        for blk in self.layer1:
            t = blk.emit(b, t)
        return t

    def load_pretrained(self, hhrnet_statedict_path, device="cpu", check=False):
        """reference :262-283: picks the stem keys (``1.`` + key) out of the teacher checkpoint"""
        hhrnet_d = torch.load(hhrnet_statedict_path, map_location=device)
        self.load_state_dict({k: hhrnet_d["1." + k] for k in self.state_dict()})
        if check:
            assert all((hhrnet_d["1." + k].to(device) == v.to(device)).all() for k, v in self.state_dict().items())

    def forward(self, x):
        raise RuntimeError("rtpe: StemHRNet runs only inside a compiled student program")


def get_pretrained_stem(hhrnet_statedict_path, device="cuda", half_precision=True):
    """reference :286-298"""
    stem = network_to_half(StemHRNet()) if half_precision else \
        torch.nn.Sequential(torch.nn.Identity(), StemHRNet())
    stem[1].load_pretrained(hhrnet_statedict_path, device, check=False)
    return stem


class AttentionStudent(CompiledModule):
    """reference :595-771"""

    def __init__(self, hhrnet_statedict_path=None, device="cuda", inplanes=48, num_heatmaps=17, ae_dims=1,
                 half_precision=True, init_fn=torch.nn.init.kaiming_normal_, trainable_stem=False,
                 bn_momentum=0.1):
        super().__init__()
        self.bn_momentum = bn_momentum
        self.num_heatmaps, self.ae_dims = num_heatmaps, ae_dims
        self.stem = StemHRNet()
        self.stem_out_chans = self.stem.layer1[-1].bn3.num_features
        self.trainable_stem = trainable_stem
        self.inplanes = inplanes
        mid = (self.stem_out_chans + inplanes) // 2
        m1, m2 = _cbr(self.stem_out_chans, mid, 3, bn_momentum=bn_momentum), _cbr(mid, inplanes, 3, bn_momentum=bn_momentum)
        self.mid_stem = torch.nn.Sequential(*m1, *m2)
        self.att_lo, self.att_mid, self.att_hi, self.att_top = self._body([1, 2, 3, 4, 5], 1)
        self.det_lo, self.det_mid, self.det_hi, self.det_top = self._body([1, 2, 3, 4], num_heatmaps + ae_dims)
        if init_fn is not None:
            self.apply(lambda module: init_weights(module, init_fn, 0.0))
        self.stem = network_to_half(self.stem) if half_precision else \
            torch.nn.Sequential(torch.nn.Identity(), self.stem)
        if hhrnet_statedict_path is not None:
            self.stem[1].load_pretrained(hhrnet_statedict_path, device, check=False)
        self._init_compiled()
        self.to(device)
        self.device = device

    def _body(self, dilations, top_chans):
        """reference :651-706: [AvgPool+CAM, AvgPool+CAM, CAM, 3x3 conv with bias]"""
        pool = lambda: torch.nn.AvgPool2d(kernel_size=3, stride=2, padding=1, count_include_pad=False)
        cam = lambda: ContextAwareModule(self.inplanes, hdc_dilations=dilations)
        top = torch.nn.Sequential(torch.nn.Conv2d(self.inplanes, top_chans, kernel_size=3, stride=1,
                                                  dilation=1, padding=1, bias=True))
        return torch.nn.ModuleList([torch.nn.Sequential(pool(), cam()), torch.nn.Sequential(pool(), cam()),
                                    torch.nn.Sequential(cam()), top])

    def load_state_dicts(self, inpath):
        """reference :708-722"""
        for name in ("mid_stem", "att_lo", "att_mid", "att_hi", "att_top"):
            getattr(self, name).load_state_dict(torch.load(inpath + name + ".statedict", map_location="cpu"))
        self.invalidate()

    def compile_program(self):
        """the forward of reference :724-771 as one program"""
        half = isinstance(self.stem[0], tofp16)
        b = ProgramBuilder(f32=not half)
        t = self.stem[1].emit(b)
        if half:
            t = b.cast(t)                                   # tofp32 of the wrapped stem
        b.f32 = True
        m = self.mid_stem
        t = b.conv(t, m[0], m[1], relu=True)
        stem_out = b.conv(t, m[3], m[4], relu=True)
        # human-mask (attention) head: hi + up(lo) + up(lo), :736-744
        hi = self.att_hi[0].emit(b, stem_out)
        mid = self.att_mid[1].emit(b, b.avgpool(stem_out))
        lo = self.att_lo[1].emit(b, b.avgpool(mid))
        att = b.fuse([(hi, 0), (lo, 2), (lo, 2)], relu=False)
        logits = b.conv(att, self.att_top[0], None)
        stem_out = b.sigmoid_add(logits, stem_out, out_flag=nat.F_OUT_PREDS)     # :755-756
        # keypoint head: det_hi for hi AND mid, det_lo on top of it, :759-769
        hi = self.det_hi[0].emit(b, stem_out)
        lo = self.det_lo[1].emit(b, b.avgpool(hi))
        det = b.fuse([(hi, 0), (lo, 1), (lo, 1)], relu=False)
        b.conv(det, self.det_top[0], None, out_flag=nat.F_OUT_REFINED, nhwc=False)
        return b.finish()

    def forward(self, x, out_hw=None, return_intermediate=False):
        """x (N,3,H,W) fp32 on the GPU, H and W multiples of 32 -> (att (N,1,H/4,W/4) = sigmoid mask,
        det (N,num_heatmaps+ae_dims,H/4,W/4)), both fp32.  ``out_hw`` and
        ``return_intermediate`` are accepted and unused, as in the reference."""
        self._check_inference(x, "AttentionStudent.forward")
        att, det = self._engine(x.device).forward(x.float(), torch.float32)
        return att, det


class AttentionStudentSteps(CompiledModule):
    """reference :786-1063: the student that ``distillation.py:137`` trains.  The stem features get the
    down-scaled ``alt`` image (the input in LAB or HSV, ``rtpe.dataloaders.rgb2lab``) concatenated, the attention head
    gates them (``stem_out * sigmoid(att)``), a second 5x5-stride-2 stem of the alt image is concatenated and three
    CAMs + a 3x3 conv give the heat maps.  Same constructor, attribute names (state-dict keys) and
    ``forward(x, out_hw=None, alt=None, att_divisor=None) -> (att, det)`` as the reference; one compiled program per
    ``att_divisor`` value."""

    def __init__(self, hhrnet_statedict_path=None, device="cuda", inplanes=48, num_heatmaps=17, ae_dims=1,
                 half_precision=True, init_fn=torch.nn.init.kaiming_normal_, trainable_stem=False, bn_momentum=0.1):
        super().__init__()
        if inplanes % 4:
            raise NotImplementedError("rtpe: AttentionStudentSteps needs inplanes to be a multiple of 4 (16-byte rows)")
        self.bn_momentum = bn_momentum
        self.num_heatmaps, self.ae_dims = num_heatmaps, ae_dims
        self.stem = StemHRNet()
        self.stem_out_chans = self.stem.layer1[-1].bn3.num_features
        self.trainable_stem = trainable_stem
        self.inplanes = inplanes
        mid = (self.stem_out_chans + inplanes) // 2
        m1, m2 = _cbr(self.stem_out_chans, mid, 3, bn_momentum=bn_momentum), _cbr(mid, inplanes, 3, bn_momentum=bn_momentum)
        self.mid_stem = torch.nn.Sequential(*m1, *m2)
        self._alt_planes = 50

        def c5(cin, cout):
            return [torch.nn.Conv2d(cin, cout, kernel_size=5, stride=2, dilation=1, padding=2, bias=False),
                    torch.nn.BatchNorm2d(cout, momentum=bn_momentum), torch.nn.ReLU(inplace=True)]
        self.alt_img_stem = torch.nn.Sequential(*c5(3, self._alt_planes), *c5(self._alt_planes, inplanes))
        self.att_lo, self.att_mid, self.att_hi, self.att_top = self._attention_body()
        self.steps = self._detection_stage()
        if init_fn is not None:
            self.apply(lambda module: init_weights(module, init_fn, 0.0))
        self.stem = network_to_half(self.stem) if half_precision else \
            torch.nn.Sequential(torch.nn.Identity(), self.stem)
        if hhrnet_statedict_path is not None:
            self.stem[1].load_pretrained(hhrnet_statedict_path, device, check=False)
        self._init_compiled()
        self.to(device)
        self.device = device

    def _attention_body(self):
        """reference :871-899"""
        c = self.inplanes + 3
        pool = lambda: torch.nn.AvgPool2d(kernel_size=3, stride=2, padding=1, count_include_pad=False)
        cam = lambda: ContextAwareModule(c, hdc_dilations=[1, 2, 3, 4])
        top = torch.nn.Sequential(torch.nn.Conv2d(c, 1, kernel_size=3, stride=1, dilation=1, padding=1, bias=True))
        return torch.nn.ModuleList([torch.nn.Sequential(pool(), cam()), torch.nn.Sequential(pool(), cam()),
                                    torch.nn.Sequential(cam()), top])

    def _detection_stage(self):
        """reference :901-948: three CAMs over [gated stem + alt (inplanes + 3) | alt stem (inplanes)], then 3x3"""
        c = 2 * self.inplanes + 3
        return torch.nn.Sequential(*[ContextAwareModule(c, hdc_dilations=[1, 2, 3]) for _ in range(3)],
                                   torch.nn.Conv2d(c, self.num_heatmaps + self.ae_dims, kernel_size=3, stride=1,
                                                   dilation=1, padding=1, bias=True))

    def load_state_dicts(self, inpath):
        """reference :950-964"""
        for name in ("mid_stem", "att_lo", "att_mid", "att_hi", "att_top"):
            getattr(self, name).load_state_dict(torch.load(inpath + name + ".statedict", map_location="cpu"))
        self.invalidate()

    def compile_program(self, variant=None):
        """the forward of reference :966-1040 as one program (``variant`` = ("att_divisor", value))"""
        att_divisor = variant[1] if variant else None
        half = isinstance(self.stem[0], tofp16)
        P = self.inplanes
        b = ProgramBuilder(f32=not half)
        t = self.stem[1].emit(b)
        if half:
            t = b.cast(t)
        b.f32 = True
        m = self.mid_stem
        t = b.conv(t, m[0], m[1], relu=True)
        # torch.cat((stem_out, alt), 1) (:1002) without a copy: [mid stem (P) | alt at 1/4 resolution (3) | 0]
        cat1 = b.new_tensor(P + 4, b.tensors[t][1])
        b.conv(t, m[3], m[4], relu=True, out=(cat1, 0), cout_store=P)
        alt = b.aux_input()
        b.resize(alt, cat1, P, 4)                               # F.interpolate(alt, (h, w), mode="bilinear"), :996-1000
        # the second stem of the alt image (:984): two 5x5 stride-2 convs
        a = self.alt_img_stem
        u = b.conv(alt, a[0], a[1], relu=True)
        # attention head (:1004-1019): hi + up(lo) + up(lo), the reference's double use of lo reproduced
        hi = self.att_hi[0].emit(b, cat1)
        mid = self.att_mid[1].emit(b, b.avgpool(cat1))
        lo = self.att_lo[1].emit(b, b.avgpool(mid))
        att = b.fuse([(hi, 0), (lo, 2), (lo, 2)], relu=False)
        logits = b.conv(att, self.att_top[0], None)
        # stem_out * att, then torch.cat((stem_out, alt_stem_out), 1) (:1040-1042): [gated (P+3) | 0 | alt stem (P)]
        cat2 = b.new_tensor(2 * P + 4, b.tensors[cat1][1])
        b.gate_mul(logits, cat1, cat2, P + 4, att_divisor, out_flag=nat.F_OUT_PREDS)
        b.conv(u, a[3], a[4], relu=True, out=(cat2, P + 4), cout_store=P)
        col_map = list(range(P + 3)) + list(range(P + 4, 2 * P + 4))
        x = self.steps[0].emit(b, cat2, col_map=col_map)
        x = self.steps[1].emit(b, x)
        x = self.steps[2].emit(b, x)
        b.conv(x, self.steps[3], None, out_flag=nat.F_OUT_REFINED, nhwc=False)
        return b.finish()

    def forward(self, x, out_hw=None, alt=None, att_divisor=None):
        """x (N,3,H,W) fp32 on the GPU (H, W multiples of 32), alt (N,3,H,W) fp32: the image in the alternative
        colour space -> (att (N,1,H/4,W/4) = sigmoid mask, det (N,num_heatmaps+ae_dims,H/4,W/4)), both fp32"""
        self._check_inference(x, "AttentionStudentSteps.forward")
        if alt is None:
            raise NotImplementedError("ATM alt is expected")        # reference :993-994
        eng = self._engine(x.device, ("att_divisor", None if att_divisor is None else float(att_divisor)))
        att, det = eng.forward(x.float(), torch.float32, aux=alt.to(x.device).float())
        return att, det
