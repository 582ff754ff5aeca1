"""PyTorch-CPU restatement of the HigherHRNet forward pass (oracle).

Test infrastructure (see oracle/__init__.py).  A *functional* re-statement:
the network is evaluated straight from an un-prefixed state dict with
``torch.nn.functional`` ops; topology (branches, blocks, modules, heads) is
read off the key names, so there is no module tree to keep in sync.

Every function cites the reference lines it follows; paths are relative to
/root/reference/rtpe/third_party/.

Numerics: in ``half=True`` mode this reproduces the half wrapper
(fp16_utils/fp16util.py:40-91): the input is cast to fp16, conv / deconv
weights and biases are fp16, BatchNorm parameters and statistics stay fp32,
every op output is an fp16 tensor (so there is one rounding after each conv,
each BN and each add), and the two results are cast back to fp32.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, pose_higher_hrnet.py:52


def strip_prefix(sd, prefix="1."):
    """helpers.py:69-70 loads the checkpoint into Sequential(tofp16, net, tofp32),
    hence the ``1.`` key prefix (students.py:275)."""
    if all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return dict(sd)


def prepare(sd, half):
    """fp16util.py:71-91 ``network_to_half``: ``.half()`` everything, then put
    every BatchNorm (weight, bias, running stats) back to fp32."""
    sd = strip_prefix(sd)
    out = {}
    bn_prefixes = {k[:-len("running_mean")] for k in sd if k.endswith("running_mean")}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        is_bn = any(k.startswith(p) and k[len(p):] in
                    ("weight", "bias", "running_mean", "running_var") for p in bn_prefixes)
        v = v.detach().float()
        out[k] = v if (is_bn or not half) else v.half()
    return out


class _R16(torch.Tensor):
    """fp32 tensor that holds fp16 values (``half="emulate"``): the result of ``a + b`` is rounded to fp16 like
    the half wrapper's add; every other op of this file rounds explicitly"""

    @staticmethod
    def wrap(t):
        return t.half().float().as_subclass(_R16)

    def __add__(self, other):
        return _R16.wrap(torch.Tensor.__add__(self.as_subclass(torch.Tensor), other.as_subclass(torch.Tensor)))


class _Net:
    """``half``: False = plain fp32; True = the half wrapper as PyTorch-CPU runs it (native fp16 tensors; WHICH
    kernels that takes depends on the host CPU: seconds per image with fp16 vector units, minutes without);
    "emulate" = the same rounding points (after every conv, BatchNorm and add) with fp32 kernels: not bit-equal to
    True - the fp32 accumulation ORDER inside a convolution differs, exactly as it does between PyTorch-CPU and
    the MI355X kernels - but statistically the same thing, and fast on any host.  Used to measure how far two
    faithful implementations of the half wrapper are apart."""

    def __init__(self, sd, half):
        self.emulate = half == "emulate"
        self.sd = prepare(sd, bool(half))
        if self.emulate:
            self.sd = {k: v.float() for k, v in self.sd.items()}
        self.half = bool(half)

    def _r(self, t):
        return _R16.wrap(t) if self.emulate else t

    # ---- leaf ops -------------------------------------------------------- #
    def conv(self, x, key, stride=1, pad=0):
        x = x.as_subclass(torch.Tensor) if self.emulate else x
        return self._r(F.conv2d(x, self.sd[key + ".weight"], self.sd.get(key + ".bias"),
                                stride=stride, padding=pad))

    def bn(self, x, key):
        s = self.sd
        x = x.as_subclass(torch.Tensor) if self.emulate else x
        return self._r(F.batch_norm(x, s[key + ".running_mean"], s[key + ".running_var"],
                                    s[key + ".weight"], s[key + ".bias"], False, 0.1, BN_EPS))

    def has(self, key):
        return (key + ".weight") in self.sd

    def count(self, prefix):
        """number of consecutive integer children ``prefix{i}.`` present"""
        n = 0
        while any(k.startswith("%s%d." % (prefix, n)) for k in self.sd):
            n += 1
        return n

    # ---- blocks ---------------------------------------------------------- #
    def basic_block(self, x, p):
        """BasicBlock.forward, pose_higher_hrnet.py:59-75"""
        out = F.relu(self.bn(self.conv(x, p + "conv1", 1, 1), p + "bn1"))
        out = self.bn(self.conv(out, p + "conv2", 1, 1), p + "bn2")
        res = x
        if self.has(p + "downsample.0"):
            res = self.bn(self.conv(x, p + "downsample.0"), p + "downsample.1")
        return F.relu(out + res)

    def bottleneck(self, x, p):
        """Bottleneck.forward, pose_higher_hrnet.py:96-116"""
        out = F.relu(self.bn(self.conv(x, p + "conv1"), p + "bn1"))
        out = F.relu(self.bn(self.conv(out, p + "conv2", 1, 1), p + "bn2"))
        out = self.bn(self.conv(out, p + "conv3"), p + "bn3")
        res = x
        if self.has(p + "downsample.0"):
            res = self.bn(self.conv(x, p + "downsample.0"), p + "downsample.1")
        return F.relu(out + res)

    def block(self, x, p):
        return self.bottleneck(x, p) if self.has(p + "conv3") else self.basic_block(x, p)

    def conv_bn_chain(self, x, p, relu_last):
        """Sequential of Sequential(conv3x3 s2, BN[, ReLU]):
        transition new branches :571-581 (ReLU everywhere) and fuse
        down-paths :213-230 (no ReLU on the last)."""
        n = self.count(p)
        for k in range(n):
            x = self.bn(self.conv(x, "%s%d.0" % (p, k), 2, 1), "%s%d.1" % (p, k))
            if relu_last or k < n - 1:
                x = F.relu(x)
        return x

    def hr_module(self, xs, p):
        """HighResolutionModule.forward, pose_higher_hrnet.py:238-256"""
        nb = self.count(p + "branches.")
        xs = list(xs)
        for i in range(nb):
            for b in range(self.count("%sbranches.%d." % (p, i))):
                xs[i] = self.block(xs[i], "%sbranches.%d.%d." % (p, i, b))
        if nb == 1:
            return xs
        outs = []
        for i in range(self.count(p + "fuse_layers.")):   # 1 if not multi_scale_output
            y = None
            for j in range(nb):
                fp = "%sfuse_layers.%d.%d." % (p, i, j)
                if j == i:
                    t = xs[j]
                elif j > i:   # 1x1 conv + BN + nearest upsample, :201-209
                    t = self.bn(self.conv(xs[j], fp + "0"), fp + "1")
                    t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
                else:         # (i-j) strided 3x3 convs, :213-230
                    t = self.conv_bn_chain(xs[j], fp, relu_last=False)
                y = t if y is None else y + t
            outs.append(F.relu(y))
        return outs

    def transition(self, ys, p, n_new):
        """_make_transition_layer :548-583 and its use in forward :646-669"""
        xs = []
        for i in range(n_new):
            tp = "%s%d." % (p, i)
            if self.has(tp + "0") and self.sd[tp + "0.weight"].dim() == 4 \
                    and (tp + "1.running_mean") in self.sd:
                # Sequential(conv3x3 s1, BN, ReLU), :557-565
                xs.append(F.relu(self.bn(self.conv(ys[-1], tp + "0", 1, 1), tp + "1")))
            elif self.has(tp + "0.0"):
                xs.append(self.conv_bn_chain(ys[-1], tp, relu_last=True))
            else:  # NoOpModule :566-568 -> forward :659
                xs.append(ys[i])
        return xs

    def stem(self, x):
        """conv1..layer1 (pose_higher_hrnet.py:638-644; rtpe/students.py:242-255 StemHRNet.forward)"""
        if self.half:
            x = x.half()                                   # tofp16, fp16util.py:50-51
            if self.emulate:
                x = x.float()
        x = F.relu(self.bn(self.conv(x, "conv1", 2, 1), "bn1"))
        x = F.relu(self.bn(self.conv(x, "conv2", 2, 1), "bn2"))
        for b in range(self.count("layer1.")):
            x = self.block(x, "layer1.%d." % b)
        return x

    def forward(self, x):
        """PoseHigherResolutionNet.forward, pose_higher_hrnet.py:637-686"""
        x = self.stem(x)
        ys = [x]
        for s in (2, 3, 4):
            n_new = self.count("stage%d.0.branches." % s)
            ys = self.transition(ys, "transition%d." % (s - 1), n_new)
            for m in range(self.count("stage%d." % s)):
                ys = self.hr_module(ys, "stage%d.%d." % (s, m))
        outs = []
        x = ys[0]
        fw = self.sd["final_layers.0.weight"]
        y = self.conv(x, "final_layers.0", 1, 1 if fw.shape[-1] == 3 else 0)   # :674
        outs.append(y)
        for i in range(self.count("deconv_layers.")):
            dp = "deconv_layers.%d." % i
            w = self.sd[dp + "0.0.weight"]                 # (Cin, Cout, k, k)
            if w.shape[0] == x.shape[1] + y.shape[1]:      # deconv_cat, :679-680
                x = torch.cat((x, y), 1)
            k = w.shape[-1]
            pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[k]   # _get_deconv_cfg :535-546
            x = self._r(F.conv_transpose2d(x.as_subclass(torch.Tensor) if self.emulate else x, w, None, stride=2,
                                           padding=pad, output_padding=opad))
            x = F.relu(self.bn(x, dp + "0.1"))
            for b in range(1, self.count(dp)):
                x = self.basic_block(x, "%s%d.0." % (dp, b))
            y = self.conv(x, "final_layers.%d" % (i + 1), 1, 1 if fw.shape[-1] == 3 else 0)
            outs.append(y)
        return [o.float().as_subclass(torch.Tensor) for o in outs]   # tofp32, fp16util.py:64-68


@torch.no_grad()
def hrnet_forward(sd, x, half=True):
    """Reference-equivalent forward on the CPU: ``x`` float32 NCHW ->
    ``[preds (N,34,H/4,W/4), refined (N,17,H/2,W/2)]`` float32."""
    return _Net(sd, half).forward(x)


class OracleNet:
    """Prepared (weights converted once) callable, used for CPU timing."""

    def __init__(self, sd, half=True):
        self.net = _Net(sd, half)

    @torch.no_grad()
    def __call__(self, x):
        return self.net.forward(x)
