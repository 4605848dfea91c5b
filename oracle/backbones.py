"""Oracle image backbones (torch CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Restates the third-party architectures the reference names in
loadImageModelClassifier.py:50-75 (``custom-cnn``, ``resnet-18``,
``resnet-50``).  ``custom-cnn`` is pinned by the reference import
(it is defined in the reference itself, :50-60); the torchvision ones are
PARITY UNPINNED (torchvision==0.19.1 is absent) and follow the published v1.5
layout: stride on the 3x3 of the bottleneck, BN eps 1e-5 / momentum 0.1,
kaiming_normal_(fan_out, relu) conv init, BN weight 1 / bias 0.
Sub-module names reproduce torchvision's so ``image_encoder.*`` state_dict keys
line up with checkpoints trained by the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Stage(nn.Sequential):
    pass


class _Residual(nn.Module):
    """One residual unit; `plan` is a list of (cin, cout, k, stride) convs."""

    def __init__(self, plan, shortcut):
        super().__init__()
        for i, (cin, cout, k, s) in enumerate(plan, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, k, stride=s, padding=k // 2, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(cout))
        self.nconv = len(plan)
        if shortcut is not None:
            cin, cout, s = shortcut
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=s, bias=False), nn.BatchNorm2d(cout))
        else:
            self.downsample = None

    def forward(self, x):
        y = x
        for i in range(1, self.nconv + 1):
            y = getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(y))
            if i < self.nconv:
                y = F.relu(y)
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


RESNET_SPECS = {
    # name: (bottleneck?, blocks per stage)
    "resnet-18": (False, (2, 2, 2, 2)),
    "resnet-50": (True, (3, 4, 6, 3)),
}


class OracleResNet(nn.Module):
    def __init__(self, name):
        super().__init__()
        bottleneck, depths = RESNET_SPECS[name]
        expansion = 4 if bottleneck else 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for li, (width, nblk) in enumerate(zip((64, 128, 256, 512), depths), start=1):
            units = []
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 1) else 1
                cout = width * expansion
                if bottleneck:
                    plan = [(cin, width, 1, 1), (width, width, 3, stride), (width, cout, 1, 1)]
                else:
                    plan = [(cin, width, 3, stride), (width, cout, 3, 1)]
                shortcut = (cin, cout, stride) if (stride != 1 or cin != cout) else None
                units.append(_Residual(plan, shortcut))
                cin = cout
            setattr(self, f"layer{li}", _Stage(*units))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Identity()
        self.num_features = cin
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
        return self.fc(torch.flatten(self.avgpool(x), 1))


class _DenseLayer(nn.Module):
    def __init__(self, cin, growth=32, bn_size=4):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=False)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, x):
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.Module):
    def __init__(self, nlayers, cin, growth=32):
        super().__init__()
        for i in range(nlayers):
            setattr(self, f"denselayer{i + 1}", _DenseLayer(cin + i * growth, growth))
        self.nlayers = nlayers

    def forward(self, x):
        feats = [x]
        for i in range(self.nlayers):
            feats.append(getattr(self, f"denselayer{i + 1}")(torch.cat(feats, 1)))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=False)
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.pool = nn.AvgPool2d(2, stride=2)


class OracleDenseNet169(nn.Module):
    """torchvision densenet169 layout (growth 32, blocks (6,12,32,32), bn_size 4, 64 init features);
    forward = features -> relu -> adaptive_avg_pool2d(1) -> flatten -> classifier (Identity) => 1664.
    PARITY UNPINNED against torchvision (absent); state_dict keys follow torchvision's."""

    def __init__(self):
        super().__init__()
        from collections import OrderedDict
        feats = OrderedDict()
        feats["conv0"] = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        feats["norm0"] = nn.BatchNorm2d(64)
        feats["relu0"] = nn.ReLU(inplace=False)
        feats["pool0"] = nn.MaxPool2d(3, stride=2, padding=1)
        c = 64
        for bi, n in enumerate((6, 12, 32, 32), start=1):
            feats[f"denseblock{bi}"] = _DenseBlock(n, c)
            c += 32 * n
            if bi < 4:
                feats[f"transition{bi}"] = _Transition(c, c // 2)
                c //= 2
        feats["norm5"] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(feats)
        self.classifier = nn.Identity()
        self.num_features = c
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        f = F.relu(self.features(x))
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(f, 1), 1))


class OracleVGG16(nn.Module):
    """torchvision vgg16 (cfg D, no BN) with the last classifier Linear dropped as loadImageModelClassifier.py:77-81
    does (`classifier = Sequential(*list(classifier.children())[:-1])`) -> 4096 features.  PARITY UNPINNED against
    torchvision (absent); module indices / state_dict keys follow it."""
    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(), nn.Dropout())
        self.num_features = 4096
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def _cna(cin, cout, k, stride, groups=1):
    """torchvision Conv2dNormActivation(norm=BatchNorm2d, act=ReLU6): Sequential(0 conv, 1 bn, 2 relu6)."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU6(inplace=False))


class _InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, t):
        super().__init__()
        hidden = inp * t
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if t != 1:
            layers.append(_cna(inp, hidden, 1, 1))
        layers += [_cna(hidden, hidden, 3, stride, groups=hidden), nn.Conv2d(hidden, oup, 1, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


MOBILENET_V2_CFG = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


class OracleMobileNetV2(nn.Module):
    """torchvision mobilenet_v2 (width 1.0) with `classifier = Identity` (loadImageModelClassifier.py:96-100) -> 1280.
    PARITY UNPINNED against torchvision (absent); module indices / state_dict keys follow it."""

    def __init__(self):
        super().__init__()
        feats = [_cna(3, 32, 3, 2)]
        cin = 32
        for t, c, n, s in MOBILENET_V2_CFG:
            for i in range(n):
                feats.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_cna(cin, 1280, 1, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Identity()
        self.num_features = 1280
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class _CNA(nn.Sequential):
    """torchvision Conv2dNormActivation: Sequential(0 conv, 1 norm[, 2 activation])."""

    def __init__(self, cin, cout, k, stride, groups, norm, act):
        layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), norm(cout)]
        if act:
            layers.append(nn.SiLU(inplace=False))
        super().__init__(*layers)


class _SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU()
        self.scale_activation = nn.Sigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class StochasticDepth(nn.Module):
    """torchvision.ops.StochasticDepth(p, "row")."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        noise = torch.empty([x.shape[0]] + [1] * (x.dim() - 1), dtype=x.dtype, device=x.device).bernoulli_(keep)
        if keep > 0:
            noise.div_(keep)
        return x * noise


class _MBConv(nn.Module):
    def __init__(self, cin, cout, expand, k, stride, sd_prob, norm):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        hidden = _make_divisible(cin * expand)
        layers = []
        if hidden != cin:
            layers.append(_CNA(cin, hidden, 1, 1, 1, norm, True))
        layers.append(_CNA(hidden, hidden, k, stride, hidden, norm, True))
        layers.append(_SqueezeExcitation(hidden, max(1, cin // 4)))
        layers.append(_CNA(hidden, cout, 1, 1, 1, norm, False))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob)

    def forward(self, x):
        r = self.block(x)
        return self.stochastic_depth(r) + x if self.use_res_connect else r


EFFICIENTNET_CFG = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
                    (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
EFFICIENTNET_SCALE = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b7": (2.0, 3.1)}


def efficientnet_blocks(name):
    """-> list of stages, each a list of (cin, cout, expand, k, stride, sd_prob); torchvision _efficientnet_conf."""
    import math
    width, depth = EFFICIENTNET_SCALE[name]
    adj = lambda c: _make_divisible(c * width)
    stages = [[(adj(i) if li == 0 else adj(o), adj(o), e, k, s if li == 0 else 1) for li in range(int(math.ceil(n * depth)))]
              for e, k, s, i, o, n in EFFICIENTNET_CFG]
    total = sum(len(st) for st in stages)
    out, bid = [], 0
    for st in stages:
        cur = []
        for cin, cout, e, k, s in st:
            cur.append((cin, cout, e, k, s, 0.2 * bid / total))
            bid += 1
        out.append(cur)
    return out, adj(32)


class OracleEfficientNet(nn.Module):
    """torchvision efficientnet_b0 / efficientnet_b7 with `classifier = Identity` (loadImageModelClassifier.py:102-112)
    -> 1280 / 2560 features.  PARITY UNPINNED against torchvision (absent); module tree / state_dict keys follow it."""

    def __init__(self, name):
        super().__init__()
        from functools import partial
        norm = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01) if name == "efficientnet-b7" else nn.BatchNorm2d
        stages, stem = efficientnet_blocks(name)
        feats = [_CNA(3, stem, 3, 2, 1, norm, True)]
        for st in stages:
            feats.append(nn.Sequential(*[_MBConv(cin, cout, e, k, s, p, norm) for cin, cout, e, k, s, p in st]))
        last = stages[-1][-1][1]
        feats.append(_CNA(last, 4 * last, 1, 1, 1, norm, True))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Identity()
        self.num_features = 4 * last
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(self.features(x), 1), 1))


def custom_cnn(common_dim):
    """loadImageModelClassifier.py:50-60."""
    return nn.Sequential(
        nn.Conv2d(3, 16, kernel_size=3, stride=2, padding=1),
        nn.ReLU(),
        nn.MaxPool2d(kernel_size=2),
        nn.AdaptiveAvgPool2d((1, 1)),
        nn.Flatten(),
        nn.Linear(16, common_dim),
    )


def apply_train_mode(model, mode, last_n_layers=1):
    """loadImageModelClassifier.py:15-35 (freeze policy)."""
    params = list(model.parameters())
    for p in params:
        p.requires_grad = False
    if mode == "frozen_weights":
        return
    if mode == "unfrozen_weights":
        for p in params:
            p.requires_grad = True
    elif mode == "last_layer_unfrozen_weights":
        for p in params[-2 * last_n_layers:]:
            p.requires_grad = True
    else:
        raise ValueError(f"Invalid backbone_train_mode: {mode}")


def build_image_encoder(name, common_dim, mode):
    """-> (module, feature_dim); loadImageModelClassifier.py:41-157."""
    if name == "custom-cnn":
        net, dim = custom_cnn(common_dim), common_dim
    elif name in RESNET_SPECS:
        net = OracleResNet(name)
        dim = net.num_features
    elif name == "vgg16":
        net = OracleVGG16()
        dim = net.num_features
    elif name == "mobilenet-v2":
        net = OracleMobileNetV2()
        dim = net.num_features
    elif name in EFFICIENTNET_SCALE:
        net = OracleEfficientNet(name)
        dim = net.num_features
    elif name == "densenet169":
        net = OracleDenseNet169()
        dim = net.num_features
        if mode == "partial":                        # loadImageModelClassifier.py:88-92
            for p in net.parameters():
                p.requires_grad = False
            for p in net.features.denseblock4.parameters():
                p.requires_grad = True
            return net, dim
    else:
        raise ValueError(f"Backbone '{name}' não implementado.")
    apply_train_mode(net, mode)
    return net, dim
