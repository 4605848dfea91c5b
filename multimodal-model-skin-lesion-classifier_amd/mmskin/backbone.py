"""Image encoders on the HIP path.

HipResNet / HipDenseNet mirror torchvision's module trees (conv1/bn1/layerN.i.convK/bnK/downsample.{0,1};
features.denseblockB.denselayerL.{norm1,conv1,norm2,conv2}, features.transitionB.{norm,conv}, ...)
so that `image_encoder.*` state_dict keys match checkpoints trained with the reference
(loadImageModelClassifier.py:65-92 builds torchvision models).  All parameters are views
into ONE flat fp32 buffer and all BN running statistics into another; a whole forward (or
backward) is a single C-ABI call into the plan executor (csrc/backbone.hip, csrc/densenet.hip).
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, ops
from ._autograd import no_second_order
from ._lib import call, ptr, stream

RESNET_DEPTHS = {"resnet-18": (False, (2, 2, 2, 2)), "resnet-50": (True, (3, 4, 6, 3))}
_DTYPES = {"bf16": _lib.BF16, "bfloat16": _lib.BF16, "fp32": _lib.F32, "float32": _lib.F32}


def default_compute_dtype():
    return os.environ.get("MMSKIN_BACKBONE_DTYPE", "bf16").lower()


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample


class _BasicBlock(nn.Module):
    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = downsample


class _Plan:
    """Owns one C plan handle + its workspace for a fixed (batch, H, W, dtype, device)."""

    def __init__(self, arch, N, H, W, dtype_id, device):
        h = ctypes.c_void_p()
        call("mmskin_backbone_create", arch.encode(), N, H, W, dtype_id, ctypes.byref(h))
        self.handle = h
        lib = _lib.load()
        self.feat_dim = lib.mmskin_backbone_feature_dim(h)
        oh, ow = ctypes.c_int(), ctypes.c_int()
        call("mmskin_backbone_feature_hw", h, ctypes.byref(oh), ctypes.byref(ow))
        self.out_hw = (oh.value, ow.value)
        self.param_numel = lib.mmskin_backbone_param_numel(h)
        self.ws_bytes = lib.mmskin_backbone_workspace_bytes(h)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device) if device is not None else None
        self.eval_key = None      # (param version, buffer version, arena address) of the last folded eval forward

    def last_conv_shape(self):
        c, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        call("mmskin_backbone_last_conv_shape", self.handle, ctypes.byref(c), ctypes.byref(oh), ctypes.byref(ow))
        return c.value, oh.value, ow.value

    def grad_segments(self):
        """(offset, numel) ranges of the flat gradient arena in the order backward completes them ([] = unsegmented)."""
        n = ctypes.c_int()
        call("mmskin_backbone_num_grad_segments", self.handle, ctypes.byref(n))
        out = []
        for i in range(n.value):
            off, numel = ctypes.c_int64(), ctypes.c_int64()
            call("mmskin_backbone_grad_segment", self.handle, i, ctypes.byref(off), ctypes.byref(numel))
            out.append((off.value, numel.value))
        return out

    def wait_grad_segment(self, index, torch_stream):
        """Make `torch_stream` wait until segment `index` of the backward enqueued last is complete."""
        call("mmskin_backbone_wait_grad_segment", self.handle, index, ctypes.c_void_p(torch_stream.cuda_stream))

    def tensor_table(self, kind):
        lib = _lib.load()
        out = []
        for i in range(lib.mmskin_backbone_num_tensors(self.handle, kind)):
            name = ctypes.create_string_buffer(128)
            off, numel, ndim = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
            shape = (ctypes.c_int64 * 4)()
            call("mmskin_backbone_tensor_info", self.handle, kind, i, name, 128, ctypes.byref(off),
                 ctypes.byref(numel), ctypes.byref(ndim), shape)
            out.append((name.value.decode(), off.value, numel.value, tuple(shape[: ndim.value])))
        return out

    def __del__(self):
        try:
            _lib.load().mmskin_backbone_destroy(self.handle)
        except Exception:
            pass


@no_second_order
class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, module, training, *params):
        ops._need_gpu(image, "image_encoder")
        u8 = image.dtype == torch.uint8
        if u8:   # raw decoded batch [N, H, W, 3]: normalisation + layout change happen inside the stem packing kernel
            if image.dim() != 4 or image.shape[-1] != 3:
                raise _lib.MMSkinError(f"uint8 images must be NHWC [N, H, W, 3], got {tuple(image.shape)}")
            image = image.contiguous()
            N, H, W, _ = image.shape
        else:
            image = image.float().contiguous()
            N, _, H, W = image.shape
        plan = module._plan_for(N, H, W, image.device)
        shape = (N, plan.feat_dim) if plan.out_hw == (1, 1) else (N, plan.feat_dim) + plan.out_hw
        feats = torch.empty(shape, device=image.device, dtype=torch.float32)
        # serving: an eval forward whose parameters / BatchNorm buffers are untouched since the previous eval forward on this
        # plan reuses the BN-folded staged weights already in the workspace (tensor version counters; a training forward
        # updates the running statistics behind torch's back, so it always invalidates)
        # (Parameters alias the flat arena through .data, which does not share its version counter: sum the parameters' own --
        # counters only grow, so the sum changes whenever any of them does.)  The C call updates running_mean / running_var
        # without touching any version counter, so every training forward -- on ANY plan of this module -- bumps
        # module._stats_gen, which is part of the key: an eval plan of another batch shape never reuses weights folded
        # with the previous epoch's statistics.
        if training:
            module._stats_gen += 1
        key = None if training else (sum(p._version for p in params), module._flat_b._version, module._flat_p.data_ptr(),
                                     module._stats_gen)
        call("mmskin_backbone_set_option", plan.handle, b"reuse_staged", int(key is not None and plan.eval_key == key))
        plan.eval_key = key
        if u8:
            norm6 = (ctypes.c_float * 6)(*module.input_mean, *module.input_std)
            call("mmskin_backbone_forward_u8", plan.handle, ptr(image), norm6, ptr(module._flat_p), ptr(module._flat_b),
                 ptr(plan.workspace), ptr(feats), int(training), stream())
        else:
            call("mmskin_backbone_forward", plan.handle, ptr(image), ptr(module._flat_p), ptr(module._flat_b),
                 ptr(plan.workspace), ptr(feats), int(training), stream())
        ctx.plan = plan
        ctx.module = module
        ctx.training_fwd = training
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        module, plan = ctx.module, ctx.plan
        if not ctx.training_fwd:
            raise _lib.MMSkinError("image_encoder backward needs a training-mode forward (batch-stat BN)")
        dfeat = dfeat.float().contiguous()
        grads = torch.empty(plan.param_numel, device=dfeat.device, dtype=torch.float32)
        # data-parallel hook (mmskin/dp.py OverlappedGradSync): told before and after the backward launches are enqueued
        sync = getattr(module, "_grad_sync", None)
        fresh = all(p.grad is None for p in module.parameters())   # no accumulation: .grad will alias `grads`
        if sync is not None:
            sync.before_encoder_backward()
        call("mmskin_backbone_backward", plan.handle, ptr(dfeat), ptr(module._flat_p), ptr(plan.workspace), ptr(grads),
             stream())
        module.last_flat_grad = grads
        if sync is not None and fresh and all(ctx.needs_input_grad[3:]):
            sync.after_encoder_backward(plan, grads)
        # Hand every parameter its gradient as a VIEW of the flat buffer (what DDP calls
        # gradient_as_bucket_view): no per-parameter copy, and the data-parallel all-reduce can run on
        # the one flat buffer.  Autograd itself gets None for these inputs.
        for need, p, (off, numel, shape) in zip(ctx.needs_input_grad[3:], module.parameters(), module._layout):
            if not need:
                continue
            view = grads[off:off + numel].view(shape)
            if p.grad is None:
                p.grad = view
            else:
                p.grad.add_(view)
        return (None, None, None) + (None,) * len(module._layout)


class _CamTailFn(torch.autograd.Function):
    """features as a function of the last conv's raw output (eval-mode BN + skip + ReLU + global average pool), so that
    `torch.autograd.grad(score, activations)` of the reference's Grad-CAM code (XAI/models/cam.py:38-43) reaches the
    activations its forward hook captured.  The forward value was already computed by the plan."""

    @staticmethod
    def forward(ctx, acts, plan, feats):
        ctx.plan = plan
        return feats.clone()

    @staticmethod
    def backward(ctx, dfeat):
        plan = ctx.plan
        with torch.no_grad():
            d = dfeat.detach().float().contiguous()
            c, oh, ow = plan.last_conv_shape()
            dx = torch.empty((d.shape[0], c, oh, ow), device=d.device, dtype=torch.float32)
            call("mmskin_backbone_last_conv_grad", plan.handle, ptr(d), ptr(plan.workspace), ptr(dx), stream())
        if torch.is_grad_enabled() and dfeat.requires_grad:
            # create_graph=True (cam.py:38-43): the FIRST-order gradient above is what Grad-CAM++ uses (it squares and cubes
            # it elementwise).  Differentiating THROUGH it a second time is not implemented on the HIP path: say so.
            dx = _CamNoSecondOrder.apply(dx, dfeat)
        return dx, None, None


class _CamNoSecondOrder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dx, dfeat):
        return dx.view_as(dx)

    @staticmethod
    def backward(ctx, g):
        raise _lib.MMSkinError(
            "second-order differentiation through the image encoder's Grad-CAM tail is not supported on the HIP path "
            "(ResNet plans compute d(features)/d(last conv output) in one kernel); the first-order gradients of "
            "autograd.grad(..., create_graph=True) are exact.  DenseNet encoders hooked at features[-1] run their tail in "
            "torch ops and do support it.")


class _FlatBackbone(nn.Module):
    """Parameter arena + plan cache shared by the plan-executed backbones.  Sub-classes build the
    torchvision-named module tree, set `self.arch`, then call `_init_flat()`."""

    # Normalisation applied to uint8 NHWC inputs (skinLesionDatasets.py:29: ImageNet mean / std on the 0..1 scale)
    input_mean = (0.485, 0.456, 0.406)
    input_std = (0.229, 0.224, 0.225)
    max_plans = 4
    # (H, W) or None.  When set, uint8 NHWC batches of another size go through the GPU A.Resize kernel first (the val / test
    # transform of skinLesionDatasets.py:116-120: Resize -> Normalize -> ToTensor, all on the device)
    resize_to = None

    def _init_flat(self, compute_dtype):
        self.compute_dtype = (compute_dtype or default_compute_dtype()).lower()
        if self.compute_dtype not in _DTYPES:
            raise ValueError(f"unknown compute dtype {self.compute_dtype}")
        self._plans = {}
        self._flat_p = None
        self._flat_b = None
        self._layout = None
        self.last_flat_grad = None
        self._stats_gen = 0      # training forwards so far (they rewrite the BatchNorm running statistics in place)
        self._repack()

    # ------------------------------------------------------------------ flat parameter arena
    def _bn_buffers(self):
        for name, m in self.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                yield name, m

    @torch.no_grad()
    def _repack(self):
        params = list(self.parameters())
        device = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=device)
        layout, off = [], 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            layout.append((off, n, tuple(p.shape)))
            off += n
        bns = [m for _, m in self._bn_buffers()]
        fb = torch.empty(sum(2 * m.num_features for m in bns), dtype=torch.float32, device=device)
        off = 0
        for m in bns:
            for key in ("running_mean", "running_var"):
                buf = getattr(m, key)
                n = buf.numel()
                fb[off:off + n].copy_(buf.reshape(-1).float())
                m._buffers[key] = fb[off:off + n]
                off += n
        self._flat_p, self._flat_b, self._layout = flat, fb, layout
        self._plans = {}

    def _packed(self):
        base = self._flat_p.data_ptr()
        for p, (off, _, _) in zip(self.parameters(), self._layout):
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                return False
        bb, off = self._flat_b.data_ptr(), 0
        for _, m in self._bn_buffers():
            for key in ("running_mean", "running_var"):
                if getattr(m, key).data_ptr() != bb + 4 * off:
                    return False
                off += m.num_features
        return True

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._repack()
        return out

    def _plan_for(self, N, H, W, device, arch=None):
        arch = arch or getattr(self, "_arch_override", None) or self.arch
        key = (N, H, W, self.compute_dtype, str(device), arch)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans[key] = self._plans.pop(key)   # most recently used last
        if plan is None:
            plan = _Plan(arch, N, H, W, _DTYPES[self.compute_dtype], device)
            table = plan.tensor_table(0)
            # a "<arch>-features" plan run on the full model's arena (Grad-CAM on DenseNet) names tensors without the prefix
            strip = "features." if arch != self.arch and arch.endswith("-features") else ""
            mine = [(n[len(strip):] if strip and n.startswith(strip) else n, off, numel, shape) for (n, _), (off, numel, shape) in
                    zip(self.named_parameters(), self._layout)]
            if [(t[0], t[1], t[2]) for t in table] != [(t[0], t[1], t[2]) for t in mine]:
                raise _lib.MMSkinError(f"{type(self).__name__} parameter layout disagrees with the C plan")
            btable = plan.tensor_table(1)
            bmine = [f"{n[len(strip):] if strip and n.startswith(strip) else n}.{k}" for n, _ in self._bn_buffers()
                     for k in ("running_mean", "running_var")]
            if [t[0] for t in btable] != bmine:
                raise _lib.MMSkinError(f"{type(self).__name__} buffer layout disagrees with the C plan")
            # the reference loop produces three shapes per epoch (full batch, ragged last train batch, ragged last
            # validation batch; train_pad_20.py:105,121) plus the evaluation passes: keep four plans, drop the oldest
            if len(self._plans) >= self.max_plans:
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = plan
        return plan

    def _hooked_last_conv(self):
        """The module Grad-CAM code finds with `find_last_conv` (XAI/models/model_loader.py:36-41) when it carries hooks."""
        last = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                last = m
        return last if last is not None and len(last._forward_hooks) > 0 else None

    def _hooked_features_tail(self):
        """DenseNet: `model.image_encoder.features[-1]` (norm5), the Grad-CAM++ target of the reference's script
        (interpretability/gradcam_plusplus.py:298), when it carries forward hooks."""
        feats = getattr(self, "features", None)
        if self.arch != "densenet169" or not isinstance(feats, nn.Sequential) or len(feats) == 0:
            return None
        return feats[-1] if len(feats[-1]._forward_hooks) > 0 else None

    def _forward_with_features_hooks(self, image, target):
        """Eval forward of DenseNet up to norm5 on the feature-map plan (same parameter arena), the hooks, then torchvision's
        tail relu -> adaptive_avg_pool2d(1) -> flatten in torch ops, so autograd.grad(score, activations, create_graph=True)
        and anything differentiated after it work as in the reference."""
        if self.training:
            raise _lib.MMSkinError("forward hooks on features[-1] (Grad-CAM) are supported in eval mode")
        self._arch_override = "densenet169-features"
        try:
            fmap = _BackboneFn.apply(image.detach(), self, False, *[p.detach() for p in self.parameters()])
        finally:
            self._arch_override = None
        acts = fmap.detach().requires_grad_(True)
        for hook in list(target._forward_hooks.values()):
            out = hook(target, (None,), acts)
            if out is not None:
                acts = out
        return torch.relu(acts).mean(dim=(2, 3))

    def _forward_with_cam_hooks(self, image, conv):
        if self.training or self.arch not in RESNET_DEPTHS:
            raise _lib.MMSkinError("forward hooks on the last conv (Grad-CAM) are supported for ResNet encoders in eval mode "
                                   "(DenseNet: hook image_encoder.features[-1])")
        image = image.detach()
        n, h, w = (image.shape[0], image.shape[1], image.shape[2]) if image.dtype == torch.uint8 else \
            (image.shape[0], image.shape[2], image.shape[3])
        plan = self._plan_for(n, h, w, image.device)
        call("mmskin_backbone_set_option", plan.handle, b"keep_raw_eval", 1)
        try:
            feats = _BackboneFn.apply(image, self, False, *[p.detach() for p in self.parameters()])
            c, oh, ow = plan.last_conv_shape()
            acts = torch.empty((n, c, oh, ow), device=image.device, dtype=torch.float32)
            call("mmskin_backbone_last_conv_export", plan.handle, ptr(plan.workspace), ptr(acts), stream())
        finally:
            call("mmskin_backbone_set_option", plan.handle, b"keep_raw_eval", 0)
        acts.requires_grad_(True)
        for hook in list(conv._forward_hooks.values()):
            out = hook(conv, (None,), acts)
            if out is not None:
                acts = out
        return _CamTailFn.apply(acts, plan, feats)

    def forward(self, image):
        if not self._packed():
            self._repack()
        if self.resize_to is not None and image.dtype == torch.uint8:
            from .preprocess import resize_u8
            image = resize_u8(image, self.resize_to)
        tail = self._hooked_features_tail()
        if tail is not None:
            return self._forward_with_features_hooks(image, tail)
        conv = self._hooked_last_conv()
        if conv is not None:
            return self._forward_with_cam_hooks(image, conv)
        training = self.training
        feats = _BackboneFn.apply(image, self, training, *self.parameters())
        if training:
            counters = [m.num_batches_tracked for _, m in self._bn_buffers()]
            if counters:
                torch._foreach_add_(counters, 1)
        return feats


class HipResNet(_FlatBackbone):
    def __init__(self, name, compute_dtype=None):
        super().__init__()
        if name not in RESNET_DEPTHS:
            raise ValueError(f"Backbone '{name}' não implementado.")
        self.arch = name
        bottleneck, depths = RESNET_DEPTHS[name]
        exp = 4 if bottleneck else 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for li, (width, nblk) in enumerate(zip((64, 128, 256, 512), depths), start=1):
            blocks = []
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 1) else 1
                cout = width * exp
                ds = None
                if stride != 1 or cin != cout:
                    ds = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))
                blocks.append((_Bottleneck if bottleneck else _BasicBlock)(cin, width, stride, ds))
                cin = cout
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Identity()
        self.num_features = cin
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._init_flat(compute_dtype)


class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)


class _DenseBlock(nn.Module):
    def __init__(self, nlayers, cin, growth, bn_size):
        super().__init__()
        for i in range(nlayers):
            setattr(self, f"denselayer{i + 1}", _DenseLayer(cin + i * growth, growth, bn_size))


class _Transition(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.pool = nn.AvgPool2d(2, stride=2)


class _Features(nn.Sequential):
    """torchvision's `densenet.features` is an nn.Sequential(OrderedDict): consumers index it (`features[-1]`,
    gradcam_plusplus.py:298).  The children are holders of parameters; the plan executor runs the arithmetic."""


def _build_densenet169_features(f):
    """Adds torchvision's densenet169 `.features` children to module f; returns the output channel count."""
    growth, bn_size, c = 32, 4, 64
    f.conv0 = nn.Conv2d(3, c, 7, stride=2, padding=3, bias=False)
    f.norm0 = nn.BatchNorm2d(c)
    f.relu0 = nn.ReLU(inplace=True)
    f.pool0 = nn.MaxPool2d(3, stride=2, padding=1)
    for bi, n in enumerate((6, 12, 32, 32), start=1):
        setattr(f, f"denseblock{bi}", _DenseBlock(n, c, growth, bn_size))
        c += growth * n
        if bi < 4:
            setattr(f, f"transition{bi}", _Transition(c, c // 2))
            c //= 2
    f.norm5 = nn.BatchNorm2d(c)
    for m in f.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
    return c


class HipDenseNet(_FlatBackbone):
    """torchvision densenet169 (growth 32, blocks 6/12/32/32, bn_size 4) with `classifier = Identity`
    (loadImageModelClassifier.py:84-92) -> 1664 features."""

    def __init__(self, name="densenet169", compute_dtype=None):
        super().__init__()
        if name != "densenet169":
            raise ValueError(f"Backbone '{name}' não implementado.")
        self.arch = name
        f = _Features()
        c = _build_densenet169_features(f)
        self.features = f
        self.classifier = nn.Identity()
        self.num_features = c
        self._init_flat(compute_dtype)


class HipDenseNetFeatures(_FlatBackbone):
    """torchvision `densenet169(...).features` as the reference's MD-Net holds it (multimodalMDNet.py:72-76):
    conv0 ... norm5, no final ReLU / pooling; forward returns the feature map [N, 1664, H/32, W/32] (fp32)."""

    def __init__(self, compute_dtype=None):
        super().__init__()
        self.arch = "densenet169-features"
        self.num_features = _build_densenet169_features(self)
        self._init_flat(compute_dtype)


VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class HipVGGFeatures(_FlatBackbone):
    """torchvision `vgg16().features` + `avgpool` on the plan executor (csrc/vgg.hip): children are named by their
    index in torchvision's Sequential ("0", "2", "5", ...), so keys read `features.0.weight`, ...; returns
    [N, 512, 7, 7] fp32."""

    def __init__(self, compute_dtype=None):
        super().__init__()
        self.arch = "vgg16-features"
        cin, idx = 3, 0
        for v in VGG16_CFG:
            if v == "M":
                self.add_module(str(idx), nn.MaxPool2d(2, 2))
                idx += 1
                continue
            conv = nn.Conv2d(cin, v, 3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            self.add_module(str(idx), conv)
            self.add_module(str(idx + 1), nn.ReLU(inplace=True))
            cin, idx = v, idx + 2
        self._init_flat(compute_dtype)


class HipVGG16(nn.Module):
    """torchvision vgg16 with the last classifier layer dropped (loadImageModelClassifier.py:77-81) -> 4096 features."""

    def __init__(self, compute_dtype=None):
        super().__init__()
        from .nn import FusedAway, HipDropout, HipLinear
        self.features = HipVGGFeatures(compute_dtype)
        self.avgpool = FusedAway("AdaptiveAvgPool2d(7) (inside the features plan)")
        self.classifier = nn.Sequential(
            HipLinear(512 * 7 * 7, 4096, fuse_relu=True), FusedAway("ReLU"), HipDropout(0.5),
            HipLinear(4096, 4096, fuse_relu=True), FusedAway("ReLU"), HipDropout(0.5))
        for m in self.classifier:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)
        self.num_features = 4096

    def forward(self, x):
        return self.classifier(self.features(x).flatten(1))


def _cna(cin, cout, k, stride, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


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


MOBILENET_V2_CFG = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


class HipMobileNetV2(_FlatBackbone):
    """torchvision mobilenet_v2 module tree (features.N[.conv.K[.J]]) with `classifier = Identity`
    (loadImageModelClassifier.py:96-100) -> 1280 features; plan executor csrc/mobilenet.hip."""

    def __init__(self, name="mobilenet-v2", compute_dtype=None):
        super().__init__()
        if name != "mobilenet-v2":
            raise ValueError(f"Backbone '{name}' não implementado.")
        self.arch = name
        feats, cin = [_cna(3, 32, 3, 2)], 32
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
        self._init_flat(compute_dtype)


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


EFFICIENTNET_CFG = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
                    (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
EFFICIENTNET_SCALE = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b7": (2.0, 3.1)}


class _SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()


class StochasticDepth(nn.Module):
    """Parameter-less holder of torchvision's StochasticDepth(p, "row"); the plan applies the per-sample mask."""

    def __init__(self, p):
        super().__init__()
        self.p = p
        self.mode = "row"


class _MBConv(nn.Module):
    def __init__(self, cin, cout, expand, k, stride, sd_prob, norm):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        hidden = _make_divisible(cin * expand)

        def cna(ci, co, kk, st, groups, act):
            layers = [nn.Conv2d(ci, co, kk, st, (kk - 1) // 2, groups=groups, bias=False), norm(co)]
            if act:
                layers.append(nn.SiLU(inplace=True))
            return nn.Sequential(*layers)
        layers = []
        if hidden != cin:
            layers.append(cna(cin, hidden, 1, 1, 1, True))
        layers.append(cna(hidden, hidden, k, stride, hidden, True))
        layers.append(_SqueezeExcitation(hidden, max(1, cin // 4)))
        layers.append(cna(hidden, cout, 1, 1, 1, False))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob)


class HipEfficientNet(_FlatBackbone):
    """torchvision efficientnet_b0 / efficientnet_b7 module tree with `classifier = Identity`
    (loadImageModelClassifier.py:102-112) -> 1280 / 2560 features; plan executor csrc/effnet.hip."""

    def __init__(self, name, compute_dtype=None):
        super().__init__()
        if name not in EFFICIENTNET_SCALE:
            raise ValueError(f"Backbone '{name}' não implementado.")
        import math
        from functools import partial
        self.arch = name
        width, depth = EFFICIENTNET_SCALE[name]
        adj = lambda c: _make_divisible(c * width)
        norm = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01) if name == "efficientnet-b7" else nn.BatchNorm2d
        stages = [[(adj(i) if li == 0 else adj(o), adj(o), e, k, s if li == 0 else 1) for li in range(int(math.ceil(n * depth)))]
                  for e, k, s, i, o, n in EFFICIENTNET_CFG]
        total = sum(len(st) for st in stages)
        feats = [nn.Sequential(nn.Conv2d(3, adj(32), 3, 2, 1, bias=False), norm(adj(32)), nn.SiLU(inplace=True))]
        bid = 0
        for st in stages:
            blocks = []
            for cin, cout, e, k, s in st:
                blocks.append(_MBConv(cin, cout, e, k, s, 0.2 * bid / total, norm))
                bid += 1
            feats.append(nn.Sequential(*blocks))
        last = stages[-1][-1][1]
        feats.append(nn.Sequential(nn.Conv2d(last, 4 * last, 1, bias=False), norm(4 * last), nn.SiLU(inplace=True)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Identity()
        self.num_features = 4 * last
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self._sd_mask = None
        self._init_flat(compute_dtype)

    def forward(self, image):
        if self.training:   # this step's stochastic-depth masks (torchvision: bernoulli(1-p) / (1-p) per sample, per block)
            u8 = image.dtype == torch.uint8
            n, h, w = (image.shape[0], image.shape[1], image.shape[2]) if u8 else (image.shape[0], image.shape[2], image.shape[3])
            plan = self._plan_for(n, h, w, image.device)
            probs = [m.stochastic_depth.p for m in self.modules() if isinstance(m, _MBConv) and m.use_res_connect]
            if any(p > 0 for p in probs):
                pr = torch.tensor(probs, device=image.device, dtype=torch.float32)[:, None]
                self._sd_mask = ((torch.rand(len(probs), n, device=image.device) >= pr).float() / (1.0 - pr)).contiguous()
                call("mmskin_backbone_set_pointer", plan.handle, b"sd_mask", ptr(self._sd_mask))
            else:
                self._sd_mask = None
                call("mmskin_backbone_set_pointer", plan.handle, b"sd_mask", None)
        return super().forward(image)


class HipCustomCNN(nn.Sequential):
    """loadImageModelClassifier.py:50-60: Conv(3,16,3,s2,p1)-ReLU-MaxPool2-GAP-Flatten-Linear(16,D)."""

    def __init__(self, common_dim):
        from .nn import FusedAway, HipLinear
        super().__init__(
            nn.Conv2d(3, 16, kernel_size=3, stride=2, padding=1),
            FusedAway("ReLU"),
            FusedAway("MaxPool2d(2)"),
            FusedAway("AdaptiveAvgPool2d(1)"),
            FusedAway("Flatten"),
            HipLinear(16, common_dim),
        )

    def forward(self, x):
        conv = self[0]
        y = ops.direct_conv2d(x, conv.weight, conv.bias, 2, 1, True)
        return self[5](ops.pool_gap(y, 2))
