"""TEST INFRASTRUCTURE ONLY -- loads the *real* reference (isarandi/metrabs, metrabs_pytorch) read-only.

This module imports the reference's PyTorch modules unmodified from ``/root/reference`` by
pre-registering small stub modules for dependencies that are absent in this image
(hydra/posepile/simplepyutils/torchvision/ultralytics).  It is used for two
things only:

* ``oracle/gen_golden.py`` -- minting the golden vectors under ``tests/golden/``;
* ``tests/test_oracle_pin.py`` -- re-checking the in-repo restatement (``oracle/cpu_ref.py``)
  against the live reference when ``/root/reference`` is present (build container only).

``/root/reference`` does not exist on the GPU box; nothing in ``-m gpu`` tests, ``smoke()`` or
``bench.py`` may import this file.  No reference source is copied: the modules are executed where
they lie.
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('METRABS_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'metrabs_pytorch'))


class RefConfig(types.SimpleNamespace):
    """Stand-in for the hydra config returned by metrabs_pytorch.util.get_config()
    (metrabs_pytorch/util.py:41-57; keys from metrabs_pytorch/config/config.yaml:1-22 and
    config_s_256.yaml:5-9)."""


DEFAULT_CONFIG = dict(
    proc_side=256, stride_train=32, stride_test=32, centered_stride=True,
    legacy_centered_stride_bug=False, depth=8, box_size_mm=2200.0, weak_perspective=False,
    mix_3d_inside_fov=0.5, affine_weights=None, transform_coords=False,
    predict_all_and_latents=False, regularize_to_manifold=False)

_config = RefConfig(**DEFAULT_CONFIG)
_loaded = {}


def set_config(**kwargs):
    """Mutates the live config object that the reference reads at call time."""
    for k, v in kwargs.items():
        if k not in DEFAULT_CONFIG:
            raise KeyError(k)
        setattr(_config, k, v)


@contextlib.contextmanager
def config(**kwargs):
    old = {k: getattr(_config, k) for k in kwargs}
    set_config(**kwargs)
    try:
        yield _config
    finally:
        set_config(**old)


class _JointInfoStub:
    """posepile.joint_info.JointInfo as far as multiperson_model.py:25,246-251 needs it:
    ``n_joints``, ``names``, ``stick_figure_edges`` and ``mirror_mapping`` (left/right swap found
    by the leading 'l'/'r' of the joint name, which is posepile's documented convention)."""

    def __init__(self, names, edges):
        import numpy as np
        self.names = [str(n) for n in names]
        self.n_joints = len(self.names)
        self.stick_figure_edges = [tuple(int(x) for x in e) for e in edges]
        index = {n: i for i, n in enumerate(self.names)}
        mapping = []
        for n in self.names:
            if n.startswith('l') and ('r' + n[1:]) in index:
                mapping.append(index['r' + n[1:]])
            elif n.startswith('r') and ('l' + n[1:]) in index:
                mapping.append(index['l' + n[1:]])
            else:
                mapping.append(index[n])
        self.mirror_mapping = np.array(mapping, dtype=np.int64)


def _install_stubs():
    import torch

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    pkg = types.ModuleType('metrabs_pytorch')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'metrabs_pytorch')]
    sys.modules['metrabs_pytorch'] = pkg

    util = types.ModuleType('metrabs_pytorch.util')
    util.get_config = lambda *a, **k: _config
    sys.modules['metrabs_pytorch.util'] = util
    pkg.util = util

    posepile = types.ModuleType('posepile')
    posepile.__path__ = []
    paths = types.ModuleType('posepile.paths')
    paths.DATA_ROOT = '/nonexistent'
    ji = types.ModuleType('posepile.joint_info')
    ji.JointInfo = _JointInfoStub
    posepile.paths, posepile.joint_info = paths, ji

    def get_joint2bone_mat(joint_info):
        """posepile.joint_info.get_joint2bone_mat (third party, not vendored; published algorithm,
        posepile/joint_info.py): one row per stick-figure edge, +1 at its first joint, -1 at its
        second, so `mat @ pose` is the bone vector."""
        import numpy as np
        edges = joint_info.stick_figure_edges
        mat = np.zeros([len(edges), joint_info.n_joints], np.float32)
        for i_bone, (j1, j2) in enumerate(edges):
            mat[i_bone, j1] = 1
            mat[i_bone, j2] = -1
        return torch.from_numpy(mat)

    ji.get_joint2bone_mat = get_joint2bone_mat
    ds3d = types.ModuleType('posepile.datasets3d')
    posepile.datasets3d = ds3d
    spu = types.ModuleType('simplepyutils')
    # plausibility_check.py:13-16 reads FLAGS.bone_length_dataset / FLAGS.bone_length_file; the tests
    # set `simplepyutils.mean_bones` and leave the dataset name empty
    spu.FLAGS = types.SimpleNamespace(bone_length_dataset='', bone_length_file='<stub>')
    spu.mean_bones = None
    spu.load_pickle = lambda path: spu.mean_bones
    sys.modules.update({'posepile': posepile, 'posepile.paths': paths,
                        'posepile.joint_info': ji, 'posepile.datasets3d': ds3d,
                        'simplepyutils': spu})

    tv = types.ModuleType('torchvision')
    tv.__path__ = []
    tvt = types.ModuleType('torchvision.transforms')
    tvt.__path__ = []
    tvf = types.ModuleType('torchvision.transforms.functional')
    tv.transforms, tvt.functional = tvt, tvf

    def tv_resize(img, size, interpolation=None, max_size=None, antialias=True):
        """torchvision.transforms.functional.resize on a float tensor with a (h, w) size (torchvision is
        not installed; its published tensor path is `torch.nn.functional.interpolate(img, size=size,
        mode='bilinear', align_corners=False, antialias=antialias)` -- transforms/_functional_tensor.py
        `resize`, default InterpolationMode.BILINEAR), which is what person_detector.py:23-24 reaches."""
        import torch.nn.functional as F
        return F.interpolate(img, size=[int(size[0]), int(size[1])], mode='bilinear',
                             align_corners=False, antialias=bool(antialias))

    tvf.resize = tv_resize
    # (multiperson_model.py:314 names torchvision's enum; the stub's resize is bilinear whatever it gets)
    tvf.InterpolationMode = types.SimpleNamespace(BILINEAR='bilinear', NEAREST='nearest', BICUBIC='bicubic')

    # ---- what metrabs_pytorch/backbones/efficientnet.py imports from torchvision (row f.4: the
    # backbone's parameter names and TF-'SAME' padding are part of the checkpoint format).  The three
    # classes with behaviour restate torchvision's published definitions (ops/misc.py,
    # ops/stochastic_depth.py, models/_utils.py); the rest are inert placeholders for the ImageNet
    # weight registry, which the reference never uses.
    class Conv2dNormActivation(torch.nn.Sequential):
        def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1,
                     norm_layer=torch.nn.BatchNorm2d, activation_layer=torch.nn.ReLU, dilation=1,
                     inplace=True, bias=None):
            if padding is None:
                padding = (kernel_size - 1) // 2 * dilation
            if bias is None:
                bias = norm_layer is None
            layers = [torch.nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                      dilation=dilation, groups=groups, bias=bias)]
            if norm_layer is not None:
                layers.append(norm_layer(out_channels))
            if activation_layer is not None:
                params = {} if inplace is None else {'inplace': inplace}
                layers.append(activation_layer(**params))
            super().__init__(*layers)
            self.out_channels = out_channels

    class SqueezeExcitation(torch.nn.Module):
        def __init__(self, input_channels, squeeze_channels, activation=torch.nn.ReLU,
                     scale_activation=torch.nn.Sigmoid):
            super().__init__()
            self.avgpool = torch.nn.AdaptiveAvgPool2d(1)
            self.fc1 = torch.nn.Conv2d(input_channels, squeeze_channels, 1)
            self.fc2 = torch.nn.Conv2d(squeeze_channels, input_channels, 1)
            self.activation = activation()
            self.scale_activation = scale_activation()

        def forward(self, x):
            scale = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
            return scale * x

    class StochasticDepth(torch.nn.Module):  # identity at inference
        def __init__(self, p, mode):
            super().__init__()
            self.p, self.mode = p, mode

        def forward(self, x):
            if self.training and self.p > 0:
                raise NotImplementedError('training-time stochastic depth is not stubbed')
            return x

    def _make_divisible(v, divisor, min_value=None):
        if min_value is None:
            min_value = divisor
        new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
        if new_v < 0.9 * v:
            new_v += divisor
        return new_v

    class Weights:
        def __init__(self, url=None, transforms=None, meta=None):
            self.url, self.transforms, self.meta = url, transforms, meta

    class WeightsEnum:
        @classmethod
        def verify(cls, obj):
            return None

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    tv.models = _mod('torchvision.models')
    _mod('torchvision.models._api', Weights=Weights, WeightsEnum=WeightsEnum)
    _mod('torchvision.models._meta', _IMAGENET_CATEGORIES=[])
    _mod('torchvision.models._utils', _make_divisible=_make_divisible,
         _ovewrite_named_param=lambda kwargs, param, new_value: kwargs.__setitem__(param, new_value),
         handle_legacy_interface=lambda **kw: (lambda fn: fn), _ModelURLs=dict)
    tv.ops = _mod('torchvision.ops', StochasticDepth=StochasticDepth)
    _mod('torchvision.ops.misc', Conv2dNormActivation=Conv2dNormActivation,
         SqueezeExcitation=SqueezeExcitation)
    _mod('torchvision.transforms._presets', ImageClassification=object,
         InterpolationMode=types.SimpleNamespace(BICUBIC='bicubic', BILINEAR='bilinear'))
    tv.utils = _mod('torchvision.utils', _log_api_usage_once=lambda *a, **k: None)
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt,
                        'torchvision.transforms.functional': tvf})

    ul = types.ModuleType('ultralytics')

    class YOLO:  # the detector network is out of scope; boxes are supplied by the caller
        """Stand-in for ultralytics.YOLO: records what person_detector.py:38-41 feeds the network and
        answers with the boxes in `YOLO.fake_boxes_xyxy_conf` (one [n, 5] tensor per image, in the
        network's padded input frame), wrapped in the attribute layout the reference reads
        (`r.boxes.xyxy`, `.xywh`, `.conf`)."""
        fake_boxes_xyxy_conf = None
        last_source = None
        last_kwargs = None

        def __init__(self, *a, **k):
            pass

        def predict(self, source, **kw):
            type(self).last_source = source.detach().clone()
            type(self).last_kwargs = dict(kw)
            out = []
            for b in type(self).fake_boxes_xyxy_conf:
                boxes = types.SimpleNamespace(
                    xyxy=b[:, :4],
                    xywh=torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2,
                                      b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], dim=1),
                    conf=b[:, 4])
                out.append(types.SimpleNamespace(boxes=boxes))
            return out

    ul.YOLO = YOLO
    sys.modules['ultralytics'] = ul

    # multiperson_model.py:154-155,171 pass a Tensor of sizes to torch.split, which torch 2.10
    # rejects; accept it the way older torch did.
    if not getattr(torch.split, '_mtr_compat', False):
        orig_split = torch.split

        def split_compat(tensor, split_size_or_sections, dim=0):
            if isinstance(split_size_or_sections, torch.Tensor):
                split_size_or_sections = [int(x) for x in split_size_or_sections]
            return orig_split(tensor, split_size_or_sections, dim)

        split_compat._mtr_compat = True
        torch.split = split_compat


class _FlexShape(list):
    """A shape that adds to tuples AND lists.  ptu.reduce_mean_masked / reduce_sum_masked
    (ptu.py:30,41) compute ``is_valid.shape + [1] * n`` -- torch.Size + list is a TypeError on
    torch >= 2.x -- while ptu.mean_stdev_masked (ptu.py:12) adds a tuple."""

    def __add__(self, other):
        return _FlexShape(list(self) + list(other))

    def __radd__(self, other):
        return _FlexShape(list(other) + list(self))


def flex_mask(mask):
    """The same boolean tensor as a subclass whose ``.shape`` is a _FlexShape: the only thing the
    reference's masked reductions need to run on this torch.  The reference code itself executes
    unmodified, every arithmetic op is torch's own."""
    import torch

    class FlexShapeMask(torch.Tensor):
        @property
        def shape(self):
            return _FlexShape(super().shape)

    return mask.as_subclass(FlexShapeMask)


@contextlib.contextmanager
def weak_perspective_runnable(ref):
    """Inside this context ptu3d.is_within_fov returns a flex_mask, so that
    ptu3d.reconstruct_absolute(weak_perspective=True) (ptu3d.py:9-33 -> :36-49 -> ptu.py:4-34) runs
    end to end.  Results come back as FlexShapeMask-typed tensors: call plain() on them."""
    orig = ref.ptu3d.is_within_fov
    ref.ptu3d.is_within_fov = lambda *a, **k: flex_mask(orig(*a, **k))
    try:
        yield
    finally:
        ref.ptu3d.is_within_fov = orig


def plain(t):
    import torch
    return t.as_subclass(torch.Tensor)


def minmax_values(t):
    """The same tensor as a subclass for which ``torch.min(t, dim=...)`` / ``torch.max(t, dim=...)``
    return the values alone, the way tf.reduce_min / reduce_max do in the TF twin
    (metrabs_tf/multiperson/plausibility_check.py:75-76).  The PyTorch port of
    is_pose_consistent_with_box (metrabs_pytorch/multiperson/plausibility_check.py:86-107) feeds the
    (values, indices) pair of torch.min straight into torch.maximum and raises; with this argument
    type the rest of the function -- the reference's own arithmetic -- runs unmodified."""
    import torch

    class ValuesOnlyMinMax(torch.Tensor):
        @classmethod
        def __torch_function__(cls, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            out = super().__torch_function__(func, types, args, kwargs)
            if func in (torch.min, torch.max) and ('dim' in kwargs or len(args) > 1):
                return out.values
            return out

    return t.as_subclass(ValuesOnlyMinMax)


def load():
    """Returns a namespace with the reference modules (ptu, ptu3d, model_util, metrabs_model,
    warping, multiperson_model, person_detector)."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError(f'reference not found at {REFERENCE_ROOT}')
    _install_stubs()
    import importlib
    _loaded['ptu'] = importlib.import_module('metrabs_pytorch.ptu')
    _loaded['ptu3d'] = importlib.import_module('metrabs_pytorch.ptu3d')
    _loaded['model_util'] = importlib.import_module('metrabs_pytorch.models.util')
    _loaded['metrabs_model'] = importlib.import_module('metrabs_pytorch.models.metrabs')
    _loaded['warping'] = importlib.import_module('metrabs_pytorch.multiperson.warping')
    _loaded['multiperson_model'] = importlib.import_module(
        'metrabs_pytorch.multiperson.multiperson_model')
    _loaded['person_detector'] = importlib.import_module(
        'metrabs_pytorch.multiperson.person_detector')
    _loaded['plausibility_check'] = importlib.import_module(
        'metrabs_pytorch.multiperson.plausibility_check')
    _loaded['efficientnet'] = importlib.import_module('metrabs_pytorch.backbones.efficientnet')
    _loaded['JointInfo'] = _JointInfoStub
    return types.SimpleNamespace(**_loaded)
