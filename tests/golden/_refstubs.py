"""Stub modules that let the reference's own Python import on a CPU-only box.

Used ONLY by tests/golden/make_goldens.py (this container, where
/root/reference exists).  Nothing here is shipped or imported by the product
or by the tests: the GPU box has no /root/reference.

What is stubbed and why it does not touch arithmetic is tabulated in
SURVEY.md section 8(c) / Appendix C.  The one stub that *is* arithmetic is
torch_scatter.scatter(reduce='sum') on int64, restated with scatter_add_
(an integer sum has exactly one right answer), and reduce='max' on fp32, restated with scatter_reduce_('amax') (a maximum
has exactly one right answer too; slots nothing indexes read 0 as in torch_scatter and are never gathered by the reference).
"""
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference'


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, other=(), **kw):
        other = dict(other, **kw)
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].update(v)
            elif isinstance(v, dict):
                d = _AttrDict()
                d.update(v)
                self[k] = d
            else:
                self[k] = v


class _ERModule(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self._cfg = _AttrDict()
        self.set_default_config()
        if config:
            self._cfg.update(config)

    @property
    def config(self):
        return self._cfg

    def set_default_config(self):
        pass


class _Registry(dict):
    def register(self, name=None, obj=None):
        if obj is not None:
            self[name] = obj
            return obj

        def deco(o):
            self[name or o.__name__] = o
            return o
        return deco


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    assert reduce in ('sum', 'add', 'max')
    index = index.expand_as(src)
    size = list(src.shape)
    size[dim] = int(index.max()) + 1 if dim_size is None else dim_size
    if reduce == 'max':
        # torch_scatter's scatter_max: the maximum of the entries sharing an index; slots nothing indexes read 0
        out = torch.full(size, float('-inf'), dtype=src.dtype).scatter_reduce_(dim, index, src, reduce='amax')
        return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)
    return torch.zeros(size, dtype=src.dtype).scatter_add_(dim, index, src)


# ttach==0.0.3 (requirement.txt:165) is neither vendored nor installed: the two transforms the reference composes
# (tools.py:133-137) are restated here from its published behaviour so that the reference's own pre_slide /
# tta_predict can run.  What this pins: the reference's window arithmetic, padding, averaging -- NOT ttach itself.
class _TtaChain:
    def __init__(self, hflip, k):
        self.hflip, self.k = hflip, k

    def augment_image(self, x):
        x = x.flip(3) if self.hflip else x
        return torch.rot90(x, self.k, (2, 3))

    def deaugment_mask(self, m):
        m = torch.rot90(m, -self.k, (2, 3))
        return m.flip(3) if self.hflip else m


class _HorizontalFlip:
    params = (False, True)


class _Rotate90:
    def __init__(self, angles):
        self.params = tuple(a // 90 for a in angles)


class _Compose:
    def __init__(self, transforms):
        assert isinstance(transforms[0], _HorizontalFlip) and isinstance(transforms[1], _Rotate90)
        self.t = transforms

    def __iter__(self):
        for f in self.t[0].params:          # itertools.product order: first transform outermost
            for k in self.t[1].params:
                yield _TtaChain(f, k)

    def __len__(self):
        return len(self.t[0].params) * len(self.t[1].params)


def install():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Logger:
        def info(self, *a, **k):
            pass
        warning = debug = error = info

    reg = types.SimpleNamespace(MODEL=_Registry())
    ever = mod('ever')
    core = mod('ever.core', registry=reg)
    mod('ever.core.registry', MODEL=reg.MODEL)
    lg = mod('ever.core.logger', get_logger=lambda *a, **k: _Logger())
    core.logger = lg
    mod('ever.core.iterator', Iterator=object)
    mod('ever.interface', ERModule=_ERModule, ConfigurableMixin=object)
    pu = mod('ever.util.param_util', freeze_params=lambda *a, **k: None,
             freeze_modules=lambda *a, **k: None,
             count_model_parameters=lambda *a, **k: 0)
    mod('ever.util', param_util=pu)
    ever.ERModule = _ERModule
    ever.registry = reg
    mod('torch_scatter', scatter=_scatter)
    mod('ttach', Compose=_Compose, HorizontalFlip=_HorizontalFlip, Rotate90=_Rotate90)
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _Dummy()

    class _Permissive(types.ModuleType):
        __path__ = []
        __all__ = []

        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return _Dummy

    import importlib.abc
    import importlib.machinery
    prefixes = ('cv2', 'skimage', 'ttach', 'torchvision', 'albumentations',
                'prettytable', 'ever.', 'mmcv', 'seaborn')

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            if name in sys.modules:
                return None
            if name.startswith(prefixes) or (name + '.').startswith(prefixes):
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Permissive(spec.name)

        def exec_module(self, module):
            pass

    sys.meta_path.append(_Finder())
    mod('segment_anything', sam_model_registry={}, SamAutomaticMaskGenerator=object,
        SamPredictor=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
