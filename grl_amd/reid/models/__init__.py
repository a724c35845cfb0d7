"""Model factory with the reference's names
(/root/reference/reid/models/__init__.py:8-49)."""
from .grl_model import resnet50_grl, ResNet50_GRL_Model
from .Siamese import Siamese
from .Siamese_video import Siamese_video


def _resnet50_baseline(*a, **k):
    # reid/models/resnet.py is a torchvision-backed baseline that is not on
    # the GRL path: registered, not provided.
    raise NotImplementedError(
        "'resnet50' (torchvision baseline) is outside the GRL hot path")


__factory = {
    'resnet50': _resnet50_baseline,
    'siamese': Siamese,
    'siamese_video': Siamese_video,
    'resnet50_grl': resnet50_grl,
}


def names():
    return sorted(__factory.keys())


def create(name, *args, **kwargs):
    if name not in __factory:
        raise KeyError("Unknown model:", name)
    return __factory[name](*args, **kwargs)
