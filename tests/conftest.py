import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session', autouse=True)
def ensure_built():
    """Fresh checkout: compile libgrl_hip.so (hipcc cross-compiles without a GPU) and the
    C oracle before the first test needs them.  Building the checker is not using it."""
    from grl_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH) or not os.path.isfile(os.path.join(ROOT, 'oracle', 'libgrl_oracle.so')):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope='session')
def synth_models():
    """Product-side nn.Module holders with the synthetic weights loaded
    (CPU tensors); tests move them to the device as needed."""
    import contextlib
    import io
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0,
                            numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    return cnn, siam, siamv
