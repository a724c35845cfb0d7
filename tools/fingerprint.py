"""Identity of the kernels a profile was taken on: sha256 of the built library and of the sources it is built from
(grl_amd/csrc/*.hip, common.h, Makefile, include/grl_hip.h).  bench.py only quotes counter traffic from a
profiles/rNN_pmc_<series>.json whose fingerprint matches what it is running (VERDICT r5 measurement item 8)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def lib_sha256(path=None):
    path = path or os.environ.get('GRL_HIP_LIB') or os.path.join(ROOT, 'grl_amd', 'libgrl_hip.so')
    return _sha([path]) if os.path.isfile(path) else None


def src_sha256():
    c = os.path.join(ROOT, 'grl_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(c, '*.hip'))) + [os.path.join(c, 'common.h'), os.path.join(c, 'Makefile'),
                                                            os.path.join(ROOT, 'include', 'grl_hip.h')]
    return _sha(files)


def fingerprint():
    return {"lib_sha256": lib_sha256(), "src_sha256": src_sha256()}


if __name__ == '__main__':
    import json
    print(json.dumps(fingerprint()))
